#!/usr/bin/env python3
"""Registers / scratch / occupancy of the hot kernels as hipcc reports them:
    python scripts/resource_usage.py [-DKL_DEV_BUILD ...]   (compiles the device code only; ~1-2 min for the full build)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'multimodal_amd', 'csrc', 'api_loop.hip')
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fno-slp-vectorize', '-std=c++17', '--cuda-device-only', '-c',
       '-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/klnmf_dev.o', src] + sys.argv[1:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip().split(' ')[0]
    if not re.search(r'k_rowpass|k_colpass', name):
        continue
    try:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    dem = re.sub(r'\(.*$', '', dem).replace('void klnmf::', '')
    print('%-48s VGPR %3s AGPR %3s scratch %4s B/lane  occupancy %s waves/SIMD  LDS %6s B' % (
        dem, g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
