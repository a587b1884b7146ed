#!/usr/bin/env python3
"""Which kernels of a built library use scratch memory (private segment), spill, and how many registers: from the code objects'
metadata notes.  A kernel with a non-zero private segment pays about 4 us more per launch on this part
(experiments/micro/boundary_probe.hip: profiles/r06_boundary_probe.txt).
    python scripts/scratch_report.py multimodal_amd/csrc/libklnmf.so [regex]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_fingerprint import device_code      # noqa: E402

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
FILT = '/usr/bin/c++filt'
lib = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for co in device_code(lib, tmp):
        txt = subprocess.run([READELF, '--notes', co], capture_output=True, text=True).stdout
        for blk in txt.split('- .agpr_count:')[1:]:
            name = re.search(r'\.name:\s+(\S+)', blk)
            priv = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk)
            vg = re.search(r'\.vgpr_count:\s+(\d+)', blk)
            sp = re.search(r'\.vgpr_spill_count:\s+(\d+)', blk)
            lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', blk)
            if name:
                rows.append((name.group(1), int(priv.group(1)) if priv else -1, int(vg.group(1)) if vg else -1,
                             int(sp.group(1)) if sp else -1, int(lds.group(1)) if lds else -1))
names = subprocess.run([FILT], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
n_scr = 0
for (raw, priv, vg, sp, lds), nm in sorted(zip(rows, names), key=lambda t: t[1]):
    nm = nm.replace('void klnmf::', '').replace('klnmf::', '').split('(')[0]
    if pat and not pat.search(nm):
        continue
    if priv > 0:
        n_scr += 1
    if priv > 0 or pat:
        print('%-60s scratch %4d B  vgpr %3d  spilled %3d  lds %6d' % (nm[:60], priv, vg, sp, lds))
print('%d kernels, %d with scratch' % (len(rows), n_scr))
