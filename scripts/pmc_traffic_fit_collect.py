#!/usr/bin/env python3
"""gpurun_out/TAG/traffic_by_kernel.json of several FIT workloads (scripts/pmc_traffic.sh around `bench.py ...`) -> the committed
file bench.py reads for `roofline.traffic` (profiles/r06_pmc_traffic.json, one entry per workload):

    python3 scripts/pmc_traffic_fit_collect.py profiles/r06_pmc_traffic.json 50000,4096,50:gpurun_out/r06_pmc_c2 90000,6144,200:gpurun_out/r06_pmc_c3 ...

Per workload: the steady-state launches of one fit iteration (the kernels launched about as often as the dominant update pass:
whole-row pass, column-split tail and its slab W rule, conversion, column pass, k_slab_sum, k_post) with their mean HBM bytes
per launch -- FETCH_SIZE x 2 (gfx950 counts 128-byte requests at 64 bytes) + WRITE_SIZE, separate passes, as
MI355X_MICROARCH.md prescribes -- and their sum as the iteration's traffic."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench      # noqa: E402


def params(name):
    m = re.search(r'<([^>]*)>', name)
    return [p.strip() for p in m.group(1).split(',')] if m else []


out_path = sys.argv[1]
workloads = []
for spec in sys.argv[2:]:
    shape, d = spec.split(':', 1)
    n_local, f, k = (int(v) for v in shape.split(','))
    t = json.load(open(os.path.join(d, 'traffic_by_kernel.json')))
    # the dominant update pass: MODE = 0 (third template parameter), the most launches, then the most bytes
    upd = [kn for kn in t if kn.startswith('k_rowpass4<') and params(kn)[2] == '0']
    whole = [kn for kn in upd if params(kn)[5] == '0']
    dom = max(whole, key=lambda kn: (t[kn]['launches_per_pass'], t[kn]['hbm_bytes_per_launch']))
    its = t[dom]['launches_per_pass']
    kernels, total = {}, 0.0
    for kn, v in t.items():
        if not any(w in kn for w in ('k_rowpass4', 'k_colpass', 'k_wrule_slabs', 'k_w8_from_wb', 'k_slab_sum', 'k_post')):
            continue                                   # (uploads -- k_tile_V -- can be launched about as often as the loop iterates)
        if v['launches_per_pass'] < 0.8 * its or v['launches_per_pass'] > 1.25 * its:
            continue                                   # uploads, the loop's first two iterations, the monitor
        label = kn
        if kn.startswith('k_rowpass4<'):
            label = kn + (' column-split' if params(kn)[5] == '1' else '')
        kernels[label] = {'hbm_bytes_per_launch': v['hbm_bytes_per_launch'], 'fetch_bytes_corrected': v['fetch_bytes_corrected'],
                          'write_bytes': v['write_bytes'], 'launches': v['launches_per_pass'],
                          'mean_duration_ms_under_pmc': v['mean_duration_ms_under_pmc']}
        total += v['hbm_bytes_per_launch']
    workloads.append({'workload': {'n_local': n_local, 'f': f, 'k': k, 'precision': 'f16'},
                      'note': 'steady state of a loop (fp8 ratio tiles); kernels launched once per iteration',
                      'hbm_bytes_per_iteration': total, 'kernels': kernels})
    print('%s: %d iterations, %.1f MB per iteration' % (shape, its, total / 1e6))
    for kn, v in sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch']):
        print('   %-70s %9.1f MB  (read %9.1f, written %9.1f)  %.3f ms' % (kn[:70], v['hbm_bytes_per_launch'] / 1e6, v['fetch_bytes_corrected'] / 1e6,
                                                                         v['write_bytes'] / 1e6, v['mean_duration_ms_under_pmc']))
json.dump({'source_hash': bench.kernel_source_hash(),
           'correction': 'bytes = 1024*(2*FETCH_SIZE + WRITE_SIZE): gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM); '
                         'FETCH_SIZE counts the L2\'s fabric requests, Infinity-Cache hits included',
           'command': 'scripts/pmc_traffic.sh TAG python3 bench.py --rows N --features F --components K --steps 8 --warmup 2 --repeats 1 --data device '
                      '--no-cpu-baseline --no-16bit-segment; scripts/pmc_traffic_fit_collect.py',
           'workloads': workloads}, open(os.path.join(ROOT, out_path), 'w'), indent=1)
