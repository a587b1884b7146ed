#!/usr/bin/env python3
"""gpurun_out/TAG/traffic_by_kernel.json (scripts/pmc_traffic.sh) -> the committed traffic files bench.py / bench_sparse.py read:
    python scripts/pmc_traffic_collect.py transform gpurun_out/r05_pmc_transform --n-local 1000000 --f 4096 --k 200
    python scripts/pmc_traffic_collect.py sparse gpurun_out/r05_pmc_sparse --n 20000 --f 110000 --k 50 --precision f64"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('kind', choices=['transform', 'sparse'])
ap.add_argument('dir')
ap.add_argument('--n-local', type=int, default=0)
ap.add_argument('--n', type=int, default=0)
ap.add_argument('--f', type=int)
ap.add_argument('--k', type=int)
ap.add_argument('--precision', default='f16')
args = ap.parse_args()
d = json.load(open(os.path.join(args.dir, 'traffic_by_kernel.json')))
if args.kind == 'transform':
    # the update pass with 16-bit tiles switched off at run time: <KT, ODD, MODE 0, EP, 8 waves, whole rows, Q8 0>
    name = [k for k in d if k.startswith('k_rowpass4') and ', 0, ' in k and k.rstrip('>').endswith('8, 0, 0')]
    name = max(name, key=lambda k: d[k]['launches_per_pass'])
    out = {'source_hash': bench.kernel_source_hash(), 'what': 'scripts/pmc_traffic.sh: FETCH_SIZE (x2 on gfx950) + WRITE_SIZE per launch',
           'workloads': [{'workload': {'n_local': args.n_local, 'f': args.f, 'k': args.k}, 'kernel': name,
                          'hbm_bytes_per_launch': d[name]['hbm_bytes_per_launch'], 'read': d[name]['fetch_bytes_corrected'],
                          'written': d[name]['write_bytes'], 'launches': d[name]['launches_per_pass'],
                          'mean_duration_ms_under_pmc': d[name]['mean_duration_ms_under_pmc']}]}
    path = os.path.join(ROOT, 'profiles', 'r06_pmc_traffic_transform.json')
else:
    per_it = {}
    steps = None
    for k, v in d.items():
        if k.startswith(('k_spb_qw<double, 1, 1>', 'k_spb_qw<float, 1, 1>', 'k_spb_n<', 'k_spb_numer', 'k_spb_wrule', 'k_sp_transpose_H', 'k_update_H',
                         'k_sp_hsum', 'k_sp_colsum', 'k_sp_dots', 'k_sp_loss')):
        # one launch of each per fit iteration
            per_it[k] = v['hbm_bytes_per_launch']
    out = {'source_hash': bench.kernel_source_hash(), 'what': 'scripts/pmc_traffic.sh: FETCH_SIZE (x2 on gfx950) + WRITE_SIZE per launch; '
           'FETCH_SIZE counts the L2\'s fabric requests, Infinity-Cache hits included (MI355X_MICROARCH.md)',
           'workloads': [{'n': args.n, 'f': args.f, 'k': args.k, 'precision': args.precision,
                          'hbm_bytes_per_iteration': sum(per_it.values()), 'kernels': per_it}]}
    path = os.path.join(ROOT, 'profiles', 'r05_pmc_traffic_sparse.json')
json.dump(out, open(path, 'w'), indent=1)
print(path, json.dumps(out['workloads'][0])[:600])
