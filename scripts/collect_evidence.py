#!/usr/bin/env python3
"""Copy the outputs of scripts/final_run_a.sh + final_run_b.sh (gpurun_out/TAG) into profiles/ under the round's names and
rebuild profiles/rNN_pmc_traffic.json from the two traffic.json files:  scripts/collect_evidence.py TAG [r02]"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else 'r04'
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
G = os.path.join(R, 'gpurun_out', tag)
P = os.path.join(R, 'profiles')


def cp(src, dst):
    shutil.copy(os.path.join(G, src), os.path.join(P, '%s_%s' % (rnd, dst)))


cp('bench_default.json', 'bench_1gpu_default.json')
cp('configs.txt', 'bench_configs.txt')
cp('pytest.log', 'gpu_tests.txt')
cp('kt_c4/c4_kernel_stats.csv', 'kernel_stats_c4.csv')
cp('kt_c5s/c5s_kernel_stats.csv', 'kernel_stats_c5_shard.csv')
os.makedirs(os.path.join(P, rnd + '_configs'), exist_ok=True)
for f in glob.glob(os.path.join(G, 'configs', '*.json')):
    shutil.copy(f, os.path.join(P, rnd + '_configs', os.path.basename(f)))
if os.path.exists(os.path.join(G, 'pmc_c4', 'traffic.json')):
    cp('pmc_c4/summary.txt', 'pmc_summary_c4.txt')
    cp('pmc_c5s/summary.txt', 'pmc_summary_c5_shard.txt')
    t4 = json.load(open(os.path.join(G, 'pmc_c4', 'traffic.json')))
    t5 = json.load(open(os.path.join(G, 'pmc_c5s', 'traffic.json')))
    main = 'k_rowpass4<KT=7,odd=1,mode=0,fp8 tiles>'
    split = 'k_rowpass4<KT=7,odd=1,mode=0,fp8 tiles,column-split>'
    slabs = [k for k in t4 if k.startswith('k_wrule_slabs')]
    col8 = [k for k in t4 if k.startswith('k_colpass_q8')]
    col = col8[0] if col8 else 'k_colpass_q2<fp8 tiles>'
    kern = {main: t4[main], col: t4[col]}
    conv = [k for k in t4 if k.startswith('k_w8')]
    if conv:
        kern['k_w8_from_wb'] = t4[conv[0]]
    post = [k for k in t4 if k.startswith('k_post')]
    if post:
        kern['k_post'] = t4[post[0]]
    if split in t4:
        kern[split] = t4[split]
    if slabs:
        kern['k_wrule_slabs'] = t4[slabs[0]]
    n_row = 983040 if split in t4 else 1000000
    sys.path.insert(0, R)
    import bench
    out = {
        # the state of multimodal_amd/csrc/ these figures were measured at: bench.py prints it beside `roofline.traffic` and says
        # whether it is the state that ran (counters cannot be read in-process; the constants go stale when the kernels change)
        'source_hash': bench.kernel_source_hash(),
        'correction': 'bytes = 1024*(2*FETCH_SIZE + WRITE_SIZE): gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)',
        'command': 'bash scripts/final_run_b.sh %s  (scripts/pmc_profile.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE '
                   'TCC_HIT_sum TCC_MISS_sum ... -- python3 bench.py --steps 6 --warmup 2 --repeats 1 --data device --no-cpu-baseline; '
                   'separate passes); scripts/collect_evidence.py' % tag,
        'workloads': [
            {'workload': {'n_local': 1000000, 'f': 4096, 'k': 200, 'precision': 'f16'},
             'note': 'steady state of a loop (from its third iteration on): fp8 ratio tiles. The update pass is three launches: whole '
                     'rows over the workgroups of the full rounds (%d rows: the roofline entry of bench.py), the column-split last '
                     'partial round and its slab W rule' % n_row,
             'kernels': kern,
             'first_iterations_other_kernels': {k: v for k, v in t4.items() if k not in kern and not k.startswith('k_wrule')},
             'algorithmic_bytes_per_rowpass_launch': n_row * 4096 * 2 + 2 * n_row * 200 * 4,
             'iteration_hbm_bytes': sum(v['hbm_bytes_per_launch'] for v in kern.values())},
            {'workload': {'n_local': 250000, 'f': 12288, 'k': 500, 'precision': 'f16'},
             'kernels': t5,
             'algorithmic_bytes_per_rowpass_launch': 250000 * 12288 * 2 + 2 * 250000 * 500 * 4},
        ]}
    json.dump(out, open(os.path.join(P, rnd + '_pmc_traffic.json'), 'w'), indent=1)
    print('iteration HBM bytes %.2f GB, whole-row launch %.2f GB' % (out['workloads'][0]['iteration_hbm_bytes'] / 1e9,
                                                                     kern[main]['hbm_bytes_per_launch'] / 1e9))
d = json.loads(open(os.path.join(G, 'bench_default.json')).read().strip().splitlines()[-1])
r, k = d['roofline'], d['kernels']
print('bench: %.1f it/s  %.3f ms  row launch %.3f ms frac %.3f (%.0f TF)  section %.3f ms  col %.3f ms  iteration frac %.3f  cpu %.4f / %.4f' % (
    d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['achieved'], k['row_pass_section']['avg_ms'],
    [v for kk, v in k.items() if kk.startswith('k_colpass')][0]['avg_launch_ms'], k['iteration_frac'],
    d['cpu_baseline']['value'], d['cpu_baseline']['optimised_cpu']['value']))
