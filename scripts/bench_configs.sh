#!/bin/bash
# One bench.py line per BASELINE.json configuration that fits one MI355X (run on the GPU box through gpurun):
# C2, C3, C4 (the default), the 1/8 shards of C4 and C5 (what one rank of the 8-GPU run holds) and C5 whole.
# Usage: scripts/bench_configs.sh TAG   -> gpurun_out/TAG/<config>.json
TAG=${1:-configs}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; python3 $R/bench.py --no-cpu-baseline --data device --repeats 3 "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python3 - $OUT/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r, c = d['roofline'], [v for k, v in d['kernels'].items() if k.startswith('k_colpass')][0]
    sec = d['kernels'].get('row_pass_section', {})
    tail = (' + split tail %.3f ms' % sec['tail_avg_ms']) if sec.get('tail_avg_ms') else ''
    print("  %.1f it/s  %.3f ms/iter  valid %s | row %.3f ms%s (%s roof: %.0f%% of peak, algorithmic; schedule %.0f GB/s) col %.3f ms (%.0f GB/s, %.0f TF) | iter %.0f TF = %.0f%% of its dtype-true t_min" % (
        d['value'], d['ms_per_step'], d['valid'], r['avg_launch_ms'], tail, r['bound'], 100 * r['frac'], r['schedule_hbm_gbs'],
        c['avg_launch_ms'], c['schedule_hbm_gbs'], c['algorithmic_tflops'], d['kernels']['iteration_algorithmic_tflops'],
        100 * d['kernels']['iteration_frac']))
except Exception as e:
    print("  (no result: %s)" % e)
PY
}
run c2 --n 50000 --f 4096 --k 50 --steps 150 --warmup 5
run c3 --n 90000 --f 6144 --k 200 --steps 40 --warmup 5
run c4_shard8 --n 125000 --f 4096 --k 200 --steps 100 --warmup 5
run c4 --n 1000000 --f 4096 --k 200 --steps 40 --warmup 3
run c5_shard8 --n 250000 --f 12288 --k 500 --steps 10 --warmup 2
run c5 --n 2000000 --f 12288 --k 500 --steps 4 --warmup 1
