#!/bin/bash
# C2 (50 000 x 4096, k = 50: the HBM-bound configuration) under the schedules the library has, interleaved on one box:
#   scripts/c2_schedules.sh [ROUNDS]   -> one line per schedule and round
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${1:-2}
ARGS="--n 50000 --f 4096 --k 50 --steps 150 --warmup 5 --repeats 3 --data device --no-cpu-baseline --no-16bit-segment"
line() { python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);c=[v for kk,v in d['kernels'].items() if kk.startswith('k_colpass')][0]
print('%-34s %7.1f it/s  %.4f ms/iter  row %.4f  col %.4f  valid %s  fp8 %s  hbm-roof frac %.3f' % ('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], c['avg_launch_ms'], d['valid'], d['config']['fp8']['timed_iterations_with_fp8_ratio_tiles'], d['kernels']['iteration_algorithmic_bytes'] / (d['ms_per_step'] * 1e-3) / 8e12))"; }
for i in $(seq 1 $ROUNDS); do
  python3 $R/bench.py $ARGS 2>/dev/null | line "default"
  KLNMF_COLPASS=1 python3 $R/bench.py $ARGS 2>/dev/null | line "COLPASS=1 (recompute from VtB)"
  KLNMF_QTILE=8 python3 $R/bench.py $ARGS 2>/dev/null | line "QTILE=8 (fp8 tiles + fp8 col)"
  KLNMF_QTILE=8 KLNMF_COL8=0 python3 $R/bench.py $ARGS 2>/dev/null | line "QTILE=8 COL8=0 (fp8 tiles, f16 col)"
  KLNMF_ROW_SPLIT=2 python3 $R/bench.py $ARGS 2>/dev/null | line "ROW_SPLIT=2"
done
