#!/bin/bash
# A/B two builds of libklnmf on the SAME device, interleaved rounds (guide rule 24).
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-3}
for i in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    KLNMF_LIB=$R/ab/libklnmf_$v.so python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys;d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);print('$v round $i: %.1f it/s  step %.3f ms  row %.3f  col %.3f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],[v for kk, v in d['kernels'].items() if kk.startswith('k_colpass')][0]['avg_launch_ms']))"
  done
done
