#!/bin/bash
# Interleaved A/B of library builds AND environment switches on one device:
#   ROUNDS=3 BENCH_ARGS="--n 1000000" scripts/ab2.sh base: nw4o1:KLNMF_ROW_NW=4 nw4o2:KLNMF_ROW_NW=4
# entry = <tag of ab/libklnmf_TAG.so, or "base" for the in-tree library>:<comma-separated ENV=VALUE list>
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-3}
for i in $(seq 1 $ROUNDS); do
  for e in "$@"; do
    tag=${e%%:*}; envs=${e#*:}
    lib=$R/ab/libklnmf_$tag.so; [ "$tag" = base ] && lib=$R/multimodal_amd/csrc/libklnmf.so
    ( export KLNMF_LIB=$lib; for kv in ${envs//,/ }; do export $kv; done
      python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --data device --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys;d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);print('$e round $i: %.1f it/s  step %.3f ms  row %.3f  col %.3f  valid %s  loss_last %.6e'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],[v for kk, v in d['kernels'].items() if kk.startswith('k_colpass')][0]['avg_launch_ms'],d['valid'],d['loss_last']))" )
  done
done
