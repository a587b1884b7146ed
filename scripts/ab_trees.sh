#!/bin/bash
# Interleaved A/B of whole TREES on one device (bench.py + host layer + library of each): the in-tree build against a copy of an
# earlier round under ab/<name>/ (git worktree of that round's commit + its built library; `ab/` travels with gpurun and is
# git-ignored).  Entries: "base" = this tree, anything else = ab/<entry>/.  Environment switches as in ab2.sh: entry:ENV=VAL,...
#   ROUNDS=3 BENCH_ARGS="--rows 125000 --steps 100 --warmup 5" scripts/ab_trees.sh r05tree: base:
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-3}
for i in $(seq 1 $ROUNDS); do
  for e in "$@"; do
    tag=${e%%:*}; envs=${e#*:}
    tree=$R/ab/$tag; [ "$tag" = base ] && tree=$R
    ( for kv in ${envs//,/ }; do export $kv; done
      cd $tree && python3 $tree/bench.py --repeats 1 --data device --no-cpu-baseline --no-16bit-segment $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys;d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);print('$e round $i: %.1f it/s  step %.4f ms  row %.4f  col %.4f  valid %s  loss_last %.6e'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],[v for kk, v in d['kernels'].items() if kk.startswith('k_colpass')][0]['avg_launch_ms'],d['valid'],d['loss_last']))" )
  done
done
