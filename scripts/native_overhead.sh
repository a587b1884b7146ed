#!/bin/bash
# What the native collective path adds per iteration on ONE GPU (KLNMF_COMM_SINGLE=1: a one-rank RCCL communicator; the
# all-reduce is a local copy, so this is the launch / boundary cost of the grouped collective + the separate decision kernel,
# a lower bound of what N ranks pay): one rank's shard of C4 and of C5 at 8 GPUs, single-process loop vs native path.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {   # label, env, args
    local label=$1; shift
    local envs=$1; shift
    out=$(env $envs python3 bench.py --no-cpu-baseline --no-16bit-segment --data device --repeats 3 "$@" 2>/dev/null | grep '^{' | tail -1)
    python3 - "$label" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print('%-34s %8.1f it/s  %.4f ms/iter  segments (ms/iter) %s  path %s ranks %s' % (sys.argv[1], d['value'], d['ms_per_step'],
      ' '.join('%.4f' % s for s in d['segments_ms_per_step']), d['config'].get('collective_path'), d['config'].get('rccl_ranks')))
PY
}
for rep in 1 2; do
run "C4/8 single-process loop"  "X=1"                 --n 125000 --f 4096 --k 200 --steps 40 --warmup 3
run "C4/8 native, 1-rank comm"  "KLNMF_COMM_SINGLE=1" --n 125000 --f 4096 --k 200 --steps 40 --warmup 3 --collective native
run "C5/8 single-process loop"  "X=1"                 --n 250000 --f 12288 --k 500 --steps 10 --warmup 3
run "C5/8 native, 1-rank comm"  "KLNMF_COMM_SINGLE=1" --n 250000 --f 12288 --k 500 --steps 10 --warmup 3 --collective native
done
