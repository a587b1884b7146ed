#!/bin/bash
# The whole fuzz campaign against the oracle on one GPU box (through gpurun; two calls: `fuzz_all.sh TAG 1`, `fuzz_all.sh TAG 2`).
# Output: gpurun_out/TAG/*.txt and one summary line per script (cases inside / outside their tolerance).
TAG=${1:-fuzz}
PART=${2:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() {   # name, timeout, command...
  local name=$1 t=$2; shift 2
  timeout -k 10 $t "$@" > $O/$name.txt 2>&1
  local rc=$?
  if grep -q "Memory access fault\|HSA_STATUS_ERROR" $O/$name.txt; then echo "$name: GPU FAULT"; exit 3; fi
  echo "$name: rc=$rc  inside $(grep -c ' ok ' $O/$name.txt)  $(grep -i 'outside\|differ' $O/$name.txt | tail -1)"
}
if [ "$PART" = 1 ]; then
  run shape_f16_f64   500 python3 scripts/shape_fuzz.py --random 40 --seed 0 --precisions f16,f64
  run shape_f32       400 python3 scripts/shape_fuzz.py --random 25 --seed 5 --precisions f32,f16
  run data_f16        300 python3 scripts/data_fuzz.py
  run sparse          200 python3 scripts/sparse_fuzz.py
  exit 0
fi
run learner         200 python3 scripts/learner_fuzz.py
run sequence        300 python3 scripts/sequence_fuzz.py --steps 100 --seed 0
run tol             300 python3 scripts/tol_fuzz.py
run degenerate      200 python3 scripts/degenerate_fuzz.py
run shard           300 python3 scripts/shard_fuzz.py
run distance        200 python3 scripts/distance_fuzz.py
run upload          200 python3 scripts/upload_fuzz.py
