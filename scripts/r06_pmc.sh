#!/bin/bash
# Round 6: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of every single-GPU configuration's fit iteration, and the
# instruction-mix / busy counters of configuration 2's row pass (the HBM-bound configuration: what holds it at a quarter of
# its roof?).   bash scripts/r06_pmc.sh PART   (1: C2 + C3 + the C4 shard; 2: C4; 3: C2 counter groups)
R=${GRAFT_REPO_ROOT:-$(pwd)}
B="--steps 8 --warmup 2 --repeats 1 --data device --no-cpu-baseline --no-16bit-segment"
case "$1" in
1) bash $R/scripts/pmc_traffic.sh r06_pmc_c2 python3 $R/bench.py --rows 50000 --components 50 $B > /dev/null
   bash $R/scripts/pmc_traffic.sh r06_pmc_c3 python3 $R/bench.py --rows 90000 --features 6144 $B > /dev/null
   bash $R/scripts/pmc_traffic.sh r06_pmc_c4shard python3 $R/bench.py --rows 125000 $B > /dev/null ;;
2) bash $R/scripts/pmc_traffic.sh r06_pmc_c4 python3 $R/bench.py $B > /dev/null ;;
3) bash $R/scripts/pmc_profile.sh r06_pmc_mix_c2 --rows 50000 --components 50 $B > /dev/null ;;
esac
cd $R && ls gpurun_out/r06_pmc_* -d
