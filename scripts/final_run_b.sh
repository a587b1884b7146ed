#!/bin/bash
# Evidence run, part B (through gpurun): PMC passes of the default workload and of the C5 shard (scripts/pmc_profile.sh).
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
scripts/pmc_profile.sh $TAG/pmc_c4 --steps 6 --warmup 2 --repeats 1 --data device --no-cpu-baseline --no-16bit-segment > $O/pmc_c4.txt 2>&1
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv") $(find $O -name "*counter_collection.csv")
scripts/pmc_profile.sh $TAG/pmc_c5s --n 250000 --f 12288 --k 500 --steps 3 --warmup 1 --repeats 1 --data device --no-cpu-baseline --no-16bit-segment > $O/pmc_c5s.txt 2>&1
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv") $(find $O -name "*counter_collection.csv")
tail -4 $O/pmc_c4.txt; tail -4 $O/pmc_c5s.txt
