#!/bin/bash
# Evidence run, part A (through gpurun): GPU tests, the default bench line, one line per BASELINE configuration, rocprofv3
# kernel stats of the default bench and of the C5 shard.  Part B: scripts/final_run_b.sh (PMC passes).  Output: gpurun_out/$TAG/.
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
scripts/bench_configs.sh $TAG/configs > $O/configs.txt 2>&1; cat $O/configs.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python3 $R/bench.py --no-cpu-baseline --no-16bit-segment > $O/kt_c4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5s -o c5s -- python3 $R/bench.py --no-cpu-baseline --data device --repeats 2 --n 250000 --f 12288 --k 500 --steps 10 --warmup 2 --no-16bit-segment > $O/kt_c5s.log 2>&1
cd $R
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; grep klnmf $f | cut -c1-160 | head -8; done
