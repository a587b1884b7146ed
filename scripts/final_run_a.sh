#!/bin/bash
# Evidence run, part A (through gpurun, two calls: `final_run_a.sh TAG 1`, `final_run_a.sh TAG 2`): (1) GPU tests + the default
# bench line; (2) one line per BASELINE configuration, rocprofv3 kernel stats of the default bench, of the C5 shard and of
# the CSR bench, the CSR bench lines, per-launch timelines of one rank's C4 shard and of C2.
# Part B: scripts/final_run_b.sh (PMC passes).  Output: gpurun_out/$TAG/.
TAG=${1:-final}
PART=${2:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$PART" = 1 ]; then
  timeout -k 10 700 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
  timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  exit 0
fi
scripts/bench_configs.sh $TAG/configs > $O/configs.txt 2>&1; cat $O/configs.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python3 $R/bench.py --no-cpu-baseline --no-16bit-segment > $O/kt_c4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5s -o c5s -- python3 $R/bench.py --no-cpu-baseline --data device --repeats 2 --n 250000 --f 12288 --k 500 --steps 10 --warmup 2 --no-16bit-segment > $O/kt_c5s.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sparse -o sparse -- python3 $R/scripts/bench_sparse.py --no-cpu-baseline > $O/kt_sparse.log 2>&1
cd $R
timeout -k 10 200 python3 scripts/bench_sparse.py --precision f64 > $O/sparse_f64.json 2> $O/sparse_f64.err; tail -1 $O/sparse_f64.err
timeout -k 10 100 python3 scripts/bench_sparse.py --precision f32 --no-cpu-baseline > $O/sparse_f32.json 2> $O/sparse_f32.err; tail -1 $O/sparse_f32.err
timeout -k 10 100 python3 scripts/timeline.py $O/tl_shard -- --n 125000 --f 4096 --k 200 --steps 30 --warmup 5 --repeats 1 --data device --no-cpu-baseline --no-16bit-segment > $O/timeline_shard.txt 2>&1
timeout -k 10 100 python3 scripts/timeline.py $O/tl_c2 -- --n 50000 --f 4096 --k 50 --steps 50 --warmup 5 --repeats 1 --data device --no-cpu-baseline --no-16bit-segment > $O/timeline_c2.txt 2>&1
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; grep klnmf $f | cut -c1-160 | head -8; done
