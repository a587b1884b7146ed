#!/bin/bash
# Round 6's evidence run (through gpurun, one call per part): scripts/final_run_r06.sh TAG PART
#   1  GPU tests + the default bench line + one line per BASELINE configuration
#   2  rocprofv3 kernel stats (C4, the C4 shard, the C5 shard) + PMC: HBM traffic of C2 / C3 / the C4 shard / C4 / the transform, counter groups of C4
#   3  interleaved A/B against round 5's tree (ab/r05tree: its bench.py + host layer + library) at the C4 shard, C4, C2, C3, the C5 shard;
#      per-iteration times of the shard (single context and the native collective branch); timelines; transform bench
#   4  (after scripts/pmc_traffic_fit_collect.py + pmc_traffic_collect.py wrote profiles/r06_pmc_traffic*.json) the bench lines that carry the traffic
#   5  the counter groups of configuration 2 (what holds the HBM-bound configuration below its roof)
# Output: gpurun_out/$TAG/.
TAG=${1:-r06_final}
PART=${2:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="--steps 8 --warmup 2 --repeats 1 --data device --no-cpu-baseline --no-16bit-segment"
case $PART in
1)
  timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
  timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  bash scripts/bench_configs.sh $TAG/configs > $O/configs.txt 2>&1; cat $O/configs.txt
  ;;
2)
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python3 $R/bench.py --no-cpu-baseline --no-16bit-segment > $O/kt_c4.log 2>&1; echo "kt_c4 rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4s -o c4s -- python3 $R/bench.py --no-cpu-baseline --data device --repeats 3 --n 125000 --steps 100 --warmup 5 --no-16bit-segment > $O/kt_c4s.log 2>&1; echo "kt_c4s rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5s -o c5s -- python3 $R/bench.py --no-cpu-baseline --data device --repeats 2 --n 250000 --f 12288 --k 500 --steps 10 --warmup 2 --no-16bit-segment > $O/kt_c5s.log 2>&1; echo "kt_c5s rc=$?"
  cd $R
  bash scripts/pmc_traffic.sh $TAG/pmc_c2 python3 $R/bench.py --rows 50000 --components 50 $B > /dev/null 2>&1; echo "pmc c2 rc=$?"
  bash scripts/pmc_traffic.sh $TAG/pmc_c3 python3 $R/bench.py --rows 90000 --features 6144 $B > /dev/null 2>&1; echo "pmc c3 rc=$?"
  bash scripts/pmc_traffic.sh $TAG/pmc_c4shard python3 $R/bench.py --rows 125000 $B > /dev/null 2>&1; echo "pmc c4 shard rc=$?"
  bash scripts/pmc_traffic.sh $TAG/pmc_c4 python3 $R/bench.py $B > /dev/null 2>&1; echo "pmc c4 rc=$?"
  bash scripts/pmc_traffic.sh $TAG/pmc_transform python3 $R/bench.py --workload transform --train-iters 0 --steps 3 --warmup 1 --repeats 1 --data device --no-cpu-baseline > /dev/null 2>&1; echo "pmc transform rc=$?"
  bash scripts/pmc_profile.sh $TAG/pmc_mix_c4 $B > /dev/null 2>&1; echo "pmc mix c4 rc=$?"
  cd $R
  rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv") $(find $O -name "*counter_collection.csv") $(find $O -name "*domain_stats.csv")
  ls $O
  ;;
3)
  ROUNDS=3 BENCH_ARGS="--rows 125000 --steps 100 --warmup 5" bash scripts/ab_trees.sh r05tree: base: > $O/ab_r05_shard.txt 2>&1; cat $O/ab_r05_shard.txt
  ROUNDS=3 BENCH_ARGS="--rows 125000 --steps 20 --warmup 3" bash scripts/ab_trees.sh r05tree: base: > $O/ab_r05_shard_20steps.txt 2>&1; cat $O/ab_r05_shard_20steps.txt
  ROUNDS=2 BENCH_ARGS="--steps 40 --warmup 3" bash scripts/ab_trees.sh r05tree: base: > $O/ab_r05_c4.txt 2>&1; cat $O/ab_r05_c4.txt
  ROUNDS=2 BENCH_ARGS="--rows 50000 --components 50 --steps 150 --warmup 5" bash scripts/ab_trees.sh r05tree: base: > $O/ab_r05_c2.txt 2>&1; cat $O/ab_r05_c2.txt
  ROUNDS=2 BENCH_ARGS="--rows 90000 --features 6144 --steps 40 --warmup 5" bash scripts/ab_trees.sh r05tree: base: > $O/ab_r05_c3.txt 2>&1; cat $O/ab_r05_c3.txt
  ROUNDS=2 BENCH_ARGS="--rows 250000 --features 12288 --components 500 --steps 10 --warmup 2" bash scripts/ab_trees.sh r05tree: base: > $O/ab_r05_c5shard.txt 2>&1; cat $O/ab_r05_c5shard.txt
  python3 scripts/iteration_times.py --rows 125000 > $O/iteration_times_shard.txt 2>&1; tail -8 $O/iteration_times_shard.txt
  python3 scripts/iteration_times.py --rows 125000 --native > $O/iteration_times_shard_native.txt 2>&1; tail -8 $O/iteration_times_shard_native.txt
  cd /tmp && export TMPDIR=/tmp
  for cfg in "shard --rows 125000" "c2 --rows 50000 --components 50" "c3 --rows 90000 --features 6144" "c5shard --rows 250000 --features 12288 --components 500"; do
    set -- $cfg; name=$1; shift
    timeout -k 10 200 python3 $R/scripts/timeline.py $O/tl_$name -- "$@" --data device --steps 30 --warmup 5 --repeats 2 --no-cpu-baseline --no-16bit-segment > $O/timeline_$name.txt 2>&1
    tail -12 $O/timeline_$name.txt
  done
  cd $R
  rm -rf $O/tl_shard $O/tl_c2 $O/tl_c3 $O/tl_c5shard
  timeout -k 10 300 python3 bench.py --workload transform > $O/bench_transform.json 2> $O/bench_transform.err; echo "transform rc=$?"
  ;;
5)
  # the instruction-mix / busy counter groups of configuration 2's row pass (profiles/r06_pmc_summary_c2.txt)
  bash scripts/pmc_profile.sh $TAG/pmc_mix_c2 --rows 50000 --components 50 $B > /dev/null 2>&1; echo "pmc mix c2 rc=$?"
  ;;
4)
  timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  timeout -k 10 300 python3 bench.py --workload transform > $O/bench_transform.json 2> $O/bench_transform.err; echo "transform rc=$?"
  bash scripts/bench_configs.sh $TAG/configs > $O/configs.txt 2>&1; cat $O/configs.txt
  ;;
esac
