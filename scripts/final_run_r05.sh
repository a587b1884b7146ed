#!/bin/bash
# Round 5's evidence run (through gpurun, one call per part): scripts/final_run_r05.sh TAG PART
#   1  GPU tests + the default bench line + one line per BASELINE configuration                    (final_run_a.sh 1, bench_configs.sh)
#   2  rocprofv3 kernel stats (C4, C5 shard, CSR) + PMC passes of C4 and the C5 shard               (final_run_a.sh 2's profiles, final_run_b.sh)
#   3  N1: transform bench (C4's matrix; C3's stack against a column-sliced dictionary) + its HBM traffic; N3: CSR bench + traffic;
#      the monitor's cost at C4 (interleaved A/B)
#   4  the fp8 regime against the oracle: scripts/monitor_calibration.py (every class; the long loops), scripts/fp8_drift_probe.py
#   5  (after scripts/collect_evidence.py + scripts/pmc_traffic_collect.py have written profiles/r05_pmc_traffic*.json) the bench lines
#      that carry `roofline.traffic` from those files: fit, transform, CSR
# Output: gpurun_out/$TAG/; scripts/collect_evidence.py TAG r05 copies what is judged into profiles/.
TAG=${1:-r05_final}
PART=${2:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
case $PART in
1)
  timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
  timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  scripts/bench_configs.sh $TAG/configs > $O/configs.txt 2>&1; cat $O/configs.txt
  ;;
2)
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python3 $R/bench.py --no-cpu-baseline --no-16bit-segment > $O/kt_c4.log 2>&1
  echo "kt_c4 rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5s -o c5s -- python3 $R/bench.py --no-cpu-baseline --data device --repeats 2 --n 250000 --f 12288 --k 500 --steps 10 --warmup 2 --no-16bit-segment > $O/kt_c5s.log 2>&1
  echo "kt_c5s rc=$?"
  cd $R
  bash scripts/final_run_b.sh $TAG
  rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv")
  for f in $(find $O -name "*kernel_stats.csv"); do echo $f; grep klnmf $f | cut -c1-160 | head -8; done
  ;;
3)
  timeout -k 10 300 python3 bench.py --workload transform > $O/bench_transform.json 2> $O/bench_transform.err; echo "transform rc=$?"
  timeout -k 10 300 python3 bench.py --workload transform --n 90000 --f 6144 --k 200 --slice 4096 --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_transform_sliced_c3.json 2> $O/bench_transform_sliced_c3.err; echo "sliced rc=$?"
  bash scripts/pmc_traffic.sh $TAG/pmc_transform python3 $R/bench.py --workload transform --train-iters 0 --steps 3 --warmup 1 --repeats 1 --data device --no-cpu-baseline > $O/pmc_transform.txt 2>&1
  cd $R
  timeout -k 10 300 python3 scripts/bench_sparse.py --precision f64 > $O/sparse_f64.json 2> $O/sparse_f64.err; echo "sparse f64 rc=$?"
  timeout -k 10 200 python3 scripts/bench_sparse.py --precision f32 --no-cpu-baseline > $O/sparse_f32.json 2> $O/sparse_f32.err; echo "sparse f32 rc=$?"
  bash scripts/pmc_traffic.sh $TAG/pmc_sparse python3 $R/scripts/bench_sparse.py --precision f64 --no-cpu-baseline > $O/pmc_sparse.txt 2>&1
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sparse -o sparse -- python3 $R/scripts/bench_sparse.py --no-cpu-baseline > $O/kt_sparse.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_transform -o transform -- python3 $R/bench.py --workload transform --no-cpu-baseline > $O/kt_transform.log 2>&1
  cd $R
  rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*agent_info.csv") $(find $O -name "*counter_collection.csv")
  # the monitor's cost: the default bench against the same loop without monitor launches, interleaved
  : > $O/ab_monitor.txt
  for rep in 1 2 3; do
    for mon in 1 0; do
      KLNMF_DEV=1 KLNMF_Q8_MONITOR=$mon timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-16bit-segment > $O/ab_mon.json 2>/dev/null
      python3 -c "
import json,sys; d=json.loads(open('$O/ab_mon.json').read().strip().splitlines()[-1]); print('monitor=$mon  %.2f it/s  %.4f ms/iteration  segments %s' % (d['value'], d['ms_per_step'], ' '.join('%.4f' % s for s in d['segments_ms_per_step'])))" >> $O/ab_monitor.txt
    done
  done
  cat $O/ab_monitor.txt
  ;;
4)
  timeout -k 10 900 python3 scripts/monitor_calibration.py --quick > $O/calibration.txt 2>&1; echo "calibration rc=$?"
  for c in "C2 kind" "C4 kind" "C5 kind" "k = 130 (fp8" "x 256, k = 8, 37" "rank 16" "k = 2: 40 000 x 256" "rank 8 data" "rank 12 data, k = 200"; do
    timeout -k 10 300 python3 scripts/monitor_calibration.py --quick --iters 150 --only "$c" 2>&1 | grep -v "^class" >> $O/calibration_150_iterations.txt
  done
  cut -c1-200 $O/calibration.txt | tail -80; cut -c1-200 $O/calibration_150_iterations.txt
  ;;
5)
  timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  timeout -k 10 300 python3 bench.py --workload transform > $O/bench_transform.json 2> $O/bench_transform.err; echo "transform rc=$?"
  timeout -k 10 300 python3 scripts/bench_sparse.py --precision f64 > $O/sparse_f64.json 2> $O/sparse_f64.err; echo "sparse f64 rc=$?"
  ;;
esac
