mkdir -p gpurun_out/r05p
timeout -k 5 60 ./experiments/sr_probe/probe_sr.bin | tail -12 > gpurun_out/r05p/probe_sat.txt; cat gpurun_out/r05p/probe_sat.txt
for c in "C4 kind" "C5 kind" "k = 130 (fp8" "rank 12 data"; do
  python scripts/monitor_calibration.py --quick --iters 150 --only "$c" 2>&1 | grep -v "^class" | cut -c1-215 >> gpurun_out/r05p/calib_sr2_long.txt
done
cat gpurun_out/r05p/calib_sr2_long.txt
for c in "constant columns" "sparse" "spikes"; do
  python scripts/monitor_calibration.py --quick --only "$c" 2>&1 | grep -v "^class" | cut -c1-215 >> gpurun_out/r05p/calib_sr2_quick.txt
done
cat gpurun_out/r05p/calib_sr2_quick.txt
