#!/bin/bash
# HBM traffic per kernel of ANY workload, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate rocprofv3
# --pmc passes (they do not fit one), kernel trace only (no other tracing domain), the program itself behind `--`.
#   scripts/pmc_traffic.sh TAG python3 /root/repo/bench.py --workload transform --steps 3 --warmup 1 --repeats 1 --data device --no-cpu-baseline
# -> gpurun_out/TAG/{fetch,write}/..., gpurun_out/TAG/traffic_by_kernel.json (scripts/pmc_traffic_summary.py)
TAG=$1
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- "$@" > $OUT/fetch.log 2>&1 && \
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- "$@" > $OUT/write.log 2>&1 && \
python3 $R/scripts/pmc_traffic_summary.py $OUT > $OUT/traffic_by_kernel.txt 2>&1
tail -30 $OUT/traffic_by_kernel.txt
