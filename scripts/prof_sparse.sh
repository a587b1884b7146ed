#!/bin/bash
# Per-kernel time of the CSR path (scripts/bench_sparse.py); prints the top of rocprofv3's kernel_stats.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o sp -- python3 $R/scripts/bench_sparse.py > /tmp/sp.log 2>&1
grep -E "ms / iteration" /tmp/sp.log
F=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then head -14 "$F" | cut -c1-160; mkdir -p $R/gpurun_out/sp; head -20 "$F" > $R/gpurun_out/sp/kernel_stats_head.csv; fi
