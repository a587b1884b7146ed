#!/bin/bash
# After scripts/pmc_profile.sh <tag>: keep only the small summaries under gpurun_out/ (the raw CSVs exceed the
# 64 MiB that gpurun copies back).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc}
mkdir -p $R/gpurun_out/keep
cp $R/gpurun_out/$TAG/summary.txt $R/gpurun_out/$TAG/traffic.json $R/gpurun_out/keep/ 2>/dev/null
rm -rf $R/gpurun_out/$TAG
