#!/bin/bash
# Derived latency counters (LDS / VMEM / instruction fetch / SMEM) of the hot kernels, one pass each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lat
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "LdsLatency" "VmemLatency" "InstrFetchLatency SmemLatency"; do
  tag=$(echo $grp | tr " " "_")
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/lat/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "rowpass4" if ("k_rowpass4" in n and ("7, 1, 0" in n or "ELi0E" in n)) else ("colpass" if "k_colpass" in n else None)
        if key:
            acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, "mean %.1f" % (sum(v) / len(v)), "n", len(v))
PY
