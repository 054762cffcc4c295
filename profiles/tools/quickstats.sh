#!/bin/bash
# quick per-kernel durations of the hipGraph run: bash scratch/quickstats.sh <tag> <bench args...>
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/qs_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 5 --warmup 2 "$@" > $O/log.txt 2>&1
find $O/raw -name "*kernel_stats.csv" | while read f; do cp $f $O/kernel_stats.csv; done
rm -rf $O/raw
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
steps=7
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total us/step (incl. setup kernels):", round(tot/steps/1000,1))
for r in rows[:45]:
    n=r["Name"]
    n=n[:100]
    print("%8.1f us/step  calls/step %5.1f  avg %8.1f us  %s"%(float(r["TotalDurationNs"])/steps/1000, int(r["Calls"])/steps, float(r["AverageNs"])/1000, n))
PY
