#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in old new old new; do
  cp $R/scratch/lib_$v.so $R/abc-net_amd/libabcnet_hip.so
  rm -rf /tmp/pp_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$v -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 10 --warmup 2 > /tmp/pp_$v.log 2>&1
  f=$(find /tmp/pp_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "pack_batch_kernel" in r["Name"] or "heads_fused_kernel" in r["Name"] and "pack" not in r["Name"]:
        print(sys.argv[2], r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
cp $R/scratch/lib_new.so $R/abc-net_amd/libabcnet_hip.so
