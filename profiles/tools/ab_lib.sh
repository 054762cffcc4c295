#!/bin/bash
set -e
mkdir -p gpurun_out
for r in 1 2; do
for v in old new; do
  cp scratch/lib_$v.so abc-net_amd/libabcnet_hip.so
  echo "== $v train" ; timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline --steps 30 --warmup 5 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done
done
cp scratch/lib_new.so abc-net_amd/libabcnet_hip.so
