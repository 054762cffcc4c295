#!/bin/bash
set -e
for r in 1 2 3; do
for v in old new; do
  cp scratch/lib_$v.so abc-net_amd/libabcnet_hip.so
  t=$(timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline --steps 60 --warmup 10 2>&1 | python -c "import sys,json; [print(json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]")
  i=$(timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline --mode infer --steps 30 --warmup 5 2>&1 | python -c "import sys,json; [print(json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]")
  echo "$v train $t infer $i"
done
done
cp scratch/lib_new.so abc-net_amd/libabcnet_hip.so
