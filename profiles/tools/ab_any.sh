#!/bin/bash
# same-box A/B of several builds on any bench.py configuration: AB_ARGS="--variant unet2" bash profiles/tools/ab_any.sh base new
# (scratch/lib_$v.so for v in "$@"; two alternating rounds, then the library in the tree is restored to the LAST one)
for r in 1 2; do
for v in "$@"; do
  cp scratch/lib_$v.so abc-net_amd/libabcnet_hip.so
  echo "== $v"; timeout -k 10 300 python bench.py $AB_ARGS --no-profile --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done
done
