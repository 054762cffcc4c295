#!/bin/bash
# same-box A/B of two builds of the library: train (unet, unet2) and inference steps
for r in 1 2; do
for v in head new; do
  cp scratch/lib_$v.so abc-net_amd/libabcnet_hip.so
  for m in "" "--variant unet2" "--mode infer" "--mode infer --dtype fp8"; do
    echo "== $v $m"; timeout -k 10 300 python bench.py $m --no-profile --no-cpu-baseline --steps 30 --warmup 5 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done
done
cp scratch/lib_new.so abc-net_amd/libabcnet_hip.so
