#!/bin/bash
# same-box A/B of several builds: scratch/lib_$v.so for v in "$@"
for r in 1 2; do
for v in "$@"; do
  cp scratch/lib_$v.so abc-net_amd/libabcnet_hip.so
  echo "== $v"; timeout -k 10 300 python bench.py --no-profile --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done
done
for v in "$@"; do
cp scratch/lib_$v.so abc-net_amd/libabcnet_hip.so
echo "== $v breakdown"
ABC_BENCH_TOP=40 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step']); print({k:v for k,v in d['kernel_breakdown_ms'].items() if 'wgrad' in k})"
done
