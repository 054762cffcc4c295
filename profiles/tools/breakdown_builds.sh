#!/bin/bash
# per-kernel breakdown lines matching a pattern for several builds: scratch/bd.sh PATTERN lib1 lib2 ...
pat=$1; shift
for v in "$@"; do
cp scratch/lib_$v.so abc-net_amd/libabcnet_hip.so
echo "== $v"
ABC_BENCH_TOP=60 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json,re
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step']); print({k:v for k,v in d['kernel_breakdown_ms'].items() if re.search('$pat',k)})"
done
