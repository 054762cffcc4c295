for r in 1 2; do
for v in "" 1; do
  echo "== nodual=$v"; ABC_BENCH_NODUAL=$v ABC_BENCH_TOP=40 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step']); print({k:v for k,v in d['kernel_breakdown_ms'].items() if 'wgrad' in k or 'bn' in k})"
done
done
