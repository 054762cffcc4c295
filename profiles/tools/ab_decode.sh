for r in 1 2; do
for args in "--mode infer" "--mode infer --decode" "--mode infer --dtype fp8" "--mode infer --dtype fp8 --decode"; do
  echo "== $args"; timeout -k 10 300 python bench.py $args --no-profile --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done
done
