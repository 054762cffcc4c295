#!/bin/bash
run() { python bench.py --no-profile --no-cpu-baseline --allow-knobs --steps 30 --warmup 5 "$@" 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; }
for r in 1 2; do
  for m in 1 0 2; do echo "== pingpong=$m"; ABC_WGRAD_PINGPONG=$m timeout -k 10 200 bash -c "$(declare -f run); run $*" || exit 1; done
done
