#!/bin/bash
# same-box A/B of an environment knob: $1 = knob name; extra args go to bench.py
K=$1; shift
for r in 1 2; do
  echo "== default"; timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" || exit 1
  echo "== $K=1"; env $K=1 timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline --allow-knobs --steps 30 --warmup 5 "$@" 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" || exit 1
done
