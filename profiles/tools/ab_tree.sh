#!/bin/bash
run() { ( cd $1 && timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline --steps 30 --warmup 5 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" ); }
for r in 1 2 3; do
  for t in scratch/wt_old2 scratch/wt_old .; do echo "== $t"; run $t || exit 1; done
done
