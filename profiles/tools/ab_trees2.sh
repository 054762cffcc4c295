#!/bin/bash
# same-box A/B of TREES on any bench.py configuration: AB_ARGS="--variant unet2" bash profiles/tools/ab_trees2.sh scratch/wt_old .
# (each tree is a checkout with its own built library: git archive <commit> | tar -x -C scratch/wt_old; (cd scratch/wt_old; ./build_hip.sh))
run() { ( cd $1 && timeout -k 10 300 python bench.py $AB_ARGS --no-profile --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" ); }
for r in 1 2; do
  for t in "$@"; do echo "== $t $AB_ARGS"; run $t || exit 1; done
done
