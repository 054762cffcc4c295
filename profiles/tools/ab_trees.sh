# same-box A/B of two TREES: build the older commit beside this one first (git archive <commit> | tar -x -C scratch/r02tree; cd scratch/r02tree; ./build_hip.sh), then run on the GPU box
# same-box A/B of the round-2 final tree (scratch/r02tree, git ffd0c80) against the working tree
mkdir -p gpurun_out/r3i
for rep in 1 2; do
  for t in r02 r03; do
    if [ $t = r02 ]; then D=scratch/r02tree; else D=.; fi
    for w in "" "--variant unet2" "--mode infer"; do
      (cd $D && python bench.py $w --no-cpu-baseline --no-profile --steps 40 --warmup 5 2>/dev/null) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t rep$rep', '$w' or 'unet', d['value'], d['ms_per_step'])"
    done
  done
done
