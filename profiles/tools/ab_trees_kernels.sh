# same-box A/B of two TREES: build the older commit beside this one first (git archive <commit> | tar -x -C scratch/r02tree; cd scratch/r02tree; ./build_hip.sh), then run on the GPU box
mkdir -p gpurun_out/r3i
for t in r02 r03; do
  if [ $t = r02 ]; then D=scratch/r02tree; else D=.; fi
  for w in unet unet2; do
    (cd $D && ABC_BENCH_TOP=300 python bench.py --variant $w --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null) > gpurun_out/r3i/kb_${t}_$w.json
  done
done
python - <<PY
import json
for w in ("unet","unet2"):
    a=json.load(open("gpurun_out/r3i/kb_r02_%s.json"%w)); b=json.load(open("gpurun_out/r3i/kb_r03_%s.json"%w))
    print(w, a["ms_per_step"], b["ms_per_step"], a["eager_step_ms_sum_of_kernels"], b["eager_step_ms_sum_of_kernels"])
    ka,kb=a["kernel_breakdown_ms"],b["kernel_breakdown_ms"]
    for k in sorted(set(ka)|set(kb), key=lambda k:-abs(kb.get(k,0)-ka.get(k,0)))[:14]:
        print("   %-52s %7.3f -> %7.3f  (%+.3f)"%(k, ka.get(k,0), kb.get(k,0), kb.get(k,0)-ka.get(k,0)))
PY
