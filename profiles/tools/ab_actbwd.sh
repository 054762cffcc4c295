#!/bin/bash
mkdir -p gpurun_out/ab
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "act_bwd" > gpurun_out/ab/k.log 2>&1; tail -3 gpurun_out/ab/k.log
timeout -k 10 900 python -m pytest tests/test_gpu_insitu_fullsize.py tests/test_gpu_model.py -x -q > gpurun_out/ab/m.log 2>&1; tail -3 gpurun_out/ab/m.log
for r in 1 2; do
  for f in "" "--no-actbwd-epilogue"; do
    echo "== train $f"; timeout -k 10 300 python bench.py $f --no-profile --no-cpu-baseline --steps 30 --warmup 5 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done
