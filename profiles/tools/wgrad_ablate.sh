for d in 0 3 7 2 4; do
  ABC_WGRAD_DBG=$d ABC_BENCH_OPS=1 ABC_BENCH_TOP=600 python bench.py --allow-knobs --steps 3 --warmup 1 --no-graph --no-cpu-baseline 2>/dev/null > gpurun_out/wg_$d.json
  python - <<PY
import json
d=json.load(open('gpurun_out/wg_$d.json'))['kernel_breakdown_ms']
print('dbg=$d', ' '.join('%s=%.0f'%(k.split('| wgrad ')[1][:22], v*1000) for k,v in d.items() if 'wgrad<' in k and ('dconv1.double_conv.0' in k or 'out_modules.*.conv1' in k or 'inc2.double_conv.0' in k or 'down3.maxpool_conv.1.double_conv.3' in k)))
PY
done
