# Round-5 evidence: run on the MI355X box from the repo root (bash profiles/collect_r05.sh [tag]).
# Per workload: a counter-free kernel trace of the hipGraph run (durations), FETCH_SIZE, WRITE_SIZE and SQ/GRBM passes of the
# same step run eagerly (--no-graph) -- each its own rocprofv3 run with --kernel-trace only, as the pool requires -- reduced by
# profiles/kernel_table.py to one row per (kernel, launch shape); then rocprofv3 --stats on the hipGraph run and the bench lines.
set -e
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-profile --no-other-configs"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for w in ${WORKLOADS:-unet unet2 infer infer8}; do
  case $w in
    unet) A="--variant unet";; unet2) A="--variant unet2";; infer) A="--mode infer";; infer8) A="--mode infer --dtype fp8";;
    inferd) A="--mode infer --decode";; infer8d) A="--mode infer --dtype fp8 --decode";;
  esac
  # durations: the hipGraph run the benchmark times (eager launches run 10-30 % longer under the tracer)
  rocprofv3 --kernel-trace --output-format csv -d $O/${w}_trace -- $B $A --steps 6 --warmup 2 > $O/${w}_trace.log 2>&1
  echo "$w trace done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${w}_fetch -- $B $A --steps 2 --warmup 1 --no-graph > $O/${w}_fetch.log 2>&1
  echo "$w fetch done"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${w}_write -- $B $A --steps 2 --warmup 1 --no-graph > $O/${w}_write.log 2>&1
  echo "$w write done"
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/${w}_sq -- $B $A --steps 2 --warmup 1 --no-graph > $O/${w}_sq.log 2>&1
  echo "$w sq done"
  python3 $R/profiles/kernel_table.py $O/${TAG}_${w}_kernel_table --trace $O/${w}_trace --fetch $O/${w}_fetch --write $O/${w}_write --sq $O/${w}_sq
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_stats -- $B $A --steps 5 --warmup 2 > $O/${w}_stats.log 2>&1
  find $O/${w}_stats -name "*kernel_stats.csv" | while read f; do cp $f $O/${TAG}_${w}_kernel_stats.csv; done
  echo "$w stats done"
  # keep the merge small: drop the raw traces
  rm -rf $O/${w}_trace $O/${w}_fetch $O/${w}_write $O/${w}_sq $O/${w}_stats
done
cd $R
for w in ${WORKLOADS:-unet unet2 infer infer8}; do
  case $w in
    unet) A="--variant unet --steps 20 --warmup 3";; unet2) A="--variant unet2 --steps 20 --warmup 3";; infer) A="--mode infer --steps 10 --warmup 3";; infer8) A="--mode infer --dtype fp8 --steps 10 --warmup 3";;
    inferd) A="--mode infer --decode --steps 10 --warmup 3";; infer8d) A="--mode infer --dtype fp8 --decode --steps 10 --warmup 3";;
  esac
  ABC_BENCH_TOP=400 python3 bench.py $A > $O/${TAG}_${w}_bench.json 2> $O/${TAG}_${w}_bench.err
  echo "$w bench done"
done
ls -la $O
