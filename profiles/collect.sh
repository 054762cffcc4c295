set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-profile"
for v in unet unet2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -- $B --variant $v --steps 5 --warmup 2 > $O/stats_$v.log 2>&1
  echo stats $v done
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${v}_$c -- $B --variant $v --steps 2 --warmup 1 --no-graph > $O/pmc_${v}_$c.log 2>&1
    echo pmc $v $c done
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_infer -- $B --mode infer --steps 5 --warmup 2 > $O/stats_infer.log 2>&1
echo stats infer done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_infer_$c -- $B --mode infer --steps 2 --warmup 1 --no-graph > $O/pmc_infer_$c.log 2>&1
  echo pmc infer $c done
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_unet_SQ -- $B --steps 2 --warmup 1 --no-graph > $O/pmc_unet_SQ.log 2>&1
echo pmc SQ done
cd $R
python3 profiles/summarise_pmc.py $O/pmc_unet.json $O/pmc_unet_FETCH_SIZE $O/pmc_unet_WRITE_SIZE $O/pmc_unet_SQ
python3 profiles/summarise_pmc.py $O/pmc_unet2.json $O/pmc_unet2_FETCH_SIZE $O/pmc_unet2_WRITE_SIZE
python3 profiles/summarise_pmc.py $O/pmc_infer.json $O/pmc_infer_FETCH_SIZE $O/pmc_infer_WRITE_SIZE
find $O -name "*kernel_stats.csv" | while read f; do cp $f $O/$(echo $f | sed "s|$O/||; s|/.*||")_kernel_stats.csv; done
# keep the merge small: drop the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O
python3 bench.py --steps 20 --warmup 3 > $O/bench_unet.json 2> $O/bench_unet.err
python3 bench.py --variant unet2 --steps 20 --warmup 3 > $O/bench_unet2.json 2> $O/bench_unet2.err
python3 bench.py --mode infer --steps 10 --warmup 3 > $O/bench_infer.json 2> $O/bench_infer.err
