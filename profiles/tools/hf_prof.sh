set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hfprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/scratch/hf_time.py 0"
cd $R
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/mem -- $P > $O/mem.log 2>&1
echo fetch done
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/mem2 -- $P > $O/mem2.log 2>&1
echo write done
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- $P > $O/sq.log 2>&1
echo sq done
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -- $P > $O/sq2.log 2>&1
echo sq2 done
python3 - <<PY
import csv, glob, collections
for d in ("mem","mem2","sq","sq2"):
    acc=collections.defaultdict(lambda: [0,0.0])
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "heads_fused_kernel" in r["Kernel_Name"] or "head_wgrad_blocked" in r["Kernel_Name"]:
                k=("fused" if "heads_fused_kernel" in r["Kernel_Name"] else "wgrad", r["Counter_Name"])
                acc[k][0]+=1; acc[k][1]+=float(r["Counter_Value"])
    for k,(n,s) in sorted(acc.items()): print(d, k, n, s/n)
PY
rm -rf $O/mem $O/mem2 $O/sq $O/sq2
