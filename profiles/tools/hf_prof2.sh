# counters of heads_fused_kernel for one debug mask: bash scratch/hf_prof2.sh <dbg>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hfprof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ABC_TOOL_LIB=$R/scratch/lib_dbg.so
cd $R
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/profiles/tools/hf_time.py $1 > $O/sq.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/profiles/tools/hf_time.py $1 > $O/sq2.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq3 -- python3 $R/profiles/tools/hf_time.py $1 > $O/sq3.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("sq","sq2","sq3"):
    acc=collections.defaultdict(lambda: [0,0.0])
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "heads_fused_kernel" in r["Kernel_Name"]:
                k=r["Counter_Name"]
                acc[k][0]+=1; acc[k][1]+=float(r["Counter_Value"])
    for k,(n,s) in sorted(acc.items()): print(d, k, n, s/n)
PY
rm -rf $O/sq $O/sq2 $O/sq3
