#!/bin/bash
# Issue-slot / wait counters of one kernel family in the train step (eager launches under rocprofv3 --pmc, one pass per counter set):
#   bash profiles/tools/pmc_kernel.sh 'conv_fast_kernel.*Li128ELi1ELi6' [bench args]   -> gpurun_out/pmc_kernel/summary.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_kernel
PAT=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 2 --warmup 1 --no-graph "$@" > $O/p$i.log 2>&1
done
python3 - <<PY > $O/summary.txt
import csv, glob, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not re.search(r"""$PAT""", k): continue
        acc[(k[:160], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k,g),c in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n=len(next(iter(c.values())))
    print(k, "grid", g, "launches", n)
    for cn,vs in sorted(c.items()): print("    %-34s %14.0f" % (cn, sum(vs)/len(vs)))
PY
rm -rf $O/p*/
cat $O/summary.txt | head -120
