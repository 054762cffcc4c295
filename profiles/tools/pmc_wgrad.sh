#!/bin/bash
# PMC comparison of the weight-gradient kernel between builds: scratch/pmc_wg.sh head pprot
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wg
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $R/scratch/lib_$v.so $R/abc-net_amd/libabcnet_hip.so
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${v}_$tag -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 2 --warmup 1 --no-graph > $O/${v}_$tag.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections, os
O="$O"
for v in "$@".split():
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O+"/%s_*/**/*counter_collection.csv"%v, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "wgrad_kernel" not in k or "Li4ELi2E" not in k: continue
            acc[(k[:200], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k,g),c in acc.items():
        n=len(next(iter(c.values())))
        print(v, k[-60:], g, "launches", n)
        print("   ", {cn: round(sum(vs)/len(vs)/1e6,3) for cn,vs in sorted(c.items())})
PY
rm -rf $O/*/
