# per-dispatch durations of the wgrad launches of one step: bash scratch/wg_trace.sh <tag> [env...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wgt_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-other-configs --allow-knobs --steps 4 --warmup 2 "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/raw/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last pack_batch kernel on
idx = [i for i, r in enumerate(rows) if "pack_batch_kernel" in r["Kernel_Name"]]
last = rows[idx[-1]:]
tot = 0
for r in last:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    if "${FILTER:-wgrad}" in n or "${FILTER:-wgrad}" == "all":
        tot += d
        print("%7.1f us  grid %-8s %s" % (d, r.get("Grid_Size_X", "?"), n.replace("(anonymous namespace)::", "")[:110]))
print("wgrad total of the step: %.1f us; step kernels: %.1f us" % (tot, sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in last)))
PY
rm -rf $O/raw
