set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 6 --warmup 3 > $O/tr.log 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
adam=[i for i,r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
print("adam launches", len(adam))
for a,b in zip(adam[-4:-1], adam[-3:]):
    seg=rows[a+1:b+1]
    dur=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in seg)/1e3
    span=(int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"]))/1e3
    gaps=[(int(seg[i+1]["Start_Timestamp"])-int(seg[i]["End_Timestamp"]))/1e3 for i in range(len(seg)-1)]
    pos=[g for g in gaps if g>0]
    print("kernels %d  sum %.0f us  span %.0f us  idle %.0f us  mean gap %.2f us  overlap(neg gaps) %d" % (len(seg), dur, span, sum(pos), sum(pos)/max(len(pos),1), sum(1 for g in gaps if g<0)))
    big=sorted(((g, seg[i]["Kernel_Name"][:50], seg[i+1]["Kernel_Name"][:50]) for i,g in enumerate(gaps)), reverse=True)[:8]
    for g,k1,k2 in big: print("   gap %.1f us after %s before %s" % (g,k1,k2))
PY
rm -rf $O/tr
