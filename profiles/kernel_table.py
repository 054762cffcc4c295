#!/usr/bin/env python3
"""Per-kernel, per-SHAPE evidence table from rocprofv3 passes of one bench.py command (eager, --no-graph):

    python profiles/kernel_table.py OUT_PREFIX --trace DIR [--fetch DIR] [--write DIR] [--sq DIR] [--top N]

  --trace : `rocprofv3 --kernel-trace --output-format csv` pass (no counters): launch durations
  --fetch : `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` pass         (own pass: TCC budget)
  --write : `rocprofv3 --pmc WRITE_SIZE --kernel-trace ...` pass
  --sq    : `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
             SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace ...` pass

A row = one (kernel instantiation, grid size) pair, i.e. one launch SHAPE (the dominant conv label alone covers 16 .. 292 us
launches).  Columns, all per launch (mean over the launches of that shape in the pass):

  us           duration from the counter-free pass (profiled passes run 2-3 % slower: never mix them, MI355X_MICROARCH.md DVFS item 2)
  hbm_mb       FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024     (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)
  hbm_gbs      hbm_mb / us ; hbm_frac = hbm_gbs / 8000 (HBM3E spec peak)
  mfma_util    SQ_VALU_MFMA_BUSY_CYCLES / (us x 2.4 GHz x 1024 SIMDs): the counter is in shader cycles summed over the SIMDs
               (= 32 per v_mfma_f32_32x32x16_bf16), so this is the fraction of the chip's matrix-pipe cycles at the NOMINAL clock
               that carried an MFMA = achieved / peak for bf16 (2.5 PFLOP/s dense = 1024 SIMDs x 1024 FLOP/clk x 2.4 GHz)
  clk_ghz      GRBM_GUI_ACTIVE / 8 XCDs / duration of the SAME (profiled) pass: the clock the chip held (reads high below ~0.3 ms)
  wait/stall/active  SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES
"""
import csv
import glob
import json
import os
import re
import sys

PEAK_HBM_GBS = 8000.0
NOMINAL_HZ = 2.4e9
SIMDS = 1024


_DEMANGLED = {}


def short(name):
    """readable kernel name: demangle (llvm-cxxfilt, when present), drop the anonymous namespace and the argument list"""
    if name in _DEMANGLED:
        return _DEMANGLED[name]
    out = name
    if name.startswith("_Z"):
        import subprocess
        for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
            try:
                # (binutils does not know the bf16 mangling DF16b: hand it the vendor-type spelling)
                out = subprocess.run([tool, name.replace("DF16b", "u6__bf16")], capture_output=True, text=True, timeout=10).stdout.strip() or name
                break
            except (OSError, subprocess.SubprocessError):
                continue
    out = out.replace("(anonymous namespace)::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(out):       # cut the argument list: the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            out = out[:i]
            break
    out = out.replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16").replace(" ", "")
    out = re.sub(r"^_ZN\d+_GLOBAL__N_1", "", out)
    _DEMANGLED[name] = out[:110]
    return _DEMANGLED[name]


def read_rows(d, pattern):
    rows = []
    for path in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(path) as f:
            rows += list(csv.DictReader(f))
    return rows


def key(row):
    return (row["Kernel_Name"], int(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0))


def durations(d):
    acc = {}
    for r in read_rows(d, "*kernel_trace.csv"):
        k = key(r)
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def counters(d):
    """{(kernel, grid): {counter: mean per launch}} ; also the profiled pass's own mean duration under '_us'"""
    acc = {}
    seen = {}
    for r in read_rows(d, "*counter_collection.csv"):
        k = key(r)
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        if "Start_Timestamp" in r and r.get("Start_Timestamp"):
            seen[(k, r.get("Dispatch_Id"))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    out = {k: {c: s / n for c, (n, s) in cs.items()} for k, cs in acc.items()}
    if not seen:
        for r in read_rows(d, "*kernel_trace.csv"):
            seen[(key(r), r.get("Dispatch_Id"))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    dur = {}
    for (k, _), us in seen.items():
        a = dur.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += us
    for k, (n, s) in dur.items():
        if k in out:
            out[k]["_us"] = s / n
    return out


def main():
    args = sys.argv[1:]
    prefix = args[0]
    opt = {}
    i = 1
    while i < len(args):
        opt[args[i].lstrip("-")] = args[i + 1]
        i += 2
    dur = durations(opt["trace"])
    fetch = counters(opt["fetch"]) if "fetch" in opt else {}
    write = counters(opt["write"]) if "write" in opt else {}
    sq = counters(opt["sq"]) if "sq" in opt else {}
    total_us = sum(n * us for n, us in dur.values())
    rows = []
    for k, (n, us) in dur.items():
        r = {"kernel": short(k[0]), "grid": k[1], "launches": n, "us": round(us, 2), "share": round(n * us / total_us, 4)}
        f, w = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
        if f is not None and w is not None:
            mb = (f * 1024 * 2 + w * 1024) / 1e6
            r.update(fetch_mb=round(f * 1024 * 2 / 1e6, 2), write_mb=round(w * 1024 / 1e6, 2), hbm_mb=round(mb, 2),
                     hbm_gbs=round(mb / us * 1e3, 1), hbm_frac=round(mb / us * 1e3 / PEAK_HBM_GBS, 4))
        s = sq.get(k)
        if s and "SQ_VALU_MFMA_BUSY_CYCLES" in s:
            r["mfma_util"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * 1e-6 * NOMINAL_HZ * SIMDS), 4)
            if "GRBM_GUI_ACTIVE" in s and "_us" in s:
                r["clk_ghz"] = round(s["GRBM_GUI_ACTIVE"] / 8.0 / (s["_us"] * 1e-6) / 1e9, 3)
            wc = s.get("SQ_WAVE_CYCLES")
            if wc:
                for name, c in (("wait", "SQ_WAIT_ANY"), ("stall", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY")):
                    if c in s:
                        r[name] = round(s[c] / wc, 3)
        rows.append(r)
    rows.sort(key=lambda r: -r["share"])
    with open(prefix + ".json", "w") as f:
        json.dump({"total_us_per_pass": round(total_us, 1), "rows": rows,
                   "columns": "see profiles/kernel_table.py; mfma_util is against the nominal 2.4 GHz x 1024 SIMD matrix-pipe cycles "
                              "(= fraction of the 2.5 PFLOP/s dense bf16 peak for bf16 MFMAs), hbm_frac against 8 TB/s"}, f, indent=1)
    top = int(opt.get("top", 40))
    cols = ["kernel", "grid", "launches", "us", "share", "hbm_mb", "hbm_gbs", "hbm_frac", "mfma_util", "clk_ghz", "wait", "stall", "active"]
    with open(prefix + ".md", "w") as f:
        f.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
        for r in rows[:top]:
            f.write("| " + " | ".join(str(r.get(c, "")) for c in cols) + " |\n")
    print("wrote %s.json / .md (%d shapes, %.1f us of kernels per pass)" % (prefix, len(rows), total_us))


if __name__ == "__main__":
    main()
