#!/usr/bin/env python3
"""Per-kernel, per-SHAPE evidence table from rocprofv3 passes of one bench.py command (eager, --no-graph):

    python profiles/kernel_table.py OUT_PREFIX --trace DIR [--fetch DIR] [--write DIR] [--sq DIR] [--top N]

  --trace : `rocprofv3 --kernel-trace --output-format csv` pass (no counters): launch durations
  --fetch : `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` pass         (own pass: TCC budget)
  --write : `rocprofv3 --pmc WRITE_SIZE --kernel-trace ...` pass
  --sq    : `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
             SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace ...` pass

A row = one launch SHAPE: the launches of one (kernel instantiation, grid size) pair whose durations lie within 15 % of each
other, told apart by their position in the step's launch sequence (the dominant conv label alone covers 16 .. 340 us
launches, and the persistent kernels use one grid size for every layer).  Columns, all per launch (mean over the launches of
that shape in the pass):

  us           duration from the counter-free pass (profiled passes run 2-3 % slower: never mix them, MI355X_MICROARCH.md DVFS item 2)
  hbm_mb       FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024     (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)
  hbm_gbs      hbm_mb / us ; hbm_frac = hbm_gbs / 8000 (HBM3E spec peak)
  mfma_util    SQ_VALU_MFMA_BUSY_CYCLES / (us x 2.4 GHz x 1024 SIMDs): the counter is in shader cycles summed over the SIMDs
               (= 32 per v_mfma_f32_32x32x16_bf16), so this is the fraction of the chip's matrix-pipe cycles at the NOMINAL clock
               that carried an MFMA = achieved / peak for bf16 (2.5 PFLOP/s dense = 1024 SIMDs x 1024 FLOP/clk x 2.4 GHz)
  mfma_util_held  the same counter against the cycles at the clock the chip HELD (clk_ghz) instead of the nominal 2.4 GHz
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
    out = out.replace("(anonymous namespace)::", "").replace("abc_cf::", "").replace("void ", "")
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
    # total work-items: the counter files carry `Grid_Size` (all dimensions), the kernel trace one column per dimension -- with the
    # X dimension alone the rows of every kernel launched on a 2-D / 3-D grid (the fused heads pass, the batched heads kernels,
    # CBAM) found no counters
    if row.get("Grid_Size") not in (None, ""):
        g = int(row["Grid_Size"])
    else:
        g = int(row.get("Grid_Size_X", 0) or 0) * max(int(row.get("Grid_Size_Y", 1) or 1), 1) * max(int(row.get("Grid_Size_Z", 1) or 1), 1)
    return (row["Kernel_Name"], g)


def steps_in(rows):
    """number of steps a pass ran: launches of the once-per-step Adam kernel (train) or NMS kernel (inference)"""
    for probe in ("adam_kernel", "nms_kernel"):
        n = sum(1 for r in rows if probe in r["Kernel_Name"])
        if n:
            return n
    return 1


def by_ordinal(rows, value):
    """{(kernel, grid): {ordinal within a step: [n, sum]}}: the launch sequence of a step is the same in every pass, so the
    i-th launch of a (kernel, grid) pair within a step is the same layer everywhere -- that is what separates the SHAPES a
    persistent kernel (one grid size for every layer) is launched on"""
    rows = sorted(rows, key=lambda r: int(r.get("Start_Timestamp") or r.get("Dispatch_Id") or 0))
    nsteps = steps_in(rows)
    seq = {}
    for r in rows:
        seq.setdefault(key(r), []).append(r)
    out = {}
    for k, rs in seq.items():
        # launches before the first step (an inference run's calibration pass runs the same kernels once more): when the count
        # is not a whole number of steps, the surplus launches are the EARLIEST ones -- dropped from the per-step statistics
        extra = len(rs) % nsteps
        if extra and len(rs) > nsteps:
            rs = rs[extra:]
        per = max(len(rs) // nsteps, 1) if len(rs) % nsteps == 0 else len(rs)   # (setup-time launches: one ordinal each)
        o = out.setdefault(k, {})
        for i, r in enumerate(rs):
            a = o.setdefault(i % per, [0, 0.0])
            a[0] += 1
            a[1] += value(r)
    return out


def dur_us(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0


def shape_clusters(ordinals):
    """ordinals {ord: [n, sum_us]} -> list of ordinal lists: launches whose mean durations lie within 15 % of each other"""
    items = sorted((s / n, o) for o, (n, s) in ordinals.items())
    clusters, first = [], None
    for us, o in items:
        if first is None or us > 1.15 * first:
            clusters.append([])
            first = us
        clusters[-1].append(o)
    return clusters


def main():
    args = sys.argv[1:]
    prefix = args[0]
    opt = {}
    i = 1
    while i < len(args):
        opt[args[i].lstrip("-")] = args[i + 1]
        i += 2
    trace = by_ordinal(read_rows(opt["trace"], "*kernel_trace.csv"), dur_us)
    passes = {}
    for name, counters_wanted in (("fetch", ("FETCH_SIZE",)), ("write", ("WRITE_SIZE",)),
                                  ("sq", ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                          "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"))):
        if name not in opt:
            continue
        rows = read_rows(opt[name], "*counter_collection.csv")
        ktr = {r.get("Dispatch_Id"): r for r in read_rows(opt[name], "*kernel_trace.csv")}
        for c in counters_wanted:
            sel = [r for r in rows if r["Counter_Name"] == c]
            for r in sel:   # timestamps for the ordering (and the pass's own duration) come from its kernel trace
                t = ktr.get(r.get("Dispatch_Id"))
                if t is not None:
                    r["Start_Timestamp"], r["End_Timestamp"] = t["Start_Timestamp"], t["End_Timestamp"]
            passes[c] = by_ordinal(sel, lambda r: float(r["Counter_Value"]))
            if c == "GRBM_GUI_ACTIVE":
                passes["_sq_us"] = by_ordinal(sel, dur_us)
    # SETUP launches are not step work: torch's fills / copies of the engine's buffer allocations, the one-off weight packing
    # of an inference run, calibration.  They are listed (role = "setup", launches counted per PASS) but kept out of the shares.
    def is_setup(name, nlaunch, nsteps):
        # (torch's own kernels, and anything launched less than once per step: packing / folding / calibration of an inference run)
        return name.startswith(("at::native::", "__amd_rocclr_", "void at::native::")) or "FillFunctor" in name or nlaunch < max(nsteps, 1)

    nsteps = steps_in(read_rows(opt["trace"], "*kernel_trace.csv"))
    setup_keys = set(k for k, o in trace.items() if is_setup(short(k[0]), sum(n for n, _s in o.values()), nsteps))
    total_us = sum(s for k, o in trace.items() if k not in setup_keys for n, s in o.values())
    rows = []
    for k, ords in trace.items():
        for cl in shape_clusters(ords):
            n = sum(ords[o][0] for o in cl)
            us = sum(ords[o][1] for o in cl) / n
            setup = k in setup_keys
            r = {"kernel": short(k[0]), "grid": k[1], "role": "setup" if setup else "step", "launches": n,
                 "launches_per_step": (None if setup else len(cl)), "us": round(us, 2), "share": (None if setup else round(n * us / total_us, 4))}

            def mean(counter):
                p = passes.get(counter, {}).get(k)
                if not p:
                    return None
                tot = [p[o] for o in cl if o in p]
                nn = sum(t[0] for t in tot)
                return sum(t[1] for t in tot) / nn if nn else None

            f, w = mean("FETCH_SIZE"), mean("WRITE_SIZE")
            if f is not None and w is not None:
                mb = (f * 1024 * 2 + w * 1024) / 1e6
                r.update(fetch_mb=round(f * 1024 * 2 / 1e6, 2), write_mb=round(w * 1024 / 1e6, 2), hbm_mb=round(mb, 2),
                         hbm_gbs=round(mb / us * 1e3, 1), hbm_frac=round(mb / us * 1e3 / PEAK_HBM_GBS, 4))
            mf = mean("SQ_VALU_MFMA_BUSY_CYCLES")
            if mf is not None:
                r["mfma_util"] = round(mf / (us * 1e-6 * NOMINAL_HZ * SIMDS), 4)
                gui, squs = mean("GRBM_GUI_ACTIVE"), mean("_sq_us")
                if gui is not None and squs:
                    r["clk_ghz"] = round(gui / 8.0 / (squs * 1e-6) / 1e9, 3)
                    # against the matrix-pipe cycles the chip actually offered at the clock it HELD during this launch (DVFS: the
                    # long MFMA-dense launches hold 1.7-2.1 GHz): the utilisation a kernel change can still move.  Only where the
                    # clock estimate is sane (short launches read high, see clk_ghz).
                    if 1.2e9 <= r["clk_ghz"] * 1e9 <= 2.6e9:
                        r["mfma_util_held"] = round(mf / (us * 1e-6 * r["clk_ghz"] * 1e9 * SIMDS), 4)
                wc = mean("SQ_WAVE_CYCLES")
                if wc:
                    for name, c in (("wait", "SQ_WAIT_ANY"), ("stall", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY")):
                        v = mean(c)
                        if v is not None:
                            r[name] = round(v / wc, 3)
            rows.append(r)
    rows.sort(key=lambda r: (r["role"] == "setup", -(r["share"] or 0.0)))
    with open(prefix + ".json", "w") as f:
        json.dump({"total_us_per_pass": round(total_us, 1), "rows": rows,
                   "columns": "see profiles/kernel_table.py; a row = the launches of one kernel instantiation and grid whose durations lie "
                              "within 15 % (one launch SHAPE); mfma_util is against the nominal 2.4 GHz x 1024 SIMD matrix-pipe cycles "
                              "(= fraction of the 2.5 PFLOP/s dense bf16 peak for bf16 MFMAs), hbm_frac against 8 TB/s"}, f, indent=1)
    top = int(opt.get("top", 48))
    cols = ["kernel", "grid", "role", "launches_per_step", "us", "share", "hbm_mb", "hbm_gbs", "hbm_frac", "mfma_util", "mfma_util_held", "clk_ghz", "wait", "stall", "active"]
    with open(prefix + ".md", "w") as f:
        f.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
        for r in rows[:top]:
            f.write("| " + " | ".join(str(r.get(c, "")) for c in cols) + " |\n")
    print("wrote %s.json / .md (%d shapes, %.1f us of kernels per pass)" % (prefix, len(rows), total_us))


if __name__ == "__main__":
    main()
