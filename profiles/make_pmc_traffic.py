#!/usr/bin/env python3
"""profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from the per-shape kernel tables of one round:

    python profiles/make_pmc_traffic.py TAG [TAG2 ...]   (reads profiles/TAG_{unet,unet2,infer,infer8}_kernel_table.json and
                                                    profiles/TAG_{..}_bench.json, rewrites profiles/pmc_traffic.json)

Per bench.py kernel label (one kernel instantiation; its launches of all shapes): mean HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE passes (corrected as profiles/kernel_table.py says) beside the ALGORITHMIC bytes per launch of the
engine's own plan (bench.py `kernel_algorithmic`), and their ratio."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
T = {"bf16": "bf16", "float": "f32", "f8": "fp8"}


def label_of(kernel):
    m = re.match(r"(conv_fast|conv_igemm)_kernel<(\w+),(\w+),(\w+),(\d+),(\d+),(\d+),(\d+)", kernel)
    if m:
        k, a, b, c, ck, bn, s, mt = m.groups()
        return "%s<%s,%s,%s,CK%s,BN%s,S%s,MT%s>" % (k, T[a], T[b], T[c], ck, bn, s, mt)
    m = re.match(r"wgrad_kernel<(\w+),(\w+),(\w+),(\d+),(\d+),(\d+)", kernel)
    if m:
        a, b, c, at, bt, s = m.groups()
        return "wgrad<%s,%s,%s,%sx%s,S%s>" % (T[a], T[b], T[c], at, bt, s)
    return None


def main(tags):
    """several tags: a later tag's tables replace an earlier one's entries workload by workload (e.g. `r04b r04c`: the training
    workloads re-collected after the last kernel change, the inference ones from the full set)"""
    out = {}
    for tag in tags:
        collect(tag, out)
    with open(os.path.join(HERE, "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote pmc_traffic.json with %d entries" % len(out))


def collect(tag, out):
    # (the fp8 graph keeps bf16 launches of the same instantiations as the bf16 graph: its own key space, "infer8:unet:...")
    for mode, variant, w in (("train", "unet", "unet"), ("train", "unet2", "unet2"), ("infer", "unet", "infer"), ("infer8", "unet", "infer8")):
        tp = os.path.join(HERE, "%s_%s_kernel_table.json" % (tag, w))
        bp = os.path.join(HERE, "%s_%s_bench.json" % (tag, w))
        if not os.path.exists(tp):
            continue
        rows = json.load(open(tp))["rows"]
        algo, pure = {}, {}
        if os.path.exists(bp):
            b = json.load(open(bp))
            calls = b.get("kernel_calls", {})
            # a convolution kernel with the producer's act_bwd pass in its epilogue ("conv_fast+act_bwd<...>") is an instantiation of
            # the same kernel family: one entry, the call-weighted mean of both labels (as bench.py's roofline row)
            fam = {}
            for k, (gf, mb) in b.get("kernel_algorithmic", {}).items():
                n = calls.get(k, 1.0)
                f = fam.setdefault(k.replace("+act_bwd<", "<"), [0.0, 0.0, 0.0])
                f[0] += n; f[1] += n * gf; f[2] += n * mb
            algo = {k: [f[1] / f[0], f[2] / f[0]] for k, f in fam.items() if f[0] > 0}
            pf = {}
            for k, mb in b.get("kernel_operand_mb", {}).items():
                n = calls.get(k, 1.0)
                f = pf.setdefault(k.replace("+act_bwd<", "<"), [0.0, 0.0])
                f[0] += n; f[1] += n * mb
            pure = {k: f[1] / f[0] for k, f in pf.items() if f[0] > 0}
        acc = {}
        for r in rows:
            lab = label_of(r["kernel"])
            if lab is None or "hbm_mb" not in r:
                continue
            a = acc.setdefault(lab, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += r["launches"]
            a[1] += r["launches"] * r["hbm_mb"] * 1e6
            a[2] += r["launches"] * r["fetch_mb"] * 1e6
            a[3] += r["launches"] * r["write_mb"] * 1e6
            a[4] += r["launches"] * r["us"]
        for lab, (n, by, fb, wb, us) in acc.items():
            e = {"bytes_per_launch": int(by / n), "fetch_bytes": int(fb / n), "write_bytes": int(wb / n), "launches_in_pass": n,
                 "avg_launch_us": round(us / n, 2), "achieved_hbm_gbs": round(by / us / 1e3, 1),
                 "source": "profiles/%s_%s_kernel_table.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate --kernel-trace-only "
                           "passes; FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024; mean over every launch of the instantiation)" % (tag, w)}
            if lab in pure:
                # the OPERANDS alone (the plan's accounting also counts what the chosen algorithm adds: split-K slabs, y_raw / dY of the
                # BatchNorm-fused weight gradient): bench.py `kernel_operand_mb`
                e["operand_bytes_per_launch"] = int(pure[lab] * 1e6)
                e["traffic_over_operands"] = round(by / n / (pure[lab] * 1e6), 3) if pure[lab] else None
            if lab in algo:
                e["algorithmic_bytes_per_launch"] = int(algo[lab][1] * 1e6)
                e["traffic_over_algorithmic"] = round(by / n / (algo[lab][1] * 1e6), 3) if algo[lab][1] else None
                e["algorithmic_gflop_per_launch"] = algo[lab][0]
            out["%s:%s:%s" % (mode, variant, lab)] = e


if __name__ == "__main__":
    main(sys.argv[1:])
