#!/usr/bin/env python3
"""Summarise rocprofv3 counter-collection output into the per-kernel tables kept in this directory.

  python profiles/summarise_pmc.py OUT.json DIR [DIR ...]          per-kernel mean of every counter found under the DIRs
  python profiles/summarise_pmc.py --traffic OUT.json SUMMARY.json LABEL=ALGO_BYTES [...]
                                                                    HBM bytes per launch for bench.py's kernel labels

A DIR is one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -- python3 bench.py ...` pass (FETCH_SIZE
and WRITE_SIZE need separate passes: they do not fit the TCC counter budget together).  Traffic follows
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KiB and counts HALF of the bytes of 16-byte-per-lane
streaming reads on gfx950 -> bytes = FETCH_SIZE x 1024 x 2; WRITE_SIZE KiB x 1024 as is.
"""
import csv
import glob
import json
import os
import re
import sys

TYPES = {"bf16": "DF16b", "f32": "f"}


def mangled_fragment(label):
    """bench.py kernel label -> substring of the mangled kernel name"""
    m = re.match(r"(conv_fast|conv_igemm)<(\w+),(\w+),(\w+),CK(\d+),BN(\d+),S(\d+),MT(\d+)>", label)
    if m:
        k, a, b, c, ck, bn, s, mt = m.groups()
        return "%s_kernelI%s%s%sLi%sELi%sELi%sELi%sE" % (k, TYPES[a], TYPES[b], TYPES[c], ck, bn, s, mt)
    m = re.match(r"wgrad<(\w+),(\w+),(\w+),(\d+)x(\d+),S(\d+)>", label)
    if m:
        a, b, c, at, bt, s = m.groups()
        return "wgrad_kernelI%s%s%sLi%sELi%sELi%sE" % (TYPES[a], TYPES[b], TYPES[c], at, bt, s)
    return label


def summarise(dirs):
    acc = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], [0, 0.0])
                    k[0] += 1
                    k[1] += float(row["Counter_Value"])
    return [dict(kernel=name, **{c: {"n": n, "mean": s / n} for c, (n, s) in sorted(cs.items())}) for name, cs in sorted(acc.items())]


def traffic(summary, labels):
    out = {}
    for label, algo in labels.items():
        frag = mangled_fragment(label)
        n = fb = wb = 0.0
        for rec in summary:
            if frag in rec["kernel"] and "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
                k = rec["FETCH_SIZE"]["n"]
                n += k
                fb += k * rec["FETCH_SIZE"]["mean"] * 1024 * 2
                wb += rec["WRITE_SIZE"]["n"] * rec["WRITE_SIZE"]["mean"] * 1024
        if n:
            out[label] = {"bytes_per_launch": int((fb + wb) / n), "fetch_bytes": int(fb / n), "write_bytes": int(wb / n),
                          "algorithmic_bytes_per_launch": int(algo), "launches": int(n),
                          "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), mean over all "
                                    "launches of the instantiation; FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of 16 B/lane streaming "
                                    "reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024"}
    return out


if __name__ == "__main__":
    if sys.argv[1] == "--traffic":
        with open(sys.argv[3]) as f:
            summ = json.load(f)
        labels = dict((a.rsplit("=", 1)[0], float(a.rsplit("=", 1)[1])) for a in sys.argv[4:])
        old = {}
        if os.path.exists(sys.argv[2]):
            with open(sys.argv[2]) as f:
                old = json.load(f)
        old.update(traffic(summ, labels))
        with open(sys.argv[2], "w") as f:
            json.dump(old, f, indent=1)
    else:
        with open(sys.argv[1], "w") as f:
            json.dump(summarise(sys.argv[2:]), f, indent=1)
