#!/usr/bin/env python3
"""Scan device ISA (hipcc -S --cuda-device-only) for the store-data hazard seen on gfx950: a vector-memory store of
more than 8 bytes whose data registers are written by a VALU instruction within the next WINDOW instructions.

    python profiles/tools/store_hazard_scan.py file.s [...]                 every store of 3+ dwords, window 2 (a report)
    python profiles/tools/store_hazard_scan.py --uncovered file.s [...]     only the stores LLVM's hazard recogniser does NOT cover --
                                                                            buffer stores WITH a scalar offset register -- and exit 1
                                                                            on a hit: the check build_hip.sh runs on the disassembly
                                                                            of the built library (llvm-objdump -d of every gfx950
                                                                            code object in libabcnet_hip.so)

Prints kernel, line, the store and the offending instruction.  (LLVM's hazard recogniser inserts the wait states for buffer stores
WITHOUT a scalar offset register and for global / flat / scratch stores; the report mode looks at every store of 3+ dwords anyway.)
Accepts compiler output (hipcc -S) and llvm-objdump disassembly alike."""
import re
import sys

WINDOW = 2
ST = re.compile(r"\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = VREG.match(tok.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def data_regs(op, args):
    a = [t.strip() for t in args.split(",")]
    if op.startswith("buffer_store"):
        return regs(a[0])                      # vdata first
    return regs(a[1]) if len(a) > 1 else set()  # global / flat / scratch: addr, data


def dst_regs(line):
    s = line.split(";")[0].strip()
    if not s.startswith("v_") or s.startswith(("v_cmp", "v_nop", "v_readlane", "v_readfirstlane")):
        return set()
    parts = s.split(None, 1)
    if len(parts) < 2:
        return set()
    return regs(parts[1].split(",")[0])


def uncovered(op, args):
    """a buffer store whose soffset operand is an SGPR (vdata, vaddr, srsrc, soffset [modifiers])"""
    if not op.startswith("buffer_store"):
        return False
    a = [t.strip() for t in args.split("//")[0].split(";")[0].split(",")]
    # srsrc is the s[..:..] quad; soffset is the operand after it
    for i, t in enumerate(a):
        if t.startswith("s[") and i + 1 < len(a):
            so = a[i + 1].split()[0]
            return bool(re.match(r"s\d+$|m0$", so))
    return False


def main(paths):
    only_uncovered = False
    if paths and paths[0] == "--uncovered":
        only_uncovered, paths = True, paths[1:]
    total = 0
    for p in paths:
        kern = "?"
        code = []
        for i, l in enumerate(open(p).read().split("\n")):
            t = l.strip()
            if t.endswith(":") and not t.startswith((".", ";")):
                kern = t[:-1]
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            code.append((i + 1, l, kern))
        for j, (ln, l, k) in enumerate(code):
            m = ST.match(l)
            if not m:
                continue
            if only_uncovered and not uncovered(m.group(1), m.group(2)):
                continue
            d = data_regs(m.group(1), m.group(2))
            for q in range(1, WINDOW + 1):
                if j + q >= len(code):
                    break
                nl = code[j + q][1]
                if nl.strip().startswith("s_nop"):
                    break
                hit = d & dst_regs(nl)
                if hit:
                    total += 1
                    print("%s:%d  %s\n    %s\n    -> +%d %s   (v%s)" % (p, ln, k[:90], l.strip(), q, nl.strip(), sorted(hit)))
                    break
    print("%d suspicious sites%s" % (total, " (stores outside LLVM's hazard recogniser)" if only_uncovered else ""))
    return 1 if (only_uncovered and total) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
