"""Same-process A/B of the 128-channel convolution forms (conv_fast.hip: lane = channel epilogue through LDS; conv_fast_lp.hip: lane = pixel epilogue from registers; "nw8": conv_fast8.hip, 8-wave workgroups)
through the C ABI on the shapes that own the headline step: interleaved rounds, median and min per form, outputs compared bit for bit.

  ABC_TOOL_LIB=scratch/lib_dbg.so python profiles/tools/ab_conv128.py [rounds] [all]      (the DEBUG flavour of the library: the forms other than
  the default exist only there)

Forms are switched with the measurement hooks abc_debug_conv_lp / abc_debug_conv_nw.
"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd  # noqa: F401
from abcnet_amd import _lib as L
if os.environ.get("ABC_TOOL_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["ABC_TOOL_LIB"])
import hiputil as U
from abcnet_amd.engine import taps_square

lib = L.load()
lib.abc_debug_conv_nw.argtypes = [C.c_int]
lib.abc_debug_conv_nw.restype = None
lib.abc_debug_conv_lp.argtypes = [C.c_int]
lib.abc_debug_conv_lp.restype = None
lib.abc_debug_conv_var.argtypes = [C.c_int]
lib.abc_debug_conv_var.restype = None
try:      # (only in a library built with profiles/tools/conv_mixed_round_r05.patch applied: the mixed last round, measured slower)
    lib.abc_debug_conv_mixed.argtypes = [C.c_int]
    lib.abc_debug_conv_mixed.restype = None
    HAVE_MIXED = True
except AttributeError:
    HAVE_MIXED = False


def set_form(v):
    """v = (lp, var, nw): abc_debug_conv_lp (0: lane = channel epilogue, the product's; 1: lane = pixel where it applies),
    abc_debug_conv_var (bit 0: 1 x 4 wave layout, bit 1: s_setprio around the MFMA groups), abc_debug_conv_nw (8: 8-wave workgroups)"""
    lib.abc_debug_conv_lp(v[0])
    lib.abc_debug_conv_var(v[1])
    lib.abc_debug_conv_nw(v[2])
    if HAVE_MIXED:
        lib.abc_debug_conv_mixed(0 if len(v) > 3 and v[3] == 0 else 1)      # (4th entry 0: no mixed last round -- the product's launch)


dt = L.BF16
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
REP = 20


def make_case(name, B, H, Cin, Cout, kind):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, H, H, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / (3 * Cin ** 0.5)
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, -(-Cout // 32) * 32, Cin)
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, H, H, Cout), dtype=torch.bfloat16, device=U.DEV)
    kw = {}
    if kind == "fwd":      # training forward: BatchNorm + ReLU of the producer on load, statistics out
        kw = dict(coef=tuple(t.to(U.DEV) for t in (torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.zeros(Cin))), stats=True)
    elif kind == "plain":  # data gradient / folded inference layer
        kw = dict()
    elif kind == "infer":
        kw = dict(out_slope=0.0)
    elif kind == "actb":   # data gradient with the producer's act_bwd in the epilogue
        yraw = (torch.randn((B, H, H, Cout), generator=g) * 1.5 + 0.3).to(torch.bfloat16).to(U.DEV)
        cs = tuple(t.to(U.DEV) for t in (torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.5, torch.zeros(Cout), torch.randn(Cout) * 0.3, torch.rand(Cout) + 0.5))
        kw = dict(stats=True, actbwd=(yraw, Cout, 0) + cs)
        bias = None
    flops = 2.0 * B * H * H * Cin * Cout * 9

    def build():
        """the descriptor under the CURRENT form (the geometry -- and the number of statistics rows -- is decided when it is built)"""
        keep = []
        o, st = U.conv(lib, x, dt, dt, B, H, H, Cin, 0, Cin, wp, bias, Cout, taps_square(3), H, H, out=out, defer=keep, **kw)
        d = keep[0][0]
        return (lambda: L.check(lib.abc_conv_fwd(C.byref(d), U.stream()), "conv_fwd")), o, st, keep
    return name, build, flops, out


CASES = [
    make_case("trunk 128->128 @96 b16 fwd (BN on load + stats)", 16, 96, 128, 128, "fwd"),
    make_case("trunk 128->128 @96 b16 plain (data gradient)", 16, 96, 128, 128, "plain"),
    make_case("trunk 128->128 @96 b16 data gradient + act_bwd", 16, 96, 128, 128, "actb"),
    make_case("heads conv1 128->1024 @96 b16 fwd", 16, 96, 128, 1024, "fwd"),
    make_case("heads conv1 data gradient 1024->128 @96 b16 + act_bwd", 16, 96, 1024, 128, "actb"),
    make_case("inference 128->128 @128 b64 folded", 64, 128, 128, 128, "infer"),
    make_case("decoder 256->128 @48 b16 fwd", 16, 48, 256, 128, "fwd"),
    make_case("128->128 @96 b20 fwd (960 tiles: last round 448 of 512, no mixed round)", 20, 96, 128, 128, "fwd"),
    make_case("128->128 @96 b12 fwd (576 tiles: last round 64 of 512)", 12, 96, 128, 128, "fwd"),
]
FORMS = [("product", (0, 0, 0, 0))] + ([("mixed round", (0, 0, 0, 1))] if HAVE_MIXED else [])
if "all" in sys.argv:
    FORMS += [("lc 1x4", (0, 1, 0, 0)), ("lc prio", (0, 2, 0, 0)), ("lc 1x4 prio", (0, 3, 0, 0)),
              ("lp", (1, 0, 0, 0)), ("lp 1x4", (1, 1, 0, 0)), ("lp prio", (1, 2, 0, 0)), ("lp 1x4 prio", (1, 3, 0, 0)), ("nw8", (0, 0, 8, 0))]


def graph_of(run):
    """REP launches as one hipGraph: launched one by one from Python the 50-us kernels are bound by the host"""
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            run()
    return g


def time_one(g):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1000


for name, build, flops, out in CASES:
    res = {f: [] for f, _ in FORMS}
    outs, graphs, keeps = {}, {}, []
    for f, v in FORMS:
        set_form(v)
        run, o, st, keep = build()
        keeps.append(keep)
        out.zero_()
        run()
        torch.cuda.synchronize()
        outs[f] = (o.clone(), None if st is None else st.clone())
        graphs[f] = graph_of(run)
    ref = outs[FORMS[0][0]]
    same = all(torch.equal(ref[0], outs[f][0]) for f, _ in FORMS[1:])
    # statistics: the forms write different numbers of partial rows and sum in different orders: compare the column sums
    st_err = 0.0
    if ref[1] is not None:
        cs = ref[1].double().sum(0)
        st_err = max(((outs[f][1].double().sum(0) - cs).abs().max() / cs.abs().max()).item() for f, _ in FORMS[1:])
    same_st = st_err < 1e-5
    for r in range(ROUNDS):
        for f, v in FORMS:
            set_form(v)      # (the launch re-derives the geometry: the form must be the one the descriptor was built under)
            res[f].append(time_one(graphs[f]))
    set_form((0, 0, 0))
    print("%s: outputs %s, statistics %s (%.1e)" % (name, "bit-equal" if same else "DIFFER", "agree" if same_st else "DIFFER", st_err))
    for f, _ in FORMS:
        med, mn = statistics.median(res[f]), min(res[f])
        print("      %-14s median %7.1f us  min %7.1f  (%.3f of 2.5 PF)" % (f, med, mn, flops / (med * 1e-6) / 2.5e15), flush=True)
