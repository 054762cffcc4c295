# tile-shape sweep of the lean convolution kernel on the non-trunk layer shapes of unet.py at b16 @ 384 x 384 (debug build: the
# ABC_CONV_MT / ABC_CONV_BN experiment switches force the tile; "default" is what abc_conv_fast_geom's time model picks):
#   python profiles/tools/conv_tile_sweep.py
import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
if os.environ.get("ABC_TOOL_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["ABC_TOOL_LIB"])
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
B = 16
SHAPES = [(96, 32, 64), (96, 64, 64), (48, 64, 128), (48, 128, 128), (48, 256, 128), (24, 128, 256), (24, 256, 256), (24, 512, 256),
          (12, 256, 512), (12, 512, 512)]
g = torch.Generator().manual_seed(1)
for H, Cin, Cout in SHAPES:
    x = torch.randn((B, H, H, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / 30
    rows_pad = -(-Cout // 32) * 32
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, rows_pad, Cin)
    sc = tuple(t.to(U.DEV) for t in (torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.zeros(Cin)))
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, H, H, Cout), dtype=torch.bfloat16, device=U.DEV)
    res = []
    for bn in (None, 128, 64):
        for mt in (None, 8, 6, 4, 2):
            if (bn is None) != (mt is None):
                continue
            for k, v in (("ABC_CONV_BN", bn), ("ABC_CONV_MT", mt)):
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = str(v)
            for form, kw in (("fwd", dict(coef=sc, stats=True)), ("plain", dict(coef=None, stats=False))):
                lst = []
                U.conv(lib, x, dt, dt, B, H, H, Cin, 0, Cin, wp, bias, Cout, taps_square(3), H, H, out=out, defer=lst, **kw)
                d, wpk, st = lst[0]
                if U.conv.last_variant != 1:
                    continue
                bn_, mt_, ck_ = L.i32(), L.i32(), L.i32()
                lib.abc_conv_tile(C.byref(d), C.byref(bn_), C.byref(mt_), C.byref(ck_))
                if bn is not None and (bn_.value != bn or mt_.value != mt):
                    continue
                run = lambda: L.check(lib.abc_conv_fwd(C.byref(d), U.stream()), "conv")
                for _ in range(5): run()
                torch.cuda.synchronize()
                # (40 launches as ONE hipGraph: launched one by one from Python, 20-us kernels are bound by the host)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(40): run()
                gr.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record(); torch.cuda.synchronize()
                res.append((form, "default" if bn is None else "forced", bn_.value, mt_.value, e0.elapsed_time(e1) / 40 * 1000))
    for form in ("fwd", "plain"):
        rows = [r for r in res if r[0] == form]
        dflt = [r for r in rows if r[1] == "default"][0]
        best = min(rows, key=lambda r: r[4])
        print("%3d x %3d  %3d -> %3d  %-5s default BN%d MT%d %6.1f us | best BN%d MT%d %6.1f us | %s" % (
            H, H, Cin, Cout, form, dflt[2], dflt[3], dflt[4], best[2], best[3], best[4],
            "  ".join("BN%d/MT%d %.1f" % (r[2], r[3], r[4]) for r in rows if r[1] == "forced")), flush=True)
