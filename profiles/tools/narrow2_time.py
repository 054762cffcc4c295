import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
for (B, H, Cin, Cout) in ((16, 384, 16, 16), (16, 192, 32, 32), (16, 192, 32, 16), (64, 512, 16, 16), (64, 256, 32, 32), (64, 256, 32, 16)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, H, H, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / 12
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, 32, Cin)
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, H, H, Cout), dtype=torch.bfloat16, device=U.DEV)
    res = {}
    for mode in ("narrow", "fast"):
        if mode == "fast": os.environ["ABC_CONV_NONARROW"] = "1"
        else: os.environ.pop("ABC_CONV_NONARROW", None)
        def run():
            return U.conv(lib, x, dt, dt, B, H, H, Cin, 0, Cin, wp, bias, Cout, taps_square(3), H, H, out=out, out_slope=0.0)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 30 * 1000, U.conv.last_variant)
    mb = 2 * B * H * H * (Cin + Cout) / 1e6
    print("conv %d->%d @%dx%d b%d: narrow %.1f us (v%d, %.2f TB/s)  fast %.1f us (v%d)  %.0f MB" % (Cin, Cout, H, H, B, res["narrow"][0], res["narrow"][1], mb / res["narrow"][0], res["fast"][0], res["fast"][1], mb), flush=True)
