import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
B, H, Cin, Cout, k = 16, 384, 32, 32, 5
g = torch.Generator().manual_seed(1)
x = torch.randn((B, H, H, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
w = torch.randn((Cout, Cin, k, k), generator=g) / 28
wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, k, 32, Cin)
bias = torch.randn(Cout).to(U.DEV)
out = torch.zeros((B, H, H, Cout), dtype=torch.bfloat16, device=U.DEV)
for mode in ("narrow", "fast"):
    if mode == "fast": os.environ["ABC_CONV_NONARROW5"] = "1"
    else: os.environ.pop("ABC_CONV_NONARROW5", None)
    for stats in (False, True):
        def run():
            return U.conv(lib, x, dt, dt, B, H, H, Cin, 0, Cin, wp, bias, Cout, taps_square(k), H, H, out=out, stats=stats)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1000
        gf = 2.0 * B * H * H * Cin * Cout * k * k / 1e9
        print("%s stats=%s: %.1f us  v%d  %.0f TFLOP/s" % (mode, stats, us, U.conv.last_variant, gf / us / 1e3), flush=True)
