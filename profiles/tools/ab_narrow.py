import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
for (B, H, Cin, Cout, act) in ((16, 384, 16, 16, False), (16, 192, 32, 32, False), (16, 384, 32, 16, False), (64, 512, 16, 16, True), (64, 256, 32, 32, True), (64, 512, 32, 16, True)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, H, H, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / 12
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, 32, Cin)
    sc = tuple(t.to(U.DEV) for t in (torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.zeros(Cin)))
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, H, H, Cout), dtype=torch.bfloat16, device=U.DEV)
    def run():
        return U.conv(lib, x, dt, dt, B, H, H, Cin, 0, Cin, wp, bias, Cout, taps_square(3), H, H, coef=sc, out=out, stats=not act)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1000
    mb = 2 * B * H * H * (Cin + Cout) / 1e6
    print("conv %d->%d @%dx%d b%d: %.1f us, %.0f MB, %.2f TB/s" % (Cin, Cout, H, H, B, us, mb, mb / us), flush=True)
