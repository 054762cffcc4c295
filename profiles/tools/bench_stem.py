import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
for (B, S, Cout, k) in ((16, 384, 16, 3), (64, 512, 16, 3), (16, 384, 32, 5)):
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((B, S, S, 1), generator=g) < 0.1).float().to(U.DEV)
    w = torch.randn((Cout, 1, k, k), generator=g) / 3
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, 1, k, 32, 1)
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, S, S, Cout), dtype=torch.bfloat16, device=U.DEV)
    def run():
        return U.conv(lib, x, L.F32, dt, B, S, S, 1, 0, 1, wp, bias, Cout, taps_square(k), S, S, out=out, stats=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1000
    mb = B * S * S * (4 + Cout * 2) / 1e6
    print("stem %dx%d b%d -> %d ch k%d: %.1f us, %.0f MB, %.2f TB/s" % (S, S, B, Cout, k, us, mb, mb / us))
