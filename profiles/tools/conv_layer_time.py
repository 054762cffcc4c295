import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
import os
B, Hh, Ww, Cin, Cout = int(os.environ.get('WI_B','16')), 96, 96, 128, 128
g = torch.Generator().manual_seed(1)
x = torch.randn((B, Hh, Ww, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
w = torch.randn((Cout, Cin, 3, 3), generator=g) / 30
wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, 128, Cin)
sc = tuple(t.to(U.DEV) for t in (torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.zeros(Cin)))
bias = torch.randn(Cout).to(U.DEV)
out = torch.zeros((B, Hh, Ww, Cout), dtype=torch.bfloat16, device=U.DEV)
for name, kw in (("train-fwd (BN on load + stats)", dict(coef=sc, stats=True)), ("plain", dict(coef=None, stats=False))):
    run = lambda: U.conv(lib, x, dt, dt, B, Hh, Ww, Cin, 0, Cin, wp, bias, Cout, taps_square(3), Hh, Ww, out=out, **kw)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1000
    print("  %-32s %.1f us = %.0f TFLOP/s" % (name, us, 2 * B * Hh * Ww * 128 * 128 * 9 / us / 1e6))
