import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
def bench(B, Hh, Cin, Cout, iters=10):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, Hh, Hh, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / 30
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, -(-Cout // 32) * 32, Cin)
    sc = tuple(t.to(U.DEV) for t in (torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.zeros(Cin)))
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, Hh, Hh, Cout), dtype=torch.bfloat16, device=U.DEV)
    def run():
        return U.conv(lib, x, dt, dt, B, Hh, Hh, Cin, 0, Cin, wp, bias, Cout, taps_square(3), Hh, Hh, coef=sc, out=out, stats=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1000
    return us, 2.0 * B * Hh * Hh * Cin * Cout * 9 / us / 1e6
for name, args in (("128->128 @96 b16 (768 tiles)", (16, 96, 128, 128)), ("128->1024 @96 b16 (6144 tiles)", (16, 96, 128, 1024)),
                   ("1024->128 @96 b16 (768 long)", (16, 96, 1024, 128)), ("128->128 @128 b64 (5632 tiles)", (64, 128, 128, 128))):
    us, tf = bench(*args)
    print("%-34s %8.1f us  %6.0f TFLOP/s  (%.3f of 2.5 PF)" % (name, us, tf, tf / 2500))
