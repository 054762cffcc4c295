import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
lib.abc_debug_conv_prof.argtypes = [C.c_void_p]
dt = L.BF16
B, H, W = 16, 384, 384
for (Cin, Cout) in ((16, 16), (32, 32)):
    Hh, Ww = (H, W) if Cin == 16 else (H // 2, W // 2)
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, Hh, Ww, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / 12
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, 32, Cin)
    sc = tuple(t.to(U.DEV) for t in (torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.zeros(Cin)))
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, Hh, Ww, Cout), dtype=torch.bfloat16, device=U.DEV)
    def run():
        return U.conv(lib, x, dt, dt, B, Hh, Ww, Cin, 0, Cin, wp, bias, Cout, taps_square(3), Hh, Ww, coef=sc, out=out, stats=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1000
    mb = 2 * B * Hh * Ww * (Cin + Cout) / 1e6
    print("conv %d->%d @%dx%d: %.1f us, %.0f MB, %.2f TB/s" % (Cin, Cout, Hh, Ww, us, mb, mb / us))
    nwg = 768
    prof = torch.zeros((nwg, 8), dtype=torch.int64, device=U.DEV)
    lib.abc_debug_conv_prof(prof.data_ptr())
    run(); torch.cuda.synchronize()
    lib.abc_debug_conv_prof(None)
    p = prof.cpu().double()
    ok = p[:, 4] > 0
    p = p[ok]
    ntiles = B * (Hh // 16) * (Ww // 16)
    per_wg = ntiles / nwg
    tick = 0.01  # us per tick (100 MHz)
    print("  WGs with data %d, tiles per WG %.1f" % (len(p), per_wg))
    print("  WG lifetime  %.1f us  -> %.2f us per tile" % (((p[:, 4] - p[:, 0]).mean() * tick), ((p[:, 4] - p[:, 0]).mean() * tick / per_wg)))
    print("  last tile: main %.2f us, epilogue %.2f us, stats %.2f us" % (((p[:, 2] - p[:, 1]).mean() * tick), ((p[:, 3] - p[:, 2]).mean() * tick), ((p[:, 4] - p[:, 3]).mean() * tick)))
    print("  kernel span %.1f us" % ((p[:, 4].max() - p[:, 0].min()) * tick))
