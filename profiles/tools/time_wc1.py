import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
B, H, W = 16, 384, 384
g_ = torch.Generator().manual_seed(1)
for k, Cout, dual in ((3, 16, True), (5, 32, False), (5, 32, True)):
    gd = torch.randn((B, H, W, Cout), generator=g_).to(torch.bfloat16).to(U.DEV)
    yd = torch.randn((B, H, W, Cout), generator=g_).to(torch.bfloat16).to(U.DEV)
    xd = torch.rand((B, H, W, 1), generator=g_).to(U.DEV)
    pcoef = tuple(torch.randn(Cout).to(U.DEV) for _ in range(3))
    out = torch.zeros((B, H, W, Cout), dtype=torch.bfloat16, device=U.DEV)
    d = L.WgradDesc()
    U.fill_src(d.p, gd, H, W, Cout, pcoef if dual else None)
    U.fill_src(d.q, xd, H, W, 1, None)
    d.dtype_p, d.dtype_q, d.dtype_c = dt, L.F32, dt
    nsplit = -(-B * H // 12)
    d.B, d.Hg, d.Wg, d.Hq, d.Wq, d.Ca, d.Cb, d.stride, d.nsplit = B, H, W, H, W, Cout, 1, 1, nsplit
    L.set_taps(d, taps_square(k))
    if dual:
        d.p2, d.ld_p2, d.cp2_off, d.p_dual, d.p_out, d.ld_pout = yd.data_ptr(), Cout, 0, 1, out.data_ptr(), Cout
        assert lib.abc_wgrad_fuses_apply(C.byref(d)) == 1
    part = torch.zeros(nsplit * k * k * 32 * 1, dtype=torch.float32, device=U.DEV)
    d.partial = part.data_ptr()
    run = lambda: L.check(lib.abc_wgrad(C.byref(d), U.stream()), "wgrad")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("wgrad_c1 %dx%d Ca=%d dual=%s: %.1f us" % (k, k, Cout, dual, e0.elapsed_time(e1) / 20 * 1000), flush=True)
