# unet2's 5x5 32 -> 32 weight gradient alone (wgrad_n32r2_kernel): python profiles/tools/time_w32.py [nsplit ...]
# (debug build: ABC_W32_DBG = phase-skipping ablations, bit 0 loads, 1 commit, 2 MFMA phase, 3 dY store)
import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
B, H, W, Cc = 16, 384, 384, 32
g_ = torch.Generator().manual_seed(1)
gd = torch.randn((B, H, W, Cc), generator=g_).to(torch.bfloat16).to(U.DEV)
yd = torch.randn((B, H, W, Cc), generator=g_).to(torch.bfloat16).to(U.DEV)
xd = torch.randn((B, H, W, Cc), generator=g_).to(torch.bfloat16).to(U.DEV)
pcoef = tuple(torch.randn(Cc).to(U.DEV) for _ in range(3))
qcoef = (torch.rand(Cc).to(U.DEV) + 0.5, torch.randn(Cc).to(U.DEV) * 0.1, torch.zeros(Cc).to(U.DEV))
out = torch.zeros((B, H, W, Cc), dtype=torch.bfloat16, device=U.DEV)
for nsplit in [int(v) for v in sys.argv[1:]] or [512]:
    d = L.WgradDesc()
    U.fill_src(d.p, gd, H, W, Cc, pcoef)
    U.fill_src(d.q, xd, H, W, Cc, qcoef)
    d.dtype_p, d.dtype_q, d.dtype_c = dt, dt, dt
    d.B, d.Hg, d.Wg, d.Hq, d.Wq, d.Ca, d.Cb, d.stride, d.nsplit = B, H, W, H, W, Cc, Cc, 1, nsplit
    L.set_taps(d, taps_square(5))
    d.p2, d.ld_p2, d.cp2_off, d.p_dual, d.p_out, d.ld_pout = yd.data_ptr(), Cc, 0, 1, out.data_ptr(), Cc
    at_, bt_ = L.i32(), L.i32(); lib.abc_wgrad_tile(C.byref(d), C.byref(at_), C.byref(bt_))
    ca_, cb_ = L.i32(), L.i32(); lib.abc_wgrad_pads(C.byref(d), C.byref(ca_), C.byref(cb_))
    part = torch.zeros(nsplit * 25 * ca_.value * cb_.value, dtype=torch.float32, device=U.DEV)
    d.partial = part.data_ptr()
    run = lambda: L.check(lib.abc_wgrad(C.byref(d), U.stream()), "wgrad")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("wgrad 32x32 5x5 dual+QT tile=(%d,%d) nsplit=%d dbg=%s: %.1f us" % (at_.value, bt_.value, nsplit, os.environ.get("ABC_W32_DBG", "0"), e0.elapsed_time(e1) / 20 * 1000), flush=True)
