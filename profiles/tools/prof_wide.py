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
B, Hh, Ww, Cin, Cout = 16, 96, 96, 128, 128
g = torch.Generator().manual_seed(1)
x = torch.randn((B, Hh, Ww, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
w = torch.randn((Cout, Cin, 3, 3), generator=g) / 30
wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, 128, Cin)
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
print("conv 128->128 @96x96 b16: %.1f us = %.0f TFLOP/s" % (us, 2 * B * Hh * Ww * 128 * 128 * 9 / us / 1e6))
nwg = 768
prof = torch.zeros((nwg, 8), dtype=torch.int64, device=U.DEV)
lib.abc_debug_conv_prof(prof.data_ptr())
run(); torch.cuda.synchronize()
lib.abc_debug_conv_prof(None)
p = prof.cpu().double()
tick = 0.01
t0 = p[:, 0].min()
start = (p[:, 0] - t0) * tick
first = start < 5.0
print("WGs started in the first 5 us: %d; later: %d" % (first.sum().item(), (~first).sum().item()))
for name, sel in (("first round", first), ("second round", ~first)):
    q = p[sel]
    print("  %s: prologue %.1f, main %.1f, epilogue %.1f, stats %.1f, total %.1f us; start %.1f..%.1f" % (
        name, ((q[:, 1] - q[:, 0]).mean() * tick), ((q[:, 2] - q[:, 1]).mean() * tick), ((q[:, 3] - q[:, 2]).mean() * tick),
        ((q[:, 4] - q[:, 3]).mean() * tick), ((q[:, 4] - q[:, 0]).mean() * tick), start[sel].min().item(), start[sel].max().item()))
print("kernel span %.1f us" % ((p[:, 4].max() - t0) * tick))
