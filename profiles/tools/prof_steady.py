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
def run_case(B, Hh, Cin, Cout, name):
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
    nwg = 512
    prof = torch.zeros((nwg, 8), dtype=torch.int64, device=U.DEV)
    lib.abc_debug_conv_prof(prof.data_ptr())
    run(); torch.cuda.synchronize()
    lib.abc_debug_conv_prof(None)
    p = prof.cpu().double(); tick = 0.01
    q = p[p[:, 7] >= 2]   # workgroups whose LAST recorded tile was in a steady-state round
    print("%s: %d WGs; last tile: round %.0f; tile start->main %.1f us, main %.1f, epilogue %.1f, stats %.1f; tile total %.1f us; kernel %.1f us" % (
        name, len(q), q[:, 7].mean(), ((q[:, 1] - q[:, 5]).mean() * tick), ((q[:, 2] - q[:, 1]).mean() * tick), ((q[:, 3] - q[:, 2]).mean() * tick),
        ((q[:, 4] - q[:, 3]).mean() * tick), ((q[:, 4] - q[:, 5]).mean() * tick), ((p[:, 4].max() - p[:, 0].min()) * tick)))
run_case(16, 96, 128, 1024, "128->1024 @96 b16 (6144 tiles)")
run_case(64, 128, 128, 128, "128->128 @128 b64 (5632 tiles)")
