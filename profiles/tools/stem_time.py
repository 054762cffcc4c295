import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
B, H, Cin, Cout = 64, 512, 16, 16
g = torch.Generator().manual_seed(1)
x = torch.randn((B, H, H, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
img = torch.rand((B, H, H), generator=g).to(U.DEV)
w0 = (torch.randn((16, 1, 3, 3), generator=g) / 3).to(U.DEV); sc0 = (torch.rand(16, generator=g) + 0.5).to(U.DEV); b0 = (torch.randn(16, generator=g) * 0.2).to(U.DEV)
w = torch.randn((Cout, Cin, 3, 3), generator=g) / 12
wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, 32, Cin)
bias = torch.randn(Cout).to(U.DEV)
out = torch.zeros((B, H, H, Cout), dtype=torch.bfloat16, device=U.DEV)
for name, stem in (("plain", None), ("stem", (img, w0, sc0, b0, 0.0))):
    def run():
        return U.conv(lib, x, dt, dt, B, H, H, Cin, 0, Cin, wp, bias, Cout, taps_square(3), H, H, out=out, out_slope=0.0, stem=stem)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1000), U.conv.last_variant, flush=True)
