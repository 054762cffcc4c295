import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
B, H = 16, 384
g = torch.Generator().manual_seed(1)
img = torch.rand((B, H, H, 1), generator=g).to(U.DEV)
for k, Cout in ((3, 16), (5, 32)):
    w = torch.randn((Cout, 1, k, k), generator=g) / k
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, 1, k, 16, 1) if False else None
    # pack through the library: [tap][Cout_pad][16]
    cpad = -(-Cout // 32) * 32
    wp = torch.zeros((k * k, cpad, 16), dtype=torch.bfloat16, device=U.DEV)
    wp[:, :Cout, 0] = w.reshape(Cout, k * k).t().to(torch.bfloat16).to(U.DEV)
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, H, H, Cout), dtype=torch.bfloat16, device=U.DEV)
    def run():
        return U.conv(lib, img, L.F32, dt, B, H, H, 1, 0, 1, wp, bias, Cout, taps_square(k), H, H, out=out, stats=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("stem %dx%d -> %d:" % (k, k, Cout), "%.1f us" % (e0.elapsed_time(e1) / 20 * 1000), U.conv.last_variant, flush=True)
    ref = torch.nn.functional.conv2d(img.permute(0, 3, 1, 2), w.to(torch.bfloat16).float().to(U.DEV), bias, padding=k // 2).permute(0, 2, 3, 1)
    print("   max err vs torch", (out.float() - ref).abs().max().item(), "of", ref.abs().max().item())
