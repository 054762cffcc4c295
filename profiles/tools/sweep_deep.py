import os, sys, ctypes as C, itertools
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
import hiputil as U
from abcnet_amd.engine import taps_square
lib = L.load()
dt = L.BF16
B = 16
shapes = [(48, 64, 128), (48, 128, 128), (48, 256, 128), (48, 128, 64), (24, 128, 256), (24, 256, 256), (24, 512, 256), (24, 256, 128),
          (12, 256, 512), (12, 512, 512), (12, 512, 256), (96, 64, 64), (96, 128, 64), (96, 64, 128)]
def bench(Hh, Cin, Cout):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, Hh, Hh, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / 30
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, -(-Cout // 32) * 32, Cin)
    sc = tuple(t.to(U.DEV) for t in (torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.zeros(Cin)))
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, Hh, Hh, Cout), dtype=torch.bfloat16, device=U.DEV)
    def run():
        return U.conv(lib, x, dt, dt, B, Hh, Hh, Cin, 0, Cin, wp, bias, Cout, taps_square(3), Hh, Hh, coef=sc, out=out, stats=True)
    try:
        for _ in range(2): run()
    except Exception as e:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1000
for (Hh, Cin, Cout) in shapes:
    os.environ.pop("ABC_CONV_MT", None); os.environ.pop("ABC_CONV_BN", None)
    base = bench(Hh, Cin, Cout)
    res = []
    for bn, mt in itertools.product((128, 64), (8, 6, 4, 2)):
        if bn > Cout: continue
        os.environ["ABC_CONV_MT"] = str(mt); os.environ["ABC_CONV_BN"] = str(bn)
        t = bench(Hh, Cin, Cout)
        if t is not None: res.append((t, bn, mt))
    res.sort()
    print("%3d^2 %3d->%3d: default %.1f us | best %s" % (Hh, Cin, Cout, base, ", ".join("BN%d/MT%d %.1f" % (b, m, t) for t, b, m in res[:4])))
