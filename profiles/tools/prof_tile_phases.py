"""In-kernel phase timestamps of the lean convolution kernel (abc_debug_conv_prof; DEBUG flavour of the library:
ABC_TOOL_LIB=scratch/lib_dbg.so python profiles/tools/prof_tile_phases.py).  Per workgroup and for its LAST tile:
tile start -> main loop start (prologue), main loop, epilogue, statistics; plus the launch's span and the spread of
workgroup start / end times.  ABC_CONV_PROF_ROUND=n (read by the DEBUG flavour) stamps the tile of round n instead of the last one: a
workgroup's last tile runs beside a partner that may already have finished, a middle round shows the steady state."""
import ctypes as C
import os
import sys

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd  # noqa: F401
from abcnet_amd import _lib as L
if os.environ.get("ABC_TOOL_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["ABC_TOOL_LIB"])
import hiputil as U
from abcnet_amd.engine import taps_square

lib = L.load()
lib.abc_debug_conv_prof.argtypes = [C.c_void_p]
lib.abc_debug_conv_lp.argtypes = [C.c_int]
dt = L.BF16


def run_case(B, Hh, Cin, Cout, name, kind="fwd"):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, Hh, Hh, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / 30
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, -(-Cout // 32) * 32, Cin)
    sc = tuple(t.to(U.DEV) for t in (torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.zeros(Cin)))
    bias = torch.randn(Cout).to(U.DEV)
    out = torch.zeros((B, Hh, Hh, Cout), dtype=torch.bfloat16, device=U.DEV)
    kw = dict(coef=sc, stats=True) if kind == "fwd" else dict()

    def run():
        return U.conv(lib, x, dt, dt, B, Hh, Hh, Cin, 0, Cin, wp, bias, Cout, taps_square(3), Hh, Hh, out=out, **kw)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    nwg = 4096
    prof = torch.zeros((nwg, 8), dtype=torch.int64, device=U.DEV)
    lib.abc_debug_conv_prof(prof.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.abc_debug_conv_prof(None)
    p = prof.cpu().double()
    p = p[p[:, 0] > 0]
    tick = 0.01    # wall_clock64: 100 MHz
    t0 = p[:, 0].min()
    rounds = p[:, 7]
    print("%s [%s]: %d workgroups, rounds per workgroup %s; launch span %.1f us" % (
        name, kind, len(p), sorted(set(int(r) + 1 for r in rounds.tolist())), (p[:, 4].max() - t0) * tick))
    print("   workgroup start spread %.2f us; end times: min %.1f median %.1f max %.1f us" % (
        (p[:, 0].max() - t0) * tick, (p[:, 4].min() - t0) * tick, (p[:, 4].median() - t0) * tick, (p[:, 4].max() - t0) * tick))
    for rd in sorted(set(rounds.tolist())):
        q = p[rounds == rd]
        print("   last tile in round %d (%4d workgroups): tile start at %.1f us; prologue %.2f us, main loop %.2f, epilogue %.2f, statistics %.2f; tile %.2f us" % (
            int(rd), len(q), ((q[:, 5] - t0).mean() * tick), ((q[:, 1] - q[:, 5]).mean() * tick), ((q[:, 2] - q[:, 1]).mean() * tick),
            ((q[:, 3] - q[:, 2]).mean() * tick), ((q[:, 4] - q[:, 3]).mean() * tick), ((q[:, 4] - q[:, 5]).mean() * tick)))


for lp in (0, 1):
    print("==== epilogue: %s" % ("lane = channel (LDS transpose)" if lp == 0 else "lane = pixel (registers)"))
    lib.abc_debug_conv_lp(lp)
    run_case(16, 96, 128, 128, "trunk 128->128 @96 b16 (768 tiles)")
    run_case(16, 96, 128, 128, "trunk 128->128 @96 b16 (768 tiles)", kind="plain")
    run_case(8, 96, 128, 128, "128->128 @96 b8 (384 tiles: one workgroup per CU)")
    run_case(16, 96, 128, 1024, "heads conv1 128->1024 @96 b16 (6144 tiles)")
    run_case(64, 128, 128, 128, "inference-shaped 128->128 @128 b64 (5632 tiles)", kind="plain")
    run_case(16, 96, 1024, 128, "heads conv1 data gradient 1024->128 @96 b16 (768 tiles)", kind="plain")
lib.abc_debug_conv_lp(0)
run_case(16, 24, 256, 256, "256->256 @24 b16")
run_case(16, 12, 512, 512, "512->512 @12 b16")
