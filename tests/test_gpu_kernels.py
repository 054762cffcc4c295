"""GPU parity tests, kernel level: every HIP kernel of the path against plain torch-CPU fp32
ops of the same arithmetic (the same ATen ops the reference model dispatches), through the
C ABI.  Tolerances: fp32 mode 1e-4 relative-to-max (exact-f32 MFMA, different summation
order only); bf16 mode 3e-2 (operands rounded to 8 significant bits)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import abcnet_amd  # noqa: E402,F401
from abcnet_amd import _lib as L  # noqa: E402
from abcnet_amd.engine import TAPS_CONVT_DGRAD, convT_phase_taps, taps_mirror, taps_square  # noqa: E402

import hiputil as U  # noqa: E402

DTS = [L.F32, L.BF16]


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return L.load()


def q(x, dt):
    """round a CPU f32 tensor to the storage dtype (so that the reference sees what the kernel sees)"""
    return x.to(U.tdt(dt)).float()


def act(x, sc, sh, sl):
    y = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    return torch.maximum(y, sl.view(1, -1, 1, 1) * y)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [
    dict(Cin=16, Cout=16, k=3, H=24, W=40),
    dict(Cin=32, Cout=64, k=3, H=32, W=32, pool=True),
    dict(Cin=64, Cout=128, k=3, H=16, W=16, coef=True),
    dict(Cin=1, Cout=16, k=3, H=40, W=24, img=True),
    dict(Cin=128, Cout=14, k=1, H=16, W=16, coef=True, f32out=True),
    dict(Cin=32, Cout=32, k=5, H=24, W=24, coef=True),
    dict(Cin=256, Cout=96, k=3, H=8, W=8),
    # > 256 tiles: every persistent workgroup walks several tiles (cross-tile prefetch, resident weights on the
    # narrow layers, general loader for the image / pooled inputs)
    dict(Cin=1, Cout=16, k=3, H=128, W=384, img=True),
    dict(Cin=16, Cout=32, k=3, H=256, W=384, pool=True),
    dict(Cin=16, Cout=16, k=3, H=192, W=256, coef=True),
    dict(Cin=128, Cout=128, k=3, H=96, W=192, coef=True),
    # unet2's 5x5 stem: 25-tap one-channel kernel; 5x5 32 -> 32 on the lean kernel over many tiles
    dict(Cin=1, Cout=32, k=5, H=40, W=24, img=True),
    dict(Cin=32, Cout=32, k=5, H=128, W=192, coef=True),
    # widths that are no multiple of 4: the scalar one-channel kernel (the four-pixel form takes whole pixel quads)
    dict(Cin=1, Cout=16, k=3, H=24, W=30, img=True),
    dict(Cin=1, Cout=32, k=5, H=16, W=42, img=True),
])
def test_conv_forward(lib, dt, case):
    g = torch.Generator().manual_seed(3)
    B, Cin, Cout, k, H, W = 2, case["Cin"], case["Cout"], case["k"], case["H"], case["W"]
    pool = case.get("pool", False)
    img = case.get("img", False)
    dt_in = L.F32 if img else dt
    x = torch.randn((B, Cin, H, W), generator=g)
    x = x if img else q(x, dt)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    coef = None
    xin = x
    if case.get("coef") or pool:
        sc = torch.rand(Cin, generator=g) * 2 - 0.6
        sh = torch.randn(Cin, generator=g) * 0.3
        sl = torch.tensor([0.0, 0.01, 1.0])[torch.randint(0, 3, (Cin,), generator=g)]
        coef = tuple(t.to(U.DEV) for t in (sc, sh, sl))
        xin = act(x, sc, sh, sl)
    if pool:
        xin = F.max_pool2d(xin, 2)
    Ho, Wo = xin.shape[2:]
    ref = F.conv2d(q(xin, dt), q(w, dt), b, padding=(k - 1) // 2)
    xd = x.permute(0, 2, 3, 1).contiguous().to(U.tdt(dt_in)).to(U.DEV)
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, k, -(-Cout // 32) * 32, Cin)
    out_dt = L.F32 if case.get("f32out") else dt
    y, st = U.conv(lib, xd, dt_in, dt, B, H, W, Cin, 0, Cin, wp, b.to(U.DEV), Cout, taps_square(k), Ho, Wo, coef=coef, pool=pool,
                   out_dt=out_dt, stats=True)
    torch.cuda.synchronize()
    got = U.to_nchw(y)
    assert U.relerr(got, ref) < U.tol(dt), U.relerr(got, ref)
    # BatchNorm partial statistics of the f32 conv outputs
    s = st.double().sum(0).cpu()
    n = B * Ho * Wo
    np.testing.assert_allclose(s[0] / n, ref.double().mean((0, 2, 3)), atol=U.tol(dt, 1e-4, 2e-2))
    np.testing.assert_allclose(s[1] / n, (ref.double() ** 2).mean((0, 2, 3)), rtol=U.tol(dt, 1e-4, 3e-2), atol=1e-4)


@pytest.mark.parametrize("case", [
    dict(Cin=16, Cout=16, H=64, W=96),
    dict(Cin=32, Cout=32, H=40, W=48, slope=0.0),
    dict(Cin=32, Cout=16, H=24, W=40, mirror=True),           # a data gradient: mirrored taps, no bias
    dict(Cin=16, Cout=32, H=20, W=40, slope=0.01),            # ragged tiles: 20 = 2 x 8 + 4 rows, 40 = 2 x 16 + 8 columns
    dict(Cin=16, Cout=16, H=72, W=88, ld=48, coff=16, ldy=40, ycoff=8),   # channel slices of wider tensors
    dict(Cin=32, Cout=32, H=256, W=384, slope=0.0),           # every wave walks several tiles
    # the 3x3 form of the 32 -> 32 kernel (conv_n32r2_kernel, R = 1: unet.py's 192 x 192 level): training forward, data gradient
    dict(Cin=32, Cout=32, H=40, W=48, coef=True, stats=True),
    dict(Cin=32, Cout=32, H=192, W=192, coef=True, stats=True),
    dict(Cin=32, Cout=32, H=40, W=48, mirror=True),
    # training forward: the producer's BatchNorm + activation on load, BatchNorm partial sums of the outputs
    dict(Cin=16, Cout=16, H=64, W=96, coef=True, stats=True),
    dict(Cin=16, Cout=16, H=200, W=136, coef=True, stats=True),   # ragged tiles, several tiles per wave
    dict(Cin=16, Cout=32, H=48, W=64, stats=True),                # a pooled (finished) input into a wider layer
    dict(Cin=32, Cout=16, H=40, W=48, stats=True),
    # unet2's 5x5 32 -> 32 levels (unet2.py:49-74): data gradient (mirrored taps), block-first conv (sums), inference
    dict(Cin=32, Cout=32, H=40, W=48, k=5, mirror=True),
    dict(Cin=32, Cout=32, H=72, W=104, k=5, stats=True),
    dict(Cin=32, Cout=32, H=128, W=192, k=5, slope=0.0),
    # second output: MaxPool2d(2) of the stored tensor (the folded inference graph's levels)
    dict(Cin=16, Cout=16, H=64, W=96, slope=0.0, pool=True),
    dict(Cin=32, Cout=32, H=40, W=48, slope=0.01, pool=True),
    dict(Cin=16, Cout=32, H=22, W=42, slope=0.0, pool=True),      # ragged tiles and an odd pooled width (21)
    # the network's first convolution (one-channel image, folded BatchNorm, ReLU) computed inside the halo staging
    dict(Cin=16, Cout=16, H=64, W=96, slope=0.0, stem=True),
    dict(Cin=16, Cout=16, H=44, W=72, slope=0.0, stem=True, pool=True),
])
def test_conv_narrow_plain_input(lib, case):
    """3x3 over a FINISHED bf16 tensor with 16 / 32 channels (the folded inference graph's narrow levels, unet.py:12,15 in eval
    mode; the narrow data gradients of training): csrc/conv_narrow.hip -- wave-private tiles, weights in registers, stores
    straight from transposed accumulators -- against F.conv2d of the same bf16 operands"""
    dt = L.BF16
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H, W, k = 2, case["Cin"], case["Cout"], case["H"], case["W"], case.get("k", 3)
    ld, coff = case.get("ld", Cin), case.get("coff", 0)
    ldy, ycoff = case.get("ldy", Cout), case.get("ycoff", 0)
    xfull = q(torch.randn((B, ld, H, W), generator=g), dt)
    x = xfull[:, coff:coff + Cin]
    coef = None
    stem = None
    if case.get("stem"):
        img = torch.rand((B, 1, H, W), generator=g)
        w0 = torch.randn((Cin, 1, 3, 3), generator=g) / 3
        sc0, b0 = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
        y0 = F.conv2d(img, w0, None, padding=1) * sc0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1)
        x = q(torch.relu(y0), dt)                 # what the fused kernel puts into its halo (bf16)
        xfull = x
        stem = (img.reshape(B, H, W).contiguous().to(U.DEV), w0.to(U.DEV).contiguous(), sc0.to(U.DEV), b0.to(U.DEV), 0.0)
    if case.get("coef"):
        sc = torch.rand(ld, generator=g) * 2 - 0.6
        sh = torch.randn(ld, generator=g) * 0.3
        sl = torch.tensor([0.0, 0.01, 1.0])[torch.randint(0, 3, (ld,), generator=g)]
        coef = tuple(t.to(U.DEV) for t in (sc, sh, sl))
        x = q(act(x, sc[coff:coff + Cin], sh[coff:coff + Cin], sl[coff:coff + Cin]), dt)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    bias = None if case.get("mirror") else torch.randn(Cout, generator=g)
    taps = taps_mirror(taps_square(k)) if case.get("mirror") else taps_square(k)
    wref = q(w, dt).flip(2, 3) if case.get("mirror") else q(w, dt)
    ref = F.conv2d(x, wref, bias, padding=k // 2)
    slope = case.get("slope")
    if slope is not None:
        ref = torch.maximum(ref, slope * ref)
    xd = xfull.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(U.DEV)
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, k, 32, Cin)
    out = torch.full((B, H, W, ldy), 7.0, dtype=torch.bfloat16, device=U.DEV)
    pooled = torch.full((B, H // 2, W // 2, Cout), 7.0, dtype=torch.bfloat16, device=U.DEV) if case.get("pool") else None
    y, st = U.conv(lib, xd, dt, dt, B, H, W, ld, coff, Cin, wp, None if bias is None else bias.to(U.DEV), Cout, taps, H, W, ldy=ldy,
                   cout_off=ycoff, out=out, out_slope=slope, coef=coef, stats=bool(case.get("stats")), pool_out=pooled, stem=stem)
    torch.cuda.synchronize()
    assert U.conv.last_variant == 5, "not served by conv_narrow"
    got = y[..., ycoff:ycoff + Cout].float().permute(0, 3, 1, 2).cpu()
    assert U.relerr(got, ref) < 1.5e-2, U.relerr(got, ref)
    if case.get("pool"):
        # max-pool of the STORED (bf16) tensor: exact against the kernel's own full-resolution output
        want = F.max_pool2d(got, 2)
        assert torch.equal(pooled.float().permute(0, 3, 1, 2).cpu(), want)
    if case.get("stats"):
        ssum = st.double().sum(0).cpu()
        n = B * H * W
        np.testing.assert_allclose(ssum[0] / n, ref.double().mean((0, 2, 3)), atol=2e-2)
        np.testing.assert_allclose(ssum[1] / n, (ref.double() ** 2).mean((0, 2, 3)), rtol=3e-2, atol=1e-4)
    # the other channels of the output tensor are untouched
    if ldy > Cout:
        rest = torch.cat([y[..., :ycoff], y[..., ycoff + Cout:]], dim=-1)
        assert (rest == 7.0).all()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Cout,W", [(14, 32), (60, 24), (1, 20), (360, 18)])
def test_head_conv1x1_writes_nchw(lib, dt, Cout, W):
    """heads' 1x1 conv (unet.py:70): BN+LeakyReLU+dropout on load, channel-offset input, NCHW f32 output"""
    from abcnet_amd.dropout import keep_mask
    g = torch.Generator().manual_seed(31)
    B, H, ld, coff, Cin = 2, 16, 256, 128, 128
    x = q(torch.randn((B, ld, H, W), generator=g), dt)
    w = torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(ld, generator=g) * 2 - 0.6, torch.randn(ld, generator=g) * 0.3
    sl = torch.full((ld,), 0.01)
    p_drop, seed = 0.2, 777
    idx = (torch.arange(B * H * W).view(B, H, W, 1) * ld + torch.arange(ld).view(1, 1, 1, ld))
    keep = keep_mask(idx, seed, p_drop).permute(0, 3, 1, 2).float()
    a = act(x, sc, sh, sl) * keep / (1 - p_drop)
    ref = F.conv2d(q(a[:, coff:coff + Cin], dt), q(w, dt), b)
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 1, -(-Cout // 32) * 32, Cin)
    coef = tuple(t.to(U.DEV) for t in (sc, sh, sl))
    y, _ = U.conv(lib, U.nhwc(x, dt), dt, dt, B, H, W, ld, coff, Cin, wp, b.to(U.DEV), Cout, [(0, 0)], H, W, coef=coef, out_dt=L.F32,
                  drop_p=p_drop, drop_seed=seed, planar_out=True)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (B, Cout, H, W)
    assert U.relerr(y.cpu(), ref) < U.tol(dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Cn", [14, 60, 1, 360])
def test_head_conv1x1_backward_from_nchw(lib, dt, Cn):
    """data and weight gradient of the heads' 1x1 conv, reading NCHW f32 dlogits with a per-channel scale"""
    g = torch.Generator().manual_seed(33)
    B, H, W, Cin = 2, 16, 24, 128
    f = q(torch.randn((B, Cin, H, W), generator=g), dt).requires_grad_(True)
    w = q(torch.randn((Cn, Cin, 1, 1), generator=g) / Cin ** 0.5, dt).requires_grad_(True)
    dl = torch.randn((B, Cn, H, W), generator=g)
    scale = 0.37
    F.conv2d(f, w).backward(q(dl * scale, dt))
    dld = dl.to(U.DEV).contiguous()
    cs = (torch.full((Cn,), scale), torch.zeros(Cn), torch.ones(Cn))
    cs = tuple(t.to(U.DEV) for t in cs)
    wd = U.pack(lib, w.detach().to(U.DEV), 1, dt, Cn, Cin, 1, Cin, Cn)
    dx, _ = U.conv(lib, dld, L.F32, dt, B, H, W, 0, 0, Cn, wd, None, Cin, [(0, 0)], H, W, coef=cs, planar_in=Cn)
    d = L.WgradDesc()
    U.fill_src(d.p, dld, H, W, 0, cs)
    d.p.planar, d.p.ctot = 1, Cn
    fd = U.nhwc(f.detach(), dt)
    U.fill_src(d.q, fd, H, W, Cin)
    d.dtype_p, d.dtype_q, d.dtype_c = L.F32, dt, dt
    d.B, d.Hg, d.Wg, d.Hq, d.Wq, d.Ca, d.Cb, d.stride, d.nsplit = B, H, W, H, W, Cn, Cin, 1, 2
    L.set_taps(d, [(0, 0)])
    ca, cb = L.i32(), L.i32()
    L.check(lib.abc_wgrad_pads(C.byref(d), C.byref(ca), C.byref(cb)), "pads")
    part = torch.zeros(2 * ca.value * cb.value, dtype=torch.float32, device=U.DEV)
    d.partial = part.data_ptr()
    L.check(lib.abc_wgrad(C.byref(d), U.stream()), "wgrad")
    dw = torch.zeros((Cn, Cin, 1), dtype=torch.float32, device=U.DEV)
    r = L.WgradReduceDesc()
    r.partial, r.nsplit, r.ntaps, r.Ca, r.Cb, r.Ca_pad, r.Cb_pad, r.dw, r.accumulate = part.data_ptr(), 2, 1, Cn, Cin, ca.value, cb.value, dw.data_ptr(), 0
    L.check(lib.abc_wgrad_reduce(C.byref(r), U.stream()), "reduce")
    db = torch.zeros(Cn, device=U.DEV)
    psw = torch.zeros(lib.abc_plane_sum_work(Cn), device=U.DEV)
    L.check(lib.abc_plane_sum(dld.data_ptr(), B, Cn, H * W, cs[0].data_ptr(), psw.data_ptr(), db.data_ptr(), U.stream()), "plane_sum")
    torch.cuda.synchronize()
    assert U.relerr(U.to_nchw(dx), f.grad) < U.tol(dt)
    assert U.relerr(dw.cpu().view(Cn, Cin, 1, 1), w.grad) < U.tol(dt)
    np.testing.assert_allclose(db.cpu(), (dl * scale).sum((0, 2, 3)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("Cn,nsplit", [(14, 3), (360, 5), (60, 1), (1, 7)])
def test_head_wgrad_fused_activation(lib, Cn, nsplit):
    """weight gradient of a head's 1x1 conv in bf16 mode (dedicated kernel): NCHW f32 dlogits with a per-channel scale
    against BN + LeakyReLU + dropout applied on load to a channel slice of the NHWC feature map (unet.py:67-70)"""
    from abcnet_amd.dropout import keep_mask
    dt = L.BF16
    g = torch.Generator().manual_seed(35)
    B, H, W, ld, coff, Cin = 2, 16, 32, 256, 128, 128
    x = q(torch.randn((B, ld, H, W), generator=g), dt)
    sc, sh = torch.rand(ld, generator=g) * 2 - 0.6, torch.randn(ld, generator=g) * 0.3
    sl = torch.full((ld,), 0.01)
    p_drop, seed = 0.2, 4242
    idx = (torch.arange(B * H * W).view(B, H, W, 1) * ld + torch.arange(ld).view(1, 1, 1, ld))
    keep = keep_mask(idx, seed, p_drop).permute(0, 3, 1, 2).float()
    a = q((act(x, sc, sh, sl) * keep / (1 - p_drop))[:, coff:coff + Cin], dt)
    dl = torch.randn((B, Cn, H, W), generator=g)
    scale = torch.rand(Cn, generator=g) + 0.2
    ref = torch.einsum("bohw,bihw->oi", q(dl * scale.view(1, -1, 1, 1), dt).double(), a.double())
    dld = dl.to(U.DEV).contiguous()
    cs = tuple(t.to(U.DEV) for t in (scale, torch.zeros(Cn), torch.ones(Cn)))
    d = L.WgradDesc()
    U.fill_src(d.p, dld, H, W, 0, cs)
    d.p.planar, d.p.ctot = 1, Cn
    xd = U.nhwc(x, dt)
    qcoef = tuple(t.to(U.DEV) for t in (sc, sh, sl))  # (kept alive: the descriptor holds raw pointers)
    U.fill_src(d.q, xd, H, W, ld, qcoef, False, p_drop, seed)
    d.dtype_p, d.dtype_q, d.dtype_c = L.F32, dt, dt
    d.B, d.Hg, d.Wg, d.Hq, d.Wq, d.Ca, d.Cb, d.stride, d.nsplit, d.cq_off = B, H, W, H, W, Cn, Cin, 1, nsplit, coff
    L.set_taps(d, [(0, 0)])
    ca, cb = L.i32(), L.i32()
    L.check(lib.abc_wgrad_pads(C.byref(d), C.byref(ca), C.byref(cb)), "pads")
    at, bt = L.i32(), L.i32()
    L.check(lib.abc_wgrad_tile(C.byref(d), C.byref(at), C.byref(bt)), "tile")
    assert (at.value, bt.value) == (0, 0)  # the head kernel took it
    part = torch.full((nsplit * ca.value * cb.value,), float("nan"), dtype=torch.float32, device=U.DEV)
    d.partial = part.data_ptr()
    # per-split row sums of the transformed dL ride along: the conv's bias gradient (f32 sums, before the bf16 rounding)
    assert lib.abc_wgrad_rowsum_ok(C.byref(d)) == 1
    rs = torch.full((nsplit, ca.value), float("nan"), dtype=torch.float32, device=U.DEV)
    d.rowsum_partial = rs.data_ptr()
    L.check(lib.abc_wgrad(C.byref(d), U.stream()), "wgrad")
    torch.cuda.synchronize()
    want_b = (dl * scale.view(1, -1, 1, 1)).double().sum((0, 2, 3))
    got_b = rs.double().sum(0).cpu()
    assert torch.isfinite(rs).all() and (got_b[:Cn] - want_b).abs().max().item() <= 1e-5 * want_b.abs().max().item() + 1e-6
    assert (got_b[Cn:] == 0).all()
    dw = torch.zeros((Cn, Cin, 1), dtype=torch.float32, device=U.DEV)
    r = L.WgradReduceDesc()
    r.partial, r.nsplit, r.ntaps, r.Ca, r.Cb, r.Ca_pad, r.Cb_pad, r.dw, r.accumulate = part.data_ptr(), nsplit, 1, Cn, Cin, ca.value, cb.value, dw.data_ptr(), 0
    L.check(lib.abc_wgrad_reduce(C.byref(r), U.stream()), "reduce")
    torch.cuda.synchronize()
    assert U.relerr(dw.cpu().view(Cn, Cin), ref) < U.tol(dt)


@pytest.mark.parametrize("dt", DTS)
def test_pool_act_materialised(lib, dt):
    """abc_pool_act == max_pool2d(act(x), 2) on a channel slice (unet.py:30)"""
    g = torch.Generator().manual_seed(41)
    B, H, W, ld, coff, Cn = 2, 12, 20, 48, 16, 24
    x = q(torch.randn((B, ld, H, W), generator=g), dt)
    sc, sh = torch.rand(ld, generator=g) * 2 - 0.6, torch.randn(ld, generator=g) * 0.3
    sl = torch.tensor([0.0, 0.01, 1.0])[torch.randint(0, 3, (ld,), generator=g)]
    ref = F.max_pool2d(act(x, sc, sh, sl)[:, coff:coff + Cn], 2)
    xd = U.nhwc(x, dt)
    coef = tuple(t.to(U.DEV) for t in (sc, sh, sl))
    a = L.ActSrc()
    U.fill_src(a, xd, H, W, ld, coef)
    out = torch.zeros((B, H // 2, W // 2, Cn), dtype=U.tdt(dt), device=U.DEV)
    L.check(lib.abc_pool_act(C.byref(a), dt, coff, Cn, B, out.data_ptr(), dt, Cn, U.stream()), "pool_act")
    torch.cuda.synchronize()
    assert U.relerr(U.to_nchw(out), q(ref, dt)) < (1e-6 if dt == L.F32 else 1e-2)


@pytest.mark.parametrize("Cout,Cin,k,qt", [(128, 128, 3, True), (64, 32, 3, True), (32, 64, 3, True), (32, 32, 5, True), (32, 32, 5, False), (16, 16, 3, True)])
def test_wgrad_fused_bn_apply(lib, Cout, Cin, k, qt):
    """abc_wgrad with p_dual: P = ca*g + cb*y_raw + cc applied on load (the BatchNorm-backward correction), the corrected
    tensor written to p_out -- against the explicit formula followed by the plain weight gradient"""
    dt = L.BF16
    g_ = torch.Generator().manual_seed(51)
    # (k = 5: unet2's 32-channel 5x5 layers, wgrad_n32r2_kernel of wgrad_narrow.hip: 4 x 3 tiles per image, border tiles on all sides
    #  and interior ones, three workgroups walking eight tiles each, with and without the transform of X; 16 x 16: the wave-per-tile
    #  kernel of wgrad_narrow.hip, more tiles than waves so that every wave walks several, image borders on all sides)
    B, H, W, ldy, coff = (3, 40, 48, Cout + 32, 16) if Cout == 16 else (2, 32 if k == 5 else 24, 48 if k == 5 else 32, Cout + 32, 16)
    gq = q(torch.randn((B, Cout, H, W), generator=g_), dt)
    yq = q(torch.randn((B, ldy, H, W), generator=g_), dt)
    ca, cb, cc = (torch.randn(Cout, generator=g_) * s_ for s_ in (1.0, 0.3, 0.05))
    dy = q(ca.view(1, -1, 1, 1) * gq + cb.view(1, -1, 1, 1) * yq[:, coff:coff + Cout] + cc.view(1, -1, 1, 1), dt)
    x = q(torch.randn((B, Cin, H, W), generator=g_), dt)
    sc, sh = torch.rand(Cin, generator=g_) * 2 - 0.6, torch.randn(Cin, generator=g_) * 0.3
    sl = torch.zeros(Cin)
    a = q(act(x, sc, sh, sl), dt) if qt else x
    w = torch.zeros((Cout, Cin, k, k), requires_grad=True)
    F.conv2d(a, w, None, padding=k // 2).backward(dy)
    gd, yd, xd = U.nhwc(gq, dt), U.nhwc(yq, dt), U.nhwc(x, dt)
    pcoef = tuple(t.to(U.DEV) for t in (ca, cc, cb))   # (scale, shift, slope) = (ca, cc, cb)
    qcoef = tuple(t.to(U.DEV) for t in (sc, sh, sl)) if qt else None
    out = torch.zeros((B, H, W, Cout), dtype=U.tdt(dt), device=U.DEV)
    d = L.WgradDesc()
    U.fill_src(d.p, gd, H, W, Cout, pcoef)
    U.fill_src(d.q, xd, H, W, Cin, qcoef)
    d.dtype_p, d.dtype_q, d.dtype_c = dt, dt, dt
    d.B, d.Hg, d.Wg, d.Hq, d.Wq, d.Ca, d.Cb, d.stride, d.nsplit = B, H, W, H, W, Cout, Cin, 1, 3
    L.set_taps(d, taps_square(k))
    d.p2, d.ld_p2, d.cp2_off, d.p_dual, d.p_out, d.ld_pout = yd.data_ptr(), ldy, coff, 1, out.data_ptr(), Cout
    assert lib.abc_wgrad_fuses_apply(C.byref(d)) == 1
    ca_, cb_ = L.i32(), L.i32()
    L.check(lib.abc_wgrad_pads(C.byref(d), C.byref(ca_), C.byref(cb_)), "pads")
    part = torch.zeros(3 * k * k * ca_.value * cb_.value, dtype=torch.float32, device=U.DEV)
    d.partial = part.data_ptr()
    L.check(lib.abc_wgrad(C.byref(d), U.stream()), "wgrad")
    dw = torch.zeros((Cout, Cin, k * k), dtype=torch.float32, device=U.DEV)
    r = L.WgradReduceDesc()
    r.partial, r.nsplit, r.ntaps, r.Ca, r.Cb, r.Ca_pad, r.Cb_pad, r.dw, r.accumulate = part.data_ptr(), 3, k * k, Cout, Cin, ca_.value, cb_.value, dw.data_ptr(), 0
    L.check(lib.abc_wgrad_reduce(C.byref(r), U.stream()), "reduce")
    torch.cuda.synchronize()
    assert U.relerr(U.to_nchw(out), dy) < 1e-2           # bf16 rounding of the same f32 formula
    assert U.relerr(dw.cpu().view(Cout, Cin, k, k), w.grad) < U.tol(dt)


@pytest.mark.parametrize("k,Cout,W,nsplit", [(3, 16, 72, 7), (5, 32, 72, 7), (3, 16, 70, 5), (5, 32, 40, 3), (5, 8, 64, 11), (3, 32, 384, 9)])
def test_wgrad_one_channel_fused_bn_apply(lib, k, Cout, W, nsplit):
    """the first convolution's weight gradient (unet.py:12 with one input channel: the plain-FMA kernel) with p_dual: the
    BatchNorm-backward correction on load of g and y_raw, the corrected tensor written out -- against the explicit formula
    followed by torch's weight gradient over the f32 image.  3 x 3 (unet.py) and 5 x 5 (unet2.py:135) stems; widths that are whole
    pixel quads run the four-pixel form (splits that cross image boundaries and leave partial 8-row passes), W = 70 the scalar one"""
    dt = L.BF16
    g_ = torch.Generator().manual_seed(52)
    B, H, ldy, coff = 2, 40, 64, 16
    gq = q(torch.randn((B, Cout, H, W), generator=g_), dt)
    yq = q(torch.randn((B, ldy, H, W), generator=g_), dt)
    ca, cb, cc = (torch.randn(Cout, generator=g_) * s_ for s_ in (1.0, 0.3, 0.05))
    dy32 = ca.view(1, -1, 1, 1) * gq + cb.view(1, -1, 1, 1) * yq[:, coff:coff + Cout] + cc.view(1, -1, 1, 1)
    img = torch.randn((B, 1, H, W), generator=g_)
    w = torch.zeros((Cout, 1, k, k), requires_grad=True)
    F.conv2d(img, w, None, padding=k // 2).backward(dy32)         # (the kernel multiplies the UNROUNDED corrected value)
    gd, yd = U.nhwc(gq, dt), U.nhwc(yq, dt)
    xd = img.reshape(B, H, W, 1).contiguous().to(U.DEV)
    pcoef = tuple(t.to(U.DEV) for t in (ca, cc, cb))
    out = torch.zeros((B, H, W, Cout), dtype=U.tdt(dt), device=U.DEV)
    d = L.WgradDesc()
    U.fill_src(d.p, gd, H, W, Cout, pcoef)
    U.fill_src(d.q, xd, H, W, 1, None)
    d.dtype_p, d.dtype_q, d.dtype_c = dt, L.F32, dt
    d.B, d.Hg, d.Wg, d.Hq, d.Wq, d.Ca, d.Cb, d.stride, d.nsplit = B, H, W, H, W, Cout, 1, 1, nsplit
    L.set_taps(d, taps_square(k))
    d.p2, d.ld_p2, d.cp2_off, d.p_dual, d.p_out, d.ld_pout = yd.data_ptr(), ldy, coff, 1, out.data_ptr(), Cout
    assert lib.abc_wgrad_fuses_apply(C.byref(d)) == 1
    at_, bt_ = L.i32(), L.i32()
    L.check(lib.abc_wgrad_tile(C.byref(d), C.byref(at_), C.byref(bt_)), "tile")
    assert (at_.value, bt_.value) == (0, 1), "not the one-channel kernel"
    ca_, cb_ = L.i32(), L.i32()
    L.check(lib.abc_wgrad_pads(C.byref(d), C.byref(ca_), C.byref(cb_)), "pads")
    part = torch.zeros(nsplit * k * k * ca_.value * cb_.value, dtype=torch.float32, device=U.DEV)
    d.partial = part.data_ptr()
    L.check(lib.abc_wgrad(C.byref(d), U.stream()), "wgrad")
    dw = torch.zeros((Cout, 1, k * k), dtype=torch.float32, device=U.DEV)
    r = L.WgradReduceDesc()
    r.partial, r.nsplit, r.ntaps, r.Ca, r.Cb, r.Ca_pad, r.Cb_pad, r.dw, r.accumulate = part.data_ptr(), nsplit, k * k, Cout, 1, ca_.value, cb_.value, dw.data_ptr(), 0
    L.check(lib.abc_wgrad_reduce(C.byref(r), U.stream()), "reduce")
    torch.cuda.synchronize()
    assert U.relerr(U.to_nchw(out), dy32) < 1e-2
    assert U.relerr(dw.cpu().view(Cout, 1, k, k), w.grad) < 2e-3


@pytest.mark.parametrize("dt", DTS)
def test_conv_transpose_into_concat(lib, dt):
    """4 parity phases == ConvTranspose2d(k3,s2) + crop of first row/col, written at a channel offset"""
    g = torch.Generator().manual_seed(5)
    B, Cin, n = 2, 64, 12
    half = Cin // 2
    x = q(torch.randn((B, Cin, n, n), generator=g), dt)
    w = torch.randn((Cin, half, 3, 3), generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(half, generator=g)
    ref = F.conv_transpose2d(x, q(w, dt), b, stride=2)[:, :, 1:, 1:]
    cat = torch.full((B, 2 * n, 2 * n, Cin), 7.0, dtype=U.tdt(dt), device=U.DEV)
    xd = U.nhwc(x, dt)
    for py in (0, 1):
        for px in (0, 1):
            wp = U.pack(lib, w.to(U.DEV), 2, dt, half, Cin, 3, half, Cin, py=py, px=px)
            U.conv(lib, xd, dt, dt, B, n, n, Cin, 0, Cin, wp, b.to(U.DEV), half, convT_phase_taps(py, px), 2 * n, 2 * n, ldy=Cin,
                   cout_off=half, grid=(n, n), om=2, oy0=py, ox0=px, out=cat)
    torch.cuda.synchronize()
    got = U.to_nchw(cat)
    assert U.relerr(got[:, half:], ref) < U.tol(dt)
    assert torch.all(got[:, :half] == 7.0)  # the skip half is untouched


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [dict(Cin=32, Cout=64, k=3), dict(Cin=16, Cout=16, k=3), dict(Cin=64, Cout=8, k=1),
                                  dict(Cin=32, Cout=32, k=5)])
def test_conv_dgrad(lib, dt, case):
    g = torch.Generator().manual_seed(7)
    B, H, W, Cin, Cout, k = 2, 16, 24, case["Cin"], case["Cout"], case["k"]
    x = torch.randn((B, Cin, H, W), generator=g, requires_grad=True)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    dy = q(torch.randn((B, Cout, H, W), generator=g), dt)
    F.conv2d(x, q(w, dt), None, padding=(k - 1) // 2).backward(dy)
    wd = U.pack(lib, w.to(U.DEV), 1, dt, Cout, Cin, k, -(-Cin // 32) * 32, Cout)
    dx, _ = U.conv(lib, U.nhwc(dy, dt), dt, dt, B, H, W, Cout, 0, Cout, wd, None, Cin, taps_mirror(taps_square(k)), H, W)
    torch.cuda.synchronize()
    assert U.relerr(U.to_nchw(dx), x.grad) < U.tol(dt)


@pytest.mark.parametrize("dt", DTS)
def test_conv_transpose_dgrad_wgrad(lib, dt):
    g = torch.Generator().manual_seed(9)
    B, Cin, n = 2, 64, 12
    half = Cin // 2
    x = q(torch.randn((B, Cin, n, n), generator=g), dt).requires_grad_(True)
    w = q(torch.randn((Cin, half, 3, 3), generator=g) / (Cin * 9) ** 0.5, dt).requires_grad_(True)
    dout = q(torch.randn((B, half, 2 * n, 2 * n), generator=g), dt)
    F.conv_transpose2d(x, w, None, stride=2)[:, :, 1:, 1:].backward(dout)
    # gradient arrives inside a concat-gradient buffer at channel offset `half`
    dcat = torch.zeros((B, 2 * n, 2 * n, Cin), dtype=U.tdt(dt), device=U.DEV)
    dcat[..., half:] = U.nhwc(dout, dt)
    wd = U.pack(lib, w.detach().to(U.DEV), 3, dt, half, Cin, 3, Cin, half)
    dx, _ = U.conv(lib, dcat, dt, dt, B, 2 * n, 2 * n, Cin, half, half, wd, None, Cin, TAPS_CONVT_DGRAD, n, n, stride=2)
    dw = U.wgrad(lib, U.nhwc(x.detach(), dt), dt, n, n, Cin, 0, Cin, None, dcat, dt, 2 * n, 2 * n, Cin, half, half, None, False, dt, B,
                 TAPS_CONVT_DGRAD, stride=2)
    torch.cuda.synchronize()
    assert U.relerr(U.to_nchw(dx), x.grad) < U.tol(dt)
    assert U.relerr(dw.cpu().view(Cin, half, 3, 3), w.grad) < U.tol(dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [dict(Cin=64, Cout=128, k=3), dict(Cin=16, Cout=16, k=3), dict(Cin=32, Cout=64, k=3, pool=True),
                                  dict(Cin=1, Cout=16, k=3, img=True), dict(Cin=1, Cout=32, k=3, img=True, H=40, W=56), dict(Cin=128, Cout=14, k=1, f32dy=True),
                                  dict(Cin=32, Cout=32, k=5),
                                  # 25 taps: tap-split weight gradient with 16-row patches; 25-tap one-channel kernel
                                  dict(Cin=32, Cout=32, k=5, H=32, W=48), dict(Cin=1, Cout=32, k=5, img=True)])
def test_conv_wgrad(lib, dt, case):
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout, k = 2, case.get("H", 24), case.get("W", 40), case["Cin"], case["Cout"], case["k"]
    pool, img = case.get("pool", False), case.get("img", False)
    Hx, Wx = (2 * H, 2 * W) if pool else (H, W)
    dt_q = L.F32 if img else dt
    x = torch.randn((B, Cin, Hx, Wx), generator=g)
    x = x if img else q(x, dt)
    sc = torch.rand(Cin, generator=g) * 2 - 0.6
    sh = torch.randn(Cin, generator=g) * 0.3
    sl = torch.tensor([0.0, 0.01, 1.0])[torch.randint(0, 3, (Cin,), generator=g)]
    coef = None if img else tuple(t.to(U.DEV) for t in (sc, sh, sl))
    a = x if img else act(x, sc, sh, sl)
    if pool:
        a = F.max_pool2d(a, 2)
    w = torch.zeros((Cout, Cin, k, k), requires_grad=True)
    dt_p = L.F32 if case.get("f32dy") else dt
    dy = torch.randn((B, Cout, H, W), generator=g)
    dy = dy if case.get("f32dy") else q(dy, dt)
    F.conv2d(q(a, dt), w, None, padding=(k - 1) // 2).backward(q(dy, dt))
    xd = x.permute(0, 2, 3, 1).contiguous().to(U.tdt(dt_q)).to(U.DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(U.tdt(dt_p)).to(U.DEV)
    dw = U.wgrad(lib, dyd, dt_p, H, W, Cout, 0, Cout, None, xd, dt_q, Hx, Wx, Cin, 0, Cin, coef, pool, dt, B, taps_square(k))
    torch.cuda.synchronize()
    assert U.relerr(dw.cpu().view(Cout, Cin, k, k), w.grad) < U.tol(dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("pooled,same", [(False, True), (True, False), (True, True)])
def test_bn_act_pool_backward(lib, dt, pooled, same):
    """conv stats -> BN finalize (fwd), then act_bwd + bn_finalize_bwd + bn_apply_bwd == autograd of
    batch_norm(train) -> relu/leaky -> (maxpool) with up to two gradient sources"""
    g = torch.Generator().manual_seed(13)
    B, Cc, H, W = 2, 32, 16, 24
    y = q(torch.randn((B, Cc, H, W), generator=g) * 1.5 + 0.3, dt).requires_grad_(True)
    gamma = (torch.rand(Cc, generator=g) * 2 - 0.5).requires_grad_(True)
    beta = (torch.randn(Cc, generator=g) * 0.3).requires_grad_(True)
    slope = 0.01
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    z = F.leaky_relu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5), slope)
    loss = 0
    d_same = q(torch.randn((B, Cc, H, W), generator=g), dt)
    d_pool = q(torch.randn((B, Cc, H // 2, W // 2), generator=g), dt)
    if same:
        loss = loss + (z * d_same).sum()
    if pooled:
        loss = loss + (F.max_pool2d(z, 2) * d_pool).sum()
    loss.backward()
    # ---- device: statistics straight from y (as the conv epilogue would deliver them)
    yd = U.nhwc(y.detach(), dt)
    yf = y.detach().double()
    part = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]).float().view(1, 2, Cc).to(U.DEV)
    f32 = lambda n, v=0.0: torch.full((n,), v, dtype=torch.float32, device=U.DEV)
    scale, shift, mean, invstd, rmd, rvd = f32(Cc), f32(Cc), f32(Cc), f32(Cc), f32(Cc), f32(Cc, 1.0)
    nbt = torch.zeros(1, dtype=torch.int64, device=U.DEV)
    gd, bd = gamma.detach().to(U.DEV), beta.detach().to(U.DEV)
    d = L.BnFwdDesc()
    d.partial, d.nblk, d.C, d.count = part.data_ptr(), 1, Cc, float(B * H * W)
    d.gamma, d.beta, d.scale, d.shift, d.mean, d.invstd = (t.data_ptr() for t in (gd, bd, scale, shift, mean, invstd))
    d.running_mean, d.running_var, d.num_batches_tracked, d.eps, d.momentum = rmd.data_ptr(), rvd.data_ptr(), nbt.data_ptr(), 1e-5, 0.1
    L.check(lib.abc_bn_finalize_fwd(C.byref(d), U.stream()), "bn_fwd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(rmd.cpu(), rm, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu(), rv, rtol=1e-5, atol=1e-6)
    assert nbt.item() == 1
    slopes = f32(Cc, slope)
    a = L.ActBwdDesc()
    gbuf = torch.zeros((B, H, W, Cc), dtype=U.tdt(dt), device=U.DEV)
    a.y_raw, a.ld_y, a.g, a.ld_g = yd.data_ptr(), Cc, gbuf.data_ptr(), Cc
    ds_d, dp_d = U.nhwc(d_same, dt), U.nhwc(d_pool, dt)
    if same:
        a.dA_same, a.ld_same = ds_d.data_ptr(), Cc
    if pooled:
        a.dA_pool, a.ld_pool = dp_d.data_ptr(), Cc
    a.scale, a.shift, a.slope, a.mean, a.invstd = (t.data_ptr() for t in (scale, shift, slopes, mean, invstd))
    a.dtype, a.B, a.H, a.W, a.C = dt, B, H, W, Cc
    nblk = lib.abc_act_bwd_blocks(C.byref(a))
    p2 = torch.zeros((nblk, 2, Cc), dtype=torch.float32, device=U.DEV)
    a.partial = p2.data_ptr()
    L.check(lib.abc_act_bwd(C.byref(a), U.stream()), "act_bwd")
    dgam, dbet, k1, k2, gs = f32(Cc), f32(Cc), f32(Cc), f32(Cc), f32(Cc)
    f = L.BnBwdDesc()
    f.partial, f.nblk, f.C, f.count, f.gamma, f.invstd = p2.data_ptr(), nblk, Cc, float(B * H * W), gd.data_ptr(), invstd.data_ptr()
    f.dgamma, f.dbeta, f.k1, f.k2, f.gscale = (t.data_ptr() for t in (dgam, dbet, k1, k2, gs))
    L.check(lib.abc_bn_finalize_bwd(C.byref(f), U.stream()), "bn_bwd")
    ap = L.BnApplyDesc()
    ap.g, ap.ld_g, ap.y_raw, ap.ld_y = gbuf.data_ptr(), Cc, yd.data_ptr(), Cc
    ap.mean, ap.invstd, ap.k1, ap.k2, ap.gscale = (t.data_ptr() for t in (mean, invstd, k1, k2, gs))
    ap.dtype, ap.C, ap.npix = dt, Cc, B * H * W
    L.check(lib.abc_bn_apply_bwd(C.byref(ap), U.stream()), "bn_apply")
    torch.cuda.synchronize()
    t = U.tol(dt, 1e-4, 3e-2)
    assert U.relerr(dgam.cpu(), gamma.grad) < t
    assert U.relerr(dbet.cpu(), beta.grad) < t
    assert U.relerr(U.to_nchw(gbuf), y.grad) < t


@pytest.mark.parametrize("Cc,nblk,batch", [(16, 2304, 0), (128, 768, 0), (32, 100, 0), (256, 2048, 0), (128, 1152, 3), (48, 4099, 0), (12, 300, 0)])
def test_bn_finalisers_many_rows(lib, Cc, nblk, batch):
    """BatchNorm finalisers over many partial rows against f64 sums, single and batched with a column stride"""
    g = torch.Generator().manual_seed(5)
    n = max(batch, 1)
    Ct = Cc * n
    part = (torch.randn((nblk, 2, Ct), generator=g) * 3 + 0.5)
    part[:, 1] = part[:, 1].abs() * 4 + 30       # (a valid sum of squares)
    pd = part.to(U.DEV)
    count = float(nblk * 7)
    f32 = lambda k, v=0.0: torch.full((k,), v, dtype=torch.float32, device=U.DEV)
    gam, bet, mean_in, istd_in = ((torch.rand(Ct, generator=g) + 0.5).to(U.DEV), torch.randn(Ct, generator=g).to(U.DEV),
                                  torch.randn(Ct, generator=g).to(U.DEV), (torch.rand(Ct, generator=g) + 0.5).to(U.DEV))
    outs_f = [f32(Ct) for _ in range(4)] + [f32(Ct), f32(Ct, 1.0)]
    outs_b = [f32(Ct) for _ in range(8)]
    nbt = torch.zeros(n, dtype=torch.int64, device=U.DEV)
    fa, ba = (L.BnFwdDesc * n)(), (L.BnBwdDesc * n)()
    for i in range(n):
        o = 4 * Cc * i
        d = fa[i]
        d.partial, d.nblk, d.C, d.count, d.rows = pd.data_ptr() + o, nblk, Cc, count, 2
        d.gamma, d.beta = gam.data_ptr() + o, bet.data_ptr() + o
        d.scale, d.shift, d.mean, d.invstd, d.running_mean, d.running_var = (t.data_ptr() + o for t in outs_f)
        d.num_batches_tracked, d.eps, d.momentum = nbt.data_ptr() + 8 * i, 1e-5, 0.1
        f = ba[i]
        f.partial, f.nblk, f.C, f.count = pd.data_ptr() + o, nblk, Cc, count
        f.gamma, f.invstd, f.mean = gam.data_ptr() + o, istd_in.data_ptr() + o, mean_in.data_ptr() + o
        f.dgamma, f.dbeta, f.k1, f.k2, f.gscale, f.ca, f.cb, f.cc = (t.data_ptr() + o for t in outs_b)
    if batch:
        L.check(lib.abc_bn_finalize_fwd_batch(fa, n, Ct, U.stream()), "bn_fwd_batch")
        L.check(lib.abc_bn_finalize_bwd_batch(ba, n, Ct, U.stream()), "bn_bwd_batch")
    else:
        L.check(lib.abc_bn_finalize_fwd(C.byref(fa[0]), U.stream()), "bn_fwd")
        L.check(lib.abc_bn_finalize_bwd(C.byref(ba[0]), U.stream()), "bn_bwd")
    torch.cuda.synchronize()
    s1, s2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    mean = s1 / count
    var = (s2 / count - mean * mean).clamp_min(0)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift, mean_d, invstd_d, rm, rv = (t.cpu() for t in outs_f)
    np.testing.assert_allclose(mean_d, mean.float(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(invstd_d, invstd.float(), rtol=2e-6)
    np.testing.assert_allclose(scale, (gam.cpu().double() * invstd).float(), rtol=2e-6)
    np.testing.assert_allclose(shift, (bet.cpu().double() - mean * gam.cpu().double() * invstd).float(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rm, 0.1 * mean.float(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv, (0.9 + 0.1 * var * count / (count - 1)).float(), rtol=1e-5)
    assert (nbt.cpu() == 1).all()
    dgam, dbet, k1, k2, gs, ca, cb, cc = (t.cpu() for t in outs_b)
    np.testing.assert_allclose(dgam, s2.float(), rtol=1e-6)
    np.testing.assert_allclose(dbet, s1.float(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(k1, (s1 / count).float(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(k2, (s2 / count).float(), rtol=1e-6)
    gsr = (gam.cpu() * istd_in.cpu())
    np.testing.assert_allclose(gs, gsr, rtol=1e-6)
    np.testing.assert_allclose(ca, gsr, rtol=1e-6)
    np.testing.assert_allclose(cb, -gsr * k2 * istd_in.cpu(), rtol=1e-5)
    np.testing.assert_allclose(cc, gsr * (mean_in.cpu() * istd_in.cpu() * k2 - k1), rtol=1e-4, atol=1e-4)


def test_adam_matches_oracle(lib, golden_dir):
    from oracle import adam_oracle
    from abcnet_amd.ops import FusedAdam
    gold = np.load(golden_dir + "/adam.npz")
    p = torch.from_numpy(gold["p0"].copy()).to(U.DEV)
    gr = torch.zeros_like(p)
    opt = FusedAdam(p, gr)
    pc, m, v = torch.from_numpy(gold["p0"].copy()), torch.zeros(p.numel()), torch.zeros(p.numel())
    for it in range(3):
        gr.copy_(torch.from_numpy(gold["g%d" % it]))
        opt.step()
        adam_oracle.adam_step(pc, torch.from_numpy(gold["g%d" % it]), m, v, it + 1)
        torch.cuda.synchronize()
        np.testing.assert_allclose(p.cpu().numpy(), gold["p%d" % (it + 1)], rtol=2e-6, atol=2e-7)  # torch.optim.Adam
        np.testing.assert_allclose(p.cpu().numpy(), pc.numpy(), rtol=2e-6, atol=2e-7)              # oracle


def test_layout_roundtrip(lib):
    g = torch.Generator().manual_seed(21)
    x = torch.randn((3, 37, 9, 11), generator=g)
    xd = x.to(U.DEV)
    nh = torch.zeros((3, 9, 11, 64), dtype=torch.float32, device=U.DEV)
    L.check(lib.abc_nchw_to_nhwc_f32(xd.data_ptr(), 37, 3, 9, 11, nh.data_ptr(), 64, 5, U.stream()), "to_nhwc")
    back = torch.zeros_like(xd)
    L.check(lib.abc_nhwc_to_nchw_f32(nh.data_ptr(), 64, 5, 37, 3, 9, 11, back.data_ptr(), U.stream()), "to_nchw")
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), x)
    assert torch.equal(nh[..., 5:42].cpu(), x.permute(0, 2, 3, 1))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [dict(C=32, ld=32, off=0), dict(C=64, ld=128, off=64, scale=True), dict(C=256, ld=512, off=256),
                                  dict(C=14, ld=14, off=0), dict(C=128, ld=128, off=0, npix=777)])
def test_colsum(lib, dt, case):
    """abc_colsum: per-channel sums over pixels of a channel slice of an NHWC tensor (bias gradients of convT / 1x1
    convs) -- vector kernel for 16-byte-aligned slices, scalar kernel otherwise (C = 14)"""
    import ctypes as C_
    g = torch.Generator().manual_seed(21)
    C, ld, off = case["C"], case["ld"], case["off"]
    npix = case.get("npix", 2 * 48 * 40)
    x = q(torch.randn((npix, ld), generator=g), dt)
    cs = (torch.rand(ld, generator=g) + 0.5) if case.get("scale") else None
    ref = x[:, off:off + C].double().sum(0)
    if cs is not None:
        ref = ref * cs[off:off + C].double()
    xd = x.to(U.tdt(dt)).to(U.DEV)
    nb = lib.abc_colsum_blocks(npix)
    work = torch.zeros(nb * C, dtype=torch.float32, device=U.DEV)
    out = torch.zeros(C, dtype=torch.float32, device=U.DEV)
    csd = None if cs is None else cs.to(U.DEV)
    L.check(lib.abc_colsum(xd.data_ptr(), dt, npix, ld, off, C, None if csd is None else csd.data_ptr(), work.data_ptr(), out.data_ptr(),
                           U.stream()), "colsum")
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-4


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C,ld,off,npix", [(32, 32, 0, 2 * 48 * 40), (32, 64, 32, 3 * 72 * 88 + 5), (64, 64, 0, 777)])
def test_colsum_weighted_by_a_one_channel_image(lib, dt, C, ld, off, npix):
    """abc_colsum_w1: column sums of d(out) and of d(out) * image[pixel] in one pass = bias and weight gradient of a 1x1 convolution
    over a one-channel input (unet2.py:62,135: res_conv of the first block), against torch's conv2d autograd and f64 sums"""
    g = torch.Generator().manual_seed(23)
    x = q(torch.randn((npix, ld), generator=g), dt)
    img = torch.rand(npix, generator=g)
    xs = x[:, off:off + C]
    w = torch.zeros((C, 1, 1, 1), requires_grad=True)
    b = torch.zeros(C, requires_grad=True)
    F.conv2d(img.view(1, 1, 1, npix), w, b).backward(xs.t().reshape(1, C, 1, npix).float())
    xd = x.to(U.tdt(dt)).to(U.DEV)
    nb = lib.abc_colsum_blocks(npix)
    work = torch.zeros(nb * 2 * C, dtype=torch.float32, device=U.DEV)
    osum = torch.zeros(C, dtype=torch.float32, device=U.DEV)
    ow = torch.zeros(C, dtype=torch.float32, device=U.DEV)
    imgd = img.to(U.DEV)
    L.check(lib.abc_colsum_w1(xd.data_ptr(), dt, npix, ld, off, C, imgd.data_ptr(), work.data_ptr(), osum.data_ptr(), ow.data_ptr(), U.stream()),
            "colsum_w1")
    torch.cuda.synchronize()
    r_sum = xs.double().sum(0)
    r_w = (xs.double() * img.double().view(-1, 1)).sum(0)
    scale = xs.abs().double().sum(0).max().item()
    assert (osum.cpu().double() - r_sum).abs().max().item() <= 1e-5 * scale + 1e-4
    assert (ow.cpu().double() - r_w).abs().max().item() <= 1e-5 * scale + 1e-4
    assert (ow.cpu() - w.grad.view(-1)).abs().max().item() <= 1e-4 * scale + 1e-3 and (osum.cpu() - b.grad).abs().max().item() <= 1e-4 * scale + 1e-3
    # the bias row is optional
    ow2 = torch.zeros_like(ow)
    L.check(lib.abc_colsum_w1(xd.data_ptr(), dt, npix, ld, off, C, imgd.data_ptr(), work.data_ptr(), None, ow2.data_ptr(), U.stream()), "colsum_w1")
    torch.cuda.synchronize()
    assert torch.equal(ow2, ow)


def test_conv_rows_packed_side_by_side_and_concat(lib):
    """abc_pack_desc.rows_total / rows_off + abc_concat_f32: three convolutions over the same input as ONE convolution whose
    packed weight holds their rows one below the other (how the eight heads' conv1, unet.py:66,116-118, run), per-block
    BatchNorm partials in one [nblk][2][3 x 128] buffer finalised with a column stride (abc_bn_finalize_fwd_batch)"""
    dt = L.BF16
    g = torch.Generator().manual_seed(77)
    B, Cin, Cout, n, H, W = 2, 128, 128, 3, 24, 32
    x = q(torch.randn((B, Cin, H, W), generator=g), dt)
    ws = [torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5 for _ in range(n)]
    bs = [torch.randn(Cout, generator=g) for _ in range(n)]
    Ct = n * Cout
    ck = lib.abc_conv_chunk(dt, Cin)
    wp = torch.zeros(9 * Cin * Ct, dtype=torch.bfloat16, device=U.DEV)
    keep = []
    for i in range(n):
        wd = ws[i].to(U.DEV)
        keep.append(wd)
        d = L.PackDesc()
        d.w, d.dst, d.mode, d.dtype_c = wd.data_ptr(), wp.data_ptr(), 0, dt
        d.Cout, d.Cin, d.kh, d.kw = Cout, Cin, 3, 3
        d.rows_pad, d.red_pad, d.red_total, d.red_off, d.ck = Cout, Cin, Cin, 0, ck
        d.rows_total, d.rows_off = Ct, Cout * i
        L.check(lib.abc_pack_conv_weights(C.byref(d), U.stream()), "pack")
    bd = [b.to(U.DEV) for b in bs]
    bias_all = torch.zeros(Ct, device=U.DEV)
    srcs = (C.c_void_p * n)(*[b.data_ptr() for b in bd])
    counts = (C.c_int32 * n)(*([Cout] * n))
    L.check(lib.abc_concat_f32(srcs, counts, n, bias_all.data_ptr(), U.stream()), "concat")
    assert torch.equal(bias_all.cpu(), torch.cat(bs))
    xd = U.nhwc(x, dt)
    y, st = U.conv(lib, xd, dt, dt, B, H, W, Cin, 0, Cin, wp, bias_all, Ct, taps_square(3), H, W, stats=True)
    torch.cuda.synchronize()
    ref = F.conv2d(x, q(torch.cat(ws), dt), torch.cat(bs), padding=1)
    got = U.to_nchw(y)
    assert U.relerr(got, ref) < U.tol(dt), U.relerr(got, ref)
    # the three layers' statistics from column slices of the one partial buffer
    nblk = st.shape[0]
    outs = [[torch.zeros(Cout, device=U.DEV) for _ in range(4)] for _ in range(n)]
    gam = torch.ones(Cout, device=U.DEV)
    bet = torch.zeros(Cout, device=U.DEV)
    arr = (L.BnFwdDesc * n)()
    for i in range(n):
        f = arr[i]
        f.partial, f.nblk, f.C, f.count, f.rows = st.data_ptr() + 4 * Cout * i, nblk, Cout, float(B * H * W), 2
        f.gamma, f.beta = gam.data_ptr(), bet.data_ptr()
        f.scale, f.shift, f.mean, f.invstd = (t.data_ptr() for t in outs[i])
        f.eps, f.momentum = 1e-5, 0.1
    L.check(lib.abc_bn_finalize_fwd_batch(arr, n, Ct, U.stream()), "bn_fwd_batch")
    torch.cuda.synchronize()
    for i in range(n):
        r = ref[:, Cout * i:Cout * (i + 1)].double()
        np.testing.assert_allclose(outs[i][2].cpu().numpy(), r.mean((0, 2, 3)).numpy(), atol=2e-2)
        np.testing.assert_allclose(outs[i][3].cpu().numpy(), (1.0 / torch.sqrt(r.var((0, 2, 3), unbiased=False) + 1e-5)).numpy(), rtol=3e-2)


def test_pack_layout_for_the_weights_direct_loop(lib):
    """abc_pack_desc.layout 1 (what abc_conv_weight_layout() asks for on the 3x3 weights-direct conv loop) is the row-major
    packing with every 32-row x 64-byte block re-ordered [kk][h][row][16 bytes]; forward and data-gradient modes, a second
    weight packed below the first (rows_off)"""
    dt = L.BF16
    g = torch.Generator().manual_seed(5)
    Cout, Cin = 96, 64
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / 24).to(U.DEV)
    for mode, rows, red in ((0, Cout, Cin), (1, Cin, Cout)):
        rows_pad = -(-rows // 32) * 32
        ref = U.pack(lib, w, mode, dt, Cout, Cin, 3, rows_pad, red)
        ck = lib.abc_conv_chunk(dt, red)
        red_pad = -(-red // ck) * ck
        got = torch.zeros_like(ref)
        d = L.PackDesc()
        d.w, d.dst, d.mode, d.dtype_c = w.data_ptr(), got.data_ptr(), mode, dt
        d.Cout, d.Cin, d.kh, d.kw = Cout, Cin, 3, 3
        d.rows_pad, d.red_pad, d.red_total, d.red_off, d.ck, d.layout = rows_pad, red_pad, red, 0, ck, 1
        L.check(lib.abc_pack_conv_weights(C.byref(d), U.stream()), "pack")
        torch.cuda.synchronize()
        want = ref.view(-1, 32, 2, 2, 8).permute(0, 3, 2, 1, 4).contiguous().view(-1)
        assert torch.equal(got, want), mode


@pytest.mark.parametrize("B,H,W,Cin,Cout,slope", [(2, 48, 48, 128, 128, 0.0), (3, 24, 32, 64, 64, 0.01), (2, 32, 16, 128, 256, 0.2),
                                                  (1, 32, 48, 256, 96, 0.0), (5, 96, 96, 128, 128, 0.0),
                                                  (2, 64, 64, 16, 16, 0.0), (3, 40, 56, 16, 16, 0.01), (16, 96, 96, 16, 16, 0.0),
                                                  (2, 32, 48, 32, 32, 0.01), (3, 64, 64, 32, 32, 0.0), (2, 36, 48, 32, 32, 0.2)])
def test_act_bwd_in_the_data_gradient_epilogue(lib, B, H, W, Cin, Cout, slope):
    """abc_conv_desc.actbwd_*: a 3x3 data-gradient convolution that stores g = dA * act'(BatchNorm(y_raw)) and the BatchNorm-backward
    partial sums of the layer it differentiates (autograd of unet.py:12-17) -- against the same convolution followed by abc_act_bwd
    (g within one bf16 rounding of it: the fused form rounds once; sums against f64 sums of the device's own g) and against torch.
    Shapes: whole 192/128/64-pixel tiles of 128- and 64-channel blocks, a 96-channel block with padding lanes, several rounds of
    persistent workgroups (5 x 96 x 96), the 16-channel levels (conv_narrow.hip: whole and ragged tiles, runs of tiles per wave), the
    5x5 32 -> 32 form of unet2.py's first level (conv_n32r2_kernel); a ragged shape of the lean kernel must be refused
    (abc_conv_actbwd_ok == 0) and then runs the plain convolution."""
    dt = L.BF16
    k = 5 if Cin == 32 and H != 36 else 3           # (H = 36: the 3x3 form of the 32 -> 32 kernel)
    g = torch.Generator().manual_seed(11)
    dy = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(U.DEV)          # gradient entering the convolution
    w = torch.randn((Cout, Cin, k, k), generator=g) / (k * Cin ** 0.5)
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, k, -(-Cout // 32) * 32, Cin)
    yraw = (torch.randn((B, H, W, Cout), generator=g) * 1.5 + 0.3).to(torch.bfloat16).to(U.DEV)
    sc = (torch.rand(Cout, generator=g) + 0.5) * (torch.randint(0, 2, (Cout,), generator=g) * 2 - 1).float()      # (both signs)
    sh = torch.randn(Cout, generator=g) * 0.5
    mu = torch.randn(Cout, generator=g) * 0.3 + 0.3
    istd = torch.rand(Cout, generator=g) + 0.5
    sl = torch.full((Cout,), slope)
    scd, shd, sld, mud, isd = (t.to(U.DEV) for t in (sc, sh, sl, mu, istd))
    taps = taps_square(k)
    gfused, part = U.conv(lib, dy, dt, dt, B, H, W, Cin, 0, Cin, wp, None, Cout, taps, H, W, stats=True,
                          actbwd=(yraw, Cout, 0, scd, shd, sld, mud, isd))
    # (16 -> 16 and 5x5 32 -> 32 channels: the narrow-level kernels, whose accumulator layout -- lane = pixel -- needs no staging for
    #  y_raw (the 16-channel one serves ragged shapes too); everything else: the lean kernel, whole tiles)
    assert U.conv.last_actbwd_ok and U.conv.last_variant == (5 if Cin in (16, 32) else 1)
    dA, _ = U.conv(lib, dy, dt, dt, B, H, W, Cin, 0, Cin, wp, None, Cout, taps, H, W)
    torch.cuda.synchronize()
    # torch on the device's own tensors
    dA_ref = F.conv2d(dy.float().cpu().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=k // 2).permute(0, 2, 3, 1)
    ybn = yraw.float().cpu() * sc + sh
    fac = torch.where(ybn > 0, torch.ones(()), sl)
    g_ref = dA_ref * fac
    got = gfused.float().cpu()
    assert U.relerr(got, g_ref) < 3e-2
    # against the separate pass on the device's own (rounded) dA: equal up to one more bf16 rounding where slope != 0, 1
    g_two = dA.float().cpu() * fac
    assert (got - g_two).abs().max().item() <= 2.0 ** -7 * g_two.abs().max().item() + 1e-6
    if slope == 0.0:
        assert torch.equal(got, g_two.to(torch.bfloat16).float())
    # partial sums: f64 sums of the values the kernel summed (its f32 g before the rounding ~ g_ref), per channel
    s1 = part[:, 0].double().sum(0).cpu()
    s2 = part[:, 1].double().sum(0).cpu()
    r1 = g_ref.double().sum((0, 1, 2))
    r2 = (g_ref.double() * (yraw.double().cpu() - mu.double())).sum((0, 1, 2)) * istd.double()
    scale1 = g_ref.abs().double().sum((0, 1, 2)) + 1e-6
    scale2 = (g_ref.abs().double() * (yraw.double().cpu() - mu.double()).abs()).sum((0, 1, 2)) * istd.double() + 1e-6
    assert ((s1 - r1).abs() / scale1).max().item() < 2e-3, ((s1 - r1).abs() / scale1).max().item()
    assert ((s2 - r2).abs() / scale2).max().item() < 2e-3, ((s2 - r2).abs() / scale2).max().item()


def test_act_bwd_epilogue_refuses_ragged_shapes(lib):
    dt = L.BF16
    B, H, W, Cin, Cout = 1, 30, 40, 64, 64
    g = torch.Generator().manual_seed(3)
    dy = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / 20
    wp = U.pack(lib, w.to(U.DEV), 0, dt, Cout, Cin, 3, 64, Cin)
    yraw = torch.randn((B, H, W, Cout), generator=g).to(torch.bfloat16).to(U.DEV)
    one = torch.ones(Cout, device=U.DEV)
    out, _ = U.conv(lib, dy, dt, dt, B, H, W, Cin, 0, Cin, wp, None, Cout, taps_square(3), H, W, stats=True,
                    actbwd=(yraw, Cout, 0, one, one, one, one, one))
    assert not U.conv.last_actbwd_ok
    ref = F.conv2d(dy.float().cpu().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1)
    assert U.relerr(out.float().cpu(), ref) < 3e-2


@pytest.mark.parametrize("B,H,W,Cin,Chalf", [(2, 24, 24, 256, 128), (16, 48, 48, 128, 64), (3, 12, 20, 512, 256), (1, 9, 11, 64, 32)])
def test_conv_transpose_phases_as_one_launch(lib, B, H, W, Cin, Chalf):
    """abc_conv_fwd_batch: the four output-parity phases of ConvTranspose2d(k3, s2) + the reference's crop (unet.py:44,51-56) issued as
    ONE launch where they share a tile geometry of the lean kernel (abc_conv_batch_ok), one by one otherwise -- the same bytes as four
    abc_conv_fwd calls either way, and torch's conv_transpose2d within the bf16 bar"""
    from abcnet_amd.engine import convT_pack_parity
    dt = L.BF16
    g = torch.Generator().manual_seed(4)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cin, Chalf, 3, 3), generator=g) / (2.0 * Cin ** 0.5)       # ConvTranspose2d weight layout
    bias = torch.randn(Chalf, generator=g).to(U.DEV)
    Hs, Ws, Ctot = 2 * H, 2 * W, 2 * Chalf
    rows_pad = -(-Chalf // 32) * 32

    def run(batched):
        cat = torch.zeros((B, Hs, Ws, Ctot), dtype=torch.bfloat16, device=U.DEV)
        items = []
        for py in (0, 1):
            for px in (0, 1):
                taps = convT_phase_taps(py, px, True, True)
                wp = U.pack(lib, w.to(U.DEV), 2, dt, Chalf, Cin, 3, rows_pad, Cin, py=convT_pack_parity(py, True), px=convT_pack_parity(px, True))
                U.conv(lib, x, dt, dt, B, H, W, Cin, 0, Cin, wp, bias, Chalf, taps, Hs, Ws, ldy=Ctot, cout_off=Chalf, grid=((Hs - py + 1) // 2, (Ws - px + 1) // 2),
                       om=2, oy0=py, ox0=px, out=cat, defer=items if batched else None)
        ok = None
        if batched:
            arr = (L.ConvDesc * 4)(*[d for d, _w, _s in items])
            ok = bool(lib.abc_conv_batch_ok(arr, 4))
            L.check(lib.abc_conv_fwd_batch(arr, 4, U.stream()), "conv_fwd_batch")
        torch.cuda.synchronize()
        return cat, ok

    one, ok = run(True)
    four, _ = run(False)
    assert torch.equal(one, four)
    if (B, H) in ((2, 24), (16, 48)):
        assert ok          # (the benchmark's up-layers: one launch)
    y = F.conv_transpose2d(x.float().cpu().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias.cpu(), stride=2)
    y = y[:, :, y.shape[2] - Hs:, y.shape[3] - Ws:]
    assert U.relerr(one[..., Chalf:].float().cpu().permute(0, 3, 1, 2), y) < 3e-2
    assert float(one[..., :Chalf].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,W,Cin,Chalf,coef", [(16, 12, 12, 512, 256, True), (2, 24, 24, 256, 128, True), (16, 48, 48, 128, 64, False), (3, 10, 20, 64, 64, True),
                                                  (1, 7, 33, 96, 72, True)])
def test_conv_transpose_all_phases_in_one_pass(lib, B, H, W, Cin, Chalf, coef):
    """abc_convt_fused_fwd (convt_fused.hip): ConvTranspose2d(k3, s2) + the reference's crop (unet.py:44,51-56) with all four output-parity
    phases in one pass over the input -- bit-identical to the four abc_conv_fwd phase calls it replaces (the same products summed in the
    same order per output), with and without the producer's BatchNorm + ReLU on load, on the benchmark's three up-layers, a ragged map
    (10 x 20: tiles that stick out in both directions) and a channel count that leaves padding rows (72 of 96); torch within the bf16 bar"""
    from abcnet_amd.engine import convT_pack_parity
    dt = L.BF16
    g = torch.Generator().manual_seed(4)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(U.DEV)
    w = torch.randn((Cin, Chalf, 3, 3), generator=g) / (2.0 * Cin ** 0.5)       # ConvTranspose2d weight layout
    bias = torch.randn(Chalf, generator=g).to(U.DEV)
    cf = tuple(t.to(U.DEV) for t in (torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3, torch.zeros(Cin))) if coef else None
    Hs, Ws, Ctot = 2 * H, 2 * W, 2 * Chalf
    rows_pad = -(-Chalf // 64) * 64
    # reference: the four phase convolutions
    four = torch.zeros((B, Hs, Ws, Ctot), dtype=torch.bfloat16, device=U.DEV)
    for py in (0, 1):
        for px in (0, 1):
            wp = U.pack(lib, w.to(U.DEV), 2, dt, Chalf, Cin, 3, -(-Chalf // 32) * 32, Cin, py=convT_pack_parity(py, True), px=convT_pack_parity(px, True))
            U.conv(lib, x, dt, dt, B, H, W, Cin, 0, Cin, wp, bias, Chalf, convT_phase_taps(py, px, True, True), Hs, Ws, ldy=Ctot, cout_off=Chalf,
                   grid=((Hs - py + 1) // 2, (Ws - px + 1) // 2), om=2, oy0=py, ox0=px, out=four, coef=cf)
    # the fused pass: nine slices in one buffer, fragment-contiguous
    ck = lib.abc_conv_chunk(dt, Cin)
    assert ck == 32
    slice_elems = (Cin // ck) * rows_pad * ck
    wall = torch.zeros(9 * slice_elems, dtype=torch.bfloat16, device=U.DEV)
    first = {(0, 0): 0, (0, 1): 1, (1, 0): 3, (1, 1): 5}
    for py in (0, 1):
        for px in (0, 1):
            n = len(convT_phase_taps(py, px, True, True))
            pd = L.PackDesc()
            dst = wall[first[(py, px)] * slice_elems:(first[(py, px)] + n) * slice_elems]
            pd.w, pd.dst, pd.mode, pd.dtype_c = w.to(U.DEV).data_ptr(), dst.data_ptr(), 2, dt
            wdev = w.to(U.DEV)
            pd.w = wdev.data_ptr()
            pd.Cout, pd.Cin, pd.kh, pd.kw, pd.py, pd.px = Chalf, Cin, 3, 3, convT_pack_parity(py, True), convT_pack_parity(px, True)
            pd.rows_pad, pd.red_pad, pd.red_total, pd.red_off, pd.ck, pd.layout = rows_pad, Cin, Cin, 0, ck, 1
            L.check(lib.abc_pack_conv_weights(C.byref(pd), U.stream()), "pack")
            torch.cuda.synchronize()
    one = torch.zeros((B, Hs, Ws, Ctot), dtype=torch.bfloat16, device=U.DEV)
    d = L.ConvTDesc()
    U.fill_src(d.src, x, H, W, Cin, cf)
    d.w, d.bias, d.y, d.dtype = wall.data_ptr(), bias.data_ptr(), one.data_ptr(), dt
    d.B, d.Hin, d.Win, d.cin_off, d.Cin = B, H, W, 0, Cin
    d.Hout, d.Wout, d.ldy, d.cout_off, d.Cout, d.Cout_pad = Hs, Ws, Ctot, Chalf, Chalf, rows_pad
    assert lib.abc_convt_fused_ok(C.byref(d)) == 1
    L.check(lib.abc_convt_fused_fwd(C.byref(d), U.stream()), "convt_fused")
    torch.cuda.synchronize()
    assert torch.equal(one, four), (one.float() - four.float()).abs().max().item()
    xa = x.float().cpu()
    if coef:
        yv = xa * cf[0].cpu() + cf[1].cpu()
        xa = torch.maximum(yv, cf[2].cpu() * yv).to(torch.bfloat16).float()
    y = F.conv_transpose2d(xa.permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias.cpu(), stride=2)
    y = y[:, :, y.shape[2] - Hs:, y.shape[3] - Ws:]
    assert U.relerr(one[..., Chalf:].float().cpu().permute(0, 3, 1, 2), y) < 3e-2
    assert float(one[..., :Chalf].abs().max()) == 0.0
    # not served: an uncropped axis, f32
    d.Hout = 2 * H + 1
    assert lib.abc_convt_fused_ok(C.byref(d)) == 0
    d.Hout, d.dtype = 2 * H, L.F32
    assert lib.abc_convt_fused_ok(C.byref(d)) == 0


@pytest.mark.parametrize("Ca,Cb,ntaps,nsplit,C_,nblk", [(128, 128, 9, 128, 128, 768), (64, 32, 9, 33, 64, 2304), (16, 16, 9, 256, 16, 512), (256, 128, 4, 17, 256, 48)])
def test_slab_reduction_and_bn_finaliser_as_one_launch(lib, Ca, Cb, ntaps, nsplit, C_, nblk):
    """abc_wgrad_reduce_bn_bwd == abc_wgrad_reduce + abc_bn_finalize_bwd of two unrelated layers, bit for bit (vector and scalar reduction
    forms, and the small-output form that is launched on its own)"""
    g = torch.Generator().manual_seed(9)
    ca_pad, cb_pad = -(-Ca // 32) * 32, -(-Cb // 32) * 32
    part = torch.randn((nsplit, ntaps, ca_pad, cb_pad), generator=g).to(U.DEV)
    bnp = (torch.randn((nblk, 2, C_), generator=g) * 2).to(U.DEV)
    f32 = lambda n, v=0.0: torch.full((n,), v, dtype=torch.float32, device=U.DEV)
    gamma, invstd, mean = (torch.rand(C_, generator=g) + 0.5).to(U.DEV), (torch.rand(C_, generator=g) + 0.5).to(U.DEV), torch.randn(C_, generator=g).to(U.DEV)

    def descs():
        dw = torch.zeros((Ca, Cb, ntaps), dtype=torch.float32, device=U.DEV)
        r = L.WgradReduceDesc()
        r.partial, r.nsplit, r.ntaps, r.Ca, r.Cb, r.Ca_pad, r.Cb_pad, r.dw, r.accumulate = part.data_ptr(), nsplit, ntaps, Ca, Cb, ca_pad, cb_pad, dw.data_ptr(), 0
        outs = [f32(C_) for _ in range(8)]
        f = L.BnBwdDesc()
        f.partial, f.nblk, f.C, f.count, f.gamma, f.invstd, f.mean = bnp.data_ptr(), nblk, C_, float(nblk * 37), gamma.data_ptr(), invstd.data_ptr(), mean.data_ptr()
        f.dgamma, f.dbeta, f.k1, f.k2, f.gscale, f.ca, f.cb, f.cc = (t.data_ptr() for t in outs)
        return r, f, dw, outs

    r1, f1, dw1, o1 = descs()
    L.check(lib.abc_wgrad_reduce_bn_bwd(C.byref(r1), C.byref(f1), U.stream()), "reduce+bn_bwd")
    r2, f2, dw2, o2 = descs()
    L.check(lib.abc_wgrad_reduce(C.byref(r2), U.stream()), "reduce")
    L.check(lib.abc_bn_finalize_bwd(C.byref(f2), U.stream()), "bn_bwd")
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2) and float(dw1.abs().max()) > 0
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    assert float(o1[0].abs().max()) > 0 and float(o1[5].abs().max()) > 0
    ref = part[:, :, :Ca, :Cb].double().sum(0).permute(1, 2, 0)
    assert U.relerr(dw1.cpu(), ref.cpu()) < 1e-5


def test_batched_packing_tile_form_is_bit_identical_to_the_single_kernel():
    """abc_pack_batch sends Conv2d forward / data-gradient packings with 32-multiple channel counts through the source-major tile
    kernel (32 x 32 x taps tiles transposed in LDS); everything else through the dest-major gather.  Both against
    abc_pack_conv_weights (the single-weight dest-major kernel): bit for bit -- both layouts, padded rows, several weights side
    by side along rows (the eight heads' conv1) and along the reduction axis (the merged data gradient), a per-row scale, 1x1 and
    3x3, and items the tile form does not take (16 channels, transposed convolutions) in the same table."""
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    st = U.stream()
    items, singles, keep = [], [], []

    def add(w, mode, Cout, Cin, k, rows_pad, red_total=None, red_off=0, rows_total=0, rows_off=0, layout=0, row_scale=None, dst=None, py=0, px=0):
        red = {0: Cin, 1: Cout, 2: Cin, 3: Cout}[mode]
        rt = red if red_total is None else red_total
        ck = lib.abc_conv_chunk(L.BF16, rt)
        red_pad = -(-red // ck) * ck
        ntaps = {0: k * k, 1: k * k, 2: (2 if py else 1) * (2 if px else 1), 3: 9}[mode]
        n = ntaps * (-(-rt // ck) * ck) * (rows_total or rows_pad)
        if dst is None:
            dst = (torch.zeros(n, dtype=torch.bfloat16, device=U.DEV), torch.zeros(n, dtype=torch.bfloat16, device=U.DEV))
            keep.append(dst)
        ds = []
        for which in (0, 1):
            d = L.PackDesc()
            d.w, d.dst, d.mode, d.dtype_c = w.data_ptr(), dst[which].data_ptr(), mode, L.BF16
            d.Cout, d.Cin, d.kh, d.kw, d.py, d.px = Cout, Cin, k, k, py, px
            d.rows_pad, d.red_pad, d.red_total, d.red_off, d.ck = rows_pad, red_pad, rt, red_off, ck
            d.rows_total, d.rows_off, d.layout = rows_total, rows_off, layout
            d.row_scale = row_scale.data_ptr() if row_scale is not None else None
            ds.append(d)
        items.append(ds[0])
        singles.append(ds[1])
        return dst

    rnd = lambda *s: torch.randn(*s, generator=g).to(U.DEV)
    w1 = rnd(128, 128, 3, 3); keep.append(w1)
    arena = rnd(10 + 64 * 64 * 9); keep.append(arena)
    wu = arena[10:].view(64, 64, 3, 3)               # a view at an odd element offset of a flat arena: not 16-byte aligned
    assert wu.data_ptr() % 16 != 0
    add(wu, 0, 64, 64, 3, 64, layout=1)
    add(wu, 1, 64, 64, 3, 64)
    add(w1, 0, 128, 128, 3, 128)
    add(w1, 1, 128, 128, 3, 128)
    add(w1, 0, 128, 128, 3, 128, layout=1)
    w2 = rnd(64, 96, 3, 3); keep.append(w2)
    add(w2, 0, 64, 96, 3, 128)                       # padded rows (Cout_pad 128)
    add(w2, 1, 64, 96, 3, 96)
    sc = (torch.rand(64, generator=g) + 0.5).to(U.DEV); keep.append(sc)
    add(w2, 0, 64, 96, 3, 64, row_scale=sc, layout=1)
    w3 = rnd(32, 64, 1, 1); keep.append(w3)
    add(w3, 0, 32, 64, 1, 32)
    # two heads' conv1 one below the other (rows_total / rows_off), and their data gradient side by side along the reduction axis
    wa, wb = rnd(128, 128, 3, 3), rnd(128, 128, 3, 3); keep += [wa, wb]
    dst = add(wa, 0, 128, 128, 3, 128, rows_total=256, rows_off=0, layout=1)
    add(wb, 0, 128, 128, 3, 128, rows_total=256, rows_off=128, layout=1, dst=dst)
    dst = add(wa, 1, 128, 128, 3, 128, red_total=256, red_off=0)
    add(wb, 1, 128, 128, 3, 128, red_total=256, red_off=128, dst=dst)
    # not tile material: 16 channels, a transposed convolution's phase and data-gradient packing
    w4 = rnd(16, 16, 3, 3); keep.append(w4)
    add(w4, 0, 16, 16, 3, 32)
    add(w4, 1, 16, 16, 3, 32)
    wt = rnd(64, 32, 3, 3); keep.append(wt)          # ConvTranspose2d weight [Cin][Cout][3][3]
    add(wt, 2, 32, 64, 3, 32, py=1, px=1)
    add(wt, 3, 32, 64, 3, 64)
    isz = lib.abc_pack_item_bytes()
    host = (C.c_char * (isz * len(items)))()
    first = 0
    for i, d in enumerate(items):
        n = lib.abc_pack_item_fill(C.addressof(host) + i * isz, C.byref(d), first)
        assert n >= 0          # (0: the item goes through the tile kernel and takes no range of the dest-major one)
        first += n
    assert 0 < first
    table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(U.DEV)
    L.check(lib.abc_pack_batch(table.data_ptr(), len(items), first, st), "pack_batch")
    for d in singles:
        L.check(lib.abc_pack_conv_weights(C.byref(d), st), "pack")
    torch.cuda.synchronize()
    for i, pair in enumerate(keep):
        if isinstance(pair, tuple):
            assert torch.equal(pair[0].view(torch.int16), pair[1].view(torch.int16)), i
            assert pair[0].float().abs().sum().item() > 0


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 40, 72), (2, 18, 9), (1, 12, 12), (2, 96, 200), (16, 48, 48), (1, 17, 131)])
def test_cbam_conv7_forward_and_backward(lib, B, H, W):
    """SpatialAttentionModule's 7x7 convolution (unet2.py:27,34) against torch: sa = sigmoid(conv2d(st)), and -- given du = d(pre-sigmoid) --
    the data gradient d(st) and the weight / bias gradients (both tile forms of abc_cbam_conv7_bwd: four pixels per thread from 48
    columns up, one below; sizes that are no multiple of the tile, many tiles per persistent workgroup)."""
    g = torch.Generator().manual_seed(11)
    st = torch.randn(B, H, W, 2, generator=g)
    du = torch.randn(B, H, W, generator=g)
    w7 = torch.randn(1, 2, 7, 7, generator=g) * 0.1
    b7 = torch.randn(1, generator=g)
    d = L.CbamConv7Desc()
    std, dud, wd, bd = st.to("cuda"), du.to("cuda"), w7.to("cuda"), b7.to("cuda")
    sa = torch.empty(B, H, W, device="cuda")
    dst = torch.full((B, H, W, 2), float("nan"), device="cuda")
    dw, db = torch.empty(98, device="cuda"), torch.empty(1, device="cuda")
    d.st, d.w7, d.b7, d.sa, d.du, d.dst = std.data_ptr(), wd.data_ptr(), bd.data_ptr(), sa.data_ptr(), dud.data_ptr(), dst.data_ptr()
    d.B, d.H, d.W = B, H, W
    nb = lib.abc_cbam_conv7_blocks(C.byref(d))
    part = torch.empty(nb, 99, device="cuda")
    d.dw_partial, d.dw7, d.db7 = part.data_ptr(), dw.data_ptr(), db.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    L.check(lib.abc_cbam_conv7_fwd(C.byref(d), stream), "conv7_fwd")
    L.check(lib.abc_cbam_conv7_bwd(C.byref(d), stream), "conv7_bwd")
    torch.cuda.synchronize()
    x = st.permute(0, 3, 1, 2).double().requires_grad_(True)
    w = w7.double().requires_grad_(True)
    b = b7.double().requires_grad_(True)
    pre = F.conv2d(x, w, b, padding=3)
    assert (sa.cpu().double() - torch.sigmoid(pre)[:, 0]).abs().max() < 1e-5
    pre.backward(du.double().unsqueeze(1))
    ref_dst = x.grad.permute(0, 2, 3, 1)
    assert (dst.cpu().double() - ref_dst).abs().max() < 1e-4 * max(1.0, ref_dst.abs().max().item())
    assert (dw.cpu().double() - w.grad.reshape(-1)).abs().max() < 2e-5 * max(1.0, w.grad.abs().max().item())
    assert abs(db.cpu().double().item() - b.grad.item()) < 2e-5 * max(1.0, abs(b.grad.item()))
