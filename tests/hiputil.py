"""Helpers for the GPU parity tests: build C-ABI descriptors over torch device tensors."""
import ctypes as C

import torch

import abcnet_amd  # noqa: F401
from abcnet_amd import _lib as L

DEV = "cuda"


def stream():
    return torch.cuda.current_stream().cuda_stream


def tdt(dt):
    return torch.bfloat16 if dt == L.BF16 else torch.float32


def nhwc(x_nchw, dt):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(tdt(dt)).to(DEV)


def to_nchw(y_nhwc):
    return y_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def fill_src(a, t, H, W, ld, coef=None, pool=False, drop_p=0.0, drop_seed=0):
    a.x = t.data_ptr()
    if coef is not None:
        a.scale, a.shift, a.slope = (c.data_ptr() for c in coef)
    a.Hx, a.Wx, a.ldx, a.pool, a.drop_p, a.drop_seed = H, W, ld, int(pool), drop_p, drop_seed


def pack(lib, w, mode, dt, Cout, Cin, k, rows_pad, red_real, py=0, px=0):
    """w: f32 device tensor in the reference layout"""
    ck = lib.abc_conv_chunk(dt, red_real)
    red_pad = -(-red_real // ck) * ck
    ntaps = {0: k * k, 1: k * k, 2: (2 if py else 1) * (2 if px else 1), 3: 9}[mode]
    dst = torch.zeros(ntaps * red_pad * rows_pad, dtype=tdt(dt), device=DEV)
    d = L.PackDesc()
    d.w, d.dst, d.mode, d.dtype_c = w.data_ptr(), dst.data_ptr(), mode, dt
    d.Cout, d.Cin, d.kh, d.kw, d.py, d.px = Cout, Cin, k, k, py, px
    d.rows_pad, d.red_pad, d.red_total, d.red_off, d.ck = rows_pad, red_pad, red_real, 0, ck
    L.check(lib.abc_pack_conv_weights(C.byref(d), stream()), "pack")
    return dst


def conv(lib, x, dt_in, dt, B, Hx, Wx, ldx, cin_off, Cin, wp, bias, Cout, taps, Hout, Wout, ldy=None, cout_off=0, coef=None,
         pool=False, stride=1, grid=None, om=1, oy0=0, ox0=0, out=None, out_dt=None, stats=False, drop_p=0.0, drop_seed=0,
         planar_in=0, planar_out=False, out_slope=None, pool_out=None, stem=None, actbwd=None, defer=None):
    """actbwd: (y_raw, ld, coff, scale, shift, slope, mean, invstd) -> abc_conv_desc.actbwd_*; conv.last_actbwd_ok tells whether the
    library honoured it (else the plain convolution ran)"""
    out_dt = dt if out_dt is None else out_dt
    ldy = Cout if ldy is None else ldy
    if out is None:
        out = torch.zeros((B, Cout, Hout, Wout) if planar_out else (B, Hout, Wout, ldy), dtype=tdt(out_dt), device=DEV)
    d = L.ConvDesc()
    fill_src(d.src, x, Hx, Wx, ldx, coef, pool, drop_p, drop_seed)
    d.src.planar, d.src.ctot = (1, planar_in) if planar_in else (0, 0)
    d.planar_out, d.ctot_out = (1, Cout) if planar_out else (0, 0)
    d.w, d.bias, d.y = wp.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr()
    d.dtype_in, d.dtype_c, d.dtype_out = dt_in, dt, out_dt
    d.B, d.Hin, d.Win = B, (Hx // 2 if pool else Hx), (Wx // 2 if pool else Wx)
    d.cin_off, d.Cin = cin_off, Cin
    gh, gw = grid if grid else (Hout, Wout)
    d.Hg, d.Wg, d.Hout, d.Wout, d.ldy, d.cout_off, d.Cout, d.Cout_pad = gh, gw, Hout, Wout, ldy, cout_off, Cout, -(-Cout // 32) * 32
    d.stride, d.om, d.oy0, d.ox0 = stride, om, oy0, ox0
    if out_slope is not None:
        d.out_act, d.out_slope = 1, out_slope
    if pool_out is not None:
        d.pool_y, d.ld_pool = pool_out.data_ptr(), pool_out.shape[-1]
    if stem is not None:   # (image, first-layer weights, scale, bias, slope): device f32 tensors
        d.stem_x, d.stem_w, d.stem_scale, d.stem_bias, d.stem_slope = stem[0].data_ptr(), stem[1].data_ptr(), stem[2].data_ptr(), stem[3].data_ptr(), stem[4]
    L.set_taps(d, taps)
    conv.last_actbwd_ok = False
    if actbwd is not None:
        yr, ld_y, coff, sc, sh, sl, mu, istd = actbwd
        d.actbwd_y, d.actbwd_ld, d.actbwd_coff = yr.data_ptr(), ld_y, coff
        d.actbwd_scale, d.actbwd_shift, d.actbwd_slope, d.actbwd_mean, d.actbwd_invstd = (t.data_ptr() for t in (sc, sh, sl, mu, istd))
        d.stats = out.data_ptr()      # (any non-null pointer: the geometry query looks at it; the real buffer is set below)
        d.stats_rows = 2
        conv.last_actbwd_ok = bool(lib.abc_conv_actbwd_ok(C.byref(d)))
        if not conv.last_actbwd_ok:
            d.actbwd_y = None
            d.stats = None
        assert stats, "actbwd needs stats=True (the BatchNorm-backward partial sums)"
    conv.last_variant = lib.abc_conv_variant(C.byref(d))
    if lib.abc_conv_weight_layout(C.byref(d)) == 1:
        # the kernel serving this descriptor reads abc_pack_desc.layout 1: re-order the row-major packing here (an independent
        # statement of the permutation; abc_pack_conv_weights' own layout-1 path is held to it in test_gpu_kernels.py)
        wp = wp.view(-1, 32, 2, 2, 8).permute(0, 3, 2, 1, 4).contiguous().view(-1)
        d.w = wp.data_ptr()
    st = None
    if stats:
        nblk = lib.abc_conv_stat_blocks(C.byref(d))
        st = torch.zeros((nblk, 2, Cout), dtype=torch.float32, device=DEV)
        d.stats = st.data_ptr()
    if defer is not None:
        defer.append((d, wp, st))      # (the launch is the caller's: abc_conv_fwd_batch over several descriptors)
        return out, st
    L.check(lib.abc_conv_fwd(C.byref(d), stream()), "conv_fwd")
    return out, st


def wgrad(lib, p, dt_p, Hp, Wp, ldp, cp_off, Ca, p_coef, q, dt_q, Hq, Wq, ldq, cq_off, Cb, q_coef, q_pool, dt, B, taps,
          stride=1, nsplit=3):
    d = L.WgradDesc()
    fill_src(d.p, p, Hp, Wp, ldp, p_coef)
    fill_src(d.q, q, Hq, Wq, ldq, q_coef, q_pool)
    d.dtype_p, d.dtype_q, d.dtype_c = dt_p, dt_q, dt
    d.B, d.Hg, d.Wg = B, Hp, Wp
    d.Hq, d.Wq = (Hq // 2, Wq // 2) if q_pool else (Hq, Wq)
    d.cp_off, d.Ca, d.cq_off, d.Cb, d.stride, d.nsplit = cp_off, Ca, cq_off, Cb, stride, nsplit
    L.set_taps(d, taps)
    ca, cb = L.i32(), L.i32()
    L.check(lib.abc_wgrad_pads(C.byref(d), C.byref(ca), C.byref(cb)), "pads")
    part = torch.zeros(nsplit * len(taps) * ca.value * cb.value, dtype=torch.float32, device=DEV)
    d.partial = part.data_ptr()
    L.check(lib.abc_wgrad(C.byref(d), stream()), "wgrad")
    dw = torch.zeros((Ca, Cb, len(taps)), dtype=torch.float32, device=DEV)
    r = L.WgradReduceDesc()
    r.partial, r.nsplit, r.ntaps, r.Ca, r.Cb, r.Ca_pad, r.Cb_pad, r.dw, r.accumulate = part.data_ptr(), nsplit, len(taps), Ca, Cb, ca.value, cb.value, dw.data_ptr(), 0
    L.check(lib.abc_wgrad_reduce(C.byref(r), stream()), "wgrad_reduce")
    return dw


def tol(dt, f32=1e-4, bf16=3e-2):
    return bf16 if dt == L.BF16 else f32


def relerr(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
