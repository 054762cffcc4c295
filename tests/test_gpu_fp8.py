"""The fp8 (e4m3) form of the BatchNorm-folded inference graph (BASELINE.json config 5;
img2smiles2.py:42-59): kernel level and graph level.

Kernel level: the 3x3 convolution of conv_fast.hip's weights-direct loop over e4m3 operands on the block-scaled MFMA
(v_mfma_scale_f32_32x32x64_f8f6f4, unit scales) against torch's conv2d of the SAME dequantised e4m3 operands -- e4m3 x e4m3
products are exact in f32, so only the summation order differs: the bf16-output form agrees to one bf16 rounding, the
e4m3-output form to one e4m3 rounding step on a small fraction of the elements (a value that sits on a rounding boundary).
Graph level: InferenceRunner(fp8=True) against the fp32 oracle and against the bf16 folded graph.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import abcnet_amd  # noqa: E402,F401
from abcnet_amd import _lib as L  # noqa: E402
from abcnet_amd.engine import taps_square  # noqa: E402
from abcnet_amd.synthetic import synthetic_images  # noqa: E402
from oracle import nms_oracle  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

DEV = "cuda"
F8 = torch.float8_e4m3fn
HEADS = uo.HEADS


def st():
    return torch.cuda.current_stream().cuda_stream


def q8(x):
    """round to e4m3 (saturating at 448) and back"""
    return x.clamp(-448.0, 448.0).to(F8).float()


def _pack_fp8(lib, w, qmul, Cout, Cin, layout):
    ck = lib.abc_conv_chunk(L.FP8, Cin)
    assert ck == 64
    dst = torch.zeros(9 * Cin * Cout, dtype=torch.uint8, device=DEV)
    d = L.PackDesc()
    d.w, d.dst, d.mode, d.dtype_c = w.data_ptr(), dst.data_ptr(), 0, L.FP8
    d.Cout, d.Cin, d.kh, d.kw = Cout, Cin, 3, 3
    d.rows_pad, d.red_pad, d.red_total, d.red_off, d.ck, d.layout = Cout, Cin, Cin, 0, ck, layout
    d.row_scale = qmul.data_ptr()
    L.check(lib.abc_pack_conv_weights(C.byref(d), st()), "pack")
    return dst


def _conv(lib, x, dt_in, cdt, out_dt, B, H, W, Cin, Cout, wp, bias, out_scale, out_quant, slope=0.0):
    tdt = {L.BF16: torch.bfloat16, L.FP8: F8}
    y = torch.zeros((B, H, W, Cout), dtype=tdt[out_dt], device=DEV)
    d = L.ConvDesc()
    d.src.x, d.src.Hx, d.src.Wx, d.src.ldx = x.data_ptr(), H, W, Cin
    d.w, d.bias, d.y = wp.data_ptr(), bias.data_ptr(), y.data_ptr()
    d.dtype_in, d.dtype_c, d.dtype_out = dt_in, cdt, out_dt
    d.B, d.Hin, d.Win, d.cin_off, d.Cin = B, H, W, 0, Cin
    d.Hg, d.Wg, d.Hout, d.Wout, d.ldy, d.cout_off, d.Cout, d.Cout_pad = H, W, H, W, Cout, 0, Cout, Cout
    d.stride, d.om = 1, 1
    d.out_act, d.out_slope = 1, slope
    d.out_scale = None if out_scale is None else out_scale.data_ptr()
    d.out_quant = None if out_quant is None else out_quant.data_ptr()
    L.set_taps(d, taps_square(3))
    assert lib.abc_conv_variant(C.byref(d)) == 1 and lib.abc_conv_weight_layout(C.byref(d)) == 1
    L.check(lib.abc_conv_fwd(C.byref(d), st()), "conv_fwd")
    return y


@pytest.mark.parametrize("case", [
    dict(B=2, H=48, W=48, Cin=128, Cout=128),
    dict(B=1, H=20, W=40, Cin=64 * 3, Cout=256),       # ragged tiles, three chunks, two n-blocks
    dict(B=4, H=96, W=192, Cin=128, Cout=256),         # 768 tiles on 512 persistent workgroups
])
def test_fp8_conv_equals_conv2d_of_the_dequantised_operands(case):
    lib = L.load()
    B, H, W, Cin, Cout = (case[k] for k in ("B", "H", "W", "Cin", "Cout"))
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn((B, Cin, H, W), generator=g)) * 3.0            # an activated tensor
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5
    fold = torch.rand(Cout, generator=g) + 0.5                                 # gamma / sqrt(var + eps) of the folded BatchNorm
    bias = torch.randn(Cout, generator=g) * 0.2
    # ---- calibration kernels: amax -> (s, 1 / s) on the device
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    amax, s_in, inv_in = (torch.zeros(1, device=DEV) for _ in range(3))
    L.check(lib.abc_absmax(xd.data_ptr(), L.BF16, xd.numel(), amax.data_ptr(), st()), "absmax")
    L.check(lib.abc_fp8_act_scale(amax.data_ptr(), 1.0, s_in.data_ptr(), inv_in.data_ptr(), st()), "act_scale")
    assert amax.item() == xd.float().abs().max().item() and abs(s_in.item() - amax.item() / 448.0) <= 1e-6 * s_in.item()
    assert abs(inv_in.item() * s_in.item() - 1.0) < 1e-6
    # ---- weight scales: per output row
    wd, fd = w.to(DEV), fold.to(DEV)
    qmul, deq = torch.zeros(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    L.check(lib.abc_fp8_weight_scales(wd.data_ptr(), Cout, Cin * 9, fd.data_ptr(), s_in.data_ptr(), qmul.data_ptr(), deq.data_ptr(), st()), "wscales")
    sw = (w.abs().amax((1, 2, 3)) * fold) / 448.0
    assert torch.allclose(qmul.cpu(), fold / sw, rtol=1e-6) and torch.allclose(deq.cpu(), sw * s_in.item(), rtol=1e-6)
    # ---- packing: layout 1 (fragment-contiguous) == the row-major packing re-ordered here
    p1 = _pack_fp8(lib, wd, qmul, Cout, Cin, 1)
    p0 = _pack_fp8(lib, wd, qmul, Cout, Cin, 0)
    assert torch.equal(p1, p0.view(-1, 32, 2, 2, 16).permute(0, 3, 2, 1, 4).contiguous().view(-1))
    wq = q8(w * qmul.cpu().view(-1, 1, 1, 1))                                   # what the packed bytes hold
    want = wq.permute(2, 3, 1, 0).reshape(9, Cin // 64, 64, Cout).permute(0, 1, 3, 2).contiguous()   # [tap][chunk][row][64]
    assert torch.equal(p0.view(F8).float().cpu().view(9, Cin // 64, Cout, 64), want)
    # ---- the convolution: e4m3 in, bf16 out and e4m3 out
    xq = (x * inv_in.item()).clamp(max=448.0).to(F8)                             # the producer's quantisation
    xq_d = xq.permute(0, 2, 3, 1).contiguous().to(DEV)
    ref = F.conv2d(xq.float(), wq, padding=1) * deq.cpu().view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    ref = torch.relu(ref)
    bd = bias.to(DEV)
    # (launched several times: the multi-tile case once failed INTERMITTENTLY, in a few hundred of 18.9 M outputs per launch -- a
    #  store-data hazard of the epilogue's 16-byte buffer stores, DESIGN.md section 3 -- and a single launch can miss such a thing)
    for _rep in range(4):
        y16 = _conv(lib, xq_d, L.FP8, L.FP8, L.BF16, B, H, W, Cin, Cout, p1, bd, deq, None)
        torch.cuda.synchronize()
        got = y16.float().cpu().permute(0, 3, 1, 2)
        assert (got - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6        # one bf16 rounding
    amax_o, s_o, inv_o = (torch.zeros(1, device=DEV) for _ in range(3))
    L.check(lib.abc_absmax(y16.data_ptr(), L.BF16, y16.numel(), amax_o.data_ptr(), st()), "absmax")
    L.check(lib.abc_fp8_act_scale(amax_o.data_ptr(), 1.0, s_o.data_ptr(), inv_o.data_ptr(), st()), "act_scale")
    y8 = _conv(lib, xq_d, L.FP8, L.FP8, L.FP8, B, H, W, Cin, Cout, p1, bd, deq, inv_o)
    torch.cuda.synchronize()
    got8 = y8.float().cpu().permute(0, 3, 1, 2)
    ref8 = q8(ref * inv_o.item())
    differ = (got8 != ref8)
    assert differ.float().mean().item() < 2e-3, differ.float().mean().item()                   # summation order on a rounding boundary
    # ... and then by ONE e4m3 step -- plus, near zero (ReLU of a cancelling sum, where the subnormal steps are 2^-9), the f32
    # summation-order noise of the two implementations: ~1e-6 of sum |x||w| (measured: up to two subnormal steps)
    noise = 8e-6 * (F.conv2d(xq.float().abs(), wq.abs(), padding=1) * deq.cpu().view(1, -1, 1, 1)).max().item() * inv_o.item()
    bad = (got8 - ref8).abs() > 0.13 * ref8.abs() + 2.0 ** -9 + noise
    exact = (ref * inv_o.item())
    assert not bad.any(), (int(bad.sum()), got8[bad][:8].tolist(), ref8[bad][:8].tolist(), exact[bad][:8].tolist())
    assert got8.max().item() <= 448.0 and not torch.isnan(got8).any()
    # ---- bf16 compute, e4m3 output (the convolution that enters the fp8 chain)
    xb = xd                                                                       # bf16 NHWC of the same x
    wb = torch.zeros(9 * Cin * Cout, dtype=torch.bfloat16, device=DEV)
    d = L.PackDesc()
    d.w, d.dst, d.mode, d.dtype_c = wd.data_ptr(), wb.data_ptr(), 0, L.BF16
    d.Cout, d.Cin, d.kh, d.kw = Cout, Cin, 3, 3
    d.rows_pad, d.red_pad, d.red_total, d.red_off, d.ck, d.layout = Cout, Cin, Cin, 0, 32, 1
    d.row_scale = fd.data_ptr()
    L.check(lib.abc_pack_conv_weights(C.byref(d), st()), "pack")
    yb8 = _conv(lib, xb, L.BF16, L.BF16, L.FP8, B, H, W, Cin, Cout, wb, bd, None, inv_o)
    torch.cuda.synchronize()
    refb = torch.relu(F.conv2d(xd.float().cpu().permute(0, 3, 1, 2), (w * fold.view(-1, 1, 1, 1)).to(torch.bfloat16).float(), padding=1)
                      + bias.view(1, -1, 1, 1))
    refb8 = q8(refb * inv_o.item())
    gotb8 = yb8.float().cpu().permute(0, 3, 1, 2)
    assert (gotb8 != refb8).float().mean().item() < 2e-3
    badb = (gotb8 - refb8).abs() > 0.13 * refb8.abs() + 2.0 ** -9 + noise
    assert not badb.any(), (int(badb.sum()), gotb8[badb][:8].tolist(), refb8[badb][:8].tolist())


def _runner(fp8, B, S, margin=1.0, heads_epilogue=False, H=None):
    from abcnet_amd.infer import InferenceRunner
    from abcnet_amd.unet import UNet
    m = UNet(1, HEADS, dtype="bf16")
    m.load_state_dict(uo.filled_state("unet", 1, HEADS, seed=0))
    m = m.to(DEV)
    return InferenceRunner(m, B, H or S, S, use_graph=True, fold_bn=True, fp8=fp8, fp8_margin=margin, heads_epilogue=heads_epilogue)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,H,W", [(2, 128, 128), (3, 160, 96), (2, 384, 512)])
def test_heads_in_the_epilogue_equal_the_separate_heads_kernel(fp8, B, H, W):
    """abc_conv_desc.heads_epi: the heads' 1x1 convolutions (unet.py:70) computed in the epilogue of the convolution that makes
    their features, the 8 x 128-channel feature tensor never written.  Same operand roundings (features to bf16 / e4m3 with the
    head's scale), same K order of the MFMAs: the eight maps equal those of the plan with the separate heads kernel BIT FOR BIT
    -- whole tiles, ragged tiles (40 x 24 and 96 x 128 maps), bf16 and e4m3.  (An opt-in plan, InferenceRunner(heads_epilogue=True):
    exact, but measured slower than conv1 + the separate heads kernel -- DESIGN.md section 3, round 3.)"""
    x = synthetic_images(B, max(H, W), seed=7)[:, :, :H, :W].contiguous().to(DEV)
    outs = []
    for epi in (True, False):
        run = _runner(fp8, B, W, heads_epilogue=epi, H=H)
        assert (run.eng.hepi is not None) == epi
        run.load_batch(x)
        run.step()
        run.step()
        torch.cuda.synchronize()
        outs.append([t.clone() for t in run.logits] + [run.atom_mask.clone(), run.omega_mask.clone()])
    # bf16 at 96 x 128 maps (whole 12-row tiles): the separate plan's conv1 runs the 16x16x32 MFMA form of the tile (conv_fast_body.hpp
    # M16, round 5), the heads-epilogue kernel the 32x32x16 form -- the same products summed in a different association inside the
    # instruction, so the features agree to a bf16 rounding instead of bit for bit: logits within 2 % of their spread, at most 1 in 10^4
    # NMS decisions different.  Everything else (e4m3; maps without whole 12-row tiles) stays exact.
    m16 = (not fp8) and (H // 4) % 12 == 0 and (W // 4) % 16 == 0
    for i, (a, b) in enumerate(zip(*outs)):
        if not m16:
            assert torch.equal(a, b)
        elif a.dtype.is_floating_point and i < 8:
            assert (a - b).abs().max().item() <= 0.02 * b.float().std().item() + 1e-3, (i, (a - b).abs().max().item(), b.float().std().item())
        else:
            assert (a != b).float().mean().item() <= 1e-4, (i, (a != b).float().mean().item())


def test_fp8_inference_graph_against_oracle_and_bf16_graph():
    """InferenceRunner(fp8=True) at 2 x 128 x 128: which launches run in e4m3, calibration on the first batch, eager step == graph
    replay, logits against the fp32 oracle and the bf16 folded graph, NMS bit-exact on the graph's own logits"""
    B, S = 2, 128
    x = synthetic_images(B, S, seed=7)
    run8, run16 = _runner(True, B, S), _runner(False, B, S)
    kinds = [op[4]["kernel"] for op in run8.eng.fwd_ops]
    # up3.conv's first convolution enters the chain (bf16 compute, e4m3 out); five 128 -> 128 convolutions and the eight heads'
    # merged conv1 compute in e4m3 and store e4m3; the heads' 1x1 convolutions read e4m3 features and write the f32 maps
    assert sum("<fp8,fp8,fp8" in k for k in kinds) == 6 and sum("<bf16,bf16,fp8" in k for k in kinds) == 1 and "heads_fwd_batch" in kinds, kinds
    assert run8.eng.hfeat_q is not None and run8.eng.hepi is None
    assert not run8.eng.fp8_calibrated
    outs = []
    for run in (run8, run16):
        run.load_batch(x.to(DEV))
        run.step()
        torch.cuda.synchronize()
        eager = [t.clone() for t in run.logits]
        run.step()            # captured
        run.step()            # replayed
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(eager, run.logits))
        outs.append([t.cpu() for t in run.logits])
    assert run8.eng.fp8_calibrated
    for r in run8.eng.fp8_recs:   # every e4m3 tensor got a scale from its bf16 twin's amax, and uses most of the range
        amax, s, inv = (t.item() for t in r.q)
        assert amax > 0 and abs(s - amax / 448.0) <= 1e-6 * s
        assert r.y.dtype == F8 and r.y.float().max().item() > 200.0
    with torch.no_grad():
        ref = uo.forward("unet", uo.filled_state("unet", 1, HEADS, seed=0), x, train=False)
    d8 = max((a - r).abs().max().item() for a, r in zip(outs[0], ref))
    d16 = max((a - r).abs().max().item() for a, r in zip(outs[1], ref))
    print("fp8 graph vs fp32 oracle: logits Linf %.4f (bf16 folded graph: %.4f)" % (d8, d16))
    assert d16 < 0.08 and d8 < 0.5, (d8, d16)
    da, db, dr, do = nms_oracle.nms(outs[0][0], outs[0][4], outs[0][6], outs[0][7])
    assert torch.equal(run8.atom_mask.cpu(), da) and torch.equal(run8.bond_mask.cpu(), db) and torch.equal(run8.omega_mask.cpu(), do)
    assert torch.equal(run8.rho_abs.cpu(), dr)


@pytest.mark.parametrize("fp8", [False, True])
def test_nms_outputs_from_the_heads_kernel_equal_the_nms_kernel(fp8):
    """img2smiles2.py:73-79 as second outputs of the heads' 1x1 kernel (abc_conv_desc.head_aux: |rho|, the omega-bin mask with its
    circular neighbours exchanged between the two lanes of a pixel) against the round-3 plan where the NMS kernel reads the stored
    maps back: logits, all four NMS outputs equal bit for bit, bf16 and e4m3 graphs, a size whose maps are not a multiple of the
    kernel's 64-pixel pairs per image row"""
    a = _runner(fp8, 3, 160)
    from abcnet_amd.infer import InferenceRunner
    assert a.nms_in_heads
    x = synthetic_images(3, 160, seed=7).to(DEV)
    b = InferenceRunner(a.model, 3, 160, 160, use_graph=True, fold_bn=True, fp8=fp8, nms_in_heads=False)
    assert not b.nms_in_heads
    for run in (a, b):
        run.load_batch(x)
        run.step()
        run.step()
    torch.cuda.synchronize()
    for i, (p, q) in enumerate(zip(a.logits, b.logits)):
        assert torch.equal(p, q), i
    for name in ("atom_mask", "bond_mask", "rho_abs", "omega_mask"):
        p, q = getattr(a, name), getattr(b, name)
        assert torch.equal(p, q), (name, int((p != q).sum()))
    assert a.omega_mask.sum().item() > 0 and a.rho_abs.abs().sum().item() > 0


@pytest.mark.parametrize("fp8", [False, True])
def test_decode_mode_stores_what_the_decoder_reads(fp8):
    """InferenceRunner(decode=True): the raw rho map and the 360 bond-type planes are not stored (img2smiles2.py:73 uses |rho| only,
    :71,112 the six-way arg max per omega bin only).  Against the plan that stores all eight maps, same weights and batch: the six
    remaining maps, the four NMS outputs, the uint8 arg-max map == torch.argmax of the stored planes viewed as (6, 60) -- first
    maximum on ties --, and the candidate lists of the device extraction, all bit for bit; bf16 and e4m3 graphs."""
    from abcnet_amd.infer import InferenceRunner
    a = _runner(fp8, 3, 160)
    x = synthetic_images(3, 160, seed=11).to(DEV)
    full = InferenceRunner(a.model, 3, 160, 160, use_graph=True, fold_bn=True, fp8=fp8, extract=True)
    dec = InferenceRunner(a.model, 3, 160, 160, use_graph=True, fold_bn=True, fp8=fp8, extract=True, decode=True)
    assert dec.decode and dec.logits[5] is None and dec.logits[6] is None and dec.btype_idx.dtype == torch.uint8
    for run in (full, dec):
        run.load_batch(x)
        run.step()
        run.step()
    torch.cuda.synchronize()
    for i in (0, 1, 2, 3, 4, 7):
        assert torch.equal(full.logits[i], dec.logits[i]), i
    for name in ("atom_mask", "bond_mask", "rho_abs", "omega_mask"):
        assert torch.equal(getattr(full, name), getattr(dec, name)), name
    ref_idx = full.logits[5].view(3, 6, 60, 40, 40).argmax(1)
    assert torch.equal(dec.btype_idx.long(), ref_idx), int((dec.btype_idx.long() != ref_idx).sum())
    assert len(set(dec.btype_idx.flatten().tolist())) > 1
    cf, cd = full.candidates(), dec.candidates()
    for f, d in zip(cf, cd):
        assert torch.equal(f["atoms"], d["atoms"]) and torch.equal(f["bonds"], d["bonds"]) and torch.equal(f["rho"], d["rho"])
        assert f["counts"] == d["counts"]
    assert sum(len(f["bonds"]) for f in cf) > 0
