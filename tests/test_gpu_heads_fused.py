"""GPU parity of the fused heads pass (csrc/heads_fused.hip + the blocked weight gradient in csrc/wgrad.hip), kernel level,
through the C ABI:

  * logits against torch's 1x1 convolution of the same bf16-rounded operands (unet.py:70);
  * d(logits) against the stand-alone loss kernel to one bf16 ulp (same formulas; hardware exp / log here) (abc_loss_fwd_bwd, itself held to the reference-exec golden)
    run on the fused kernel's own logits, through the packed row order -- and the reduced loss against it and the oracle;
  * g (gradient w.r.t. the BatchNorm outputs), the BatchNorm-backward sums, conv2.weight.grad and conv2.bias.grad against
    torch autograd of the ORACLE's loss (oracle/loss_oracle.py, train.py:95-137) over the same graph in f32.
Tolerances are relative L2 of whole tensors: the kernel rounds d(logits), the data gradient and g to bf16 (8 significant bits).
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import abcnet_amd  # noqa: E402,F401
from abcnet_amd import _lib as L  # noqa: E402
from abcnet_amd.dropout import keep_mask  # noqa: E402
from abcnet_amd.engine import head_offsets  # noqa: E402
from abcnet_amd.ops import FusedLoss  # noqa: E402
from abcnet_amd.synthetic import synthetic_targets  # noqa: E402
from oracle import loss_oracle  # noqa: E402

HEADS = [1, 14, 3, 2, 1, 360, 60, 60]
DEV = "cuda"


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _run(B, hw, drop_p, seed=3, keepmask=False):
    lib = L.load()
    g = torch.Generator().manual_seed(seed)
    npix, ld = B * hw * hw, 1024
    feat = (torch.randn((npix, ld), generator=g) * 1.5).to(torch.bfloat16)
    sc = torch.rand(ld, generator=g) * 0.8 + 0.4
    sh = torch.randn(ld, generator=g) * 0.3
    sl = torch.full((ld,), 0.01)
    mean = torch.randn(ld, generator=g) * 0.2
    invstd = torch.rand(ld, generator=g) + 0.5
    w2 = [torch.randn((c, 128), generator=g) * 0.15 for c in HEADS]
    b2 = [torch.randn((c,), generator=g) * 0.5 for c in HEADS]
    s = torch.rand(10, generator=g) * 0.4 - 0.2
    tg = synthetic_targets(B, hw, seed=1, n_atoms=12, n_bonds=14)
    dseed = 0x1234567

    dev = lambda t: t.to(DEV).contiguous()
    d = L.HeadsFusedDesc()
    keep = {k: dev(v) for k, v in dict(feat=feat, sc=sc, sh=sh, sl=sl, mean=mean, invstd=invstd).items()}
    d.feat, d.ld = keep["feat"].data_ptr(), ld
    d.scale, d.shift, d.slope, d.mean, d.invstd = (keep[k].data_ptr() for k in ("sc", "sh", "sl", "mean", "invstd"))
    d.drop_p, d.drop_seed, d.drop_salt = drop_p, dseed, None
    w2d, b2d = [dev(t) for t in w2], [dev(t) for t in b2]
    pack = torch.zeros(lib.abc_heads_fused_pack_bytes(), dtype=torch.uint8, device=DEV)
    logits = [torch.zeros((B, c, hw, hw), device=DEV) for c in HEADS]
    tgd = [dev(t) for t in tg]
    for i in range(8):
        d.w2[i], d.b2[i], d.logits[i] = w2d[i].data_ptr(), b2d[i].data_ptr(), logits[i].data_ptr()
    d.w2_pack = pack.data_ptr()
    (d.t_atom, d.t_types, d.t_charges, d.t_hs, d.t_bond, d.t_btypes, d.t_rho, d.t_omega) = (t.data_ptr() for t in tgd)
    d.B, d.h, d.w = B, hw, hw
    nchunk = lib.abc_heads_fused_chunks(C.byref(d))
    assert nchunk == npix // 128
    dl = torch.zeros(lib.abc_heads_fused_dl_elems(C.byref(d)), dtype=torch.bfloat16, device=DEV)
    gbuf = torch.zeros((npix, ld), dtype=torch.bfloat16, device=DEV)
    bnp = torch.zeros((nchunk, 2, ld), device=DEV)
    nlb = lib.abc_heads_fused_loss_blocks(C.byref(d))
    lp = torch.zeros((nlb, 16), dtype=torch.float64, device=DEV)
    d.dl, d.g, d.bn_partial, d.loss_partial = dl.data_ptr(), gbuf.data_ptr(), bnp.data_ptr(), lp.data_ptr()
    work = torch.zeros(lib.abc_heads_fused_wgrad_floats(C.byref(d)), device=DEV)
    d.wgrad_work = work.data_ptr()
    kmask = torch.zeros((3, npix, 2, 8), dtype=torch.uint8, device=DEV) if keepmask else None
    if keepmask:
        d.keep_mask = kmask.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.abc_heads_fused_pack(C.byref(d), st), "pack")
    L.check(lib.abc_heads_fused_fwd_bwd(C.byref(d), st), "fwd_bwd")
    # loss finalisation on the fused partial sums -> per-channel factors
    off = head_offsets(HEADS)
    chan_scale = torch.zeros(sum(HEADS), device=DEV)
    out = torch.zeros(17, dtype=torch.float64, device=DEV)
    sdev, ds = dev(s), torch.zeros(10, device=DEV)
    f = L.LossFinDesc()
    f.partial, f.nblk, f.s, f.ds, f.out = lp.data_ptr(), nlb, sdev.data_ptr(), ds.data_ptr(), out.data_ptr()
    f.chan_scale, f.nchan, f.grad_scale = chan_scale.data_ptr(), chan_scale.numel(), 1.0
    for i in range(8):
        f.chan_off[i], f.head_c[i] = off[i], HEADS[i]
    L.check(lib.abc_loss_finalize(C.byref(f), st), "loss_finalize")
    # weight / bias gradients
    d.chan_scale = chan_scale.data_ptr()
    dw2 = [torch.zeros((c, 128), device=DEV) for c in HEADS]
    db2 = [torch.zeros((c,), device=DEV) for c in HEADS]
    for i in range(8):
        d.chan_off[i], d.dw2[i], d.db2[i] = off[i], dw2[i].data_ptr(), db2[i].data_ptr()
    L.check(lib.abc_heads_fused_wgrad(C.byref(d), st), "wgrad")
    torch.cuda.synchronize()

    # ---- stand-alone loss kernel on the fused kernel's logits
    class E:
        pass

    e = E()
    e.lib, e.B, e.h, e.w, e.heads, e.head_off = lib, B, hw, hw, HEADS, off
    e.logits = logits
    e.dlogits = [torch.zeros_like(t) for t in logits]
    e.chan_scale = torch.zeros(sum(HEADS), device=DEV)
    ds2 = torch.zeros(10, device=DEV)
    fl = FusedLoss(e, tgd, sdev.data_ptr(), ds2.data_ptr())
    fl.run(st)
    torch.cuda.synchronize()

    # ---- torch autograd of the oracle's loss over the same graph (f32, same bf16-rounded operands, same dropout mask)
    x = feat.float()
    y = x * sc + sh
    a = torch.maximum(y, sl * y)
    idx = torch.arange(npix * ld, dtype=torch.int64).view(npix, ld)
    km = keep_mask(idx, dseed, drop_p).float() / (1.0 - drop_p) if drop_p > 0 else torch.ones_like(a)
    a = (a * km).to(torch.bfloat16).float()
    leaves, preds, ws, bs = [], [], [], []
    for i, c in enumerate(HEADS):
        ai = a[:, 128 * i:128 * (i + 1)].reshape(B, hw, hw, 128).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        wi = w2[i].to(torch.bfloat16).float().requires_grad_(True)
        bi = b2[i].clone().requires_grad_(True)
        p = F.conv2d(ai, wi.view(c, 128, 1, 1), bi)
        p.retain_grad()
        leaves.append(ai); preds.append(p); ws.append(wi); bs.append(bi)
    total, _, _ = loss_oracle.abc_loss(preds, tg, s)
    total.backward()
    return dict(lib=lib, d=d, B=B, hw=hw, ld=ld, npix=npix, nchunk=nchunk, logits=logits, dl=dl, g=gbuf, bnp=bnp, out=out, ds=ds,
                chan_scale=chan_scale, off=off, dw2=dw2, db2=db2, alone=e, alone_out=fl.out, alone_ds=ds2,
                ref=dict(total=total.item(), preds=preds, leaves=leaves, ws=ws, bs=bs, y=y, km=km, x=x, mean=mean, invstd=invstd, sl=sl),
                kmask=kmask, keep=(keep, w2d, b2d, pack, tgd, lp, work, sdev))


@pytest.mark.parametrize("B,hw,drop_p,keepmask", [(2, 32, 0.2, False), (1, 48, 0.0, False), (2, 32, 0.2, True)])
def test_fused_heads_pass(B, hw, drop_p, keepmask):
    """keepmask: abc_heads_fused_desc.keep_mask set -- the fused pass leaves the wide heads' dropout keep bits for its conv2
    weight gradient (checked against the host mirror of the hash below, and through dW2 of heads 5-7 like the hashed form)"""
    r = _run(B, hw, drop_p, keepmask=keepmask)
    lib, ref = r["lib"], r["ref"]
    # 1. logits
    for i, c in enumerate(HEADS):
        got, want = r["logits"][i].cpu(), ref["preds"][i].detach()
        assert (got - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item()), ("logits", i)
    # 2. the loss: against the stand-alone kernel on the same logits, and the oracle
    o, oa = r["out"].cpu(), r["alone_out"].cpu()
    # (the fused kernel evaluates exp / log / 1/x with the hardware approximations, ~1e-6 relative: loss_math.hpp)
    assert torch.allclose(o, oa, rtol=5e-6, atol=1e-12), (o, oa)
    assert torch.allclose(r["ds"].cpu(), r["alone_ds"].cpu(), rtol=2e-5, atol=1e-8)
    assert torch.allclose(r["chan_scale"].cpu(), r["alone"].chan_scale.cpu(), rtol=1e-6, atol=0)
    assert abs(o[0].item() - ref["total"]) <= 2e-5 * abs(ref["total"])
    # 3. d(logits): the blocked bf16 buffer holds bf16(the stand-alone kernel's values), row by row
    nchunk, row0 = r["nchunk"], 0
    dl = r["dl"].cpu()
    for i, c in enumerate(HEADS):
        rows = lib.abc_heads_fused_rows(i)
        blk = dl[row0 * nchunk * 128:(row0 + rows) * nchunk * 128].view(nchunk, rows, 128)
        want = r["alone"].dlogits[i].cpu().to(torch.bfloat16)            # [B][c][HW]
        want = want.view(B, c, -1).permute(1, 0, 2).reshape(c, nchunk, 128)  # channel, chunk, pixel
        for m in range(rows):
            ch = lib.abc_heads_fused_chan_of_row(i, m)
            if ch < 0:
                assert not blk[:, m, :].any(), ("padding row not zero", i, m)
            else:
                # equal up to the hardware exp / log (a few f32 ulps before the bf16 rounding: at most one bf16 ulp apart,
                # and only where the f32 value sat on a rounding boundary)
                a_, b_ = blk[:, m, :].float(), want[ch].float()
                assert ((a_ - b_).abs() <= 2.0 ** -7 * b_.abs() + 1e-30).all(), ("d(logits)", i, m, ch)
                assert (a_ != b_).float().mean().item() <= 0.02, ("d(logits) mismatches", i, m, ch, (a_ != b_).float().mean().item())
        row0 += rows
    # 4. g and the BatchNorm-backward sums: autograd's d(loss)/d(features) through LeakyReLU' and the dropout mask
    npix, ld = r["npix"], r["ld"]
    cs = r["chan_scale"].cpu()
    g = r["g"].float().cpu()
    lk = torch.where(ref["y"] > 0, torch.ones_like(ref["y"]), ref["sl"].expand_as(ref["y"])) * ref["km"]
    xhat = (ref["x"] - ref["mean"]) * ref["invstd"]
    part = r["bnp"].cpu().double().sum(0)
    for i, c in enumerate(HEADS):
        da = ref["leaves"][i].grad.permute(0, 2, 3, 1).reshape(npix, 128)
        want = da * lk[:, 128 * i:128 * (i + 1)]
        got = g[:, 128 * i:128 * (i + 1)] * cs[r["off"][i]]
        assert rel(got, want) <= 1e-2, ("g", i, rel(got, want))
        s1, s2 = want.double().sum(0), (want.double() * xhat[:, 128 * i:128 * (i + 1)].double()).sum(0)
        k1, k2 = part[0, 128 * i:128 * (i + 1)] * cs[r["off"][i]].double(), part[1, 128 * i:128 * (i + 1)] * cs[r["off"][i]].double()
        scale = want.double().abs().sum(0).mean()   # (sums of signed terms: error relative to the sum of magnitudes)
        assert (k1 - s1).abs().max().item() <= 5e-3 * scale.item(), ("bn sum g", i)
        assert (k2 - s2).abs().max().item() <= 1e-2 * scale.item(), ("bn sum g xhat", i)
    # 5. conv2 weight / bias gradients
    for i, c in enumerate(HEADS):
        assert rel(r["dw2"][i].cpu(), ref["ws"][i].grad) <= 6e-3, ("dW2", i, rel(r["dw2"][i].cpu(), ref["ws"][i].grad))
        assert rel(r["db2"][i].cpu(), ref["bs"][i].grad) <= 6e-3, ("db2", i, rel(r["db2"][i].cpu(), ref["bs"][i].grad))
    # 6. the keep bits: byte kk of half h of a pixel = channels 16 kk + 8 h .. + 7 of the head's 128 features
    if keepmask:
        km = (ref["km"] > 0).view(npix, 8, 128)                       # [pixel][head][channel]
        bits = r["kmask"].cpu()                                        # [3][pixel][h][kk]
        for i in (5, 6, 7):
            want = km[:, i, :].view(npix, 8, 2, 8).permute(0, 2, 1, 3)   # [pixel][h][kk][j]
            wbyte = (want.to(torch.int32) << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8)
            assert torch.equal(bits[i - 5], wbyte), ("keep bits", i)


@pytest.mark.parametrize("keep_logits", [True, False])
def test_sparse_targets_give_the_same_step_bit_for_bit(keep_logits):
    """Trainer.use_sparse_targets (abc_heads_fused_desc.target_flags): the fused heads pass reads a head's target planes only in the
    32-pixel groups the rasteriser flagged; everywhere else the same loads hit 512 zero bytes.  Same records through the sparse
    rasteriser + flags and through the plain rasteriser + every plane: the loss terms, d(logits) in the stored logits' gradient
    chain -- i.e. EVERY parameter gradient -- and the parameters after three steps (three different batches: the sparse form erases
    its previous drawing) are bit-identical; and the flags did switch reads off (most groups carry no target).  Since round 5 a wave
    without a target of a softmax head also SKIPS that head's loss / data gradient / epilogue (run_head_skip: logits + zeros) -- the
    stored logits must be the same bits too, and with keep_logits=False (nothing of the head's forward is observable) the skipped
    waves write zeros only."""
    from abcnet_amd.raster import TargetRasterizer, parse_record
    from abcnet_amd.synthetic import random_annotations, synthetic_images
    from abcnet_amd.train import Trainer
    from abcnet_amd.unet import UNet
    B, S = 2, 192
    h = S // 4
    res = []
    for sparse in (False, True):
        m = UNet(1, HEADS, dtype="bf16", dropout_p=0.2)
        m.reset_parameters(seed=77)
        m = m.to(DEV)
        tr = Trainer(m, B, S, S, use_graph=False, keep_logits=keep_logits)
        rz = TargetRasterizer(B, h, max_atoms=64, max_bonds=64, targets=tr.targets, sparse=sparse)
        if sparse:
            tr.use_sparse_targets(rz)
            with pytest.raises(Exception, match="sparse"):
                tr.load_batch(synthetic_images(B, S, seed=1).to(DEV), [t.clone() for t in tr.targets])
        grads, losses, logits = [], [], []
        for step in range(3):
            tr.load_batch(synthetic_images(B, S, seed=11 + step).to(DEV))
            rz.load([parse_record(*random_annotations(12 + 9 * step, 10 + 11 * step, 4000 + 10 * step + i, size=S), h=h) for i in range(B)])
            rz.run()
            tr.step()
            torch.cuda.synchronize()
            grads.append(m._flat_grad.clone())
            losses.append(tr.loss.out.clone())
            if keep_logits:
                logits.append([t.clone() for t in tr.eng.logits])
        frac = None
        if sparse:
            f = rz.group_flags.cpu().numpy().astype("uint32")
            frac = float(((f & 0xFF) != 0).mean())
        res.append((grads, losses, m._flat.detach().clone(), frac, logits))
        del tr, rz, m
    (g0, l0, p0, _, z0), (g1, l1, p1, frac, z1) = res
    for step in range(3):
        if keep_logits:
            for i in range(8):
                assert torch.equal(z0[step][i], z1[step][i]), ("stored logits", step, i)
        assert torch.equal(l0[step], l1[step]), ("loss terms", step, l0[step], l1[step])
        assert torch.equal(g0[step], g1[step]), ("gradients", step, (g0[step] - g1[step]).abs().max().item())
    assert torch.equal(p0, p1)
    assert 0.0 < frac < 1.0, frac      # (48 x 48 maps with up to 62 items: most groups carry something; still not all)
