"""GPU parity tests, model level: the HIP U-Net (through the drop-in nn.Module and through the
fused training step) against the oracle and the committed golden vectors.

Bar (BASELINE.json north_star): logits within 1e-3 of the PyTorch reference in fp32 mode
on identical inputs.  bf16 mode is the throughput mode and is held to its own, measured,
looser bound (the reference itself moves by 0.3-0.5 under bf16 autocast)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import abcnet_amd  # noqa: E402,F401
from abcnet_amd import _lib as L  # noqa: E402
from abcnet_amd.synthetic import synthetic_images, synthetic_targets  # noqa: E402
from abcnet_amd.unet import UNet  # noqa: E402
from abcnet_amd.unet2 import UNet as UNet2  # noqa: E402
from oracle import loss_oracle, nms_oracle  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

HEADS = uo.HEADS
DEV = "cuda"
PRE_BN_BIAS = ("double_conv.0.bias", "double_conv.3.bias", "conv1.bias")


def make_model(dtype="fp32", dropout_p=0.0, seed=0, variant="unet"):
    cls = UNet if variant == "unet" else UNet2
    m = cls(1, HEADS, dtype=dtype, dropout_p=dropout_p)
    m.load_state_dict(uo.filled_state(variant, 1, HEADS, seed=seed))
    return m.to(DEV)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_forward_matches_golden_fp32(mode, golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_unet_64.npz"))
    m = make_model()
    m.train(mode == "train")
    x = synthetic_images(2, 64, seed=7).to(DEV)
    with torch.no_grad():
        ys = m(x)
    assert isinstance(ys, list) and len(ys) == 8
    for i, y in enumerate(ys):
        ref = gold["%s_head%d" % (mode, i)]
        assert tuple(y.shape) == ref.shape and y.dtype == torch.float32
        err = np.abs(y.cpu().numpy() - ref).max()
        assert err < 1e-3, (mode, i, err)
    if mode == "train":
        sd = m.state_dict()
        for k in gold.files:
            if k.startswith("rs_"):
                np.testing.assert_allclose(sd[k[3:]].cpu().numpy(), gold[k], rtol=1e-4, atol=1e-5)
        assert int(sd["inc1.double_conv.1.num_batches_tracked"]) == int(gold["nbt"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_forward_matches_oracle_384_fp32(mode, golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_unet_384.npz"))
    m = make_model()
    m.train(mode == "train")
    x = synthetic_images(2, 384, seed=7)
    with torch.no_grad():
        ys = m(x.to(DEV))
        ref = uo.forward("unet", uo.filled_state("unet", 1, HEADS, seed=0), x, train=(mode == "train"))
    for i, (y, r) in enumerate(zip(ys, ref)):
        err = (y.cpu() - r).abs().max().item()
        assert err < 1e-3, (mode, i, err)
        f = y.cpu().reshape(-1)
        step = max(f.numel() // 257, 1)
        np.testing.assert_allclose(f[::step][:257].double().numpy(), gold["%s_head%d_sample" % (mode, i)], atol=1e-3)


def test_forward_bf16_bound():
    """bf16 throughput mode: eval-mode logits stay within a few 1e-2 of the fp32 oracle at fresh weights,
    train-mode (34 re-normalisations) within the deviation the reference itself shows under autocast"""
    x = synthetic_images(2, 128, seed=7)
    for mode, bound in (("eval", 0.08), ("train", 0.6)):
        m = make_model("bf16")
        m.train(mode == "train")
        with torch.no_grad():
            ys = m(x.to(DEV))
            ref = uo.forward("unet", uo.filled_state("unet", 1, HEADS, seed=0), x, train=(mode == "train"))
        worst = max((y.cpu() - r).abs().max().item() for y, r in zip(ys, ref))
        assert worst < bound, (mode, worst)


def _oracle_grads(x, targets, dropout_masks=None, dtype=torch.float32, variant="unet"):
    sd0 = uo.filled_state(variant, 1, HEADS, seed=0)
    sd = uo.clone_state({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd0.items()}, requires_grad=True)
    dm = None if dropout_masks is None else [m.to(dtype) for m in dropout_masks]
    preds = uo.forward(variant, sd, x.to(dtype), train=True, dropout_masks=dm)
    total, weighted, terms = loss_oracle.abc_loss(preds, [t.to(dtype) for t in targets], sd["s"])
    total.backward()
    return sd, total, weighted, preds


def _check_grads(get_grad, sd32, sd64):
    """Gradient parity bar.  ReLU / max-pool decisions flip on 1e-7 perturbations, so the reference's OWN
    fp32 gradients differ from its fp64 gradients by up to ~1e-2 (relative L2) in the earliest layers
    (measured: 4e-3..1.5e-2 at inc1/inc2, 1e-6 at the heads), and ONE flipped ReLU among the 262,144 outputs of a
    decoder block moves every gradient upstream of it by ~2e-3 (measured).  Any fp32 implementation whose forward
    differs in the last bits flips a different handful of decisions, so the end-to-end bar is
        ||g_hip - g_f64|| <= max(2.5 * ||g_cpu32 - g_f64|| + 1e-3 * ||g_f64||, 2e-2 * ||g_f64||)      per parameter
    (2e-2 = what the reference's own fp32 run shows against fp64 in its worst layer, 1.5e-2, with margin),
    where for the 1-D BN gamma/beta gradients (a flipped pixel lands in exactly ONE entry with its full d(loss)/d(act))
    the two largest entry deviations are set aside first,
    while the flip-free statement -- every backward kernel is exact with respect to its own inputs -- is
    test_backward_chain_is_exact_in_situ below (1e-6) and tests/test_gpu_kernels.py (1e-4)."""
    bad = []
    for name, t in sd64.items():
        if t.grad is None or name.endswith(PRE_BN_BIAS):
            continue
        ref = t.grad.double()
        floor = (sd32[name].grad.double() - ref).norm().item()
        got = get_grad(name).double().cpu()
        dev = (got - ref).abs().flatten()
        if ref.ndim == 1 and dev.numel() > 8:
            dev[dev.topk(2).indices] = 0.0
        e = dev.norm().item()
        if not e <= max(2.5 * floor + 1e-3 * ref.norm().item(), 2e-2 * ref.norm().item()):
            bad.append((name, e / (ref.norm().item() + 1e-30), floor / (ref.norm().item() + 1e-30)))
    assert not bad, bad[:8]


def test_compat_path_autograd_matches_oracle():
    """the reference training-loop body (model(x) -> torch loss -> backward) on the drop-in module"""
    B, S = 2, 128
    x = synthetic_images(B, S, seed=7)
    tg = synthetic_targets(B, S // 4, seed=1)
    sd, total, _, _ = _oracle_grads(x, tg)
    sd64 = _oracle_grads(x, tg, dtype=torch.float64)[0]
    m = make_model()
    m.train()
    preds = m(x.to(DEV))
    loss, _, _ = loss_oracle.abc_loss(preds, [t.to(DEV) for t in tg], m.s)  # torch ops on device = the reference's loss code
    opt = torch.optim.Adam(m.parameters(), lr=2.5e-4, weight_decay=1e-8)   # train.py:55 on the 159 named tensors
    opt.zero_grad()
    loss.backward()
    assert abs(loss.item() - total.item()) < 1e-4 * abs(total.item())
    named = dict(m.named_parameters())
    assert list(named) == [k for k, v in sd.items() if v.requires_grad] and len(named) == 159
    assert all(p.grad is not None and p.grad.shape == p.shape for p in named.values())
    _check_grads(lambda name: named[name].grad, sd, sd64)
    # optimizer.step() (train.py:141) on the named tensors moves the arena the kernels read, as the oracle's Adam does
    from oracle import adam_oracle
    before = {k: v.detach().clone() for k, v in named.items()}
    opt.step()
    for name in ("out_modules.5.conv2.weight", "dconv1.double_conv.1.weight", "up2.up.weight", "s"):
        p = before[name].cpu().reshape(-1).clone()
        n = p.numel()
        adam_oracle.adam_step(p, named[name].grad.cpu().reshape(-1), torch.zeros(n), torch.zeros(n), 1)
        assert (named[name].detach().cpu().reshape(-1) - p).abs().max().item() < 1e-6, name
        off, cnt = m._lay_p[name]
        assert torch.equal(m._flat[off:off + cnt], named[name].detach().reshape(-1)), "parameter is not a view of the arena"
    with torch.no_grad():
        m.eval()
        y2 = m(x.to(DEV))
    assert all(torch.isfinite(t).all() for t in y2)


def test_named_parameters_follow_the_arena_across_devices():
    """unet.py:78-98 surface: 159 named tensors / 10,698,575 values, state_dict round trip, .to(device) keeps them views of
    ONE arena, per-tensor optimiser groups work"""
    m = UNet(1, HEADS)
    assert sum(p.numel() for p in m.parameters()) == 10698575 and len(list(m.parameters())) == 159
    m = m.to(DEV)
    base = m._flat.data_ptr()
    for (name, p), (off, cnt, shape) in zip(m.named_parameters(), m._param_slices):
        assert p.is_cuda and p.data_ptr() == base + 4 * off and tuple(p.shape) == shape, name
    # per-layer learning rates (anything per-tensor) work on the named tensors
    groups = [{"params": [p for n, p in m.named_parameters() if n.startswith("out_modules")], "lr": 1e-3},
              {"params": [p for n, p in m.named_parameters() if not n.startswith("out_modules")], "lr": 1e-4}]
    torch.optim.Adam(groups)
    with pytest.raises(L.AbcNetHipError):
        m.half()
    with pytest.raises(L.AbcNetHipError):
        m(torch.zeros(1, 1, 64, 64))   # CPU tensor: no fallback


def test_three_input_channels_match_oracle():
    """unet.py:122-134 builds UNet(in_channels=3, ...): the first convolution reads the NCHW image channel-planar"""
    B, S = 2, 64
    x = synthetic_images(B, S, seed=7, in_channels=3)
    tg = synthetic_targets(B, S // 4, seed=1)
    sd0 = uo.filled_state("unet", 3, HEADS, seed=0)
    m = UNet(3, HEADS, dtype="fp32", dropout_p=0.0)
    m.load_state_dict(sd0)
    m = m.to(DEV)
    for mode in ("eval", "train"):
        m.train(mode == "train")
        with torch.no_grad():
            ys = m(x.to(DEV))
            ref = uo.forward("unet", uo.clone_state(sd0), x, train=(mode == "train"))
        err = max((y.cpu() - r).abs().max().item() for y, r in zip(ys, ref))
        assert err < 1e-3, (mode, err)
    m.train()
    preds = m(x.to(DEV))
    loss, _, _ = loss_oracle.abc_loss(preds, [t.to(DEV) for t in tg], m.s)
    loss.backward()
    sd = uo.clone_state(sd0, requires_grad=True)
    total, _, _ = loss_oracle.abc_loss(uo.forward("unet", sd, x, train=True), tg, sd["s"])
    total.backward()
    g, r = dict(m.named_parameters())["inc1.double_conv.0.weight"].grad.cpu(), sd["inc1.double_conv.0.weight"].grad
    assert g.shape == (16, 3, 3, 3)
    assert (g - r).norm().item() <= 2e-2 * r.norm().item(), ((g - r).norm().item(), r.norm().item())


def test_fused_train_step_matches_oracle():
    """fast path: forward + fused loss + backward + fused Adam in HIP vs oracle forward/backward + oracle Adam"""
    from abcnet_amd.train import Trainer
    from abcnet_amd.dropout import head_keep_masks
    from oracle import adam_oracle
    B, S = 2, 128
    x = synthetic_images(B, S, seed=7)
    tg = synthetic_targets(B, S // 4, seed=1)
    m = make_model(dropout_p=0.2)
    tr = Trainer(m, B, S, S, use_graph=False)
    masks = head_keep_masks(B, S // 4, S // 4, 8, tr.eng.dropout_seed(1), 0.2)   # the mask of the first forward
    sd, total, weighted, _ = _oracle_grads(x, tg, dropout_masks=masks)
    sd64 = _oracle_grads(x, tg, dropout_masks=masks, dtype=torch.float64)[0]
    p0 = m._flat.data.clone()
    tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
    tr.step()
    torch.cuda.synchronize()
    res = tr.loss_value()
    assert abs(res["total"] - total.item()) < 2e-4 * abs(total.item()), (res["total"], total.item())
    for k_or, k_us in (("atom_t", "atom_t"), ("bond_t", "bond_t"), ("atom_types", "atom_types"), ("atom_charges", "atom_charges"),
                       ("bond_types", "bond_types"), ("bond_rhos", "bond_rhos"), ("bond_omega", "bond_omega"), ("atom_hs", "atom_hs")):
        assert abs(res[k_us] - weighted[k_or].item()) < 5e-4 * abs(weighted[k_or].item()) + 1e-6, k_or
    _check_grads(lambda n: m.grad_of(n), sd, sd64)
    # one Adam step from the ORACLE gradients must land where the fused optimiser landed
    worst = 0.0
    for name, t in sd.items():
        if t.grad is None or name.endswith(PRE_BN_BIAS):
            continue
        off, n = m._lay_p[name]
        p = p0[off:off + n].cpu().clone()
        adam_oracle.adam_step(p, t.grad.reshape(-1).float(), torch.zeros(n), torch.zeros(n), 1)
        worst = max(worst, (m._flat.data[off:off + n].cpu() - p).abs().max().item())
    assert worst < 6e-4, worst  # |delta| per step is lr=2.5e-4; sign flips of ~zero gradients allowed


def test_backward_chain_is_exact_in_situ():
    """Flip-free gradient parity: run one fused step (lr = 0 so the weights stay put) and check every link of the
    backward chain against torch ops applied to the engine's OWN tensors: data gradients == conv_transpose2d(dY, W),
    BN/ReLU backward (dgamma, dbeta, dY) == the closed form, and d(loss)/d(activation) == the oracle's autograd."""
    import torch.nn.functional as F
    from abcnet_amd.train import Trainer
    B, S = 2, 128
    x = synthetic_images(B, S, seed=7)
    tg = synthetic_targets(B, S // 4, seed=1)
    m = make_model(dropout_p=0.0)
    tr = Trainer(m, B, S, S, lr=0.0, use_graph=False)
    tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
    tr.step()
    torch.cuda.synchronize()
    sdm = m.state_dict()
    recs = {r.cname: r for r in tr.eng.recs if r.kind == "conv"}
    for cn in ("dconv2.double_conv.3", "dconv2.double_conv.0", "dconv1.double_conv.3", "up3.conv.double_conv.3",
               "down5.maxpool_conv.1.double_conv.3", "inc2.double_conv.3"):
        r = recs[cn]
        dY = r.dY.float().permute(0, 3, 1, 2)
        ref = F.conv_transpose2d(dY, sdm[cn + ".weight"], padding=1)
        got = r.src.producer.grad_same[0].float().permute(0, 3, 1, 2)
        assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-9, cn
    for cn, bn in (("dconv1.double_conv.3", "dconv1.double_conv.4"), ("up2.conv.double_conv.0", "up2.conv.double_conv.1")):
        r = recs[cn]
        dA = r.grad_same[0].float()[..., r.grad_same[2]:r.grad_same[2] + r.cout]
        y = r.y.float()[..., r.coff:r.coff + r.cout]
        G = dA * ((y * r.scale + r.shift) > 0).float()
        n = G.shape[0] * G.shape[1] * G.shape[2]
        xh = (y - r.mean) * r.invstd
        dbeta, dgamma = G.sum((0, 1, 2)), (G * xh).sum((0, 1, 2))
        assert (dbeta - m.grad_of(bn + ".bias")).abs().max().item() <= 1e-5 * dbeta.abs().max().item()
        assert (dgamma - m.grad_of(bn + ".weight")).abs().max().item() <= 1e-5 * dgamma.abs().max().item()
        dY = sdm[bn + ".weight"] * r.invstd * (G - dbeta / n - xh * dgamma / n)
        assert (dY - r.dY.float()).abs().max().item() <= 1e-5 * dY.abs().max().item()
    # against the oracle's autograd: d(loss)/d(logits) has no decision in between (continuous in the logits);
    # d(loss)/d(trunk) has the head's own ReLU in between, where a flipped decision moves the 3x3x128 trunk gradients
    # under it (0.44 % of this tensor per flip) -- all but 2 % of the elements must agree to 2e-5 of the largest
    sd = uo.clone_state(uo.filled_state("unet", 1, HEADS, seed=0), requires_grad=True)
    preds, trunk = uo.forward("unet", sd, x, train=True, return_trunk=True)
    trunk.retain_grad()
    for p_ in preds:
        p_.retain_grad()
    loss_oracle.abc_loss(preds, tg, sd["s"])[0].backward()
    for i, p_ in enumerate(preds):
        hc = HEADS[i]
        cs = tr.eng.chan_scale[tr.eng.head_off[i]:tr.eng.head_off[i] + hc]
        dl = (tr.eng.dlogits[i].float() * cs.view(1, -1, 1, 1)).cpu()
        assert (dl - p_.grad).abs().max().item() <= 1e-4 * p_.grad.abs().max().item() + 1e-9, i
    got = tr.eng.trunk.producer.grad_same[0].float().permute(0, 3, 1, 2).cpu()
    off = ((got - trunk.grad).abs() > 2e-5 * trunk.grad.abs().max()).float().mean().item()
    assert off <= 2e-2, off


def test_fused_loss_matches_golden(golden_dir):
    """loss kernel alone on the seeded logits/targets of tests/golden/loss_128.npz"""
    from abcnet_amd.engine import head_offsets
    from abcnet_amd.ops import FusedLoss
    gold = np.load(os.path.join(golden_dir, "loss_128.npz"))
    g = torch.Generator().manual_seed(11)
    preds = [torch.randn((2, c, 128, 128), generator=g) * 2.0 for c in HEADS]
    tg = synthetic_targets(2, 128, seed=1)
    s = torch.rand(10, generator=g) * 0.4 - 0.2

    class E:  # the slice of Engine the loss wrapper needs
        pass

    e = E()
    e.lib, e.B, e.h, e.w, e.heads, e.head_off = L.load(), 2, 128, 128, HEADS, head_offsets(HEADS)
    e.logits = [p.to(DEV).contiguous() for p in preds]
    e.dlogits = [torch.zeros_like(t) for t in e.logits]
    e.chan_scale = torch.zeros(sum(HEADS), device=DEV)
    sdev, ds = s.to(DEV), torch.zeros(10, device=DEV)
    fl = FusedLoss(e, [t.to(DEV) for t in tg], sdev.data_ptr(), ds.data_ptr())
    fl.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    r = fl.result()
    assert abs(r["total"] - gold["loss"].item()) < 1e-5 * abs(gold["loss"].item())
    np.testing.assert_allclose(ds.cpu().double().numpy(), gold["ds"], rtol=1e-4, atol=1e-7)
    for i, c in enumerate(HEADS):
        gl = (e.dlogits[i] * e.chan_scale[e.head_off[i]]).cpu()
        f = gl.reshape(-1)
        step = max(f.numel() // 1031, 1)
        np.testing.assert_allclose(f[::step][:1031].double().numpy(), gold["dlogit%d_sample" % i], rtol=2e-3, atol=1e-7)
        assert abs(gl.double().norm().item() - gold["dlogit%d_norm" % i].item()) < 1e-4 * gold["dlogit%d_norm" % i].item()


def test_nms_matches_golden(golden_dir):
    from abcnet_amd.ops import nms_peaks
    gold = np.load(os.path.join(golden_dir, "nms_128.npz"))
    g = torch.Generator().manual_seed(13)
    a = torch.randn((2, 1, 128, 128), generator=g) * 2
    b = torch.randn((2, 1, 128, 128), generator=g) * 2
    rho = torch.randn((2, 60, 128, 128), generator=g) * 3
    _ = torch.randn((2, 360, 128, 128), generator=g)
    om = torch.round(torch.randn((2, 60, 128, 128), generator=g) * 4) / 4
    am, bm, r, omm = nms_peaks(a.to(DEV), b.to(DEV), rho.to(DEV), om.to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(np.packbits(am.cpu().numpy().astype(np.uint8)), gold["atom_mask"])
    assert np.array_equal(np.packbits(bm.cpu().numpy().astype(np.uint8)), gold["bond_mask"])
    assert np.array_equal(np.packbits(omm.cpu().numpy().astype(np.uint8)), gold["omega_mask"])
    ra, rb, rr, ro = nms_oracle.nms(a, b, rho, om)
    assert torch.equal(r.cpu(), rr) and torch.equal(am.cpu(), ra) and torch.equal(omm.cpu(), ro)


def test_inference_prologue_matches_oracle():
    """img2smiles2.py:42-79 end to end: eval forward + NMS on the HIP path vs oracle forward + oracle NMS.
    Peak masks are decisions on logits that differ by ~1e-5, so allow a handful of flips."""
    x = synthetic_images(2, 128, seed=7)
    m = make_model()
    m.eval()
    with torch.no_grad():
        am, bm, r, om = m.nms(x.to(DEV))
        ref = uo.forward("unet", uo.filled_state("unet", 1, HEADS, seed=0), x, train=False)
    ra, rb, rr, ro = nms_oracle.nms(ref[0], ref[4], ref[6], ref[7])
    assert (am.cpu() != ra).sum().item() <= 2 and (bm.cpu() != rb).sum().item() <= 2
    assert (om.cpu() != ro).float().mean().item() < 1e-3
    assert (r.cpu() - rr).abs().max().item() < 1e-3


def test_inference_runner_graph_matches_module_path():
    """InferenceRunner (static buffers, packed once, hipGraph replay) == model.nms() bit for bit, and the logits it
    leaves behind are the oracle's within the 1e-3 bar; a second batch through the same graph gives ITS results."""
    from abcnet_amd.infer import InferenceRunner
    m = make_model()
    m.eval()
    run = InferenceRunner(m, 2, 128, 128, use_graph=True, fold_bn=False)   # (the module path keeps BatchNorm on load: same arithmetic)
    for seed in (7, 8, 9):   # step 0 eager, step 1 captures, step 2 replays
        x = synthetic_images(2, 128, seed=seed)
        run.load_batch(x.to(DEV))
        run.step()
        torch.cuda.synchronize()
        got = [t.clone() for t in (run.atom_mask, run.bond_mask, run.rho_abs, run.omega_mask)]
        lg = [t.clone() for t in run.logits]
        with torch.no_grad():
            want = m.nms(x.to(DEV))
        for g, w in zip(got, want):
            assert torch.equal(g, w)
        ref = uo.forward("unet", uo.filled_state("unet", 1, HEADS, seed=0), x, train=False)
        assert max((a.cpu() - b).abs().max().item() for a, b in zip(lg, ref)) < 1e-3


def test_state_dict_roundtrip_and_module_prefix():
    m = make_model()
    sd = m.state_dict()
    ref = uo.filled_state("unet", 1, HEADS, seed=0)
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(sd[k].cpu(), ref[k]), k
    wrapped = torch.nn.DataParallel(m)
    sd2 = wrapped.state_dict()
    assert all(k.startswith("module.") for k in sd2)
    m2 = UNet(1, HEADS).to(DEV)
    m2.load_state_dict(sd2)  # train.py:435 checkpoint -> img2smiles2.py:43-44
    assert torch.equal(m2._flat.data, m._flat.data)


def test_fails_loudly_on_cpu_tensor():
    m = UNet(1, HEADS)
    with pytest.raises(L.AbcNetHipError):
        m(torch.zeros(1, 1, 64, 64))


# ---------------------------------------------------------------------------------------------------------------
# unet2 (CBAM + residual variant, /root/reference/src/unet2.py) -- BASELINE.json config 3

@pytest.mark.parametrize("mode", ["eval", "train"])
def test_unet2_forward_matches_golden_fp32(mode, golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_unet2_64.npz"))
    m = make_model(variant="unet2")
    m.train(mode == "train")
    x = synthetic_images(2, 64, seed=7).to(DEV)
    with torch.no_grad():
        ys = m(x)
    for i, y in enumerate(ys):
        ref = gold["%s_head%d" % (mode, i)]
        assert tuple(y.shape) == ref.shape
        err = np.abs(y.cpu().numpy() - ref).max()
        assert err < 1e-3, (mode, i, err)
    if mode == "train":
        sd = m.state_dict()
        for k in gold.files:
            if k.startswith("rs_"):
                np.testing.assert_allclose(sd[k[3:]].cpu().numpy(), gold[k], rtol=1e-4, atol=1e-5)


def test_unet2_forward_384_and_bf16_bound(golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_unet2_384.npz"))
    x = synthetic_images(2, 384, seed=7)
    m = make_model(variant="unet2")
    m.train()
    with torch.no_grad():
        ys = m(x.to(DEV))
    for i, y in enumerate(ys):
        f = y.cpu().reshape(-1)
        step = max(f.numel() // 257, 1)
        np.testing.assert_allclose(f[::step][:257].double().numpy(), gold["train_head%d_sample" % i], atol=1e-3)
    xb = synthetic_images(2, 128, seed=7)
    mb = make_model("bf16", variant="unet2")
    mb.eval()
    with torch.no_grad():
        yb = mb(xb.to(DEV))
        ref = uo.forward("unet2", uo.filled_state("unet2", 1, HEADS, seed=0), xb, train=False)
    assert max((a.cpu() - b).abs().max().item() for a, b in zip(yb, ref)) < 0.1


def test_unet2_fused_train_step_matches_oracle():
    from abcnet_amd.train import Trainer
    B, S = 2, 128
    x = synthetic_images(B, S, seed=7)
    tg = synthetic_targets(B, S // 4, seed=1)
    sd, total, weighted, _ = _oracle_grads(x, tg, variant="unet2")
    sd64 = _oracle_grads(x, tg, dtype=torch.float64, variant="unet2")[0]
    m = make_model(variant="unet2")
    tr = Trainer(m, B, S, S, use_graph=False)
    tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
    tr.step()
    torch.cuda.synchronize()
    res = tr.loss_value()
    assert abs(res["total"] - total.item()) < 2e-4 * abs(total.item()), (res["total"], total.item())
    _check_grads(lambda n: m.grad_of(n), sd, sd64)


def test_trainer_checkpoint_resume_is_bit_exact():
    """Trainer.state_dict() / load_state_dict(): 2 steps + save + 2 steps == restore into a FRESH trainer + 2 steps, bit
    for bit (every kernel of the step, the split-K reductions included, is order-deterministic); the model part of the
    checkpoint is the reference's own state_dict layout"""
    from abcnet_amd.train import Trainer
    x, tg = synthetic_images(2, 64, seed=7), synthetic_targets(2, 16, seed=1)

    def fresh():
        m = make_model(dtype="bf16", dropout_p=0.2)
        tr = Trainer(m, 2, 64, 64, use_graph=True, metrics=True)
        tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
        return m, tr

    m1, t1 = fresh()
    for _ in range(2):
        t1.step()
    torch.cuda.synchronize()
    ck = t1.state_dict()
    assert list(ck["model"].keys()) == list(uo.filled_state("unet", 1, HEADS, seed=0).keys())
    for _ in range(2):
        t1.step()
    m2, t2 = fresh()
    t2.load_state_dict(ck)
    for _ in range(2):
        t2.step()
    torch.cuda.synchronize()
    assert torch.equal(m1._flat.data, m2._flat.data)
    assert torch.equal(m1._flat_buf, m2._flat_buf) and torch.equal(m1._counters, m2._counters)
    assert torch.equal(t1.opt.m, t2.opt.m) and torch.equal(t1.opt.v, t2.opt.v) and torch.equal(t1.opt.step_t, t2.opt.step_t)
    assert torch.equal(t1.metrics.totals, t2.metrics.totals)
    assert t1.loss_value()["total"] == t2.loss_value()["total"]


def test_dropout_mask_changes_every_step_also_inside_a_graph():
    """nn.Dropout (unet.py:69) draws a fresh mask per forward; the launch plan is static (and captured), so the step
    dependence is a device counter: same weights + same input must give DIFFERENT train-mode logits on consecutive
    forwards, eagerly and when replayed from the hipGraph, and each must be the oracle's under that step's mask"""
    from abcnet_amd.dropout import head_keep_masks
    B, S = 2, 64
    x = synthetic_images(B, S, seed=7)
    m = make_model(dropout_p=0.2)
    m.train()
    eng = m._engine_for(x.to(DEV), True)
    st = torch.cuda.current_stream().cuda_stream
    eng.img.copy_(x.to(DEV))
    eng.run_pack(st)
    outs = []
    for step in range(1, 3):
        eng.run_forward(st)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in eng.logits])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.run_forward(torch.cuda.current_stream().cuda_stream)
    for step in range(3, 5):
        g.replay()
        torch.cuda.synchronize()
        outs.append([t.clone() for t in eng.logits])
    for a, b in zip(outs[:-1], outs[1:]):
        assert (a[5] - b[5]).abs().max().item() > 1e-3      # consecutive forwards differ
    # BN running statistics move between the forwards but batch statistics (train mode) do not: every forward is the
    # oracle's forward under its own mask
    sd0 = uo.filled_state("unet", 1, HEADS, seed=0)
    for step, got in zip(range(1, 5), outs):
        masks = head_keep_masks(B, S // 4, S // 4, 8, eng.dropout_seed(step), 0.2)
        ref = uo.forward("unet", uo.clone_state(sd0), x, train=True, dropout_masks=masks)
        assert max((a.cpu() - r).abs().max().item() for a, r in zip(got, ref)) < 1e-3, step


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_batched_heads_equal_one_by_one_launches(variant):
    """bf16 throughput mode launches everything of the 8 heads batched (1x1 forward / data gradient / weight gradient,
    the BN -> LeakyReLU -> Dropout backward as one pass over 8 x 128 channels, batched finalisers and slab reductions).
    The same step with one launch per head (Trainer(batched_heads=False)) must give the same logits bit for bit and the same
    gradients up to the f32 summation order of the BatchNorm partial sums."""
    from abcnet_amd.train import Trainer
    B, S = 2, 64
    x, tg = synthetic_images(B, S, seed=7), synthetic_targets(B, S // 4, seed=1)

    def one_step(batched):
        m = make_model(dtype="bf16", dropout_p=0.2, variant=variant)
        tr = Trainer(m, B, S, S, lr=0.0, use_graph=False, fused_heads=False, batched_heads=batched)   # (the fused pass has its own test below)
        tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
        tr.step()
        torch.cuda.synchronize()
        kinds = set(op[4]["kernel"] for op in tr.eng.fwd_ops + tr.eng.bwd_ops)
        return [t.clone() for t in tr.eng.logits], m._flat_grad.clone(), tr.loss_value()["total"], kinds, dict(m._lay_p)

    lg_b, g_b, loss_b, kinds_b, lay = one_step(True)
    lg_s, g_s, loss_s, kinds_s, _ = one_step(False)
    assert "heads_fwd_batch" in kinds_b and "heads_wgrad_batch" in kinds_b and "heads_fwd_batch" not in kinds_s
    for a, b in zip(lg_b, lg_s):
        assert torch.equal(a, b)
    assert loss_b == loss_s
    # The merged pass sums the BatchNorm partials in a different partition, so the correction coefficients differ in
    # their last f32 bits; every bf16 re-quantisation downstream (dY, dA, g of ~25 layers, BN-backward cancellation in
    # each) then rounds a few elements the other way.  At the heads the two runs agree to f32 accuracy; the encoder
    # ends up inside the bf16 gradient noise floor (the bf16 bar of tests/test_gpu_kernels.py is 3e-2).
    for name, (off, n) in lay.items():
        a, b = g_b[off:off + n].double(), g_s[off:off + n].double()
        rel = (a - b).norm().item() / (b.norm().item() + 1e-30)
        if name.startswith("out_modules.") and "conv1" not in name:
            assert rel <= 1e-5, (name, rel)       # 1x1 convs and BN parameters of the heads: upstream of any bf16 re-rounding
        elif n == 1:
            # a ONE-element gradient (the 7x7 spatial-attention bias of unet2: a near-cancelling sum over all pixels) has no norm to
            # average the noise over -- and the gradient of CBAM's global max-pool lands on ONE pixel per (image, channel), which the
            # two runs' roundings may pick differently (worst seen: 1.6e-1)
            assert rel <= 1.0, (name, rel)
        else:
            assert rel <= 1e-1, (name, rel)   # (worst seen: 2.4e-2 unet, 5.6e-2 unet2 -- the first block, ~60 bf16 roundings away)


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_act_bwd_epilogue_step_equals_separate_passes(variant):
    """The default bf16 step lets a layer's activation / BatchNorm-statistics backward pass (bn_act.hip act_bwd) ride in the epilogue
    of the data gradient that produces its input wherever that layer has this one reader (abc_conv_desc.actbwd_*,
    Engine._actbwd_target).  Against the same step with every act_bwd as a launch of its own (Trainer(actbwd_epilogue=False)): same
    forward bit for bit; gradients up to the f32 order of the BatchNorm partial sums and the one bf16 rounding of dA the fused form
    skips (bounds as for the batched heads: the bf16 gradient noise floor)."""
    from abcnet_amd.train import Trainer
    B, S = 2, 128
    x, tg = synthetic_images(B, S, seed=7), synthetic_targets(B, S // 4, seed=1)

    def one_step(fused):
        m = make_model(dtype="bf16", dropout_p=0.2, variant=variant)
        tr = Trainer(m, B, S, S, lr=0.0, use_graph=False, actbwd_epilogue=fused)
        tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
        tr.step()
        torch.cuda.synchronize()
        names = [op[2] for op in tr.eng.bwd_ops]
        return [t.clone() for t in tr.eng.logits], m._flat_grad.clone(), tr.loss_value()["total"], names, dict(m._lay_p)

    lg_f, g_f, loss_f, names_f, lay = one_step(True)
    lg_s, g_s, loss_s, names_s, _ = one_step(False)
    nf = sum("+ act_bwd" in n for n in names_f)
    assert nf >= 4 and not any("+ act_bwd" in n for n in names_s), (nf, names_f)
    assert sum(n.startswith("act_bwd") for n in names_s) == sum(n.startswith("act_bwd") for n in names_f) + nf
    for a, b in zip(lg_f, lg_s):
        assert torch.equal(a, b)
    assert loss_f == loss_s
    for name, (off, n) in lay.items():
        a, b = g_f[off:off + n].double(), g_s[off:off + n].double()
        rel = (a - b).norm().item() / (b.norm().item() + 1e-30)
        if name.startswith("out_modules."):
            assert rel <= 1e-5, (name, rel)       # upstream of the first fused launch
        elif n == 1:
            assert rel <= 1.0, (name, rel)
        elif "_attention." in name:
            # unet2's CBAM: the gradient of a global max-pool sits on ONE pixel per (image, channel) and the two runs' roundings may
            # pick different ones (seen: 1.03e-1 on inc2's channel MLP, ~60 bf16 roundings from the heads); the kernels themselves are
            # held to autograd in situ (tests/test_gpu_insitu_fullsize.py)
            assert rel <= 2.5e-1, (name, rel)
        else:
            assert rel <= 1e-1, (name, rel)


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_merged_reduction_and_finaliser_launches_are_the_same_step(variant):
    """Trainer(merge_reduce=True), the default: a layer's split-K slab reduction waits for the next BatchNorm-backward finaliser of the plan
    and shares its launch (abc_wgrad_reduce_bn_bwd).  Same kernels' bodies on the same inputs, only grouped and ordered differently:
    gradients, logits and loss equal those of the plan with separate launches BIT FOR BIT."""
    from abcnet_amd.train import Trainer
    B, S = 2, 128
    x, tg = synthetic_images(B, S, seed=7), synthetic_targets(B, S // 4, seed=1)

    def one_step(merge):
        m = make_model(dtype="bf16", dropout_p=0.2, variant=variant)
        tr = Trainer(m, B, S, S, lr=0.0, use_graph=False, merge_reduce=merge)
        tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
        tr.step()
        torch.cuda.synchronize()
        return [t.clone() for t in tr.eng.logits], m._flat_grad.clone(), tr.loss_value()["total"], [op[2] for op in tr.eng.bwd_ops]

    lg_m, g_m, loss_m, names_m = one_step(True)
    lg_s, g_s, loss_s, names_s = one_step(False)
    nm = sum(" reduce + bn_bwd " in n for n in names_m)
    assert nm >= 15 and not any(" reduce + bn_bwd " in n for n in names_s) and len(names_s) == len(names_m) + nm, (nm, len(names_m), len(names_s))
    assert torch.equal(g_m, g_s) and loss_m == loss_s
    for a, b in zip(lg_m, lg_s):
        assert torch.equal(a, b)


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_fused_heads_step_equals_unfused_step(variant):
    """The Trainer's default bf16 step runs the heads' conv2 + loss + way back as ONE pass (csrc/heads_fused.hip).  Against
    the same step on the separate kernels (fused_heads=False): logits to f32 rounding order, the loss to the hardware
    exp / log of the fused kernel, conv2's gradients to the bf16 rounding of d(logits) (the separate kernels read them in f32), the rest
    inside the bf16 gradient noise floor, as in the test above."""
    from abcnet_amd.train import Trainer
    B, S = 2, 128
    x, tg = synthetic_images(B, S, seed=7), synthetic_targets(B, S // 4, seed=1)

    def one_step(fused):
        m = make_model(dtype="bf16", dropout_p=0.2, variant=variant)
        tr = Trainer(m, B, S, S, lr=0.0, use_graph=False, fused_heads=fused)
        assert (tr.eng.hf is not None) == fused
        tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
        tr.step()
        torch.cuda.synchronize()
        return [t.clone() for t in tr.eng.logits], m._flat_grad.clone(), tr.loss_value(), dict(m._lay_p)

    lg_f, g_f, loss_f, lay = one_step(True)
    lg_u, g_u, loss_u, _ = one_step(False)
    for a, b in zip(lg_f, lg_u):   # (same bf16 operands; the fused kernel's accumulators start from the bias: f32 rounding order)
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())
    for k in loss_u:               # (hardware exp / log in the fused kernel: loss_math.hpp)
        assert abs(loss_f[k] - loss_u[k]) <= 1e-5 * abs(loss_u[k]) + 1e-12, k
    for name, (off, n) in lay.items():
        a, b = g_f[off:off + n].double(), g_u[off:off + n].double()
        rel = (a - b).norm().item() / (b.norm().item() + 1e-30)
        if name == "s":
            assert rel <= 1e-6, (name, rel)
        elif name.startswith("out_modules.") and "conv2" in name:
            assert rel <= 1e-2, (name, rel)
        elif n == 1:
            assert rel <= 1.0, (name, rel)      # (one-element gradients: see the test above; worst seen 4.8e-1)
        else:
            # (worst seen: 1.2e-1 on a CBAM MLP bias of unet2's first block -- the far end of the chain, a near-cancelling sum)
            assert rel <= (1e-1 if variant == "unet" else 2e-1), (name, rel)


def test_fused_step_without_stored_logits_is_the_same_step():
    """Trainer(keep_logits=False): the fused heads pass does not store the eight output maps (abc_heads_fused_desc.logits[i] =
    NULL) -- loss and every gradient bit-equal to the step that stores them, eng.logits untouched; with metrics=True the
    maps are kept whatever the flag says (the meters read them)"""
    from abcnet_amd.train import Trainer
    B, S = 2, 128
    x, tg = synthetic_images(B, S, seed=7), synthetic_targets(B, S // 4, seed=1)

    def one_step(**kw):
        m = make_model(dtype="bf16", dropout_p=0.2)
        tr = Trainer(m, B, S, S, lr=0.0, use_graph=False, **kw)
        assert tr.eng.hf is not None
        tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
        tr.step()
        torch.cuda.synchronize()
        return [t.clone() for t in tr.eng.logits], m._flat_grad.clone(), tr.loss_value()

    lg_k, g_k, loss_k = one_step()
    lg_n, g_n, loss_n = one_step(keep_logits=False)
    lg_m, g_m, loss_m = one_step(keep_logits=False, metrics=True)
    assert torch.equal(g_k, g_n) and torch.equal(g_k, g_m)
    assert loss_k == loss_n == loss_m
    assert all(float(t.abs().max()) == 0.0 for t in lg_n)       # (never written)
    assert all(torch.equal(a, b) for a, b in zip(lg_k, lg_m))
    assert any(float(t.abs().max()) > 0.0 for t in lg_k)


@pytest.mark.parametrize("prefix", ["dconv2", "dconv1", "up1.conv", "inc2", "down2.maxpool_conv.1"])
def test_unet2_block_is_exact_in_situ(prefix):
    """Flip-free parity of unet2's CBAM + residual block (unet2.py:6-74), every abc_cbam_* entry by its own output: one
    fused step (fp32, lr = 0), then torch ops / autograd on the ENGINE's tensors of one block --
      forward : ca (abc_cbam_channel_fwd), [mean, max] over channels (abc_cbam_spatial_stats), sa (abc_cbam_conv7_fwd),
                relu(sa * ca * z + r) (abc_cbam_apply_fwd);
      backward: g = dOut * [out > 0] and du (abc_cbam_bwd1), d[mean, max] + the 7x7 weight / bias gradients
                (abc_cbam_conv7_bwd), the MLP gradients and the pool gradients (abc_cbam_bwd2 + abc_cbam_channel_bwd),
                d(y2) after abc_cbam_bwd3 + BatchNorm backward, BN2's dgamma / dbeta.
    Blocks with an identity residual (dconv2, dconv1, inc2), a 1x1 residual over the concat (up1.conv, 512 -> 256) and a
    pooled consumer + pooled input (down2)."""
    import torch.nn.functional as F
    from abcnet_amd.train import Trainer
    B, S = 2, 64
    x = synthetic_images(B, S, seed=7)
    tg = synthetic_targets(B, S // 4, seed=1)
    m = make_model(dropout_p=0.0, variant="unet2")
    tr = Trainer(m, B, S, S, lr=0.0, use_graph=False)
    tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
    tr.step()
    torch.cuda.synchronize()
    eng = tr.eng
    blk = [u for k, u in eng.units2 if k == "blk" and u.prefix == prefix][0]
    rec2 = blk.rec2
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    p = prefix + ".double_conv"
    mlp = p + ".5.channel_attention.shared_MLP"
    c7 = p + ".5.spatial_attention.conv2d"
    C_, H, W = blk.cout, blk.H, blk.W

    def nchw(t):
        return t.float().permute(0, 3, 1, 2).contiguous()

    leaf = lambda t: t.detach().clone().requires_grad_(True)
    y2 = leaf(nchw(rec2.y[..., rec2.coff:rec2.coff + C_]))
    gamma, beta = leaf(sd[rec2.bname + ".weight"]), leaf(sd[rec2.bname + ".bias"])
    w1, b1, w2, b2 = (leaf(sd[mlp + k]) for k in (".0.weight", ".0.bias", ".2.weight", ".2.bias"))
    w7, b7 = leaf(sd[c7 + ".weight"]), leaf(sd[c7 + ".bias"])
    rt, ld_r, c_r, pooled_r = blk.res
    r_full = nchw(rt[..., c_r:c_r + C_])
    if pooled_r:
        r_full = F.max_pool2d(r_full, 2)
    r = leaf(r_full)
    z = F.batch_norm(y2, None, None, gamma, beta, training=True, eps=1e-5)
    zc = (y2.detach() * rec2.scale.view(1, -1, 1, 1) + rec2.shift.view(1, -1, 1, 1))
    assert (z.detach() - zc).abs().max().item() <= 2e-5 * zc.abs().max().item()

    def mlp_f(v):
        return F.linear(F.relu(F.linear(v, w1, b1)), w2, b2)

    # (unet2.py:9-10,19-21: AdaptiveAvgPool2d(1) / AdaptiveMaxPool2d(1); unet2.py:31-33: torch.mean / torch.max over channels)
    ca = torch.sigmoid(mlp_f(F.adaptive_avg_pool2d(z, 1).flatten(1)) + mlp_f(F.adaptive_max_pool2d(z, 1).flatten(1)))
    o1 = ca[:, :, None, None] * z
    st = torch.cat([torch.mean(o1, dim=1, keepdim=True), torch.max(o1, dim=1, keepdim=True)[0]], 1)
    sa = torch.sigmoid(F.conv2d(st, w7, b7, padding=3))
    out = F.relu(sa * o1 + r)

    def close(got, ref, tol=2e-5, what=""):
        err = (got.float() - ref.float()).abs().max().item()
        assert err <= tol * ref.float().abs().max().item() + 1e-9, (prefix, what, err, ref.float().abs().max().item())

    close(blk.ca, ca.detach(), what="ca")
    close(blk.st.permute(0, 3, 1, 2), st.detach(), what="st")
    close(blk.sa.unsqueeze(1), sa.detach(), what="sa")
    got_out = nchw(blk.out[..., blk.coff_out:blk.coff_out + C_])
    close(got_out, out.detach(), what="out")
    # ---- backward: the engine's own d(out)
    dOut = torch.zeros_like(out)
    if blk.grad_same is not None:
        t, ld, co = blk.grad_same
        dOut = dOut + nchw(t[..., co:co + C_])
    if blk.grad_pool is not None:
        t, ld, co = blk.grad_pool
        oo = got_out.detach().clone().requires_grad_(True)      # route through the max-pool of the ENGINE's output
        F.max_pool2d(oo, 2).backward(nchw(t[..., co:co + C_]))
        dOut = dOut + oo.grad
    # decisions (relu mask) from the engine's own output, so that no last-bit difference can flip one
    mask = (got_out > 0).float()
    (sa * o1 + r).backward(dOut * mask)
    bw = blk.bw
    close(nchw(bw["g"]), dOut * mask, what="g")
    close(nchw(bw["g"]), r.grad, what="d_residual")
    close(nchw(bw["dz"]), y2.grad, tol=1e-4, what="d_y2")
    close(m.grad_of(c7 + ".weight"), w7.grad, tol=1e-4, what="dw7")
    close(m.grad_of(c7 + ".bias"), b7.grad, tol=1e-4, what="db7")
    for k, t in ((".0.weight", w1), (".0.bias", b1), (".2.weight", w2), (".2.bias", b2)):
        close(m.grad_of(mlp + k), t.grad, tol=1e-4, what="mlp" + k)
    close(m.grad_of(rec2.bname + ".weight"), gamma.grad, tol=1e-4, what="dgamma2")
    close(m.grad_of(rec2.bname + ".bias"), beta.grad, tol=1e-4, what="dbeta2")


@pytest.mark.parametrize("dtype,size,tol", [("fp32", 64, 1e-3), ("bf16", 128, 0.08)])
def test_folded_inference_graph_matches_oracle(dtype, size, tol):
    """the eval graph with every BatchNorm folded into the convolution in front of it (weights packed times
    gamma / sqrt(running_var + eps), bias (b - running_mean) * that + beta, activation in the conv's epilogue, consumers load
    finished tensors) gives the reference's eval forward (img2smiles2.py:49,56-59): fp32 within the 1e-3 gate, bf16 within
    the bound of the un-folded bf16 graph; and the NMS on top is the oracle's NMS of these logits, bit for bit"""
    from abcnet_amd.infer import InferenceRunner
    B = 2
    x = synthetic_images(B, size, seed=7)
    m = make_model(dtype)
    run = InferenceRunner(m, B, size, size, use_graph=True, fold_bn=True)
    assert run.fold_bn and all(op[4]["kernel"] != "bn_eval" for op in run.eng.pack_ops)
    run.load_batch(x.to(DEV))
    run.step()
    run.step()   # captured graph
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = uo.forward("unet", uo.filled_state("unet", 1, HEADS, seed=0), x, train=False)
    worst = max((y.cpu() - r).abs().max().item() for y, r in zip(run.logits, ref))
    assert worst < tol, worst
    plain = InferenceRunner(m, B, size, size, use_graph=False, fold_bn=False)
    plain.load_batch(x.to(DEV))
    plain.step()
    torch.cuda.synchronize()
    dev = max((a - b).abs().max().item() for a, b in zip(run.logits, plain.logits))
    assert dev < tol, dev
    got = [t.cpu() for t in run.logits]
    da, db, dr, do = nms_oracle.nms(got[0], got[4], got[6], got[7])
    assert torch.equal(run.atom_mask.cpu(), da) and torch.equal(run.bond_mask.cpu(), db) and torch.equal(run.omega_mask.cpu(), do)
    # new weights: refresh() re-folds
    sd = uo.filled_state("unet", 1, HEADS, seed=3)
    m.load_state_dict(sd)
    run.refresh()
    run.step()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref2 = uo.forward("unet", sd, x, train=False)
    assert max((y.cpu() - r).abs().max().item() for y, r in zip(run.logits, ref2)) < tol


@pytest.mark.parametrize("tag,cin,H,W", [("odd", 1, 72, 88), ("rgb", 3, 64, 64), ("odd_rgb", 3, 104, 40), ("odd2", 1, 72, 88), ("odd2b", 1, 104, 40)])
def test_general_shapes_match_reference_golden_and_oracle(tag, cin, H, W, golden_dir):
    """the reference's general code paths on the HIP kernels: input sizes that are not multiples of 32 (unet.py:51-56: per level
    and axis the transposed conv's first row is cropped, or nothing is; MaxPool2d floors) and in_channels = 3
    (unet.py:122-134).  fp32 logits within the 1e-3 gate of the reference-generated golden and of the oracle, eval and train;
    gradients of the reference loop body (loss.backward() through the module) against the oracle's autograd"""
    # (tags odd2*: unet2.py's general pad path, unet2.py:104-109 -- CBAM blocks, residuals and pooled gradients on odd-sized levels)
    variant = "unet2" if tag.startswith("odd2") else "unet"
    gold = np.load(os.path.join(golden_dir, "shapes_%s.npz" % variant))
    x = synthetic_images(2, max(H, W), seed=7, in_channels=cin)[:, :, :H, :W].contiguous()
    sd0 = uo.filled_state(variant, cin, HEADS, seed=0)
    m = (UNet2 if variant == "unet2" else UNet)(cin, HEADS, dtype="fp32", dropout_p=0.0)
    m.load_state_dict(sd0)
    m = m.to(DEV)
    for mode in ("eval", "train"):
        m.load_state_dict(sd0)
        m.train(mode == "train")
        with torch.no_grad():
            ys = m(x.to(DEV))
            ref = uo.forward(variant, uo.clone_state(sd0), x, train=(mode == "train"))
        for i, (y, r) in enumerate(zip(ys, ref)):
            assert tuple(y.shape) == tuple(r.shape) == tuple(gold["%s_%s_head%d_shape" % (tag, mode, i)])
            assert (y.cpu() - r).abs().max().item() < 1e-3, (tag, mode, i)
            f = y.cpu().reshape(-1)
            step = max(f.numel() // 257, 1)
            np.testing.assert_allclose(f[::step][:257].double().numpy(), gold["%s_%s_head%d_sample" % (tag, mode, i)], atol=1e-3)
    # gradients under the golden's surrogate loss, through autograd of the drop-in module
    m.load_state_dict(sd0)
    m.train()
    ys = m(x.to(DEV))
    loss = sum((y ** 2).mean() for y in ys)
    loss.backward()
    assert abs(loss.item() - gold["%s_loss" % tag].item()) <= 1e-4 * abs(gold["%s_loss" % tag].item())
    sd = uo.clone_state(sd0, requires_grad=True)
    sum((y ** 2).mean() for y in uo.forward(variant, sd, x, train=True)).backward()
    named = dict(m.named_parameters())
    extra = ("down2.maxpool_conv.1.double_conv.5.channel_attention.shared_MLP.0.weight", "up1.conv.res_conv.weight",
                "inc2.double_conv.5.spatial_attention.conv2d.weight") if variant == "unet2" else ()
    for k in ("inc1.double_conv.0.weight", "down3.maxpool_conv.1.double_conv.3.weight", "up1.up.weight", "up2.up.weight", "up3.up.weight",
              "up2.up.bias", "up2.conv.double_conv.0.weight", "dconv2.double_conv.4.weight", "out_modules.5.conv2.weight") + extra:
        g, r = named[k].grad.cpu().double(), sd[k].grad.double()
        assert (g - r).norm().item() <= 2e-2 * r.norm().item() + 1e-12, (tag, k, (g - r).norm().item(), r.norm().item())
        assert abs(g.norm().item() - gold["%s_gnorm/%s" % (tag, k)].item()) <= 2e-2 * gold["%s_gnorm/%s" % (tag, k)].item() + 1e-12


def test_multi_replica_dataparallel_matches_per_chunk_oracle():
    """train.py:50 / img2smiles2.py:43 wrap the model in nn.DataParallel; with several device ids torch replicates the module
    every forward, scatters the batch and gathers the outputs.  Here with device_ids=[0, 0] (two replicas, one GPU: the first
    runs on the master's arenas, the second on a shadow model) -- forward == the oracle applied to each chunk on its own
    (every replica normalises its own chunk: torch's DataParallel semantics), and loss.backward() leaves the SUM of the two
    replicas' gradients in the master's parameters, as autograd through Broadcast does for any module."""
    B, S = 4, 64
    x = synthetic_images(B, S, seed=7)
    sd0 = uo.filled_state("unet", 1, HEADS, seed=0)
    m = make_model(dropout_p=0.0)
    dp = torch.nn.DataParallel(m, device_ids=[0, 0])
    m.train()
    ys = dp(x.to(DEV))
    assert len(ys) == 8 and ys[5].shape == (B, 360, S // 4, S // 4)
    loss = sum((y ** 2).mean() for y in ys)
    loss.backward()
    sd = uo.clone_state(sd0, requires_grad=True)
    halves = [uo.forward("unet", sd, x[i:i + 2], train=True) for i in (0, 2)]
    ref = [torch.cat([a, b], 0) for a, b in zip(*halves)]
    for y, r in zip(ys, ref):
        assert (y.detach().cpu() - r.detach()).abs().max().item() < 1e-3
    sum((y ** 2).mean() for y in ref).backward()
    named = dict(m.named_parameters())
    for k in ("inc1.double_conv.0.weight", "down3.maxpool_conv.1.double_conv.3.weight", "up2.up.weight", "dconv2.double_conv.4.weight",
              "out_modules.5.conv2.weight", "out_modules.0.bn.bias"):
        g, r = named[k].grad.cpu().double(), sd[k].grad.double()
        assert (g - r).norm().item() <= 2e-2 * r.norm().item() + 1e-12, (k, (g - r).norm().item(), r.norm().item())
    assert not m._dp_busy and all(not s._dp_busy for pool in m._dp_shadows.values() for s in pool) and len(m._dp_shadows[torch.device("cuda", 0)]) == 1
    # eval through the wrapper: running statistics (updated by the replica on the master's arenas only), no gradients
    m.eval()
    with torch.no_grad():
        ye = dp(x.to(DEV))
    sde = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    re = uo.forward("unet", sde, x, train=False)
    assert max((a.cpu() - b).abs().max().item() for a, b in zip(ye, re)) < 1e-3
    assert int(sde["inc1.double_conv.1.num_batches_tracked"]) == 1


def test_dataparallel_forward_without_backward_returns_its_engine():
    """a train-mode forward under nn.DataParallel whose graph is dropped without a backward (an exception in the loss, a loss
    that is only looked at) gives its engine owner back when the graph dies: six such forwards in a row work (five leaked
    checkouts per device used to raise), the master keeps updating ITS running statistics, nothing stays checked out"""
    import gc
    B, S = 4, 64
    x = synthetic_images(B, S, seed=7).to(DEV)
    m = make_model(dropout_p=0.0)
    dp = torch.nn.DataParallel(m, device_ids=[0, 0])
    m.train()
    for i in range(6):
        ys = dp(x)
        _ = float(ys[0].mean())     # inspected, never back-propagated
        del ys
        gc.collect()
        assert not m._dp_busy, i
    assert int(m.state_dict()["inc1.double_conv.1.num_batches_tracked"]) == 6
    assert all(not s._dp_busy for pool in m._dp_shadows.values() for s in pool) and len(m._dp_shadows[torch.device("cuda", 0)]) == 1
