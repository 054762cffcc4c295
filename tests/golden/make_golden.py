#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the development container (needs /root/reference); the fixtures it
writes are data (inputs are regenerated from seeds, outputs are stored) and are
what travels to the GPU box.  No reference source text is stored.

What is produced (all float32 unless noted):
  model_<variant>_64.npz   full head maps for a 2x1x64x64 seeded image, eval and
                           train mode (dropout p=0), + updated BN running stats
                           samples; weights = oracle.unet_oracle.filled_state(seed=0)
  model_<variant>_384.npz  per-head statistics + strided samples at 2x1x384x384
  grads_<variant>_512.npz  gradient L2 norms + leading samples of every parameter
                           under the REAL loss (exec of train.py:95-137) at
                           1x1x512x512 (the slice hard-codes 128x128 maps)
  loss_128.npz             the 8 weighted loss terms, total and dL/dlogit samples
                           for seeded logits/targets at [2,.,128,128]
  nms_128.npz              NMS masks (bit-packed) from exec of img2smiles2.py:61-79
  metrics_128.npz          sum / count of the 17 training meters after one update (exec of
                           train.py:95-105 + 145-215 with the reference's meter.AverageMeter)
  raster_128.npz           the 8 target maps of utils.py:83-228 for seeded annotation strings (sparse)
  decode_128.npz           atom / bond candidate lists of img2smiles2.py:113-191 for seeded head maps
  adam.npz                 one torch.optim.Adam step (train.py:55 settings)
  shapes_unet.npz          unet.py at an input size that is not a multiple of 32 (72x88, 104x40) and with 3 input channels:
                           head-map samples / statistics (eval, train) and gradient norms under a surrogate loss
  calibrated_<variant>.npz BatchNorm running statistics after CALIB_STEPS train-mode forwards of the reference module (statistics that
                           match the activations: the eval maps then have their train-mode range), eval head maps with them at 64x64
                           (full) and for two images of the 512x512 benchmark batch (samples, statistics, NMS decisions)
  trained_unet.npz         the FROZEN trained fixture (trained_unet_state.npz: unet.py trained on the device ONCE, parameters rounded to
                           bf16 -- make_trained_fixture.py) loaded into the reference's UNet: eval head maps at 64x64 (full, two drawn
                           molecules) and, for the 16 sampled images of config 5's accuracy batch (drawn_molecules(64, 512, seed=777)
                           [0::4]), strided samples, statistics and the NMS decisions of img2smiles2.py:61-79 (bit-packed)
  meta.json                state_dict key/shape lists, parameter counts
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import abcnet_amd  # noqa: E402
from abcnet_amd.synthetic import synthetic_images, synthetic_targets  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

HEADS = [1, 14, 3, 2, 1, 360, 60, 60]
TERMS = ["atom_targets_loss", "bond_targets_loss", "atom_types_loss", "atom_charges_loss", "bond_types_loss",
         "bond_rhos_loss", "bond_omega_types_loss", "atom_hs_loss"]
PRED_NAMES = ["atom_targets_pred", "atom_types_pred", "atom_charges_pred", "atom_hs_pred", "bond_targets_pred",
              "bond_types_pred", "bond_rhos_pred", "bond_omega_types_pred"]
TGT_NAMES = ["atom_targets", "atom_types", "atom_charges", "atom_hs", "bond_targets", "bond_types", "bond_rhos",
             "bond_omega_types"]


def ref_module(variant):
    import importlib
    return importlib.import_module("unet" if variant == "unet" else "unet2")


def slice_text(path, lo, hi):
    with open(path) as f:
        lines = f.readlines()[lo - 1:hi]
    # de-indent to column 0 (the slices sit inside loops)
    ind = min(len(l) - len(l.lstrip()) for l in lines if l.strip())
    return "".join(l[ind:] if l.strip() else l for l in lines)


class _Box:
    pass


def run_loss_slice(preds, targets, s):
    ns = {"torch": torch}
    for n, v in zip(PRED_NAMES, preds):
        ns[n] = v
    for n, v in zip(TGT_NAMES, targets):
        ns[n] = v
    ns["atom_type_weights"] = torch.tensor([1, 0.1, 0.1, 0.1, 1, 1, 1, 1, 1, 10, 10, 10, 10, 10]).reshape([1, 14, 1, 1])
    model = _Box()
    model.module = _Box()
    model.module.s = s
    ns["model"] = model
    exec(slice_text(os.path.join(REF, "train.py"), 95, 137), ns)
    return ns["loss"], [ns[t] for t in TERMS]


def sample(t, n=257):
    f = t.detach().reshape(-1)
    step = max(f.numel() // n, 1)
    return f[::step][:n].double().numpy()


def model_goldens(variant):
    mod = ref_module(variant)
    sd = uo.filled_state(variant, 1, HEADS, seed=0)
    out = {}
    for size, tag in ((64, "64"), (384, "384")):
        x = synthetic_images(2, size, seed=7)
        res = {}
        for mode in ("eval", "train"):
            m = mod.UNet(1, HEADS)
            m.load_state_dict(sd, strict=True)
            for om in m.out_modules:
                if hasattr(om, "drop"):
                    om.drop.p = 0.0
            m.train(mode == "train")
            with torch.no_grad():
                ys = m(x)
            for i, y in enumerate(ys):
                if size == 64:
                    res["%s_head%d" % (mode, i)] = y.numpy()
                else:
                    res["%s_head%d_sample" % (mode, i)] = sample(y)
                    res["%s_head%d_stats" % (mode, i)] = np.array(
                        [y.min().item(), y.max().item(), y.double().mean().item(), y.double().norm().item()])
            if mode == "train":
                msd = m.state_dict()
                for k in ("inc1.double_conv.1.running_mean", "inc1.double_conv.1.running_var",
                          "dconv2.double_conv.4.running_mean", "dconv2.double_conv.4.running_var",
                          "out_modules.5.bn.running_mean", "out_modules.5.bn.running_var"):
                    res["rs_" + k] = msd[k].numpy()
                res["nbt"] = np.array(msd["inc1.double_conv.1.num_batches_tracked"].item())
        np.savez(os.path.join(HERE, "model_%s_%s.npz" % (variant, tag)), **res)
        print("wrote model", variant, tag)
    return sd


def grad_goldens(variant):
    mod = ref_module(variant)
    sd = uo.filled_state(variant, 1, HEADS, seed=0)
    m = mod.UNet(1, HEADS)
    m.load_state_dict(sd, strict=True)
    for om in m.out_modules:
        if hasattr(om, "drop"):
            om.drop.p = 0.0
    m.train()
    x = synthetic_images(1, 512, seed=7)
    tg = synthetic_targets(1, 128, seed=1)
    preds = m(x)
    loss, terms = run_loss_slice(list(preds), tg, m.s)
    loss.backward()
    res = {"loss": np.array(loss.item()), "terms": np.array([t.item() for t in terms])}
    for k, p in m.named_parameters():
        g = p.grad
        res["norm/" + k] = np.array(g.double().norm().item())
        res["head/" + k] = g.reshape(-1)[:64].double().numpy()
    np.savez(os.path.join(HERE, "grads_%s_512.npz" % variant), **res)
    print("wrote grads", variant, loss.item())


def shape_goldens():
    """unet.py on the shapes its general code paths exist for: an input whose size is NOT a multiple of 32 (the pad / crop of
    unet.py:51-56 then crops on some levels and not on others) and in_channels = 3 (unet.py:122-134's own self-check).
    Stored: strided samples + statistics of the eight head maps in eval and train mode, and (train mode) the gradient
    norms / leading samples of a few parameters under the surrogate loss sum_i mean(head_i ** 2)."""
    res = {}
    # (unet2.py's general pad path, unet2.py:104-109, at a size that is not a multiple of 32: shapes_unet2.npz)
    for tag, cin, H, W in (("odd", 1, 72, 88), ("rgb", 3, 64, 64), ("odd_rgb", 3, 104, 40), ("odd2", 1, 72, 88), ("odd2b", 1, 104, 40)):
        variant = "unet2" if tag.startswith("odd2") else "unet"
        mod = ref_module(variant)
        sd = uo.filled_state(variant, cin, HEADS, seed=0)
        x = synthetic_images(2, max(H, W), seed=7, in_channels=cin)[:, :, :H, :W].contiguous()
        for mode in ("eval", "train"):
            m = mod.UNet(cin, HEADS)
            m.load_state_dict(sd, strict=True)
            for om in m.out_modules:
                if hasattr(om, "drop"):
                    om.drop.p = 0.0
            m.train(mode == "train")
            ys = m(x)
            for i, y in enumerate(ys):
                res["%s_%s_head%d_shape" % (tag, mode, i)] = np.array(y.shape)
                res["%s_%s_head%d_sample" % (tag, mode, i)] = sample(y)
                res["%s_%s_head%d_stats" % (tag, mode, i)] = np.array(
                    [y.min().item(), y.max().item(), y.double().mean().item(), y.double().norm().item()])
            if mode == "train":
                loss = sum((y ** 2).mean() for y in ys)
                loss.backward()
                res["%s_loss" % tag] = np.array(loss.item())
                named = dict(m.named_parameters())
                for k in ("inc1.double_conv.0.weight", "down3.maxpool_conv.1.double_conv.3.weight", "up1.up.weight", "up2.up.weight",
                          "up3.up.weight", "up2.up.bias", "up2.conv.double_conv.0.weight", "dconv2.double_conv.4.weight",
                          "out_modules.5.conv2.weight") + (("down2.maxpool_conv.1.double_conv.5.channel_attention.shared_MLP.0.weight",
                                                            "up1.conv.res_conv.weight", "inc2.double_conv.5.spatial_attention.conv2d.weight")
                                                           if variant == "unet2" else ()):
                    g = named[k].grad
                    res["%s_gnorm/%s" % (tag, k)] = np.array(g.double().norm().item())
                    res["%s_ghead/%s" % (tag, k)] = g.reshape(-1)[:64].double().numpy()
    np.savez(os.path.join(HERE, "shapes_unet.npz"), **{k: v for k, v in res.items() if not k.startswith("odd2")})
    np.savez(os.path.join(HERE, "shapes_unet2.npz"), **{k: v for k, v in res.items() if k.startswith("odd2")})
    print("wrote shapes")


def loss_goldens():
    g = torch.Generator().manual_seed(11)
    preds = [(torch.randn((2, c, 128, 128), generator=g) * 2.0).requires_grad_(True) for c in HEADS]
    tg = synthetic_targets(2, 128, seed=1)
    s = (torch.rand(10, generator=g) * 0.4 - 0.2).requires_grad_(True)
    loss, terms = run_loss_slice(preds, tg, s)
    loss.backward()
    res = {"loss": np.array(loss.item()), "loss_dtype": np.array(str(loss.dtype)),
           "terms": np.array([t.item() for t in terms]), "ds": s.grad.double().numpy()}
    for i, p in enumerate(preds):
        res["dlogit%d_sample" % i] = sample(p.grad, 1031)
        res["dlogit%d_norm" % i] = np.array(p.grad.double().norm().item())
    np.savez(os.path.join(HERE, "loss_128.npz"), **res)
    print("wrote loss", loss.item(), loss.dtype)


def metrics_goldens():
    """train.py:95-105 (activations) + train.py:145-215 (the 17 meters) executed on the seeded inputs with the
    reference's own meter.AverageMeter; stored: sum and count of every meter after ONE update"""
    import re
    from meter import AverageMeter
    from abcnet_amd.synthetic import correlated_logits
    tg = synthetic_targets(2, 128, seed=3)
    preds = correlated_logits(tg, seed=19)
    ns = {"torch": torch}
    for n, v in zip(PRED_NAMES, preds):
        ns[n] = v
    for n, v in zip(TGT_NAMES, tg):
        ns[n] = v
    text = slice_text(os.path.join(REF, "train.py"), 145, 215)
    names = []
    for m in re.finditer(r"(train_\w+)\.update", text):
        if m.group(1) not in names:
            names.append(m.group(1))
    for n in names:
        ns[n] = AverageMeter()
    exec(slice_text(os.path.join(REF, "train.py"), 95, 105), ns)
    exec(text, ns)
    res = {"names": np.array(names), "sum": np.array([float(ns[n].sum) for n in names]),
           "count": np.array([float(ns[n].count) for n in names])}
    np.savez(os.path.join(HERE, "metrics_128.npz"), **res)
    print("wrote metrics")
    for n in names:
        print("  %-36s sum %12.4f count %12.4f" % (n, ns[n].sum, ns[n].count))


def decode_goldens():
    """img2smiles2.py:61-79 (NMS) + 113-191 (candidate extraction) executed on seeded head maps; stored per image: the
    accepted atoms (x, y, type, charge, hs) and the bond candidates (x, y, delta_x, delta_y, type) exactly as the
    reference lists hold them.  The slice sits inside `for j in range(B)` and uses `continue`, so it is executed
    wrapped in that loop; the vocab look-ups (atom_type_devocab / atom_charge_devocab) are identity maps here."""
    from abcnet_amd.synthetic import correlated_logits
    tg = synthetic_targets(2, 128, seed=3)
    lg = correlated_logits(tg, seed=29, centre_noise=0.5)
    ns = {"torch": torch, "np": np}
    for n, v in zip(PRED_NAMES, lg):
        ns[n] = v
    ns["imgs"] = torch.zeros(2, 1, 512, 512)
    exec(slice_text(os.path.join(REF, "img2smiles2.py"), 61, 79), ns)

    class _Ident(dict):
        def __missing__(self, k):
            return k
    ns["atom_type_devocab"], ns["atom_charge_devocab"] = _Ident(), _Ident()
    ns["results"] = []
    ns["collected"] = []
    body = slice_text(os.path.join(REF, "img2smiles2.py"), 113, 191)
    src = "for j in range(2):\n" + "".join("    " + l if l.strip() else l for l in body.splitlines(True))
    src += "\n    collected.append((atoms_position_list, atoms_type_list, atoms_charge_list, atoms_hs_list, bonds_position_list, bonds_property_list, bonds_delta_list))\n"
    exec(src, ns)
    res = {}
    for j, (ap, aty, ach, ahs, bp, bpr, bd) in enumerate(ns["collected"]):
        res["atoms%d" % j] = np.concatenate([np.array(ap, dtype=np.int64).reshape(-1, 2), np.array(aty, dtype=np.int64).reshape(-1, 1),
                                             np.array(ach, dtype=np.int64).reshape(-1, 1), np.array(ahs, dtype=np.int64).reshape(-1, 1)], axis=1)
        res["bond_pos%d" % j] = np.array(bp, dtype=np.int64).reshape(-1, 2)
        res["bond_type%d" % j] = np.array(bpr, dtype=np.int64)
        res["bond_delta%d" % j] = np.array(bd, dtype=np.float64).reshape(-1, 2)
    np.savez_compressed(os.path.join(HERE, "decode_128.npz"), **res)
    print("wrote decode", [(res["atoms%d" % j].shape, res["bond_pos%d" % j].shape) for j in range(2)])


def raster_goldens():
    """utils.py:83-228 (target rasteriser inside MolecularImageDataset.__getitem__) executed on seeded annotation strings
    with the vocabularies of utils.py:12-15; stored: the 8 target maps as sparse (index, value) lists.
    The slice uses `np.math.atan`, which numpy >= 2 no longer has: the namespace's `np` is numpy plus a `math`
    attribute (the standard library module, which is what np.math was)."""
    import math
    import types
    from oracle.raster_oracle import random_annotations
    npx = types.SimpleNamespace(**{k: getattr(np, k) for k in dir(np) if not k.startswith("__")})
    npx.math = math
    res = {}
    cases = [(40, 45, 101, 1, 1, 0, 0), (25, 30, 102, 0.8317, 1, 43, 0), (60, 70, 103, 1, 0.9071, 0, 23)]
    for ci, (na, nb, seed, sx, sy, ddx, ddy) in enumerate(cases):
        atoms_string, bonds_string = random_annotations(na, nb, seed, size=int(512 * min(sx, sy)) - 1)
        ns = {"np": npx, "atoms_string": atoms_string, "bonds_string": bonds_string, "scale_x": sx, "scale_y": sy, "ddx": ddx, "ddy": ddy}
        exec(slice_text(os.path.join(REF, "utils.py"), 12, 15), ns)
        exec(slice_text(os.path.join(REF, "utils.py"), 83, 228), ns)
        maps = [ns[n] for n in ("atom_target", "atom_type", "atom_charge", "atom_hs", "bond_target", "bond_type", "bond_rho", "bond_omega_type")]
        for mi, m in enumerate(maps):
            flat = m.reshape(-1)
            nz = np.flatnonzero(flat)
            res["c%d_m%d_idx" % (ci, mi)] = nz.astype(np.int64)
            res["c%d_m%d_val" % (ci, mi)] = flat[nz]
            res["c%d_m%d_dtype" % (ci, mi)] = np.array(str(m.dtype))
        res["c%d_args" % ci] = np.array([na, nb, seed, sx, sy, ddx, ddy], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "raster_128.npz"), **res)
    print("wrote raster", [len(res["c%d_m5_idx" % c]) for c in range(3)])


def nms_goldens():
    g = torch.Generator().manual_seed(13)
    ns = {"torch": torch}
    ns["atom_targets_pred"] = torch.randn((2, 1, 128, 128), generator=g) * 2
    ns["bond_targets_pred"] = torch.randn((2, 1, 128, 128), generator=g) * 2
    ns["bond_rhos_pred"] = torch.randn((2, 60, 128, 128), generator=g) * 3
    ns["bond_types_pred"] = torch.randn((2, 360, 128, 128), generator=g)
    # quantise the omega logits so that exact ties between neighbouring bins occur
    ns["bond_omega_types_pred"] = torch.round(torch.randn((2, 60, 128, 128), generator=g) * 4) / 4
    exec(slice_text(os.path.join("/root/reference/src", "img2smiles2.py"), 61, 79), ns)
    res = {
        "atom_mask": np.packbits(ns["atom_targets_pred"].numpy().astype(np.uint8)),
        "bond_mask": np.packbits(ns["bond_targets_pred"].numpy().astype(np.uint8)),
        "omega_mask": np.packbits(ns["bond_omega_types_pred2"].numpy().astype(np.uint8)),
        "rho_sample": sample(ns["bond_rhos_pred"], 1031),
        "counts": np.array([ns["atom_targets_pred"].sum().item(), ns["bond_targets_pred"].sum().item(),
                            ns["bond_omega_types_pred2"].sum().item()]),
    }
    np.savez(os.path.join(HERE, "nms_128.npz"), **res)
    print("wrote nms", res["counts"])


def adam_goldens():
    g = torch.Generator().manual_seed(17)
    p = torch.randn(4099, generator=g).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=2.5e-4, weight_decay=1e-8)
    res = {"p0": p.detach().clone().numpy()}
    for it in range(3):
        p.grad = torch.randn(4099, generator=g) * (10.0 ** (it - 1))
        res["g%d" % it] = p.grad.clone().numpy()
        opt.step()
        res["p%d" % (it + 1)] = p.detach().clone().numpy()
    np.savez(os.path.join(HERE, "adam.npz"), **res)
    print("wrote adam")


def calibrated_goldens(variant):
    """BatchNorm running statistics that MATCH the activations, made by the reference module itself: `CALIB_STEPS` train-mode
    forwards of mod.UNet over seeded images (nn.BatchNorm2d's own momentum update), then eval head maps with those
    statistics: full at 2x1x64x64, and for images 0 and 21 of the 64x1x512x512 benchmark batch (config 5) strided samples,
    statistics and the NMS decisions of img2smiles2.py:61-79 (bit-packed)."""
    mod = ref_module(variant)
    sd = uo.filled_state(variant, 1, HEADS, seed=0)
    m = mod.UNet(1, HEADS)
    m.load_state_dict(sd, strict=True)
    for om in m.out_modules:
        if hasattr(om, "drop"):
            om.drop.p = 0.0
    m.train()
    with torch.no_grad():
        for i in range(uo.CALIB_STEPS):
            m(synthetic_images(uo.CALIB_BATCH, uo.CALIB_SIZE, seed=uo.CALIB_SEED0 + i))
    msd = m.state_dict()
    res = {"bn_stats": torch.cat([msd[k].reshape(-1) for k in uo.bn_stat_keys(msd)]).numpy(),
           "calib": np.array([uo.CALIB_STEPS, uo.CALIB_SIZE, uo.CALIB_BATCH, uo.CALIB_SEED0]),
           "nbt": np.array(msd["inc1.double_conv.1.num_batches_tracked"].item())}
    m.eval()
    with torch.no_grad():
        ys = m(synthetic_images(2, 64, seed=7))
        for i, y in enumerate(ys):
            res["eval64_head%d" % i] = y.numpy()
        x = synthetic_images(64, 512, seed=7)[[0, 21]]
        ys = m(x)
    for i, y in enumerate(ys):
        res["eval512_head%d_sample" % i] = sample(y, 4099)
        res["eval512_head%d_stats" % i] = np.array([y.min().item(), y.max().item(), y.double().mean().item(), y.double().norm().item()])
    ns = {"torch": torch, "atom_targets_pred": ys[0], "bond_targets_pred": ys[4], "bond_rhos_pred": ys[6], "bond_types_pred": ys[5],
          "bond_omega_types_pred": ys[7]}
    exec(slice_text(os.path.join(REF, "img2smiles2.py"), 61, 79), ns)
    res["nms512_atom"] = np.packbits(ns["atom_targets_pred"].numpy().astype(np.uint8))
    res["nms512_bond"] = np.packbits(ns["bond_targets_pred"].numpy().astype(np.uint8))
    res["nms512_omega"] = np.packbits(ns["bond_omega_types_pred2"].numpy().astype(np.uint8))
    res["nms512_counts"] = np.array([ns["atom_targets_pred"].sum().item(), ns["bond_targets_pred"].sum().item(),
                                     ns["bond_omega_types_pred2"].sum().item()])
    np.savez_compressed(os.path.join(HERE, "calibrated_%s.npz" % variant), **res)
    print("wrote calibrated", variant, res["nms512_counts"],
          [(float(res["eval512_head%d_stats" % i][0]), float(res["eval512_head%d_stats" % i][1])) for i in range(8)])


TRAINED_SAMPLE = tuple(range(0, 64, 4))      # (tests/test_gpu_trained.py: SAMPLE)


def trained_goldens():
    """the frozen trained fixture through THE REFERENCE: tests/golden/trained_unet_state.npz (its parameters are bf16-representable
    f32 values) loaded into unet.UNet with strict=True, eval mode; what the oracle has to reproduce bit for bit and what the device
    graphs of config 5 are measured against (tests/test_gpu_trained.py)"""
    from abcnet_amd.synthetic import drawn_molecules
    sys.path.insert(0, HERE)
    from make_trained_fixture import unpack_state
    mod = ref_module("unet")
    sd = unpack_state(os.path.join(HERE, "trained_unet_state.npz"))
    m = mod.UNet(1, HEADS)
    m.load_state_dict(sd, strict=True)
    m.eval()
    res = {"sample": np.array(TRAINED_SAMPLE)}
    with torch.no_grad():
        x64, _ = drawn_molecules(2, 64, seed=778, n_atoms=(2, 4), margin=8, min_dist=12, max_bond=40)
        for i, y in enumerate(m(x64)):
            res["eval64_head%d" % i] = y.numpy()
        x, _ = drawn_molecules(64, 512, seed=777)
        x = x[list(TRAINED_SAMPLE)]
        ys = [[] for _ in range(8)]
        for b in range(0, len(TRAINED_SAMPLE), 4):      # (four images at a time: 8 GB of activations otherwise)
            for i, y in enumerate(m(x[b:b + 4])):
                ys[i].append(y)
        ys = [torch.cat(t) for t in ys]
    for i, y in enumerate(ys):
        res["eval512_head%d_sample" % i] = np.stack([sample(y[b], 4099) for b in range(y.shape[0])])
        res["eval512_head%d_stats" % i] = np.array([[y[b].min().item(), y[b].max().item(), y[b].double().mean().item(), y[b].double().norm().item()]
                                                    for b in range(y.shape[0])])
    ns = {"torch": torch, "atom_targets_pred": ys[0], "bond_targets_pred": ys[4], "bond_rhos_pred": ys[6], "bond_types_pred": ys[5],
          "bond_omega_types_pred": ys[7]}
    exec(slice_text(os.path.join(REF, "img2smiles2.py"), 61, 79), ns)
    res["nms512_atom"] = np.packbits(ns["atom_targets_pred"].numpy().astype(np.uint8))
    res["nms512_bond"] = np.packbits(ns["bond_targets_pred"].numpy().astype(np.uint8))
    res["nms512_omega"] = np.packbits(ns["bond_omega_types_pred2"].numpy().astype(np.uint8))
    res["nms512_counts"] = np.array([ns["atom_targets_pred"].sum().item(), ns["bond_targets_pred"].sum().item(),
                                     ns["bond_omega_types_pred2"].sum().item()])
    np.savez_compressed(os.path.join(HERE, "trained_unet.npz"), **res)
    print("wrote trained", res["nms512_counts"], [(float(res["eval512_head%d_stats" % i][:, 0].min()), float(res["eval512_head%d_stats" % i][:, 1].max())) for i in range(8)])


def meta():
    out = {}
    for variant in ("unet", "unet2"):
        m = ref_module(variant).UNet(1, HEADS)
        sd = m.state_dict()
        out[variant] = {
            "keys": list(sd.keys()),
            "shapes": [list(v.shape) for v in sd.values()],
            "dtypes": [str(v.dtype) for v in sd.values()],
            "n_params": sum(p.numel() for p in m.parameters()),
            "n_param_tensors": len(list(m.parameters())),
        }
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(out, f)
    print("wrote meta", {k: v["n_params"] for k, v in out.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trained":
        trained_goldens()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "calibrated":
        for v in ("unet", "unet2"):
            calibrated_goldens(v)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("metrics", "decode", "raster", "shapes"):   # (added after the other fixtures: regenerate one alone)
        {"metrics": metrics_goldens, "decode": decode_goldens, "raster": raster_goldens, "shapes": shape_goldens}[sys.argv[1]]()
        sys.exit(0)
    meta()
    shape_goldens()
    metrics_goldens()
    decode_goldens()
    raster_goldens()
    adam_goldens()
    nms_goldens()
    loss_goldens()
    for v in ("unet", "unet2"):
        model_goldens(v)
        grad_goldens(v)
        calibrated_goldens(v)
    trained_goldens()
