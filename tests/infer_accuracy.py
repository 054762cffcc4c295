"""Test infrastructure shared by tests/test_gpu_calibrated.py and tests/test_gpu_trained.py: config 5's graph (img2smiles2.py:42-79,
eval forward + peak NMS [+ candidate extraction, img2smiles2.py:113-191]) on the device against the oracle on the host, in
terms that mean something whatever the scale of the maps:

  * every head's logits RELATIVE TO THAT HEAD'S RANGE in the oracle's map: L-inf / (max - min), rms / std;
  * NMS decisions as (missed + spurious) peaks out of the oracle's peaks, per mask;
  * candidate lists as (missed + spurious + wrong-class) entries out of the oracle's list.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import abcnet_amd  # noqa: E402,F401
from oracle import decode_oracle, nms_oracle  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

DEV = "cuda"
HEAD_NAMES = ("atom", "atom_types", "charges", "hs", "bond", "bond_types", "rho", "omega")


def oracle_maps(sd, x):
    """fp32 oracle: eval maps + NMS decisions of the images x (CPU)"""
    with torch.no_grad():
        ref = uo.forward("unet", sd, x, train=False)
        nms = nms_oracle.nms(ref[0], ref[4], ref[6], ref[7])
    return ref, nms


def _list_diff(ref_rows, got_rows, nkey):
    """rows = lists of int tuples; the first nkey entries identify a candidate (position [+ bin]), the rest are its classes.
    returns (missed, spurious, wrong_class)"""
    r = {tuple(t[:nkey]): tuple(t[nkey:]) for t in ref_rows}
    g = {tuple(t[:nkey]): tuple(t[nkey:]) for t in got_rows}
    missed = sum(1 for k in r if k not in g)
    spurious = sum(1 for k in g if k not in r)
    wrong = sum(1 for k in r if k in g and g[k] != r[k])
    return missed, spurious, wrong


def measure(model_dev, x, sample, oracle, fp8=False, fold_bn=True, extract=False, gold_samples=None):
    """model_dev: the bf16 device model (possibly perturbed); x [B,1,S,S] CPU images; sample: indices compared;
    oracle = oracle_maps(true state, x[sample]).  gold_samples: optional (list of 8 sample vectors, list of 8 (min, max)) of the
    reference's own maps for the first images of `sample` (tests/golden/calibrated_*.npz)."""
    from abcnet_amd.infer import InferenceRunner
    B, _, S, _ = x.shape
    run = InferenceRunner(model_dev, B, S, S, use_graph=True, fold_bn=fold_bn, fp8=fp8, extract=extract)
    run.load_batch(x.to(DEV))
    run.step()
    run.step()
    torch.cuda.synchronize()
    idx = torch.tensor(list(sample), device=DEV)
    got = [t[idx].cpu() for t in run.logits]
    ref, (ra, rb, rr, ro) = oracle
    res = {"fp8": bool(fp8), "fold_bn": bool(fold_bn), "heads": {}}
    for i, (g, r) in enumerate(zip(got, ref)):
        rng = (r.max() - r.min()).item()
        res["heads"][HEAD_NAMES[i]] = {
            "range": rng,
            "linf_over_range": (g - r).abs().max().item() / rng,
            "rms_over_std": ((g - r).double().pow(2).mean().sqrt() / r.double().std()).item(),
        }
    res["worst_linf_over_range"] = max(h["linf_over_range"] for h in res["heads"].values())
    res["worst_rms_over_std"] = max(h["rms_over_std"] for h in res["heads"].values())
    if gold_samples is not None:
        import numpy as np
        worst = 0.0
        for i, (gs, (lo, hi)) in enumerate(zip(*gold_samples)):
            f = got[i][:2].reshape(-1)
            step = max(f.numel() // len(gs), 1)
            d = np.abs(f[::step][:len(gs)].double().numpy() - gs).max() / (hi - lo)
            worst = max(worst, float(d))
        res["golden_sample_linf_over_range"] = worst
    masks = {"atom": (run.atom_mask[idx].cpu(), ra), "bond": (run.bond_mask[idx].cpu(), rb), "omega": (run.omega_mask[idx].cpu(), ro)}
    for k, (g, r) in masks.items():
        g, r = g.bool(), r.bool()
        missed, spurious, n = int((r & ~g).sum()), int((~r & g).sum()), int(r.sum())
        res[k + "_peaks"] = {"oracle": n, "missed": missed, "spurious": spurious, "rate": (missed + spurious) / max(n, 1)}
    rho_rng = (rr.max() - rr.min()).item()
    res["rho_abs_linf_over_range"] = (run.rho_abs[idx].cpu() - rr).abs().max().item() / rho_rng
    # the device NMS on the device's own logits is exact (the decisions differ from the oracle's only through the logits)
    da, db, dr, do = nms_oracle.nms(got[0], got[4], got[6], got[7])
    res["nms_on_device_logits_exact"] = bool(torch.equal(masks["atom"][0], da) and torch.equal(masks["bond"][0], db)
                                             and torch.equal(masks["omega"][0], do) and torch.equal(run.rho_abs[idx].cpu(), dr))
    if extract:
        lists = run.candidates()
        tot = {"atoms_oracle": 0, "atoms_missed": 0, "atoms_spurious": 0, "atoms_wrong_class": 0,
               "bonds_oracle": 0, "bonds_missed": 0, "bonds_spurious": 0, "bonds_wrong_class": 0, "truncated": 0, "rho_worst": 0.0}
        for j, b in enumerate(sample):
            oa, ob, orho = decode_oracle.extract(ra[j, 0], rb[j, 0], ref[1][j], ref[2][j], ref[3][j], ref[5][j], rr[j], ref[7][j])
            L = lists[b]
            tot["truncated"] += int(L["truncated"])
            m, s, w = _list_diff(oa.tolist(), L["atoms"].tolist(), 2)
            tot["atoms_oracle"] += len(oa)
            tot["atoms_missed"] += m
            tot["atoms_spurious"] += s
            tot["atoms_wrong_class"] += w
            m, s, w = _list_diff(ob.tolist(), L["bonds"].tolist(), 3)
            tot["bonds_oracle"] += len(ob)
            tot["bonds_missed"] += m
            tot["bonds_spurious"] += s
            tot["bonds_wrong_class"] += w
            # rho of the candidates both lists hold
            gk = {tuple(t[:3]): float(v) for t, v in zip(L["bonds"].tolist(), L["rho"].tolist())}
            for t, v in zip(ob.tolist(), orho.tolist()):
                if tuple(t[:3]) in gk:
                    tot["rho_worst"] = max(tot["rho_worst"], abs(gk[tuple(t[:3])] - v))
        tot["atoms_rate"] = (tot["atoms_missed"] + tot["atoms_spurious"] + tot["atoms_wrong_class"]) / max(tot["atoms_oracle"], 1)
        tot["bonds_rate"] = (tot["bonds_missed"] + tot["bonds_spurious"] + tot["bonds_wrong_class"]) / max(tot["bonds_oracle"], 1)
        res["candidates"] = tot
    del run
    torch.cuda.empty_cache()
    return res
