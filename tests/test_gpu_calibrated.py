"""Config 5 (img2smiles2.py:42-79: eval forward + peak NMS at 512 x 512, batch 64) with BatchNorm running statistics that
MATCH the activations -- an accuracy test that can fail.

With oracle.filled_state()'s random running statistics the eval forward collapses (the atom heat-map spans 0.05, a third
of all pixels are "peaks", and any logit deviation looks small in absolute terms).  tests/golden/calibrated_unet.npz holds
the statistics the REFERENCE module ends up with after 60 train-mode forwards (its own nn.BatchNorm2d update) and the
reference's eval maps with them: range +-1..2 per head.  Here the bf16 BatchNorm-folded graph and its fp8 (e4m3) form are
held, at the benchmarked size, to

  * logits RELATIVE TO EACH HEAD'S RANGE (L-inf / (max - min) and rms / std of the reference map), and
  * NMS decisions as (missed + spurious) peaks out of the oracle's peaks, per mask,

against hard ceilings (tests/golden/calibrated_deviation.json: the measured values and the ceilings derived from them),
directly against the reference-generated samples for images 0 and 21 of the batch and against the oracle (bit-equal to
the reference on this fixture, tests/test_oracle_golden.py) for four images.  `test_a_five_percent_error_in_one_conv_is_caught`
proves the bounds discriminate: one 128-channel convolution's weights scaled by 1.05 in the device model breaks them.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import abcnet_amd  # noqa: E402,F401
from abcnet_amd.synthetic import synthetic_images  # noqa: E402
from oracle import nms_oracle  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

HEADS = uo.HEADS
DEV = "cuda"
GOLD = os.path.join(HERE, "golden", "calibrated_unet.npz")
BOUNDS = os.path.join(HERE, "golden", "calibrated_deviation.json")
SAMPLE = (0, 21, 42, 63)
HEAD_NAMES = ("atom", "atom_types", "charges", "hs", "bond", "bond_types", "rho", "omega")


def _state(variant="unet"):
    gold = np.load(os.path.join(HERE, "golden", "calibrated_%s.npz" % variant))
    return uo.calibrated_state(variant, 1, HEADS, seed=0, stats=gold["bn_stats"]), gold


def _model(sd, dtype, variant="unet"):
    if variant == "unet2":
        from abcnet_amd.unet2 import UNet
    else:
        from abcnet_amd.unet import UNet
    m = UNet(1, HEADS, dtype=dtype, dropout_p=0.2)
    m.load_state_dict(sd)
    return m.to(DEV).eval()


def _gsample(t, n=4099):
    f = t.detach().reshape(-1)
    step = max(f.numel() // n, 1)
    return f[::step][:n].double().cpu().numpy()


_ORACLE = {}


def _oracle_maps():
    """the oracle's eval maps and NMS decisions for the four sampled images of the benchmark batch (computed once per session)"""
    if "ref" not in _ORACLE:
        sd, _ = _state()
        x = synthetic_images(64, 512, seed=7)[list(SAMPLE)]
        with torch.no_grad():
            ref = uo.forward("unet", sd, x, train=False)
            _ORACLE["ref"] = ref
            _ORACLE["nms"] = nms_oracle.nms(ref[0], ref[4], ref[6], ref[7])
    return _ORACLE["ref"], _ORACLE["nms"]


def measure(fp8=False, fold_bn=True, perturb=None):
    """run config 5's graph on the calibrated weights; perturb = (parameter name, factor): the DEVICE model's tensor scaled
    (the oracle keeps the true weights)"""
    from abcnet_amd.infer import InferenceRunner
    B, S = 64, 512
    sd, gold = _state()
    sd_dev = uo.clone_state(sd)
    if perturb is not None:
        sd_dev[perturb[0]] = sd_dev[perturb[0]] * perturb[1]
    m = _model(sd_dev, "bf16")
    x = synthetic_images(B, S, seed=7)
    run = InferenceRunner(m, B, S, S, use_graph=True, fold_bn=fold_bn, fp8=fp8)
    run.load_batch(x.to(DEV))
    run.step()
    run.step()
    torch.cuda.synchronize()
    idx = torch.tensor(SAMPLE, device=DEV)
    got = [t[idx].cpu() for t in run.logits]
    ref, (ra, rb, rr, ro) = _oracle_maps()
    res = {"fp8": bool(fp8), "fold_bn": bool(fold_bn), "perturb": list(perturb) if perturb else None, "heads": {}}
    for i, (g, r) in enumerate(zip(got, ref)):
        rng = (r.max() - r.min()).item()
        res["heads"][HEAD_NAMES[i]] = {
            "range": rng,
            "linf_over_range": (g - r).abs().max().item() / rng,
            "rms_over_std": ((g - r).double().pow(2).mean().sqrt() / r.double().std()).item(),
        }
    res["worst_linf_over_range"] = max(h["linf_over_range"] for h in res["heads"].values())
    res["worst_rms_over_std"] = max(h["rms_over_std"] for h in res["heads"].values())
    # directly against the reference-generated samples (images 0 and 21 = the first two of SAMPLE)
    worst = 0.0
    for i in range(8):
        st = gold["eval512_head%d_stats" % i]
        d = np.abs(_gsample(got[i][:2]) - gold["eval512_head%d_sample" % i]).max() / (st[1] - st[0])
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
    del run, m
    torch.cuda.empty_cache()
    return res


def _bounds():
    with open(BOUNDS) as f:
        return json.load(f)


CHECKED = ("worst_linf_over_range", "worst_rms_over_std", "golden_sample_linf_over_range")


def _violations(got, ceil):
    bad = []
    for k in CHECKED:
        if got[k] > ceil[k]:
            bad.append((k, got[k], ceil[k]))
    for k in ("atom", "bond", "omega"):
        if got[k + "_peaks"]["rate"] > ceil[k + "_peak_rate"]:
            bad.append((k + "_peak_rate", got[k + "_peaks"]["rate"], ceil[k + "_peak_rate"]))
    return bad


def test_calibrated_eval_fp32_matches_reference_maps():
    """the exact-f32 module forward on the calibrated statistics against the reference's own eval maps (64 x 64, full): 1e-3"""
    sd, gold = _state()
    m = _model(sd, "fp32")
    with torch.no_grad():
        ys = m(synthetic_images(2, 64, seed=7).to(DEV))
    for i, y in enumerate(ys):
        err = float(np.abs(y.cpu().numpy() - gold["eval64_head%d" % i]).max())
        assert err < 1e-3, (i, err)


@pytest.mark.parametrize("key", ["bf16", "fp8"])
def test_inference_accuracy_on_calibrated_statistics(key):
    """config 5's graph (b64 @ 512 x 512; bf16 folded / e4m3) under the hard ceilings of calibrated_deviation.json"""
    ceil = _bounds()["ceilings"][key]
    got = measure(fp8=(key == "fp8"))
    print("calibrated %s: %s" % (key, json.dumps({k: got[k] for k in CHECKED + ("atom_peaks", "bond_peaks", "omega_peaks")})), file=sys.stderr)
    assert got["nms_on_device_logits_exact"], "device NMS != oracle NMS on the device's own logits"
    bad = _violations(got, ceil)
    assert not bad, bad


@pytest.mark.parametrize("key", ["bf16", "fp8"])
def test_a_five_percent_error_in_one_conv_is_caught(key):
    """the same measurement with ONE 128-channel convolution's weights 5 % off in the device model (trunk, dconv1's first conv;
    and one head's conv1) must violate the ceilings: the bounds discriminate"""
    ceil = _bounds()["ceilings"][key]
    for name in ("dconv1.double_conv.0.weight", "out_modules.0.conv1.weight"):
        got = measure(fp8=(key == "fp8"), perturb=(name, 1.05))
        bad = _violations(got, ceil)
        assert bad, ("a 5 %% error in %s passes the %s ceilings" % (name, key), {k: got[k] for k in CHECKED})


if __name__ == "__main__":
    if "--measure" in sys.argv:
        import time
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(HERE)), "gpurun_out", "calibrated_measured.json")
        res = {}
        for key, kw in (("bf16", {}), ("fp8", {"fp8": True}), ("bf16_unfolded", {"fold_bn": False}),
                        ("bf16_trunk_x1.05", {"perturb": ("dconv1.double_conv.0.weight", 1.05)}),
                        ("fp8_trunk_x1.05", {"fp8": True, "perturb": ("dconv1.double_conv.0.weight", 1.05)}),
                        ("bf16_head0_x1.05", {"perturb": ("out_modules.0.conv1.weight", 1.05)}),
                        ("fp8_head0_x1.05", {"fp8": True, "perturb": ("out_modules.0.conv1.weight", 1.05)}),
                        ("bf16_trunk_x1.01", {"perturb": ("dconv1.double_conv.0.weight", 1.01)})):
            t0 = time.time()
            res[key] = measure(**kw)
            print(key, "%.1f s" % (time.time() - t0), json.dumps(res[key]), flush=True)
            os.makedirs(os.path.dirname(out), exist_ok=True)
            with open(out, "w") as f:
                json.dump(res, f, indent=1)
