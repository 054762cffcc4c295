"""Config 5 (img2smiles2.py:42-79: eval forward + peak NMS at 512 x 512, batch 64) on the reference-generated fixture with
BatchNorm running statistics that MATCH the activations (tests/golden/calibrated_unet.npz: the statistics the REFERENCE module
holds after 60 train-mode forwards, and its eval maps with them: +-1..2 per head instead of the 0.05 the random statistics of
oracle.filled_state() give).

What this fixture can and cannot show.  It is anchored in the reference import (the oracle reproduces it bit for bit,
tests/test_oracle_golden.py), so the EXACT-f32 path is held to it at 1e-3.  For reduced precision it measures something else than
kernel quality: a random-weight BatchNorm + ReLU network amplifies perturbations ~1.2x per layer (23 layers: ~80x; 1e-3 relative
noise on the weights moves the fp32 maps by 8 % of their spread), so the reference's OWN bf16 autocast run deviates from its fp32
run by 0.16-0.23 of a head's standard deviation on these weights, and a quarter of the NMS decisions flip.  The bf16 graph is
therefore held to the reference arithmetic's own deviation (measured in the test, on the host, same images: it must not be worse),
and the discriminating accuracy test -- hard ceilings a 5 % error breaks -- runs on TRAINED weights, tests/test_gpu_trained.py.
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
sys.path.insert(0, HERE)

import abcnet_amd  # noqa: E402,F401
from abcnet_amd.synthetic import synthetic_images  # noqa: E402
from oracle import nms_oracle  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
import infer_accuracy as IA  # noqa: E402

HEADS = uo.HEADS
DEV = "cuda"
SAMPLE = (0, 21, 42, 63)
_CACHE = {}


def _state(variant="unet"):
    gold = np.load(os.path.join(HERE, "golden", "calibrated_%s.npz" % variant))
    return uo.calibrated_state(variant, 1, HEADS, seed=0, stats=gold["bn_stats"]), gold


def _model(sd, dtype):
    from abcnet_amd.unet import UNet
    m = UNet(1, HEADS, dtype=dtype, dropout_p=0.2)
    m.load_state_dict(sd)
    return m.to(DEV).eval()


def _setup():
    """images, fp32 oracle maps, and the ORACLE UNDER bf16 AUTOCAST on the same images (the reference arithmetic's own deviation)"""
    if "x" not in _CACHE:
        sd, gold = _state()
        x = synthetic_images(64, 512, seed=7)
        xs = x[list(SAMPLE)]
        oracle = IA.oracle_maps(sd, xs)
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            ac = [t.float() for t in uo.forward("unet", sd, xs, train=False)]
        ref, (ra, rb, _rr, ro) = oracle
        auto = {"heads": {}}
        for i, (a, r) in enumerate(zip(ac, ref)):
            auto["heads"][IA.HEAD_NAMES[i]] = ((a - r).double().pow(2).mean().sqrt() / r.double().std()).item()
        aa, ab, _ar, ao = nms_oracle.nms(ac[0], ac[4], ac[6], ac[7])
        for k, g, r in (("atom", aa, ra), ("bond", ab, rb), ("omega", ao, ro)):
            g, r = g.bool(), r.bool()
            auto[k + "_rate"] = (int((r & ~g).sum()) + int((~r & g).sum())) / max(int(r.sum()), 1)
        gs = ([gold["eval512_head%d_sample" % i] for i in range(8)], [tuple(gold["eval512_head%d_stats" % i][:2]) for i in range(8)])
        _CACHE.update(x=x, sd=sd, gold=gold, oracle=oracle, auto=auto, gs=gs)
    return _CACHE


def measure(fp8=False, fold_bn=True):
    c = _setup()
    m = _model(c["sd"], "bf16")
    res = IA.measure(m, c["x"], SAMPLE, c["oracle"], fp8=fp8, fold_bn=fold_bn, gold_samples=c["gs"])
    res["oracle_autocast"] = c["auto"]
    del m
    torch.cuda.empty_cache()
    return res


def test_calibrated_eval_fp32_matches_reference_maps():
    """the exact-f32 module forward on the calibrated statistics against the reference's own eval maps (64 x 64, full): 1e-3 --
    with maps that span +-1..2, where filled_state()'s eval maps spanned 0.05"""
    sd, gold = _state()
    m = _model(sd, "fp32")
    with torch.no_grad():
        ys = m(synthetic_images(2, 64, seed=7).to(DEV))
    for i, y in enumerate(ys):
        ref = gold["eval64_head%d" % i]
        assert ref.max() - ref.min() > 1.0, i
        err = float(np.abs(y.cpu().numpy() - ref).max())
        assert err < 1e-3, (i, err)


def test_bf16_graph_is_no_worse_than_the_reference_under_autocast():
    """config 5's bf16 BatchNorm-folded graph (b64 @ 512 x 512) on the calibrated random-weight fixture: per head, its rms deviation
    from the fp32 oracle (relative to the head's spread) does not exceed what the reference arithmetic itself shows under
    torch.autocast(bfloat16) on the same images; nor do its NMS decision flips.  (Measured: HIP 0.12-0.17, autocast 0.16-0.23.)"""
    got = measure()
    auto = got["oracle_autocast"]
    print("calibrated bf16: %s" % json.dumps({"hip": {h: round(v["rms_over_std"], 4) for h, v in got["heads"].items()},
                                              "autocast": {h: round(v, 4) for h, v in auto["heads"].items()},
                                              "peaks": {k: (round(got[k + "_peaks"]["rate"], 4), round(auto[k + "_rate"], 4)) for k in ("atom", "bond", "omega")}}),
          file=sys.stderr)
    assert got["nms_on_device_logits_exact"], "device NMS != oracle NMS on the device's own logits"
    for h, v in got["heads"].items():
        assert v["rms_over_std"] <= 1.05 * auto["heads"][h], (h, v["rms_over_std"], auto["heads"][h])
    for k in ("atom", "bond", "omega"):
        assert got[k + "_peaks"]["rate"] <= 1.1 * auto[k + "_rate"] + 0.01, (k, got[k + "_peaks"]["rate"], auto[k + "_rate"])
    # and the chaos is real: were these weights well-conditioned, bf16 would sit at ~1e-2 (it does on trained weights)
    assert min(auto["heads"].values()) > 0.05


def test_fp8_graph_on_the_calibrated_fixture():
    """the e4m3 form on the same fixture: held to 1.6x the reference's autocast deviation per head (measured 1.2-1.6x the bf16
    graph's own: 0.19-0.27) -- recorded for completeness; the fp8 accuracy statement is test_gpu_trained.py's"""
    got = measure(fp8=True)
    auto = got["oracle_autocast"]
    print("calibrated fp8: %s" % json.dumps({h: round(v["rms_over_std"], 4) for h, v in got["heads"].items()}), file=sys.stderr)
    assert got["nms_on_device_logits_exact"]
    for h, v in got["heads"].items():
        assert v["rms_over_std"] <= 1.6 * auto["heads"][h], (h, v["rms_over_std"], auto["heads"][h])
