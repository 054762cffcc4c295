"""Config 5 (img2smiles2.py:42-79 + 113-191) on TRAINED weights -- the accuracy statement about the bf16 and fp8 (e4m3) inference
graphs that can fail.

The weights are a FROZEN fixture: unet.py trained ONCE on the device for 3000 steps on drawn molecules (tests/trained_fixture.py;
tests/golden/make_trained_fixture.py), its parameters rounded to bf16 and committed (tests/golden/trained_unet_state.npz).  That
gives what the path serves in practice: peaked atom / bond heat maps, heavy-tailed activations for the per-tensor e4m3
calibration to cope with, and -- unlike any random-weight network, tests/test_gpu_calibrated.py -- a function that does not
amplify a 1e-3 perturbation eighty-fold.  The expected outputs come from THE REFERENCE run on those very weights
(tests/golden/make_golden.py trained -> trained_unet.npz: eval maps and NMS decisions of the 16 sampled images of the accuracy
batch); the oracle reproduces them bit for bit (tests/test_oracle_golden.py, and re-checked here on the GPU box's host).  At the
benchmarked size (b64 @ 512 x 512), against those maps:

  * every head's logits relative to the head's range;
  * NMS decisions and the extracted candidate lists (atoms: position, type, charge, hs; bonds: position, omega bin, type, rho) as
    missed + spurious (+ wrong class) out of the oracle's, under hard ceilings (tests/golden/trained_deviation.json);
  * the same with ONE 128-channel convolution 5 % off in the device model must break the ceilings.

The ceilings were set ONCE for the frozen weights (measured x 2, x 1.8 for e4m3, with floors) and are NOT regenerated when kernels
change: a kernel change that moves the device's maps on these weights beyond them is a regression to look at, not a fixture to
re-fit.  Training on the device at test time is a separate test ("the fixture learns"): its trajectory moves with every kernel
change and nothing else is measured on it.
"""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import abcnet_amd  # noqa: E402,F401
from abcnet_amd.synthetic import drawn_molecules  # noqa: E402
import infer_accuracy as IA  # noqa: E402
import trained_fixture as TF  # noqa: E402

HEADS = TF.HEADS
DEV = "cuda"
BOUNDS = os.path.join(HERE, "golden", "trained_deviation.json")
TRAIN_STEPS = 3000
SAMPLE = tuple(range(0, 64, 4))          # 16 of the 64 images: ~200 atoms, ~170 bonds to count decisions on
FROZEN = os.path.join(HERE, "golden", "trained_unet_state.npz")
GOLD = os.path.join(HERE, "golden", "trained_unet.npz")
_CACHE = {}


def _sample(t, n):
    f = t.detach().reshape(-1)
    step = max(f.numel() // n, 1)
    return f[::step][:n].double().numpy()


def _trained():
    """(device model, frozen state on the CPU, eval images, oracle maps of the sampled images -- held to the reference's, below --,
    the fixture's training record): once per session"""
    if "m" not in _CACHE:
        import numpy as np
        sys.path.insert(0, os.path.join(HERE, "golden"))
        from make_trained_fixture import unpack_state
        from abcnet_amd.unet import UNet
        sd_cpu = unpack_state(FROZEN)
        m = UNet(1, HEADS, dtype="bf16", dropout_p=0.2)
        m.load_state_dict(sd_cpu)
        m = m.to(DEV).eval()
        x, _notes = drawn_molecules(64, 512, seed=777)
        oracle = IA.oracle_maps(sd_cpu, x[list(SAMPLE)])
        # the oracle's maps of THIS host are the reference's maps of the dev container (trained_unet.npz): samples within 1e-5 (bit-equal
        # where the host's thread count matches), every NMS decision equal up to 2 ties per mask
        gold = np.load(GOLD)
        assert list(gold["sample"]) == list(SAMPLE)
        ref, (ra, rb, _rr, ro) = oracle
        for i, y in enumerate(ref):
            mine = np.stack([_sample(y[b], 4099) for b in range(y.shape[0])])
            np.testing.assert_allclose(mine, gold["eval512_head%d_sample" % i], rtol=0, atol=2e-5)
        for name, mk in (("atom", ra), ("bond", rb), ("omega", ro)):
            got = np.packbits(mk.numpy().astype(np.uint8))
            assert int(np.unpackbits(got ^ gold["nms512_" + name]).sum()) <= 2, name
        with open(os.path.join(HERE, "golden", "trained_unet_state.json")) as f:
            info = json.load(f)
        _CACHE.update(m=m, sd=sd_cpu, x=x, oracle=oracle, info=info)
    c = _CACHE
    return c["m"], c["sd"], c["x"], c["oracle"], c["info"]


def _perturbed(sd, name, factor):
    from abcnet_amd.unet import UNet
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2[name] = sd2[name] * factor
    m = UNet(1, HEADS, dtype="bf16", dropout_p=0.2)
    m.load_state_dict(sd2)
    return m.to(DEV).eval()


def measure(fp8=False, perturb=None, fold_bn=True):
    m, sd, x, oracle, _info = _trained()
    dev_model = m if perturb is None else _perturbed(sd, *perturb)
    res = IA.measure(dev_model, x, SAMPLE, oracle, fp8=fp8, fold_bn=fold_bn, extract=True)
    res["perturb"] = list(perturb) if perturb else None
    return res


CHECKED = ("worst_linf_over_range", "worst_rms_over_std")


def _violations(got, ceil):
    bad = [(k, got[k], ceil[k]) for k in CHECKED if got[k] > ceil[k]]
    for h, v in got["heads"].items():       # every head against ITS OWN ceiling (an error in one head's branch moves that head only)
        if v["rms_over_std"] > ceil["heads_rms_over_std"][h]:
            bad.append(("rms_over_std:" + h, v["rms_over_std"], ceil["heads_rms_over_std"][h]))
    for k in ("atom", "bond", "omega"):
        if got[k + "_peaks"]["rate"] > ceil[k + "_peak_rate"]:
            bad.append((k + "_peak_rate", got[k + "_peaks"]["rate"], ceil[k + "_peak_rate"]))
    for k in ("atoms_rate", "bonds_rate"):
        if got["candidates"][k] > ceil["candidate_" + k]:
            bad.append(("candidate_" + k, got["candidates"][k], ceil["candidate_" + k]))
    return bad


def _bounds():
    with open(BOUNDS) as f:
        return json.load(f)


def test_training_on_drawn_molecules_learns():
    """the recipe that made the frozen fixture, run again on THIS build (tests/trained_fixture.py: the reference's model, loss and
    optimiser on the HIP path, 3000 steps): the loss falls, the network finds the atoms and bonds.  Nothing else is measured on
    these weights -- their low-order bits move with every kernel change."""
    m, _sd, info = TF.train_unet(steps=TRAIN_STEPS, log=lambda s: print(s, file=sys.stderr, flush=True))
    first, last = info["loss"][0][1], info["loss"][-1][1]
    assert last < 0.05 * first, info["loss"]
    mt = info["meters"]          # (running meters of the last training steps: train.py:145-215 on the device)
    assert mt["atom_targets_recall"] > 0.9 and mt["atom_targets_precision"] > 0.9 and mt["bond_targets_recall"] > 0.9, mt
    assert mt["atom_types_acc"] > 0.9, mt
    del m
    torch.cuda.empty_cache()


def test_the_frozen_fixture_is_a_trained_network():
    """the committed weights are what the header says: the record of their training shows the loss falling and the meters above 0.9, and
    on UNSEEN drawings at another size (512 x 512) the reference's maps have about as many atom peaks as atoms drawn (8 .. 22 per image),
    not the thousands a random-weight network produces"""
    _m, _sd, _x, (ref, (ra, rb, _rr, _ro)), info = _trained()
    assert info["loss"][-1][1] < 0.05 * info["loss"][0][1], info["loss"]
    assert info["meters"]["atom_targets_recall"] > 0.9 and info["meters"]["bond_targets_recall"] > 0.9
    n = int(ra.sum().item()) / len(SAMPLE)
    assert 6 <= n <= 30, n


def test_fp32_forward_on_trained_weights_is_within_1e_3():
    """the north-star tolerance (logits within 1e-3 of the fp32 reference arithmetic) on TRAINED weights -- peaked maps, logits
    spanning tens of units, heavy-tailed activations -- with the exact-f32 HIP path: eval forward of two 512 x 512 drawings against
    the oracle, every head, absolute"""
    from abcnet_amd.unet import UNet
    _m, sd, x, (ref, _nms), _info = _trained()
    m32 = UNet(1, HEADS, dtype="fp32", dropout_p=0.2)
    m32.load_state_dict(sd)
    m32 = m32.to(DEV).eval()
    with torch.no_grad():
        ys = m32(x[list(SAMPLE[:2])].to(DEV))
    worst, span = 0.0, 0.0
    for i, (y, r) in enumerate(zip(ys, ref)):
        err = (y.cpu() - r[:2]).abs().max().item()
        worst, span = max(worst, err), max(span, (r.max() - r.min()).item())
        assert err < 1e-3, (i, err)
    print("fp32 on trained weights: worst |dlogit| %.2e over maps spanning up to %.1f" % (worst, span), file=sys.stderr)
    del m32
    torch.cuda.empty_cache()


@pytest.mark.parametrize("key", ["bf16", "fp8"])
def test_inference_accuracy_on_trained_weights(key):
    ceil = _bounds()["ceilings"][key]
    got = measure(fp8=(key == "fp8"))
    print("trained %s: %s" % (key, json.dumps({k: got[k] for k in CHECKED + ("atom_peaks", "bond_peaks", "omega_peaks", "candidates")})), file=sys.stderr)
    assert got["nms_on_device_logits_exact"], "device NMS != oracle NMS on the device's own logits"
    assert got["candidates"]["truncated"] == 0
    bad = _violations(got, ceil)
    assert not bad, bad


@pytest.mark.parametrize("key", ["bf16", "fp8"])
def test_a_five_percent_error_in_one_conv_is_caught(key):
    """one 128-channel convolution's weights 5 % off in the DEVICE model (the trunk's dconv1; the bond-type head's conv1) must
    violate the ceilings the intact graph meets"""
    ceil = _bounds()["ceilings"][key]
    # (e4m3's 3-bit mantissa costs the fp8 graph ~8 % rms of a head's spread by itself -- as much as a 3-4 % weight error -- so
    #  its ceilings can only tell a 10 % error apart; the bf16 graph's tell 5 % -- and 1 % -- apart)
    factor = 1.10 if key == "fp8" else 1.05
    # (the trunk, the bond-type branch and the omega branch: an error confined to one head's branch has to be seen by that head's ceiling)
    for name in ("dconv1.double_conv.0.weight", "out_modules.5.conv1.weight", "out_modules.7.conv1.weight"):
        got = measure(fp8=(key == "fp8"), perturb=(name, factor))
        assert _violations(got, ceil), ("a %g x error in %s passes the %s ceilings" % (factor, name, key), {k: got[k] for k in CHECKED}, got["candidates"])


def write_bounds(res, out_dir):
    """tests/golden/trained_deviation.json from the measurements of `--measure` (res: its trained_measured_<steps>.json)"""
    # the ceilings file: measured values x a margin (the trained weights move a little from build to build: the device
    # step is bit-reproducible within a build, not across kernel changes), with floors where the measured count is ~0
    # (head_floor: a head whose deviation is ~0.1 % of its spread -- rho in bf16 -- moves by more than the margin between two trainings
    #  of the fixture (0.0011 / 0.0014 / 0.0029 after three kernel changes); 0.5 % still fails a 1 % error in one trunk convolution,
    #  which puts rho at 0.0077 and every other head at 0.020-0.034)
    def ceilings(m, f, atom_floor, cand_floor, bond_floor=0.03, head_floor=0.005):
        pk = lambda k, floor: max(floor, f * m[k + "_peaks"]["rate"])
        return {"worst_linf_over_range": f * m["worst_linf_over_range"], "worst_rms_over_std": f * m["worst_rms_over_std"],
                "heads_rms_over_std": {h: max(head_floor, f * v["rms_over_std"]) for h, v in m["heads"].items()},
                "atom_peak_rate": pk("atom", atom_floor), "bond_peak_rate": pk("bond", bond_floor), "omega_peak_rate": pk("omega", 0.03),
                "candidate_atoms_rate": max(cand_floor, f * m["candidates"]["atoms_rate"]),
                "candidate_bonds_rate": max(0.05, f * m["candidates"]["bonds_rate"])}
    # (e4m3: two trainings of the same fixture, before and after a kernel change that moved the low-order bits of the training
    #  trajectory, gave per-head deviations +-50 % apart -- atom 0.093 / 0.058, atom types 0.035 / 0.054 of a head's spread -- and
    #  0 / 4 missed + spurious atom peaks of ~190: the fp8 margin is 1.8 with decision floors of 4-5 %; a 10 % error in one
    #  convolution still lands at 1.8-3.5 x the measured values)
    prop = {"how": "python tests/test_gpu_trained.py --measure on an MI355X: the FROZEN fixture (tests/golden/trained_unet_state.npz: unet.py trained %d steps on "
                   "drawn molecules ONCE, parameters rounded to bf16), config 5's graph at b64 @ 512 x 512 against the reference's maps of the same weights "
                   "(trained_unet.npz, reproduced by the oracle) on %d images; ceilings = measured x 2 (bf16) / x 1.8 (fp8), with floors on the decision rates "
                   "(2 %% / 3 %% bf16, 4 %% / 5 %% fp8) and on a head's rms / std (0.5 %% bf16, 2 %% fp8)" % (res["train_steps"], len(SAMPLE)),
            "policy": "set ONCE for the frozen weights (round 5); NOT to be regenerated when kernels change -- a build that exceeds them has regressed",
            "measured": {k: res[k] for k in res if k not in ("train_steps", "info")}, "training": res["info"],
            "ceilings": {"bf16": ceilings(res["bf16"], 2.0, 0.02, 0.03), "fp8": ceilings(res["fp8"], 1.8, 0.04, 0.05, 0.05, head_floor=0.02)}}
    with open(os.path.join(out_dir, "trained_deviation.json"), "w") as f:
        json.dump(prop, f, indent=1)


if __name__ == "__main__":
    if "--bounds-from" in sys.argv:      # (no GPU: the ceilings file again from a measurement file, e.g. after a change of the margins)
        src = sys.argv[sys.argv.index("--bounds-from") + 1]
        with open(src) as f:
            write_bounds(json.load(f), os.path.dirname(os.path.abspath(src)))
        sys.exit(0)
    if "--measure" in sys.argv:
        import time
        if "--steps" in sys.argv:
            TRAIN_STEPS = int(sys.argv[sys.argv.index("--steps") + 1])
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(HERE)), "gpurun_out", "trained_measured_%d.json" % TRAIN_STEPS)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        res = {"train_steps": TRAIN_STEPS}
        t0 = time.time()
        info = _trained()[4]
        res["info"] = {"loss": info["loss"], "meters": info["meters"], "train_s": info["train_s"], "fixture": "tests/golden/trained_unet_state.npz"}
        print("trained in %.1f s" % (time.time() - t0), json.dumps(res["info"]), flush=True)
        for key, kw in (("bf16", {}), ("fp8", {"fp8": True}), ("bf16_unfolded", {"fold_bn": False}),
                        ("bf16_trunk_x1.05", {"perturb": ("dconv1.double_conv.0.weight", 1.05)}),
                        ("fp8_trunk_x1.05", {"fp8": True, "perturb": ("dconv1.double_conv.0.weight", 1.05)}),
                        ("fp8_trunk_x1.10", {"fp8": True, "perturb": ("dconv1.double_conv.0.weight", 1.10)}),
                        ("bf16_head5_x1.05", {"perturb": ("out_modules.5.conv1.weight", 1.05)}),
                        ("fp8_head5_x1.10", {"fp8": True, "perturb": ("out_modules.5.conv1.weight", 1.10)}),
                        ("bf16_head7_x1.05", {"perturb": ("out_modules.7.conv1.weight", 1.05)}),
                        ("fp8_head7_x1.10", {"fp8": True, "perturb": ("out_modules.7.conv1.weight", 1.10)}),
                        ("bf16_trunk_x1.01", {"perturb": ("dconv1.double_conv.0.weight", 1.01)})):
            t0 = time.time()
            res[key] = measure(**kw)
            print(key, "%.1f s" % (time.time() - t0), json.dumps(res[key]), flush=True)
            with open(out, "w") as f:
                json.dump(res, f, indent=1)
        write_bounds(res, os.path.dirname(out))
