"""GPU parity at the configurations the benchmark measures (BASELINE.json configs 2, 3 and 5), in the dtype it measures
them in (bf16):

  config 2 / 3: ONE train step of unet.py / unet2.py at 384x384, batch 16, bf16 -- loss, head logits and every
                parameter's gradient against the fp32 ORACLE (CPU) on the same batch, same weights, same dropout mask;
  config 5    : the img2smiles2.py heat-map path at 512x512, batch 64, bf16 -- logits and the NMS masks of a sample of
                the batch against the oracle's eval forward + oracle NMS.

bf16 cannot meet the 1e-3 fp32 gate (the reference itself moves by 0.3-0.5 in its train-mode logits under bf16
autocast), so these tests hold the throughput mode to MEASURED bounds: tests/golden/
bf16_deviation.json is the output of `python tests/test_gpu_fullsize.py --measure` on an MI355X (committed; it also
carries, beside each number, the deviation of the ORACLE ITSELF run under torch.autocast(bfloat16) on the same batch),
and a test fails when the HIP path deviates from the fp32 oracle by more than 1.5x what was measured -- or by more
than the oracle's own autocast run does, where that is recorded.
"""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import abcnet_amd  # noqa: E402,F401
from abcnet_amd.synthetic import synthetic_images, synthetic_targets  # noqa: E402
from oracle import loss_oracle, nms_oracle  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

HEADS = uo.HEADS
DEV = "cuda"
PRE_BN_BIAS = ("double_conv.0.bias", "double_conv.3.bias", "conv1.bias")
BOUNDS = os.path.join(HERE, "golden", "bf16_deviation.json")


def _model(variant, dtype, dropout_p):
    if variant == "unet2":
        from abcnet_amd.unet2 import UNet
    else:
        from abcnet_amd.unet import UNet
    m = UNet(1, HEADS, dtype=dtype, dropout_p=dropout_p)
    m.load_state_dict(uo.filled_state(variant, 1, HEADS, seed=0))
    return m.to(DEV)


def _grad_stats(get, ref_sd):
    """per-parameter relative L2 deviation of the gradient; returns (worst, median, name of worst, whole-model relative L2)"""
    rels, num, den = [], 0.0, 0.0
    for name, t in ref_sd.items():
        if t.grad is None or name.endswith(PRE_BN_BIAS):
            continue
        ref = t.grad.double()
        got = get(name).double().cpu()
        e, n = (got - ref).norm().item(), ref.norm().item()
        rels.append((e / (n + 1e-30), name))
        num += e * e
        den += n * n
    rels.sort()
    return rels[-1][0], rels[len(rels) // 2][0], rels[-1][1], (num / den) ** 0.5


def measure_train(variant, with_autocast=False):
    """one bf16 train step at the benchmark shape vs the fp32 oracle (and, optionally, the oracle under bf16 autocast)"""
    from abcnet_amd.dropout import head_keep_masks
    from abcnet_amd.train import Trainer
    B, S = 16, 384
    x = synthetic_images(B, S, seed=7)
    tg = synthetic_targets(B, S // 4, seed=1)
    drop = 0.2 if variant == "unet" else 0.0      # unet2's heads have no Dropout (unet2.py:116-126)
    m = _model(variant, "bf16", drop)
    tr = Trainer(m, B, S, S, lr=0.0, use_graph=False)
    masks = head_keep_masks(B, S // 4, S // 4, 8, tr.eng.dropout_seed(1), 0.2) if drop > 0 else None
    tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
    tr.step()
    torch.cuda.synchronize()
    hip_loss = tr.loss_value()["total"]
    hip_logits = [t.cpu().clone() for t in tr.eng.logits]

    def oracle(autocast):
        sd = uo.clone_state(uo.filled_state(variant, 1, HEADS, seed=0), requires_grad=True)
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                preds = uo.forward(variant, sd, x, train=True, dropout_masks=masks)
            preds = [p.float() for p in preds]
        else:
            preds = uo.forward(variant, sd, x, train=True, dropout_masks=masks)
        total, _, _ = loss_oracle.abc_loss(preds, tg, sd["s"])
        total.backward()
        return sd, total.item(), [p.detach() for p in preds]

    sd, ref_loss, ref_logits = oracle(False)
    worst, med, wname, whole = _grad_stats(lambda n: m.grad_of(n), sd)
    out = {"loss_rel": abs(hip_loss - ref_loss) / abs(ref_loss), "loss_hip": hip_loss, "loss_oracle_fp32": ref_loss,
           "logits_linf": max((a - b).abs().max().item() for a, b in zip(hip_logits, ref_logits)),
           "grad_rel_l2_worst": worst, "grad_rel_l2_median": med, "grad_worst_param": wname, "grad_rel_l2_whole_model": whole}
    if with_autocast:
        sda, la, lga = oracle(True)
        w2, m2, n2, wh2 = _grad_stats(lambda n: sda[n].grad, sd)
        out["oracle_autocast"] = {"loss_rel": abs(la - ref_loss) / abs(ref_loss),
                                  "logits_linf": max((a - b).abs().max().item() for a, b in zip(lga, ref_logits)),
                                  "grad_rel_l2_worst": w2, "grad_rel_l2_median": m2, "grad_worst_param": n2,
                                  "grad_rel_l2_whole_model": wh2}
    return out


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_benchmark_step_is_bit_reproducible(variant):
    """The step the benchmark times (b16 @ 384 x 384, bf16, hipGraph replay, the default plan), run twice from the same state on the
    same batch for three steps: parameters, BatchNorm buffers, Adam moments, gradients and logits equal BIT FOR BIT.  Every kernel is
    order-deterministic by construction (no float atomics, fixed-order split-K), so any difference is a race or a hardware hazard -- the
    kind of fault that shows in a few hundred of 19 M outputs of one launch in three (the store-data hazard of DESIGN.md section 3 was
    found by an fp8 kernel test going red intermittently); the launch shapes that exist only at this size (persistent multi-tile
    workgroups, the 1024-row merged weight gradient, 2 M-thread fused heads pass) are otherwise only held to tolerances."""
    from abcnet_amd.train import Trainer
    B, S = 16, 384
    x = synthetic_images(B, S, seed=7).to(DEV)
    tg = [t.to(DEV) for t in synthetic_targets(B, S // 4, seed=1)]

    def run():
        m = _model(variant, "bf16", 0.2 if variant == "unet" else 0.0)
        tr = Trainer(m, B, S, S, lr=2.5e-4, use_graph=True)
        tr.load_batch(x, tg)
        for _ in range(3):
            tr.step()
        torch.cuda.synchronize()
        return (m._flat.data.clone(), m._flat_buf.clone(), m._flat_grad.clone(), tr.opt.m.clone(), tr.opt.v.clone(),
                [t.clone() for t in tr.eng.logits], tr.loss_value()["total"])

    a, b = run(), run()
    for name, ta, tb in zip(("parameters", "buffers", "gradients", "adam m", "adam v"), a[:5], b[:5]):
        assert torch.equal(ta, tb), (name, int((ta != tb).sum()))
    for i, (la, lb) in enumerate(zip(a[5], b[5])):
        assert torch.equal(la, lb), ("logits", i, int((la != lb).sum()))
    assert a[6] == b[6]


SAMPLE = (0, 21, 42, 63)


def measure_infer(fold_bn=True, fp8=False):
    """config 5: eval forward + NMS at 512x512, batch 64, bf16 (the BatchNorm-folded graph the benchmark runs) or its fp8 (e4m3)
    form; a sample of the batch against the oracle"""
    from abcnet_amd.infer import InferenceRunner
    B, S = 64, 512
    x = synthetic_images(B, S, seed=7)
    m = _model("unet", "bf16", 0.2)
    run = InferenceRunner(m, B, S, S, use_graph=True, fold_bn=fold_bn, fp8=fp8)
    run.load_batch(x.to(DEV))
    run.step()
    run.step()      # the second step is the captured graph
    torch.cuda.synchronize()
    idx = torch.tensor(SAMPLE)
    with torch.no_grad():
        ref = uo.forward("unet", uo.filled_state("unet", 1, HEADS, seed=0), x[idx], train=False)
        ra, rb, rr, ro = nms_oracle.nms(ref[0], ref[4], ref[6], ref[7])
    got = [t[idx.to(DEV)].cpu() for t in run.logits]
    linf = max((a - b).abs().max().item() for a, b in zip(got, ref))
    flips = lambda g, r: int((g[idx.to(DEV)].cpu() != r).sum().item())
    # decisions of the device NMS on the DEVICE's logits equal the oracle NMS applied to those same logits: bit-exact
    da, db, dr, do = nms_oracle.nms(got[0], got[4], got[6], got[7])
    exact = (torch.equal(run.atom_mask[idx.to(DEV)].cpu(), da) and torch.equal(run.bond_mask[idx.to(DEV)].cpu(), db)
             and torch.equal(run.omega_mask[idx.to(DEV)].cpu(), do) and torch.equal(run.rho_abs[idx.to(DEV)].cpu(), dr))
    n_pix = len(SAMPLE) * (S // 4) ** 2
    return {"fold_bn": bool(fold_bn), "fp8": bool(fp8), "logits_linf": linf, "atom_mask_flips": flips(run.atom_mask, ra), "bond_mask_flips": flips(run.bond_mask, rb),
            "omega_mask_flips": flips(run.omega_mask, ro), "pixels": n_pix, "omega_entries": 60 * n_pix,
            "atom_peaks_oracle": int(ra.sum().item()), "omega_peaks_oracle": int(ro.sum().item()),
            "nms_on_device_logits_exact": bool(exact)}


@pytest.mark.parametrize("fp8", [False, True])
def test_benchmark_inference_is_bit_reproducible(fp8):
    """config 5's graph (b64 @ 512 x 512, BatchNorm folded, bf16 and e4m3), built twice and replayed four times each on the same
    batch: the eight maps and the NMS outputs equal bit for bit between all eight replays (see the training twin above: the net for
    races and hazards; the 128 x 128 maps under 12-row tiles also put one tile in eleven on the row-short epilogue path)"""
    from abcnet_amd.infer import InferenceRunner
    B, S = 64, 512
    x = synthetic_images(B, S, seed=7).to(DEV)
    ref = None
    for _build in range(2):
        m = _model("unet", "bf16", 0.2)
        run = InferenceRunner(m, B, S, S, use_graph=True, fold_bn=True, fp8=fp8)
        run.load_batch(x)
        for rep in range(5):
            run.step()
            torch.cuda.synchronize()
            if rep == 0:
                continue      # (eager; the others replay the captured graph)
            out = [t.clone() for t in run.logits] + [run.atom_mask.clone(), run.bond_mask.clone(), run.omega_mask.clone(), run.rho_abs.clone()]
            if ref is None:
                ref = out
            else:
                for i, (a, b) in enumerate(zip(ref, out)):
                    assert torch.equal(a, b), (_build, rep, i, int((a != b).sum()))
        del run, m
        torch.cuda.empty_cache()


def _bounds():
    with open(BOUNDS) as f:
        return json.load(f)


@pytest.mark.parametrize("variant", ["unet", "unet2"])
def test_bf16_train_step_at_benchmark_config(variant):
    b = _bounds()["train_" + variant]
    got = measure_train(variant)
    report = {k: got[k] for k in ("loss_rel", "logits_linf", "grad_rel_l2_worst", "grad_rel_l2_median", "grad_rel_l2_whole_model")}
    for k, v in report.items():
        assert v <= 1.5 * b[k] + 1e-12, (variant, k, v, b[k], got["grad_worst_param"])
    # and never worse than what the reference arithmetic itself does under bf16 autocast (recorded beside the bound),
    # for the quantities where that comparison is meaningful end to end
    ac = b["oracle_autocast"]
    assert got["loss_rel"] <= max(ac["loss_rel"], 1.5 * b["loss_rel"])
    assert got["grad_rel_l2_whole_model"] <= max(ac["grad_rel_l2_whole_model"], 1.5 * b["grad_rel_l2_whole_model"])


@pytest.mark.parametrize("key", ["infer_unet", "infer_unet_fp8"])
def test_inference_at_benchmark_config(key):
    """bf16 folded graph / its fp8 (e4m3) form at 512 x 512, batch 64 (`bench.py --mode infer [--dtype fp8]`)"""
    b = _bounds()[key]
    got = measure_infer(fp8=key.endswith("fp8"))
    assert got["nms_on_device_logits_exact"], "device NMS != oracle NMS on the device's own logits"
    assert got["logits_linf"] <= 1.5 * b["logits_linf"], (got["logits_linf"], b["logits_linf"])
    # (no bound on the mask flips here any more: with filled_state()'s random running statistics the eval maps are almost
    #  constant -- the atom map spans 0.05 -- a quarter of the pixels are "peaks" and the flip counts are noise.  The accuracy of
    #  the bf16 / fp8 graphs is held on weights where it means something: tests/test_gpu_calibrated.py (reference-generated
    #  statistics that match the activations) and tests/test_gpu_trained.py (a trained network: hard ceilings relative to each
    #  head's range and to the oracle's peak / candidate lists, which a 5 % error in one convolution breaks))


if __name__ == "__main__":
    if "--measure" in sys.argv:
        import time
        out = sys.argv[sys.argv.index("--measure") + 1] if len(sys.argv) > sys.argv.index("--measure") + 1 else BOUNDS
        res = {"how": "python tests/test_gpu_fullsize.py --measure on an MI355X (bf16, the shapes of BASELINE.json configs 2, 3, 5); "
                      "deviations are against the fp32 oracle on the same inputs; oracle_autocast = the oracle itself under "
                      "torch.autocast('cpu', bfloat16) against its own fp32 run; host threads: %d" % torch.get_num_threads()}
        for key, fn in (("infer_unet", measure_infer), ("infer_unet_fp8", lambda: measure_infer(fp8=True)),
                        ("infer_unet_unfolded", lambda: measure_infer(False)), ("train_unet", lambda: measure_train("unet", True)),
                        ("train_unet2", lambda: measure_train("unet2", True))):
            t0 = time.time()
            res[key] = fn()
            print(key, "%.1f s" % (time.time() - t0), json.dumps(res[key]), flush=True)
            with open(out, "w") as f:
                json.dump(res, f, indent=1)
