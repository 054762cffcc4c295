"""GPU parity: the device-side training meters (abc_metrics_update, through the C ABI) against the golden
vectors generated from the reference (train.py:145-215 executed with meter.AverageMeter) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import abcnet_amd  # noqa: E402,F401
from abcnet_amd.ops import METER_NAMES, FusedMetrics  # noqa: E402
from abcnet_amd.synthetic import correlated_logits, synthetic_images, synthetic_targets  # noqa: E402
from oracle import loss_oracle  # noqa: E402
from oracle import metrics_oracle as mo  # noqa: E402

DEV = "cuda"


def _check(res, want, tol_count=0.0):
    for n in METER_NAMES:
        num, den = want[n]
        # counting meters are exact integers / half-integers; the rho MAE numerator is a float sum
        tol = 1e-6 * max(1.0, abs(num.item())) if n == "bond_rhos_mae" else tol_count
        assert abs(res[n]["sum"] - num.item()) <= tol + 1e-9, (n, res[n]["sum"], num.item())
        assert abs(res[n]["count"] - den.item()) <= tol_count + 1e-9, (n, res[n]["count"], den.item())


def test_metrics_match_golden_and_oracle(golden_dir):
    gold = np.load(os.path.join(golden_dir, "metrics_128.npz"))
    tg = synthetic_targets(2, 128, seed=3)
    lg = correlated_logits(tg, seed=19)
    fm = FusedMetrics([t.to(DEV).contiguous() for t in lg], [t.to(DEV) for t in tg])
    fm.run()
    res = fm.result()
    assert METER_NAMES == mo.METER_NAMES == [n[len("train_"):] for n in gold["names"]]
    for n, s, c in zip(METER_NAMES, gold["sum"], gold["count"]):
        assert abs(res[n]["sum"] - s) <= 1e-5 * max(1.0, abs(s)), (n, res[n]["sum"], s)
        assert abs(res[n]["count"] - c) <= 1e-5 * max(1.0, abs(c)), (n, res[n]["count"], c)
    _check(res, mo.metrics(loss_oracle.activations(lg), tg))
    # a second update accumulates like AverageMeter (sum += num, count += den); reset() starts over
    fm.run()
    res2 = fm.result()
    for n in METER_NAMES:
        assert abs(res2[n]["sum"] - 2 * res[n]["sum"]) <= 1e-9 * max(1.0, abs(res[n]["sum"]))
        assert abs(res2[n]["count"] - 2 * res[n]["count"]) <= 1e-9 * max(1.0, res[n]["count"])
        assert res2[n]["val"] == res[n]["val"] or (np.isnan(res2[n]["val"]) and np.isnan(res[n]["val"]))
    fm.reset()
    fm.run()
    _check(fm.result(), mo.metrics(loss_oracle.activations(lg), tg))


@pytest.mark.parametrize("B,h", [(1, 32), (3, 96), (16, 96)])
def test_metrics_other_shapes_match_oracle(B, h):
    """ragged sizes (pixel count not a multiple of the workgroup), borders, the benchmark's own shape; empty targets in
    one image (every denominator that can be zero stays a clean zero, not NaN, in the table)"""
    tg = synthetic_targets(B, h, seed=5)
    for t in tg:
        t[0].zero_()          # image 0 has no atoms and no bonds at all
    lg = correlated_logits(tg, seed=23)
    fm = FusedMetrics([t.to(DEV).contiguous() for t in lg], [t.to(DEV) for t in tg])
    fm.run()
    _check(fm.result(), mo.metrics(loss_oracle.activations(lg), tg))


def test_trainer_updates_meters_every_step():
    """Trainer(metrics=True): the meters ride inside the captured step (no host sync) and see the logits of THAT step"""
    from abcnet_amd.train import Trainer
    from abcnet_amd.unet import UNet
    from oracle import unet_oracle as uo
    m = UNet(1, uo.HEADS, dtype="fp32", dropout_p=0.0)
    m.load_state_dict(uo.filled_state("unet", 1, uo.HEADS, seed=0))
    m = m.to(DEV)
    B, S = 2, 64
    x, tg = synthetic_images(B, S, seed=7), synthetic_targets(B, S // 4, seed=1)
    tr = Trainer(m, B, S, S, use_graph=True, metrics=True)
    tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
    acc = None
    for step in range(3):   # eager, capture, replay
        tr.step()
        torch.cuda.synchronize()
        want = mo.metrics(loss_oracle.activations([t.cpu() for t in tr.eng.logits]), tg)
        acc = want if acc is None else {k: (acc[k][0] + want[k][0], acc[k][1] + want[k][1]) for k in want}
        _check(tr.metrics.result(), acc)


def test_metrics_fail_loudly_on_cpu_tensors():
    tg = synthetic_targets(1, 32, seed=5)
    lg = correlated_logits(tg, seed=23)
    with pytest.raises(Exception):
        FusedMetrics(lg, tg)
