"""The bounds-check build (aux): a DEBUG plan puts every buffer of the launch plan between two 4 KB guard
bands (Engine(guards=True) / Trainer(guards=True) / InferenceRunner(guards=True)) and `check_guards()` verifies that no kernel
stored outside its tensors.  The C ABI takes raw pointers: loads go through range-checked buffer descriptors, stores do not --
this is the check that the indexing of every launch of a plan (ragged tiles, channel slices of concat buffers, crop rules of
the transposed convolutions at odd sizes, the fp8 graph's byte tensors) stays inside its operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import abcnet_amd  # noqa: E402,F401
from abcnet_amd.synthetic import synthetic_images, synthetic_targets  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

HEADS = uo.HEADS
DEV = "cuda"


def _model(variant, dtype):
    if variant == "unet2":
        from abcnet_amd.unet2 import UNet
    else:
        from abcnet_amd.unet import UNet
    m = UNet(1, HEADS, dtype=dtype)
    m.load_state_dict(uo.filled_state(variant, 1, HEADS, seed=0))
    return m.to(DEV)


@pytest.mark.parametrize("variant,dtype,B,H,W", [("unet", "bf16", 2, 128, 128), ("unet", "fp32", 1, 72, 88), ("unet2", "bf16", 2, 104, 72),
                                                 ("unet2", "fp32", 1, 64, 64)])
def test_train_plan_stores_stay_inside_its_buffers(variant, dtype, B, H, W):
    from abcnet_amd.train import Trainer
    m = _model(variant, dtype)
    tr = Trainer(m, B, H, W, use_graph=False, guards=True)
    x = synthetic_images(B, max(H, W), seed=7)[:, :, :H, :W].contiguous()
    tg = [t[..., :H // 4, :W // 4].contiguous() for t in synthetic_targets(B, max(H, W) // 4, seed=1)]
    tr.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
    tr.step()
    tr.step()
    torch.cuda.synchronize()
    assert tr.eng.check_guards() > 150
    assert torch.isfinite(torch.tensor(tr.loss_value()["total"]))
    # ... and the check does see a stray store
    raw = tr.eng._guarded[7][0]
    raw[tr.eng.GUARD_BYTES - 1] = 0
    with pytest.raises(RuntimeError, match="out-of-bounds"):
        tr.eng.check_guards()


@pytest.mark.parametrize("fp8", [False, True])
def test_inference_plans_store_inside_their_buffers(fp8):
    from abcnet_amd.infer import InferenceRunner
    m = _model("unet", "bf16")
    run = InferenceRunner(m, 2, 160, 96, use_graph=False, fold_bn=True, fp8=fp8, guards=True)
    run.load_batch(synthetic_images(2, 160, seed=7)[:, :, :, :96].contiguous().to(DEV))
    run.step()
    torch.cuda.synchronize()
    assert run.eng.check_guards() > 60
