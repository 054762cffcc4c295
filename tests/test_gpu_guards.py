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


def test_a_second_trainer_cannot_move_the_reserved_cus_under_the_first():
    """Trainer(reserve_cus=n) sets a PROCESS-wide value that the first plan's persistent grids and BatchNorm partial-row buffers were
    sized for: a default-constructed Trainer leaves it alone (and plans for it), one that asks for another value is refused while the
    first lives, a direct library call is caught before the next launch, and the step under reserved CUs stays inside its buffers"""
    import gc
    from abcnet_amd import _lib as L
    from abcnet_amd import engine as E
    from abcnet_amd.train import Trainer
    lib = L.load()
    B, S = 2, 64
    x = synthetic_images(B, S, seed=7)
    tg = synthetic_targets(B, S // 4, seed=1)
    import weakref
    gc.collect()
    # (plans that other tests of this session keep alive -- cached fixtures -- stay out of it: they are not launched meanwhile, and the
    #  process-wide value is back at 0 when this test ends)
    saved, E.Engine._live = E.Engine._live, weakref.WeakSet()
    m1 = _model("unet", "bf16")
    t1 = Trainer(m1, B, S, S, use_graph=False, guards=True, reserve_cus=8)
    try:
        assert lib.abc_get_reserved_cus() == 8 and t1.eng.reserved_cus == 8
        t1.load_batch(x.to(DEV), [t.to(DEV) for t in tg])
        t1.step()
        t2 = Trainer(_model("unet", "bf16"), B, S, S, use_graph=False)      # default: does not reset the process-wide value
        assert lib.abc_get_reserved_cus() == 8 and t2.eng.reserved_cus == 8
        with pytest.raises(L.AbcNetHipError, match="plan"):
            Trainer(_model("unet", "bf16"), B, S, S, use_graph=False, reserve_cus=0)
        L.check(lib.abc_set_reserved_cus(0), "set")         # behind the engine's back
        with pytest.raises(L.AbcNetHipError, match="reserved-CU"):
            t1.step()
        L.check(lib.abc_set_reserved_cus(8), "set")
        t1.step()
        torch.cuda.synchronize()
        assert t1.eng.check_guards() > 150
        del t2
    finally:
        del t1, m1
        gc.collect()
        L.check(lib.abc_set_reserved_cus(0), "set")
        left = [e for e in E.Engine._live]
        E.Engine._live = saved
    assert not left, "plans of this test are still alive: %d" % len(left)
