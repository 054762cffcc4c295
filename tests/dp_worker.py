"""One data-parallel rank of tests/test_gpu_00_dataparallel.py (started by abcnet_amd.distributed.launch_ranks, i.e. the
way multi_gpu_train.py:30-53 starts main_worker: one fresh process per rank, init_process_group, model on its GPU,
parameters of rank 0 everywhere, then the training loop).

    python dp_worker.py OUT_DIR VARIANT DTYPE SIZE BATCH STEPS BUCKET_MB

RCCL ("nccl") when every rank has a GPU of its own; on a one-GPU box the ranks share device 0 and exchange over gloo
(ABC_DP_SHARED_DEVICE=1), which runs the same bucketed all-reduce between the same hipGraph segments."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import abcnet_amd  # noqa: E402,F401
from abcnet_amd import distributed as D  # noqa: E402
from abcnet_amd.synthetic import synthetic_images, synthetic_targets  # noqa: E402
from abcnet_amd.train import Trainer  # noqa: E402

HEADS = [1, 14, 3, 2, 1, 360, 60, 60]


def _eval_batches(rank, batch, size):
    """two test batches of rank `rank`: images + targets + logits-independent (the meters compare predictions with targets)"""
    return [(synthetic_images(batch, size, seed=300 + 10 * rank + j), synthetic_targets(batch, size // 4, seed=400 + 10 * rank + j)) for j in range(2)]


def main():
    out, variant, dtype, size, batch, steps, bucket_mb = sys.argv[1:8]
    size, batch, steps, bucket_mb = int(size), int(batch), int(steps), float(bucket_mb)
    shared = os.environ.get("ABC_DP_SHARED_DEVICE") == "1"
    rank, world = D.init_process_group(backend="gloo" if shared else "nccl", device=0 if shared else None)
    dev = torch.device("cuda", torch.cuda.current_device())
    if variant == "unet2":
        from abcnet_amd.unet2 import UNet
    else:
        from abcnet_amd.unet import UNet
    model = UNet(1, HEADS, dtype=dtype, dropout_p=0.2)
    model.reset_parameters(seed=1000 + rank)       # DIFFERENT on every rank: the broadcast below must make rank 0's win
    with torch.no_grad():                            # non-trivial BatchNorm buffers too
        model._flat_buf.add_(0.01 * (rank + 1))
    model = model.to(dev)
    D.broadcast_parameters(model._flat, model._flat_buf, counters=model._counters)   # DDP constructor, multi_gpu_train.py:52
    p0 = model._flat.clone()
    tr = Trainer(model, batch, size, size, use_graph=True, bucket_mb=bucket_mb, broadcast_buffers="lazy")
    x = synthetic_images(batch, size, seed=7 + rank)
    tg = synthetic_targets(batch, size // 4, seed=1 + rank)
    tr.load_batch(x.to(dev), [t.to(dev) for t in tg])
    res = {"rank": rank, "world": world, "backend": dist.get_backend(), "p0": p0.cpu(), "n_buckets": len(tr.buckets),
           "n_segments": len(tr._segments), "drop_seed": tr.eng.drop_seed,
           "exchange": tr.reducer.mode, "exchange_fallback": tr.reducer.fallback_reason}
    tr.step()                                        # eager
    torch.cuda.synchronize()
    res["grad_step1"] = model._flat_grad.cpu().clone()      # mean over ranks of the per-rank gradients
    res["loss_step1"] = tr.loss_value()["total"]
    lm = D.reduce_mean(torch.tensor([res["loss_step1"]], dtype=torch.float64, device=dev), world)   # multi_gpu_train.py:116
    res["loss_mean_step1"] = lm.item()
    # checkpointing the way the reference does it (multi_gpu_train.py:318-319): `if rank == 0: save`, between two steps.
    # state_dict() must not be a collective (rank 0 alone would hang or mis-pair with the next all-reduce) ...
    if rank == 0:
        ck0 = tr.state_dict()
        res["buffers_ckpt_rank0_alone"] = torch.cat([ck0["model"][k].reshape(-1).float().cpu() for k in ck0["model"] if "running_" in k])
    else:
        # ... and on another rank it refuses to hand out that rank's own statistics as if they were DDP's
        try:
            tr.state_dict()
            res["other_rank_refused"] = False
        except RuntimeError:
            res["other_rank_refused"] = True
    tr.accumulate_loss()
    for _ in range(steps - 1):                       # captured: one hipGraph per segment between two all-reduce launches
        tr.step()
        tr.accumulate_loss()                         # device-side running sum: no host sync per step
    torch.cuda.synchronize()
    res["loss_mean_async"] = tr.read_loss_mean()     # ONE collective + host sync for all steps (multi_gpu_train.py:114-116)
    res["loss_last"] = tr.loss_value()["total"]
    res["graphs"] = tr._graphs is not None
    res["params"] = model._flat.cpu().clone()
    res["buffers_own"] = model._flat_buf.cpu().clone()      # this rank's own running statistics (its shard's)
    tr.sync_buffers()                                # "lazy": a collective on ALL ranks -- rank 0's buffers arrive here
    ck = tr.state_dict()
    res["buffers_ckpt"] = torch.cat([ck["model"][k].reshape(-1).float().cpu() for k in ck["model"] if "running_" in k])
    res["nbt_ckpt"] = int(ck["model"]["inc1.double_conv.1.num_batches_tracked"])
    res["adam_m"] = tr.opt.m.cpu().clone()
    # the periodic eval pass (train.py:217-433 per rank, multi_gpu_train.py:280-302 across ranks): this rank's two test batches
    ev = tr.evaluate(_eval_batches(rank, batch, size))
    res["eval"] = ev
    tr.step()                                        # the training plan / graphs survive an eval pass
    torch.cuda.synchronize()
    res["loss_after_eval"] = tr.loss_value()["total"]
    torch.save(res, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
