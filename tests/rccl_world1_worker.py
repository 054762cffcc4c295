"""RCCL mechanics on a ONE-GPU box (started by tests/test_gpu_00_dataparallel.py as a fresh process): backend "nccl" with
world size 1 and the exchange forced on (Trainer(force_exchange=True)), so that every bucket really goes through RCCL on
the communication stream between the hipGraph segments of the backward plan -- reduce-scatter + all-gather, all-to-all +
sum + all-gather, and plain all-reduce.  A sum over one rank is the identity: parameters after 4 steps (3 of them graph
replays) must equal, bit for bit, those of the plain single-process Trainer.

    python rccl_world1_worker.py OUT_FILE SIZE BATCH STEPS
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import abcnet_amd  # noqa: E402,F401
from abcnet_amd import distributed as D  # noqa: E402
from abcnet_amd.synthetic import synthetic_images, synthetic_targets  # noqa: E402
from abcnet_amd.train import Trainer  # noqa: E402
from abcnet_amd.unet import UNet  # noqa: E402

HEADS = [1, 14, 3, 2, 1, 360, 60, 60]


def main():
    out, size, batch, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = D.init_process_group(backend="nccl")
    assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device())
    x = synthetic_images(batch, size, seed=7).to(dev)
    tg = [t.to(dev) for t in synthetic_targets(batch, size // 4, seed=1)]
    res = {"backend": dist.get_backend()}

    def run(x=x, tg=tg, batch=batch, size=size, timed=5, bucket_mb=4.0, **kw):
        m = UNet(1, HEADS, dtype="bf16")
        m.reset_parameters(seed=1)
        m = m.to(dev)
        tr = Trainer(m, batch, size, size, use_graph=True, bucket_mb=bucket_mb, **kw)
        tr.load_batch(x, tg)
        for _ in range(steps):
            tr.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(timed):
            tr.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / timed * 1e3
        return m._flat.clone(), tr, ms

    p_plain, tr0, ms0 = run()
    res["plain"] = {"segments": len(tr0._segments), "ms_per_step": ms0}
    for mode in D.GradReducer.MODES:
        p, tr, ms = run(exchange=mode, force_exchange=True)
        res[mode] = {"used": tr.reducer.mode, "fallback": tr.reducer.fallback_reason, "buckets": len(tr.buckets),
                     "segments": len(tr._segments), "graphs": tr._graphs is not None, "ms_per_step": ms,
                     "identical_to_plain": bool(torch.equal(p, p_plain))}
    # what the exchange machinery costs at the BENCHMARK workload (b16 @ 384 x 384, 8 MB buckets) when there is nothing to exchange:
    # graph segmentation, RCCL launches on the communication stream, event joins
    del tr0, tr
    torch.cuda.empty_cache()
    xb = synthetic_images(16, 384, seed=7).to(dev)
    tgb = [t.to(dev) for t in synthetic_targets(16, 96, seed=1)]
    bench = {}
    for mode in (None,) + tuple(D.GradReducer.MODES):
        kw = {} if mode is None else {"exchange": mode, "force_exchange": True}
        _p, trb, ms = run(x=xb, tg=tgb, batch=16, size=384, timed=20, bucket_mb=8.0, **kw)
        bench["plain" if mode is None else mode] = {"ms_per_step": ms, "segments": len(trb._segments), "buckets": len(trb.buckets)}
        del trb, _p
        torch.cuda.empty_cache()
    res["bench_b16_384"] = bench
    with open(out, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
