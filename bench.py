#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): training images/sec of unet.py at 384x384, bf16, batch 16 per GPU.

  python bench.py --gpus N --steps K --warmup W
  N>1, either: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (ranks come from the env)
       or:     python bench.py --gpus N ...   -- then THIS process only starts N fresh rank processes (before it has
               touched the GPU; multi_gpu_train.py:30-36's mp.spawn) and exits non-zero unless all N joined.

A step = pack weights + forward + fused loss + backward + gradient all-reduce (N>1) + fused Adam on one
batch of synthetic inputs already resident in HBM (data: Bernoulli ink images + rasterised random
atoms/bonds honouring the reference tensor contract; random-init weights of the reference architecture).
Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     -- the dominant kernel's achieved TFLOP/s (algorithmic flops / HIP-event time, measured live)
  cpu_baseline -- the oracle (CPU restatement of the reference path) timed on this host, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HEADS = [1, 14, 3, 2, 1, 360, 60, 60]
MFMA_PEAK = {"bf16": 2500.0, "fp32": 157.3, "fp8": 5000.0}  # dense TFLOP/s, MI355X_MICROARCH.md (fp8: the block-scaled MFMA)
HBM_PEAK = 8000.0  # GB/s
# SURVEY.md section 8(d) / BASELINE.md section 3: per-layer roofline rate R = 1 / sum_l max(bytes_l / 8 TB/s, flops_l / 2.5 PF) in
# images/s per GPU (bf16, every conv reads its input and writes its output once, train = 3 x forward + loss + Adam bytes),
# with the algorithmic bytes and flops per image it is built from; keyed (mode, variant, size)
SURVEY_ROOFLINE = {
    ("train", "unet", 384): {"R_img_s": 10800.0, "mb_per_img": 527.0, "gflop_per_img": 158.6},
    ("train", "unet2", 384): {"R_img_s": 8400.0, "mb_per_img": 685.0, "gflop_per_img": 223.6},
    ("infer", "unet", 512): {"R_img_s": 19800.0, "mb_per_img": 277.0, "gflop_per_img": 93.98},
}


def cpu_baseline(size, seconds_budget=40.0, variant="unet"):
    """oracle = torch-CPU restatement of the reference train step (bitwise-pinned to the reference import);
    bounded sample: fwd+loss+bwd of batch 4 at the benchmark resolution, median of 5 warm iterations (SURVEY.md section 8d;
    fewer only when the host is so slow that five would take more than `seconds_budget`, and the sample string says how many)"""
    from abcnet_amd.synthetic import synthetic_images, synthetic_targets
    from oracle import loss_oracle
    from oracle import unet_oracle as uo
    B = 4
    x = synthetic_images(B, size, seed=7)
    tg = synthetic_targets(B, size // 4, seed=1)
    sd0 = uo.filled_state(variant, 1, HEADS, seed=0)
    times = []
    t_start = time.time()
    for it in range(6):
        sd = uo.clone_state(sd0, requires_grad=True)
        t0 = time.time()
        preds = uo.forward(variant, sd, x, train=True)
        total, _, _ = loss_oracle.abc_loss(preds, tg, sd["s"])
        total.backward()
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_start > seconds_budget and times:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle fwd+loss+bwd (no optimiser), fp32, batch %d at %dx%d, median of %d warm iterations" % (B, size, size, len(times))}


def cpu_baseline_infer(size, seconds_budget=40.0, variant="unet"):
    """oracle eval forward + oracle NMS (img2smiles2.py:56-79 restated) on the host, batch 4 at the benchmark resolution"""
    from abcnet_amd.synthetic import synthetic_images
    from oracle import nms_oracle
    from oracle import unet_oracle as uo
    B = 4
    x = synthetic_images(B, size, seed=7)
    sd = uo.filled_state(variant, 1, HEADS, seed=0)
    times = []
    t_start = time.time()
    with torch.no_grad():
        for it in range(6):
            t0 = time.time()
            p = uo.forward(variant, sd, x, train=False)
            nms_oracle.nms(p[0], p[4], p[6], p[7])
            dt = time.time() - t0
            if it > 0:
                times.append(dt)
            if time.time() - t_start > seconds_budget and times:
                break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle eval forward + NMS, fp32, batch %d at %dx%d, median of %d warm iterations" % (B, size, size, len(times))}


def pmc_traffic(kernel, mode="train", variant="unet"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/README.md says
    how they were collected and corrected; keyed by mode:variant:label because one instantiation serves different
    shapes in different workloads); None when no pass exists for this kernel label."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    rec = table.get("%s:%s:%s" % (mode, variant, kernel))
    return None if rec is None else rec["bytes_per_launch"]


def dominant_family(prof):
    """(label, record) of the kernel FAMILY with the largest share of the step among those that do arithmetic.  A convolution kernel
    with the producing layer's act_bwd pass in its epilogue ("conv_fast+act_bwd<...>", abc_conv_desc.actbwd_*) is the same tile and
    main loop as the plain instantiation -- one row, priced on the convolution's flops alone (the epilogue's extra work counts against
    it); both labels stay in the breakdown."""
    fam = {}
    for k, v in prof.items():
        f = fam.setdefault(k.replace("+act_bwd<", "<"), {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "labels": []})
        for q in ("calls", "ms", "flops", "bytes"):
            f[q] += v[q]
        f["labels"].append(k)
    dom = max((k for k in fam if fam[k]["flops"] > 0), key=lambda k: fam[k]["ms"])
    return dom, fam[dom]


def other_configs(dev, steps=20, warmup=5):
    """BASELINE.json's configs 3 and 5 (and the headline's step with the device rasteriser in it) through the same harness, AFTER the
    headline's timed region and report (nothing here can perturb it): unet2.py train step b16 @ 384x384 (unet2.py:129-173) and the img2smiles2.py heat-map path b64 @ 512x512 in bf16
    (img2smiles2.py:42-79), each on its own model / Trainer / InferenceRunner, freed afterwards.  One record per config: the same
    wall-clock measurement as the headline (K steps between synchronisations) and the dominant kernel family's fraction of the
    dense bf16 MFMA peak from an instrumented eager pass."""
    import gc
    from abcnet_amd.synthetic import synthetic_images, synthetic_targets
    from abcnet_amd.train import Trainer
    from abcnet_amd.infer import InferenceRunner
    recs = []
    for mode, variant, size, batch in (("train", "unet2", 384, 16), ("infer", "unet", 512, 64), ("train_raster", "unet", 384, 16)):
        t_cfg = time.perf_counter()
        rz = None
        if variant == "unet2":
            from abcnet_amd.unet2 import UNet
        else:
            from abcnet_amd.unet import UNet
        model = UNet(1, HEADS, dtype="bf16")
        model.reset_parameters(seed=1234)
        model = model.to(dev)
        imgs = synthetic_images(batch, size, seed=7)
        if mode == "infer":
            tr = InferenceRunner(model, batch, size, size)
            tr.load_batch(imgs.to(dev))
            workload = "img2smiles2.py heat-map path on unet.py (eval forward + peak NMS), %dx%d, batch %d/GPU" % (size, size, batch)
        else:
            tr = Trainer(model, batch, size, size)
            tr.load_batch(imgs.to(dev), [t.to(dev) for t in synthetic_targets(batch, size // 4, seed=1)])
            workload = variant + ".py train step (pack+fwd+fused loss+bwd+Adam), %dx%d, batch %d/GPU, dropout 0.2" % (size, size, batch)
            if mode == "train_raster":
                # the headline's step with the loader's half of it on the device: the eight target maps rasterised EVERY step from compact
                # records (utils.py:83-228; 30 atoms + 32 bonds per image) instead of resident dense maps, the fused heads pass reading
                # targets by the rasteriser's group flags -- strictly more work in the timed region than the headline
                from abcnet_amd.raster import TargetRasterizer, parse_record
                from abcnet_amd.synthetic import random_annotations
                rz = TargetRasterizer(batch, size // 4, max_atoms=64, max_bonds=64, targets=tr.targets, sparse=True)
                tr.use_sparse_targets(rz)
                rz.load([parse_record(*random_annotations(30, 32, 900 + i, size=size), h=size // 4) for i in range(batch)])
                _step = tr.step

                def step_with_raster(_rz=rz, _s=_step):
                    _rz.run()
                    _s()
                tr.step = step_with_raster
                workload = "unet.py train step + device rasteriser every step (records -> 8 target maps + group flags; sparse target reads), %dx%d, batch %d/GPU, dropout 0.2" % (size, size, batch)
        torch.cuda.synchronize()
        for _ in range(warmup):
            tr.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        dom, r = dominant_family(tr.profile(iters=2))
        ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
        sr = SURVEY_ROOFLINE[("train" if mode == "train_raster" else mode, variant, size)]
        val = batch * steps / el
        recs.append({"workload": workload, "value": round(val, 2), "unit": "images/sec", "ms_per_step": round(1000 * el / steps, 3), "steps": steps,
                     "warmup": warmup, "dtype": "bf16", "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": MFMA_PEAK["bf16"],
                                                                    "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK["bf16"], 4)},
                     "survey_roofline_frac": round(val / sr["R_img_s"], 4), "wall_s": None})
        del tr, model, imgs, rz
        gc.collect()
        torch.cuda.empty_cache()
        recs[-1]["wall_s"] = round(time.perf_counter() - t_cfg, 1)
    return recs


def experiment_knobs():
    """ABC_* environment variables other than this script's own hooks: switches of the library / engine that change
    WHICH kernels run.  A benchmark line measured under one is not the product's: refused unless --allow-knobs, and
    then echoed into the JSON line."""
    own = ("ABC_BENCH_",)
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("ABC_") and not k.startswith(own)}


def launch_ranks(a):
    """--gpus N without a launcher: start the N ranks ourselves.  This parent never initialises HIP
    (torch.cuda.device_count() does not, on this image); the children are fresh interpreters."""
    sys.path.insert(0, ROOT)
    import abcnet_amd  # noqa: F401
    from abcnet_amd import distributed as D
    ndev = torch.cuda.device_count()
    if "ABC_BENCH_DEVICE" not in os.environ and ndev < a.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (a.gpus, ndev))
    codes = D.launch_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus, timeout=float(os.environ.get("ABC_BENCH_TIMEOUT", 3000)))
    if any(c != 0 for c in codes):
        raise SystemExit("bench.py --gpus %d: rank exit codes %s" % (a.gpus, codes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train = the headline metric (configs 2-4); infer = config 5, img2smiles2.py heat-map path (eval forward + NMS)")
    ap.add_argument("--size", type=int, default=None, help="default 384 (train) / 512 (infer)")
    ap.add_argument("--batch", type=int, default=None, help="per GPU; default 16 (train) / 64 (infer)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="fp8 (--mode infer only): the e4m3 form of the BatchNorm-folded graph -- the 128-channel 3x3 convolutions at the output "
                         "resolution on the block-scaled MFMA, the rest bf16 (InferenceRunner(fp8=True)); calibrated on the benchmark batch")
    ap.add_argument("--variant", default="unet", choices=["unet", "unet2"], help="unet.py (headline) or unet2.py (config 3)")
    ap.add_argument("--metrics", action="store_true", help="also update the 17 training meters of train.py:145-215 on the device every step")
    ap.add_argument("--extract", action="store_true", help="(--mode infer) also build the atom / bond candidate lists of "
                    "img2smiles2.py:113-191 on the device inside the step")
    ap.add_argument("--raster", action="store_true", help="rasterise the targets on the device every step from compact records "
                    "(utils.py:83-228 on the GPU) instead of keeping pre-rasterised maps resident")
    ap.add_argument("--dense-targets", action="store_true", help="(--raster) the round-4 form: zero all eight maps and draw every step, the "
                    "fused heads pass reads every target plane (TargetRasterizer(sparse=False)): the A/B of the sparse form, not the default")
    ap.add_argument("--no-logits", action="store_true", help="(train) do not store the eight output maps: nothing reads them without "
                    "--metrics (Trainer(keep_logits=False)); NOT the default -- the headline line stores them as the reference does")
    ap.add_argument("--no-actbwd-epilogue", action="store_true", help="(train) every act_bwd pass as a launch of its own "
                    "(Trainer(actbwd_epilogue=False)): the A/B of the fused data-gradient epilogue, not the default")
    ap.add_argument("--no-fused-convt", action="store_true", help="(train) the ConvTranspose forward as four batched phase convolutions "
                    "(Trainer(fused_convt=False)): the A/B of the one-pass kernel, not the default")
    ap.add_argument("--no-merge-reduce", action="store_true", help="(train) slab reductions and BatchNorm-backward finalisers as launches of "
                    "their own (Trainer(merge_reduce=False)): the A/B of the merged launch, not the default")
    ap.add_argument("--exchange", default=None, choices=["all_reduce", "rs_ag", "direct"],
                    help="(train, N > 1) how a gradient bucket is summed over the ranks (Trainer(exchange=...)); default all_reduce.  "
                         "A mode asked for here that the run fell back from is an ERROR, not a silent substitution")
    ap.add_argument("--bucket-mb", type=float, default=8.0, help="(train, N > 1) gradient bucket size in MB of f32 (Trainer(bucket_mb=...))")
    ap.add_argument("--reserve-cus", type=int, default=0, help="(N > 1) leave this many of the 256 CUs out of the persistent convolution "
                    "grids (2 workgroups per CU x 256 VGPRs fill a SIMD's register file: a communication kernel cannot co-reside with "
                    "them), so that RCCL's kernels start at once instead of behind a draining workgroup (abc_set_reserved_cus)")
    ap.add_argument("--no-nms-in-heads", action="store_true", help="(infer) the round-3 plan: the NMS kernel reads the stored rho / omega maps back "
                    "(InferenceRunner(nms_in_heads=False)): the A/B of the heads kernel's second outputs, not the default")
    ap.add_argument("--decode", action="store_true", help="(infer) store only what the decoder of img2smiles2.py:104-191 reads: |rho| instead of the "
                    "raw rho map, the bond types as their six-way arg max per omega bin (uint8) instead of 360 f32 planes "
                    "(InferenceRunner(decode=True)); candidate lists unchanged")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="the default single-GPU headline run also times BASELINE.json's configs 3 (unet2.py "
                    "train) and 5 (inference heat-map path, bf16) after the headline and reports them as other_configs; this skips them")
    ap.add_argument("--allow-knobs", action="store_true", help="run although ABC_* experiment switches are set (they are echoed)")
    a = ap.parse_args()
    knobs = experiment_knobs()
    if knobs and not a.allow_knobs:
        raise SystemExit("bench.py: experiment switches are set (%s); unset them or pass --allow-knobs" % ", ".join(knobs))
    if a.size is None:
        a.size = 384 if a.mode == "train" else 512
    if a.batch is None:
        a.batch = 16 if a.mode == "train" else 64

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a)
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    # stdout carries ONE JSON line; anything libraries print meanwhile (gloo / RCCL banners) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # (testing hooks: ABC_BENCH_DEVICE pins every rank to one device and ABC_BENCH_BACKEND=gloo replaces RCCL, so that
    #  the N > 1 code path -- bucketed all-reduce between graph segments -- can be exercised on a 1-GPU box)
    if "ABC_BENCH_DEVICE" in os.environ:
        local = int(os.environ["ABC_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import abcnet_amd  # noqa: F401
    from abcnet_amd import distributed as D
    from abcnet_amd.synthetic import synthetic_images, synthetic_targets
    from abcnet_amd.train import Trainer
    if a.variant == "unet2":
        from abcnet_amd.unet2 import UNet
    else:
        from abcnet_amd.unet import UNet

    nodual = bool(os.environ.get("ABC_BENCH_NODUAL"))      # (measurement hook, echoed: the BatchNorm-backward apply as passes of their own)
    if nodual:
        knobs = dict(knobs, ABC_BENCH_NODUAL="1")
    backend = None
    if world > 1:
        D.init_process_group(backend=os.environ.get("ABC_BENCH_BACKEND"), rank=rank, world_size=world, device=local)
        backend = dist.get_backend()
        if dist.get_world_size() != a.gpus:
            raise SystemExit("--gpus %d but %d ranks joined" % (a.gpus, dist.get_world_size()))

    if a.dtype == "fp8" and (a.mode != "infer" or a.variant != "unet"):
        raise SystemExit("--dtype fp8 is the inference graph of unet.py (--mode infer)")
    model = UNet(1, HEADS, dtype="bf16" if a.dtype == "fp8" else a.dtype)
    model.reset_parameters(seed=1234)  # identical random init on every rank (and re-broadcast below)
    model = model.to(dev)
    if world > 1:
        D.broadcast_parameters(model._flat, model._flat_buf, counters=model._counters)
    # each rank owns its shard of the synthetic stream (weak scaling: fixed batch per GPU)
    imgs = synthetic_images(a.batch, a.size, seed=7 + rank)
    if a.mode == "infer":
        from abcnet_amd.infer import InferenceRunner
        tr = InferenceRunner(model, a.batch, a.size, a.size, use_graph=not a.no_graph, extract=a.extract, fp8=(a.dtype == "fp8"),
                             nms_in_heads=not a.no_nms_in_heads, decode=a.decode)
        tr.load_batch(imgs.to(dev))
    else:
        tr = Trainer(model, a.batch, a.size, a.size, use_graph=not a.no_graph, metrics=a.metrics, keep_logits=not a.no_logits,
                     actbwd_epilogue=not a.no_actbwd_epilogue, merge_reduce=not a.no_merge_reduce, bucket_mb=a.bucket_mb,
                     exchange=a.exchange or "all_reduce", reserve_cus=a.reserve_cus, dual_wgrad=not nodual, fused_convt=not a.no_fused_convt)
        if world > 1 and a.exchange is not None and tr.reducer.mode != a.exchange:
            raise SystemExit("bench.py --exchange %s: the reducer runs %r (%s)" % (a.exchange, tr.reducer.mode, tr.reducer.fallback_reason))
        tgs = synthetic_targets(a.batch, a.size // 4, seed=1 + rank)
        tr.load_batch(imgs.to(dev), [t.to(dev) for t in tgs])
        if a.raster:
            # the data path a real loader would use: a few KB of records per batch, maps built where the loss reads them
            from abcnet_amd.raster import TargetRasterizer, parse_record
            from abcnet_amd.synthetic import random_annotations   # seeded annotation strings in the reference's format
            # (sparse: the maps are zeroed once, later steps erase what the previous records drew, and the fused heads pass reads the
            #  target planes only where the rasteriser's group flags say there is something: SURVEY K9)
            rz = TargetRasterizer(a.batch, a.size // 4, max_atoms=64, max_bonds=64, targets=tr.targets, sparse=not a.dense_targets)
            if rz.sparse:
                tr.use_sparse_targets(rz)
            rz.load([parse_record(*random_annotations(30, 32, 900 + 16 * rank + i, size=a.size), h=a.size // 4) for i in range(a.batch)])
            _step = tr.step

            def step_with_raster():
                rz.run()
                _step()
            tr.step = step_with_raster
    torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    loss = tr.loss_value()["total"] if a.mode == "train" else float(tr.atom_mask.sum().item())
    # N > 1: how much of the gradient exchange backward did NOT hide -- three extra steps AFTER the timed region with the launch
    # stream's wait for the communication stream (GradReducer.finish) bracketed by HIP events
    exposed = None
    if a.mode == "train" and world > 1:
        tr.reducer.measure_exposed = True
        for _ in range(3):
            tr.step()
        ex = tr.reducer.exposed_ms()
        tr.reducer.measure_exposed = False
        t = torch.tensor([sum(ex) / max(len(ex), 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed = round(t.item(), 4)
    # what makes a multi-GPU line checkable from the line alone: who joined, on which device, over which backend and
    # exchange, and that the replicas hold identical parameters after the averaged updates (after the clock stopped)
    chk = model._flat.double().sum().item()
    import socket
    me = {"rank": rank, "device": torch.cuda.current_device(), "host": socket.gethostname(), "name": torch.cuda.get_device_name(), "param_checksum": chk}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        if a.mode == "train" and not all(r["param_checksum"] == ranks[0]["param_checksum"] for r in ranks):
            raise SystemExit("replicas diverged: parameter checksums %s" % [r["param_checksum"] for r in ranks])
        if backend == "nccl" and len(set((r["host"], r["device"]) for r in ranks)) != world:     # (local indices repeat across nodes)
            raise SystemExit("ranks share a GPU: (host, device) %s" % [(r["host"], r["device"]) for r in ranks])

    if a.mode == "train":
        metric = "training images/sec (%dx%d, b%d/GPU)" % (a.size, a.size, a.batch)
        workload = a.variant + ".py train step (pack+fwd+fused loss+bwd+allreduce+Adam), %dx%d, batch %d/GPU, dropout 0.2" % (a.size, a.size, a.batch)
    else:
        metric = "inference images/sec, heat-map only (%dx%d, b%d/GPU)" % (a.size, a.size, a.batch)
        workload = "img2smiles2.py heat-map path on %s.py (eval forward + peak NMS), %dx%d, batch %d/GPU" % (a.variant, a.size, a.size, a.batch)
    out = {
        "metric": metric, "value": round(world * a.batch * a.steps / el, 2), "unit": "images/sec",
        "n_gpus": world, "ranks_joined": dist.get_world_size() if world > 1 else 1, "backend": backend,
        "rank_devices": [r["device"] for r in ranks], "replica_checksum": ranks[0]["param_checksum"],
        "exchange": (tr.reducer.mode if (a.mode == "train" and world > 1) else None),
        "exchange_fallback": (tr.reducer.fallback_reason if (a.mode == "train" and world > 1) else None),
        "bucket_mb": (a.bucket_mb if (a.mode == "train" and world > 1) else None), "n_buckets": (len(tr.buckets) if (a.mode == "train" and world > 1) else None),
        "exposed_exchange_ms": exposed, "reserved_cus": a.reserve_cus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000 * el / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": world * a.batch, "parallelism": "dp%d" % world, "graph": not a.no_graph, "device_meters": bool(a.metrics),
                   "device_rasteriser": bool(a.raster), "device_extraction": bool(a.extract),
                   "logits_stored": ("decoder's maps only: |rho|, bond-type arg max (uint8), the other six heads raw" if (a.mode == "infer" and a.decode)
                                     else bool(a.mode != "train" or a.metrics or not a.no_logits)),
                   "actbwd_epilogue": bool(a.mode == "train" and not a.no_actbwd_epilogue), "env_knobs": knobs},
        ("final_loss" if a.mode == "train" else "atom_peaks"): round(loss, 4),
    }

    if a.mode == "infer" and a.variant == "unet":
        # what the reduced-precision graph costs in accuracy, measured on TRAINED weights (tests/test_gpu_trained.py): the e4m3 graph
        # holds the peak decisions, not the maps -- a throughput line in fp8 is not a drop-in for the bf16 one
        try:
            with open(os.path.join(ROOT, "tests", "golden", "trained_deviation.json")) as f:
                m = json.load(f)["measured"]["fp8" if a.dtype == "fp8" else "bf16"]
            out["config"]["accuracy"] = {"source": "tests/golden/trained_deviation.json (device-trained unet.py, b64 @ 512x512, vs the fp32 oracle)",
                                         "worst_head_rms_over_std": round(m["worst_rms_over_std"], 4),
                                         "atom_peaks_missed_spurious_of": [m["atom_peaks"]["missed"], m["atom_peaks"]["spurious"], m["atom_peaks"]["oracle"]],
                                         "omega_mask_rate": round(m["omega_peaks"]["rate"], 4),
                                         "note": ("e4m3: decision-level accuracy only (atom / bond peaks within ~2 % missed + spurious), the maps carry 5-9 % rms noise"
                                                  if a.dtype == "fp8" else "bf16: maps within 1 % rms of the fp32 oracle")}
        except (OSError, KeyError, ValueError):
            pass
    sr = SURVEY_ROOFLINE.get((a.mode, a.variant, a.size))
    if sr is not None and a.dtype in ("bf16", "fp8"):
        per_gpu = out["value"] / world
        out["survey_roofline"] = {"R_img_s_per_gpu": sr["R_img_s"], "frac": round(per_gpu / sr["R_img_s"], 4),
                                  "hbm_frac": round(per_gpu * sr["mb_per_img"] * 1e6 / (HBM_PEAK * 1e9), 4),
                                  "mfma_frac": round(per_gpu * sr["gflop_per_img"] * 1e9 / (MFMA_PEAK["bf16"] * 1e12), 4),
                                  "source": "SURVEY.md section 8(d): measured img/s per GPU / per-layer roofline rate, and the two plain "
                                            "fractions (algorithmic bytes/img x img/s / 8 TB/s, algorithmic flops/img x img/s / 2.5 PF)"}
    # (the instrumented pass and the CPU leg are reported beside the measurement; a failure there must not lose the line)
    try:
        if rank == 0 and not a.no_profile:
            prof = tr.profile(iters=3)
            tot = sum(r["ms"] for r in prof.values())
            dom, r = dominant_family(prof)
            ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
            # (a mixed graph: the dominant kernel is priced against the peak of ITS operand type)
            peak = MFMA_PEAK["fp8" if "<fp8,fp8," in dom else ("bf16" if a.dtype == "fp8" else a.dtype)]
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": pmc_traffic(dom, "infer8" if (a.mode == "infer" and a.dtype == "fp8") else a.mode, a.variant),
                               "algorithmic_mb_per_launch": round(r["bytes"] / r["calls"] / 1e6, 2),
                               "launches_per_step": r["calls"], "avg_launch_us": round(1000 * r["ms"] / r["calls"], 2),
                               "algorithmic_gflop_per_launch": round(r["flops"] / r["calls"] / 1e9, 3)}
            if len(r["labels"]) > 1:
                out["roofline"]["labels"] = {k: {"launches_per_step": prof[k]["calls"], "avg_launch_us": round(1000 * prof[k]["ms"] / prof[k]["calls"], 2),
                                                 "achieved": round(prof[k]["flops"] / (prof[k]["ms"] * 1e-3) / 1e12, 1)} for k in r["labels"]}
            out["kernel_breakdown_ms"] = {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("ABC_BENCH_TOP", 12))]}
            if os.environ.get("ABC_BENCH_TOP"):
                out["kernel_calls"] = {k: v["calls"] for k, v in prof.items()}
                # per launch: algorithmic GFLOP and MB of every kernel label (the plan's own accounting; profiles/ joins it
                # with the PMC bytes)
                out["kernel_operand_mb"] = {k: round(v.get("operand_bytes", v["bytes"]) / max(v["calls"], 1) / 1e6, 2) for k, v in prof.items()}
                out["kernel_algorithmic"] = {k: [round(v["flops"] / max(v["calls"], 1) / 1e9, 3), round(v["bytes"] / max(v["calls"], 1) / 1e6, 2)]
                                             for k, v in prof.items()}
            out["eager_step_ms_sum_of_kernels"] = round(tot, 3)
            flops_step = sum(v["flops"] for v in prof.values())
            bytes_step = sum(v["bytes"] for v in prof.values())
            out["whole_step"] = {"algorithmic_tflop": round(flops_step / 1e12, 3), "algorithmic_gb": round(bytes_step / 1e9, 3),
                                 "mfma_frac": round(flops_step / (el / a.steps) / 1e12 / MFMA_PEAK["bf16" if a.dtype == "fp8" else a.dtype], 4),
                                 "hbm_frac": round(bytes_step / (el / a.steps) / 1e9 / HBM_PEAK, 4)}
        headline = (a.mode, a.variant, a.size, a.batch, a.dtype) == ("train", "unet", 384, 16, "bf16") and not (a.metrics or a.raster or a.no_graph)
        if rank == 0 and world == 1 and headline and not a.no_other_configs:
            # (the headline's objects are released first: the other configs start from the state a fresh process would give them)
            del tr, model
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(dev)
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline if a.mode == "train" else cpu_baseline_infer)(a.size, variant=a.variant)

    except Exception as e:  # noqa: BLE001
        if rank == 0:
            out["report_error"] = "%s: %s" % (type(e).__name__, e)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
