"""Training-step harness: the hot loop of /root/reference/src/train.py:83-141 (and its DDP
form, multi_gpu_train.py:72-119) on the HIP kernels.

One step = pack weights -> forward -> fused activation+loss+dlogits -> backward ->
(bucketed gradient all-reduce over RCCL, overlapped with backward) -> fused Adam.
No host synchronisation inside the step; the loss value stays on the device until asked for.
The static launch plan is replayed from hipGraphs: one graph per segment between two
all-reduce launch points (a single graph when world size is 1).
"""
from __future__ import annotations

import os

import torch

from . import _lib as L
from . import distributed as D
from .ops import FusedAdam, FusedHeadsLoss, FusedLoss, FusedMetrics


class Trainer:
    def __init__(self, model, batch, height, width, lr=2.5e-4, weight_decay=1e-8, use_graph=True, bucket_mb=8.0,
                 process_group=None, device=None, metrics=False, broadcast_buffers="lazy", fused_heads=True, keep_logits=True,
                 batched_heads=True, exchange="all_reduce", force_exchange=False, guards=False, actbwd_epilogue=True, merge_reduce=True,
                 reserve_cus=None, dual_wgrad=True, fused_convt=True):
        """broadcast_buffers: how DDP's per-forward buffer broadcast (multi_gpu_train.py:52, broadcast_buffers=True) is
        mirrored when world > 1 -- "step": rank 0's BatchNorm buffers are broadcast at the start of every step, literally
        as DDP does; "lazy" (default): when sync_buffers() is called -- evaluate() calls it; before a checkpoint of a rank
        other than 0 or an eval forward outside evaluate() the caller does (state_dict() itself performs NO collective:
        rank 0 saves alone as the reference does, any other rank raises until sync_buffers() has run since the last step)
        -- which is observably the same because train-mode arithmetic never reads the buffers; False: never (each rank keeps
        its own shard's statistics).
        keep_logits=False (fused heads only, ignored with metrics=True): the eight output maps are not stored by the step
        (`eng.logits` is stale) -- the loss and every gradient are unchanged.
        batched_heads=False: one launch per head (the plain form the batched / merged heads launches are tested against).
        actbwd_epilogue=False: every act_bwd pass as a launch of its own (Engine(actbwd_epilogue=...)).
        merge_reduce=False: every slab reduction and BatchNorm-backward finaliser as a launch of its own (Engine(merge_reduce=...)).
        exchange: how a gradient bucket is summed over the ranks -- "all_reduce" (default: one RCCL all-reduce per bucket,
        RCCL picks the algorithm), "rs_ag" (reduce-scatter + all-gather in place on the arena), "direct" (all-to-all + local
        sum + all-gather); see distributed.GradReducer.  The two alternatives are self-checked against all_reduce on scratch
        tensors of the plan's real bucket sizes when the reducer is built and fall back (reducer.fallback_reason) -- neither
        has run at world > 1 on RCCL hardware yet, which is why they are not the default.
        reserve_cus: leave this many CUs out of the persistent convolution grids so that RCCL's kernels can start beside
        them (engine.set_reserved_cus: PROCESS-wide and baked into every plan's grids and buffer sizes, so it is applied before
        the plan is built and REFUSED when plans built under another value are alive; None (default) = leave the process's
        current value alone -- 0 unless somebody set it).
        dual_wgrad=False: the BatchNorm-backward apply as passes of their own (Engine(dual_wgrad=...): the A/B of that fusion).
        fused_convt=False: the ConvTranspose forward as four batched phase convolutions (Engine(fused_convt=...): the A/B of convt_fused.hip).
        force_exchange: segment the plan and run the bucket exchanges although the group has one rank (testing RCCL's launch
        mechanics between graph segments on a one-GPU box)."""
        if not torch.cuda.is_available():
            raise L.AbcNetHipError("Trainer needs an MI355X; abcnet_amd has no CPU fallback")
        self.model = model
        dev = torch.device(device or next(model.parameters()).device)
        if dev.type != "cuda":
            raise L.AbcNetHipError("Trainer: the model must live on a GPU (got %s); abcnet_amd has no CPU fallback" % dev)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        dist_on = torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size(process_group) if dist_on else 1
        self.rank = torch.distributed.get_rank(process_group) if dist_on else 0
        self.group = process_group
        if broadcast_buffers not in ("step", "lazy", False):
            raise ValueError("broadcast_buffers must be 'step', 'lazy' or False")
        self.broadcast_buffers = broadcast_buffers if self.world > 1 else False
        # every rank draws its OWN dropout masks, as the unseeded processes of multi_gpu_train.py:36 do (rank 0 keeps the
        # single-process stream)
        model.dropout_seed = D.rank_dropout_seed(model.dropout_seed_base, self.rank)
        model.train()
        from .engine import set_reserved_cus
        self.reserve_cus = L.load().abc_get_reserved_cus() if reserve_cus is None else set_reserved_cus(reserve_cus)
        with torch.cuda.device(dev):
            x0 = torch.zeros((batch, model.n_channels, height, width), device=dev)
            # (fused_heads: the heads' 1x1 convolutions, the loss and the way back as one pass where the engine can -- bf16)
            # (guards: the debug plan -- every buffer between guard bands, Engine.check_guards())
            self.eng = model._engine_for(x0, True, fused_heads=fused_heads, batched_heads=batched_heads, guards=guards,
                                         actbwd_epilogue=actbwd_epilogue, merge_reduce=merge_reduce, dual_wgrad=dual_wgrad, fused_convt=fused_convt)
        eng = self.eng
        h, w = eng.h, eng.w
        B = batch
        shapes = [(B, 1, h, w), (B, 14, h, w), (B, 3, h, w), (B, 2, h, w), (B, 1, h, w), (B, 6, 60, h, w), (B, 60, h, w), (B, 60, h, w)]
        dts = [torch.float32] * 6 + [torch.float64] * 2
        self.targets = [torch.zeros(s, dtype=dt, device=dev) for s, dt in zip(shapes, dts)]
        off_s, _ = model._lay_p["s"]
        extra = {"keep_logits": bool(keep_logits or metrics)} if eng.hf is not None else {}
        self.loss = (FusedHeadsLoss if eng.hf is not None else FusedLoss)(
            eng, self.targets, model._flat.data.data_ptr() + 4 * off_s, model._flat_grad.data_ptr() + 4 * off_s, grad_scale=1.0 / self.world, **extra)
        # train.py:145-215: the 17 meters, updated every step on the device (no host round trips); off by default
        self.metrics = FusedMetrics(eng.logits, self.targets) if metrics else None
        self.lr, self.wd = lr, weight_decay
        self.opt = FusedAdam(model._flat.data, model._flat_grad, lr=lr, weight_decay=weight_decay)
        # ---- gradient buckets: which backward op finalises which parameter
        names = list(model._lay_p.keys())
        sizes = [model._lay_p[n][1] for n in names]
        ready = {n: -1 for n in names}
        for j, (_fn, _ref, _what, writes, _meta) in enumerate(eng.bwd_ops):
            for n in writes:
                ready[n] = max(ready[n], j)
        rd = [ready[n] for n in names]
        self.buckets = D.plan_buckets(rd, sizes, int(bucket_mb * (1 << 20) / 4))
        store = getattr(model, "_grad_store", None)
        align = 128 * self.world
        self.exchange_on = self.world > 1 or (bool(force_exchange) and dist_on)
        if self.exchange_on and exchange != "all_reduce" and store is not None and store.numel() % align == 0 \
                and store.data_ptr() == model._flat_grad.data_ptr():
            # bucket boundaries on multiples of 128 x world elements of the padded store: every bucket splits evenly
            self.buckets = D.align_buckets(self.buckets, sizes, rd, align, store.numel())
            self.reducer = D.GradReducer(store, self.buckets, process_group, mode=exchange, force=force_exchange)
        else:
            self.reducer = D.GradReducer(model._flat_grad, self.buckets, process_group, mode="all_reduce", force=force_exchange)
        self.use_graph = use_graph
        self._graphs = None
        self._segments = self._plan_segments()
        self.steps = 0
        self._buffers_synced_at = 0   # value of self.steps when sync_buffers() last ran (lazy broadcast_buffers)
        # running mean of the loss over ranks and steps WITHOUT a per-step host sync (multi_gpu_train.py:114-116 calls
        # barrier() + reduce_mean(loss) + .item() every step): the device accumulates, read_loss_mean() reads every N steps
        self._loss_acc = torch.zeros(2, dtype=torch.float64, device=dev)   # [sum of totals, count]

    # ------------------------------------------------------------------ data
    def load_batch(self, imgs, targets=None):
        """copy a batch into the static input buffers (what a loader's H2D copy would target directly); targets=None: the images
        only (the targets come from a rasteriser that draws into self.targets)"""
        self.eng.img.copy_(imgs.reshape(self.eng.img.shape), non_blocking=True)
        if targets is None:
            return
        if getattr(self, "_rasterizer", None) is not None:
            raise L.AbcNetHipError("load_batch(dense targets) under use_sparse_targets(): the rasteriser's group flags would no longer "
                                   "describe the maps; load records into the rasteriser, or call use_sparse_targets(None) first")
        for dst, src in zip(self.targets, targets):
            dst.copy_(src, non_blocking=True)

    def use_sparse_targets(self, rasterizer):
        """rasterizer: a TargetRasterizer(sparse=True, targets=self.targets) -- the fused heads pass then reads only the target planes of
        the 32-pixel groups the rasteriser drew into (FusedHeadsLoss.use_target_flags); None: back to reading every plane.  The loss
        and every gradient are unchanged bit for bit."""
        if not hasattr(self.loss, "use_target_flags"):
            raise L.AbcNetHipError("sparse targets need the fused heads pass (bf16, fused_heads=True)")
        if rasterizer is None:
            self.loss.use_target_flags(None)
            self._rasterizer = None
        else:
            if not getattr(rasterizer, "sparse", False) or any(a.data_ptr() != b.data_ptr() for a, b in zip(rasterizer.targets, self.targets)):
                raise L.AbcNetHipError("use_sparse_targets: a TargetRasterizer(sparse=True) over this Trainer's own target tensors")
            self.loss.use_target_flags(rasterizer.group_flags)
            self._rasterizer = rasterizer
        self._graphs = None

    def reset_optimizer(self, lr):
        """train.py:84-85: a NEW Adam (moments reset) at the learning-rate drop"""
        self.lr = lr
        self.opt = FusedAdam(self.model._flat.data, self.model._flat_grad, lr=lr, weight_decay=self.wd)
        self._graphs = None

    # ------------------------------------------------------------------ plan
    def _plan_segments(self):
        """list of segments; a segment = list of callables(stream); after segment k the buckets in
        self._seg_buckets[k] are complete and their all-reduce is launched"""
        eng = self.eng
        pre = [eng.run_pack, eng.run_forward, self.loss.run]
        if self.metrics is not None:
            pre.append(self.metrics.run)
        cut = sorted(set(b[2] for b in self.buckets)) if self.exchange_on else []
        segs, seg_b = [], []
        cur = list(pre)
        start = 0
        ops = eng.bwd_ops
        early = [b for b in self.buckets if b[2] < 0]
        for c in [c for c in cut if c >= 0]:
            chunk = ops[start:c + 1]
            cur.append(lambda st, chunk=chunk: eng._run(chunk, st))
            segs.append(cur)
            seg_b.append([b for b in self.buckets if b[2] == c])
            cur = []
            start = c + 1
        tail = ops[start:]
        if tail:
            cur.append(lambda st, chunk=tail: eng._run(chunk, st))
        segs.append(cur)
        seg_b.append([])
        self._early = early
        self._seg_buckets = seg_b
        return segs

    def _run_segment(self, k):
        st = torch.cuda.current_stream(self.dev).cuda_stream
        for fn in self._segments[k]:
            fn(st)

    def _capture(self):
        graphs = []
        for k in range(len(self._segments)):
            if not self._segments[k]:
                graphs.append(None)
                continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._run_segment(k)
            graphs.append(g)
        gopt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gopt, capture_error_mode="thread_local"):
            self.opt.step()
        self._graphs = (graphs, gopt)

    # ------------------------------------------------------------------ step
    def sync_buffers(self):
        """rank 0's BatchNorm running statistics / num_batches_tracked to every rank (DDP broadcast_buffers).
        A COLLECTIVE: every rank of the group has to call it."""
        if self.world > 1:
            with torch.cuda.device(self.dev):
                D.broadcast_buffers(self.model._flat_buf, self.model._counters, group=self.group)
        self._buffers_synced_at = self.steps

    def accumulate_loss(self):
        """add this step's total loss to the device-side running sum (one tiny kernel, no host sync, graph-safe)"""
        self._loss_acc[0] += self.loss.total_device()
        self._loss_acc[1] += 1

    def read_loss_mean(self, reset=True):
        """mean of the accumulated step losses over steps AND ranks (multi_gpu_train.py:116's reduce_mean, taken every N
        steps instead of every step).  A COLLECTIVE when world > 1; one host sync."""
        acc = self._loss_acc.clone()
        if self.world > 1:
            torch.distributed.all_reduce(acc, op=torch.distributed.ReduceOp.SUM, group=self.group)
        s, n = acc.tolist()
        if reset:
            self._loss_acc.zero_()
        return s / max(n, 1.0)

    def step(self):
        """one optimisation step on the batch currently in the static buffers"""
        with torch.cuda.device(self.dev):
            if self.broadcast_buffers == "step":
                D.broadcast_buffers(self.model._flat_buf, self.model._counters, group=self.group)
            self._step()

    def _step(self):
        if self.use_graph and self._graphs is None and self.steps >= 1:
            torch.cuda.synchronize()
            self._capture()
        graphs = self._graphs
        for k in range(len(self._segments)):
            if graphs is not None:
                if graphs[0][k] is not None:
                    graphs[0][k].replay()
            else:
                self._run_segment(k)
            if k == 0:
                for lo, hi, _ in self._early:
                    self.reducer.bucket_ready(lo, hi)
            for lo, hi, _ in self._seg_buckets[k]:
                self.reducer.bucket_ready(lo, hi)
        self.reducer.finish()
        if graphs is not None:
            graphs[1].replay()
        else:
            self.opt.step()
        self.steps += 1

    def loss_value(self):
        return self.loss.result()

    # ------------------------------------------------------------------ periodic evaluation
    def evaluate(self, batches, fold_bn=None):
        """The eval pass the reference runs every 100 steps (train.py:217-433; multi_gpu_train.py:121-316 per rank):
        `model.eval()` forward over this rank's test batches with the 17 meters of train.py:335-393 (the same arithmetic as the
        training meters, csrc/metrics.hip) accumulated on the device, then the cross-rank reduction of multi_gpu_train.py:
        280-302 as one collective (distributed.reduce_meters).  batches: iterable of (imgs [B,C,H,W], the 8 target maps) of the
        Trainer's batch shape, host or device tensors.  The training plan, its graphs and the optimiser are untouched; the eval
        engine reads the CURRENT weights and -- as every DDP rank does, broadcast_buffers -- rank 0's running statistics
        (sync_buffers(): a COLLECTIVE when world > 1, so every rank has to call evaluate(), as every rank runs the reference's
        eval loop).  One host sync at the end.  Returns {meter: {"sum", "count", "avg": over all ranks' images,
        "rank_mean": the reference's mean of per-rank averages}}."""
        from .ops import METER_NAMES
        model, eng_t = self.model, self.eng
        with torch.cuda.device(self.dev):
            if self.world > 1 and self.broadcast_buffers:
                self.sync_buffers()
            if getattr(self, "_eval", None) is None:
                x0 = torch.zeros_like(eng_t.img).reshape(eng_t.B, model.n_channels, eng_t.H, eng_t.W)
                fold = (model.VARIANT == "unet") if fold_bn is None else bool(fold_bn)
                eng = model._engine_for(x0, False, fold_bn=fold)
                tg = [torch.zeros_like(t) for t in self.targets]
                self._eval = (eng, tg, FusedMetrics(eng.logits, tg))
            eng, tg, meters = self._eval
            st = torch.cuda.current_stream().cuda_stream
            meters.reset()
            eng.run_pack(st)                    # the weights moved since the last call: re-pack (and re-fold BatchNorm) once
            for imgs, targets in batches:
                eng.img.copy_(imgs.reshape(eng.img.shape), non_blocking=True)
                for dst, src in zip(tg, targets):
                    dst.copy_(src, non_blocking=True)
                eng.run_forward(st)
                meters.run(st)
            glob, rank_mean = D.reduce_meters(meters.totals, self.group)
            glob, rank_mean = glob.cpu(), rank_mean.cpu()
        out = {}
        for i, n in enumerate(METER_NAMES):
            s, c = glob[i, 0].item(), glob[i, 1].item()
            out[n] = {"sum": s, "count": c, "avg": s / c if c else float("nan"), "rank_mean": rank_mean[i].item()}
        return out

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """everything a bit-exact resume needs: the model in the REFERENCE's state_dict layout (loads into
        /root/reference/src/unet.py as is, train.py:435), plus what the reference does not save -- the Adam moments and
        step counter, the learning rate, the dropout step counter and the running meters"""
        # NO collective in here: the reference saves on one rank only (`if rank == 0: torch.save(...)`,
        # multi_gpu_train.py:318-319), and rank 0's BatchNorm buffers ARE the ones DDP's broadcast_buffers would have
        # handed to everybody.  Any other rank holds its own shard's statistics until sync_buffers() -- a collective, to be
        # called on ALL ranks -- has run since the last step; its checkpoint would silently differ from DDP's, so it raises.
        if self.broadcast_buffers and self.rank != 0 and self._buffers_synced_at != self.steps:
            raise RuntimeError("Trainer.state_dict() on rank %d: this rank's BatchNorm buffers are its own shard's (DDP would have "
                               "broadcast rank 0's).  Save from rank 0 only, or call sync_buffers() on ALL ranks first." % self.rank)
        sd = {"model": {k: v.clone() for k, v in self.model.state_dict().items()},
              "adam_m": self.opt.m.clone(), "adam_v": self.opt.v.clone(), "adam_step": self.opt.step_t.clone(),
              "lr": self.lr, "weight_decay": self.wd, "steps": self.steps, "drop_salt": self.eng.drop_salt.clone(),
              "drop_seed": self.eng.drop_seed}
        if self.metrics is not None:
            sd["meters"] = self.metrics.totals.clone()
        return sd

    def load_state_dict(self, sd):
        self.model.load_state_dict(sd["model"])
        if sd["lr"] != self.lr or sd["weight_decay"] != self.wd:
            self.wd = sd["weight_decay"]
            self.reset_optimizer(sd["lr"])
        self.opt.m.copy_(sd["adam_m"])
        self.opt.v.copy_(sd["adam_v"])
        self.opt.step_t.copy_(sd["adam_step"])
        # the dropout stream continues where it stopped; the seed itself is baked into the launch descriptors, so a
        # checkpoint written under another seed (another rank's) is continued through the device-side salt
        delta = (int(sd.get("drop_seed", self.eng.drop_seed)) - self.eng.drop_seed) & 0xFFFFFFFF
        salt = (int(sd["drop_salt"].item()) + delta) & 0xFFFFFFFF
        self.eng.drop_salt.fill_(salt - (1 << 32) if salt >= (1 << 31) else salt)
        self.steps = max(self.steps, 1) if self._graphs is not None else self.steps
        # a checkpoint's buffers are the ones DDP would hold (rank 0's): a resume on all ranks leaves every rank in sync
        self._buffers_synced_at = self.steps
        if self.metrics is not None and "meters" in sd:
            self.metrics.totals.copy_(sd["meters"])

    # ------------------------------------------------------------------ measurement
    def profile(self, iters=3):
        """Eager (no graph) steps with a HIP event pair around EVERY launch, recorded on the stream the
        kernels are launched on.  Returns {kernel: {calls, ms, flops, bytes}} per step (averaged)."""
        eng = self.eng
        torch.cuda.set_device(self.dev)
        stream = torch.cuda.current_stream(self.dev)
        st = stream.cuda_stream
        groups = [eng.pack_ops, eng.fwd_ops, None, eng.bwd_ops]
        acc = {}
        by_op = os.environ.get("ABC_BENCH_OPS")  # (diagnostics: one row per launch label instead of per kernel)
        for _ in range(iters):
            marks = []
            for ops in groups:
                if ops is None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    self.loss.run(st)
                    e1.record(stream)
                    if eng.hf is not None:   # features read twice, targets, logits + blocked d(logits) + g written
                        marks.append(("heads_fused+finalize", 2.0 * 2 * eng.B * eng.h * eng.w * 128 * 501,
                                      float(eng.B * eng.h * eng.w * (1024 * 2 * 3 + 501 * 4 + 768 * 2 + 381 * 4 + 120 * 8)), e0, e1))
                    else:
                        marks.append(("loss_fwd_bwd+finalize", 0.0, float(eng.B * eng.h * eng.w * (501 * 4 * 2 + 381 * 4 + 120 * 8)), e0, e1))
                    continue
                for fn, ref, what, _w, meta in ops:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    rc = fn(ref, st)
                    e1.record(stream)
                    if rc != 0:
                        L.check(rc, what)
                    marks.append((meta["kernel"] + " | " + what if by_op else meta["kernel"], meta["flops"], meta["bytes"], e0, e1,
                                  meta.get("operand_bytes", meta["bytes"])))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            self.opt.step(st)
            e1.record(stream)
            marks.append(("adam", 0.0, float(self.model._flat.numel() * 28), e0, e1))
            torch.cuda.synchronize()
            for mk in marks:
                k, fl, by, a, b = mk[:5]
                r = acc.setdefault(k, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "operand_bytes": 0.0})
                r["calls"] += 1
                r["ms"] += a.elapsed_time(b)
                r["flops"] += fl
                r["bytes"] += by
                r["operand_bytes"] += mk[5] if len(mk) > 5 else by
        for r in acc.values():
            for f in ("calls", "ms", "flops", "bytes", "operand_bytes"):
                r[f] /= iters
        self.steps += iters
        return acc
