"""Data-parallel scaffolding: one process per GPU, RCCL (torch.distributed backend "nccl" on
ROCm) over xGMI.  Mirrors the call surface of /root/reference/src/multi_gpu_train.py:24-75
(init_process_group, DistributedSampler partitioning, reduce_mean, gradient averaging) without
DistributedDataParallel: the model keeps every gradient in ONE flat f32 arena, so the exchange is
a handful of large contiguous all-reduces issued on a side stream as soon as the backward plan
has produced a bucket (reverse arena order: heads first), overlapping with the remaining
backward kernels.  xGMI is point-to-point, so few large messages beat many small ones.

Everything here is device-agnostic (works on CPU tensors over gloo), which is how the N>1 path
is tested without GPUs.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time

# the host driver of this pool supports dmabuf IPC only; ROCr reads this when the process first touches the GPU, so it
# has to be in the environment BEFORE any torch.cuda call (it is exported on the boxes already; rank_env() below hands
# it to child ranks explicitly)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def init_process_group(backend=None, rank=None, world_size=None, master_addr="127.0.0.1", master_port=None, device=None):
    """multi_gpu_train.py:44-48 (init_process_group + torch.cuda.set_device(local_rank)), with env-driven defaults
    (torchrun / launch_ranks) instead of a fixed port.  Selects this rank's GPU FIRST: every launch of the library goes
    to the current device's stream, and RCCL wants one distinct device per rank.
    device: explicit device index (testing: several ranks on one GPU over gloo); default LOCAL_RANK."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
    os.environ.setdefault("MASTER_ADDR", master_addr)
    os.environ.setdefault("MASTER_PORT", str(master_port or 29512))
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", rank % max(ndev, 1)))
        if not 0 <= device < ndev:
            raise RuntimeError("rank %d wants GPU %d but only %d are visible" % (rank, device, ndev))
        torch.cuda.set_device(device)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    return rank, world_size


def rank_dropout_seed(base, rank):
    """dropout hash seed of data-parallel rank `rank` (rank 0 keeps `base`)"""
    return (base ^ ((rank * 0x632BE5AB) & 0xFFFFFFFF)) & 0xFFFFFFFF


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    """environment of child rank `rank` (what torchrun would export), rendezvous on 127.0.0.1"""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def launch_ranks(argv, world, timeout=None, env=None, rank0_stdout=None):
    """multi_gpu_train.py:30-36 (`mp.spawn(main_worker, nprocs=device_count)`) as FRESH child processes: one
    `python argv...` per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set.  To be called from a parent that has
    NOT touched the GPU (a process that has initialised HIP must never fork/exec ranks).  Rank 0's stdout is the
    parent's (or `rank0_stdout`), the other ranks' stdout goes to stderr.  Returns the list of exit codes; when one rank
    fails or the timeout passes, the remaining ranks are terminated (exact PIDs) and their code is reported as -15/-9."""
    port = free_port()
    procs = []
    for r in range(world):
        out = (rank0_stdout if rank0_stdout is not None else None) if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=rank_env(r, world, port, env), stdout=out))
    t0 = time.time()
    codes = [None] * world
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = any(c not in (None, 0) for c in codes)
        late = timeout is not None and time.time() - t0 > timeout
        if failed or late:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    return codes


def sampler_indices(n, world, rank, epoch, seed=0, shuffle=True):
    """torch.utils.data.DistributedSampler semantics (multi_gpu_train.py:62-63,72-73): permutation seeded
    by seed+epoch, padded to a multiple of world by wrapping, rank takes indices rank::world."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = -(-n // world) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // len(idx))))[:pad]
    return idx[rank:total:world]


def reduce_mean(t, world):
    """multi_gpu_train.py:24-28"""
    r = t.clone()
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    return r / world


def reduce_meters(totals, group=None):
    """The cross-rank half of the periodic eval pass (multi_gpu_train.py:280-302: `barrier()` + eleven `reduce_mean` calls on
    eleven scalars) as ONE all-reduce.  totals: [n, 2] float64, this rank's (sum, count) per meter (AverageMeter's fields,
    meter.py:12-16).  Returns (global_totals [n, 2] = sums over the ranks -- the meter over ALL test images --, rank_mean [n] =
    mean over the ranks of each rank's own average, which is what the reference prints: multi_gpu_train.py:280-302 averages the
    per-rank averages, equal to the global one only when every rank's counts are equal; ranks whose count is 0 are left out).
    Device-agnostic (CPU tensors over gloo in the tests); a COLLECTIVE when world > 1."""
    t = totals.to(torch.float64)
    valid = (t[:, 1] != 0).to(torch.float64)
    avg = torch.where(t[:, 1] != 0, t[:, 0] / torch.where(t[:, 1] != 0, t[:, 1], torch.ones_like(t[:, 1])), torch.zeros_like(t[:, 0]))
    n = t.shape[0]
    pack = torch.cat([t.reshape(-1), avg, valid])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=group)
    glob = pack[:2 * n].reshape(n, 2)
    nv = pack[3 * n:]
    rank_mean = torch.where(nv > 0, pack[2 * n:3 * n] / torch.where(nv > 0, nv, torch.ones_like(nv)), torch.full_like(nv, float("nan")))
    return glob, rank_mean


def plan_buckets(ready, sizes, bucket_elems):
    """Split the flat arena [0, sum(sizes)) into contiguous buckets, walking from the END of the arena
    (gradients of the last-registered tensors, the heads, are produced first by backward).

    ready[i]  = index of the backward op after which tensor i's gradient is final
    sizes[i]  = numel of tensor i (arena order)
    returns   = list of (lo, hi, ready_op) sorted by ready_op"""
    n = len(sizes)
    offs = [0] * (n + 1)
    for i, s in enumerate(sizes):
        offs[i + 1] = offs[i] + s
    buckets = []
    hi_i = n
    while hi_i > 0:
        lo_i = hi_i
        acc = 0
        while lo_i > 0 and (acc < bucket_elems):
            lo_i -= 1
            acc += sizes[lo_i]
        buckets.append((offs[lo_i], offs[hi_i], max(ready[lo_i:hi_i])))
        hi_i = lo_i
    buckets.sort(key=lambda b: b[2])
    return buckets


def align_buckets(buckets, sizes, ready, align, padded_total):
    """Move the bucket boundaries of plan_buckets() up to multiples of `align` elements (the last one to `padded_total`, the
    size of the padded gradient store), so that every bucket splits evenly over the ranks of a reduce-scatter.  A boundary
    need not coincide with a tensor boundary: a bucket is complete once every tensor it OVERLAPS is (ready = their max)."""
    n = len(sizes)
    offs = [0] * (n + 1)
    for i, s in enumerate(sizes):
        offs[i + 1] = offs[i] + s
    cuts = sorted(set(-(-lo // align) * align for lo, _hi, _r in buckets if lo > 0))
    cuts = [0] + [c for c in cuts if 0 < c < padded_total] + [padded_total]
    out = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        rd = [ready[i] for i in range(n) if offs[i] < hi and offs[i + 1] > lo]
        out.append((lo, hi, max(rd) if rd else -1))
    out.sort(key=lambda b: b[2])
    return out


class GradReducer:
    """Bucketed SUM over the ranks of a flat gradient tensor; the 1/world factor is folded into the loss (abc_loss_finalize
    grad_scale), so no separate scaling pass touches the gradients.  Exchange of one bucket, `mode`:

      "rs_ag"      reduce_scatter_tensor + all_gather_into_tensor, both in place on the arena (every rank reduces 1/world
                   of the bucket).  Buckets must split evenly: align_buckets().
      "direct"     all_to_all_single (rank r receives everybody's r-th shard: world-1 concurrent point-to-point transfers,
                   one per xGMI link) + a local fixed-order sum + all_gather_into_tensor -- no ring at all
      "all_reduce" one all_reduce per bucket (RCCL picks the algorithm) -- the plain alternative and the fallback

    Whatever the mode, every rank ends with bit-identical sums (each shard is reduced by ONE rank, then copied).  The chosen
    mode is self-checked once against all_reduce on a small tensor at construction (a collective: all ranks construct the
    reducer together, as they construct the Trainer); a mode the backend does not serve falls back to "all_reduce"."""

    MODES = ("rs_ag", "direct", "all_reduce")

    def __init__(self, flat_grad, buckets, group=None, mode="rs_ag", force=False):
        """force: run the exchange although the group has ONE rank (a sum over one rank: the identity) -- the RCCL launch
        mechanics between the hipGraph segments on a one-GPU box (tests/rccl_world1_worker.py)"""
        if mode not in self.MODES:
            raise ValueError("GradReducer mode %r" % (mode,))
        self.g, self.buckets, self.group = flat_grad, buckets, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.cuda = flat_grad.is_cuda
        self.dev = flat_grad.device
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self._pending = []
        # measure_exposed = True: finish() brackets the launch stream's wait for the communication stream with two events; their
        # distance is the part of the exchange that backward did NOT hide (exposed_ms(): one value per step since it was switched on)
        self.measure_exposed = False
        self._exposed = []
        self.by_ready = {}
        for b in buckets:
            self.by_ready.setdefault(b[2], []).append(b)
        self.mode = "all_reduce"
        self.fallback_reason = None
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        if self.active and mode != "all_reduce":
            if any((hi - lo) % self.world for lo, hi, _ in buckets):
                self.fallback_reason = "bucket sizes do not divide by the world size"
            else:
                self._stage = None
                if mode == "direct":
                    self._stage = torch.empty(max(hi - lo for lo, hi, _ in buckets), dtype=flat_grad.dtype, device=self.dev)
                self.mode = mode
                self.fallback_reason = self._self_check()
                if self.fallback_reason is not None:
                    self.mode = "all_reduce"

    # -- one bucket, synchronous on the current (communication) stream / thread
    def _exchange(self, view, mode, stage=None):
        w, r = self.world, self.rank
        if mode == "all_reduce":
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            return
        n = view.numel() // w
        shard = view[r * n:(r + 1) * n]
        if mode == "rs_ag":
            dist.reduce_scatter_tensor(shard, view, op=dist.ReduceOp.SUM, group=self.group)
        else:
            recv = stage[:w * n]
            dist.all_to_all_single(recv, view, group=self.group)
            torch.sum(recv.view(w, n), dim=0, out=shard)       # rows in rank order on every rank: a fixed summation order
        dist.all_gather_into_tensor(view, shard, group=self.group)

    def _self_check(self):
        """the chosen mode against all_reduce, first on 64 x world elements (does the backend serve it at all?), then on
        a scratch tensor of every DISTINCT REAL bucket size of the plan (the aliased in-place collectives at the sizes, shard
        boundaries and padded tail they will run with; small-integer values, so that the sums are exact in any order and the
        comparison is bit for bit); returns None or the reason for falling back"""
        w = self.world
        sizes = [64 * w] + sorted(set(hi - lo for lo, hi, _ in self.buckets))
        try:
            for n in sizes:
                t = (torch.arange(n, dtype=torch.int64, device=self.dev) % 7 + 1).to(self.g.dtype) * (self.rank + 1)
                want = t.clone()
                dist.all_reduce(want, op=dist.ReduceOp.SUM, group=self.group)
                stage = torch.empty_like(t) if self.mode == "direct" else None
                self._exchange(t, self.mode, stage)
                ok = torch.tensor([1.0 if torch.equal(t, want) else 0.0], dtype=torch.float32, device=self.dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
                if ok.item() != 1.0:
                    return "self-check against all_reduce failed at %d elements" % n
        except (RuntimeError, NotImplementedError, ValueError) as e:  # the backend does not serve it: all ranks fail alike
            return "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:120])
        return None

    def bucket_ready(self, lo, hi):
        if not self.active:
            return
        view = self.g[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._exchange(view, self.mode, self._stage if self.mode == "direct" else None)
        elif self.mode == "all_reduce":
            self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._exchange(view, self.mode, self._stage if self.mode == "direct" else None)

    def after_op(self, op_index):
        for lo, hi, _ in self.by_ready.get(op_index, ()):
            self.bucket_ready(lo, hi)

    def finish(self):
        if not self.active:
            return
        if self.cuda:
            main = torch.cuda.current_stream(self.dev)
            if self.measure_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                main.wait_stream(self.comm_stream)
                e1.record(main)
                self._exposed.append((e0, e1))
            else:
                main.wait_stream(self.comm_stream)
        else:
            for w in self._pending:
                w.wait()
            self._pending = []

    def exposed_ms(self):
        """per step since measure_exposed was set: milliseconds the launch stream spent waiting for the communication stream after
        the last backward kernel (host sync)"""
        if self.cuda:
            torch.cuda.synchronize(self.dev)
        out = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return out


def broadcast_parameters(flat_params, flat_buffers=None, src=0, group=None, counters=None):
    """DDP constructor semantics (multi_gpu_train.py:52): rank 0's parameters and buffers win"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(flat_params, src=src, group=group)
    broadcast_buffers(flat_buffers, counters, src=src, group=group)


def broadcast_buffers(flat_buffers, counters=None, src=0, group=None):
    """DDP(broadcast_buffers=True) (the default multi_gpu_train.py:52 runs with): before every forward rank 0's BatchNorm
    running statistics and num_batches_tracked replace every other rank's, so rank 0's trajectory is THE trajectory:
    its checkpoint holds statistics that only ever saw rank 0's shard, and an eval forward on any rank uses them.
    Train-mode arithmetic never reads the buffers, so doing this right before they are read (eval, state_dict) is
    observably the same as doing it before every forward; Trainer(broadcast_buffers=...) offers both."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if flat_buffers is not None and flat_buffers.numel():
        dist.broadcast(flat_buffers, src=src, group=group)
    if counters is not None and counters.numel():
        dist.broadcast(counters, src=src, group=group)
