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

import torch
import torch.distributed as dist


def init_process_group(backend=None, rank=None, world_size=None, master_addr="127.0.0.1", master_port=None):
    """multi_gpu_train.py:44-45, with env-driven defaults (torchrun) instead of a fixed port"""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
    os.environ.setdefault("MASTER_ADDR", master_addr)
    os.environ.setdefault("MASTER_PORT", str(master_port or 29512))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    return rank, world_size


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


class GradReducer:
    """bucketed all-reduce(SUM) of a flat gradient tensor; the 1/world factor is folded into the loss
    (abc_loss_finalize grad_scale), so no separate scaling pass over the gradients is needed."""

    def __init__(self, flat_grad, buckets, group=None):
        self.g, self.buckets, self.group = flat_grad, buckets, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self._pending = []
        self.by_ready = {}
        for b in buckets:
            self.by_ready.setdefault(b[2], []).append(b)

    def bucket_ready(self, lo, hi):
        if self.world == 1:
            return
        view = self.g[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def after_op(self, op_index):
        for lo, hi, _ in self.by_ready.get(op_index, ()):
            self.bucket_ready(lo, hi)

    def finish(self):
        if self.world == 1:
            return
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w in self._pending:
                w.wait()
            self._pending = []


def broadcast_parameters(flat_params, flat_buffers=None, src=0, group=None):
    """DDP constructor semantics (multi_gpu_train.py:52): rank 0's parameters and buffers win"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(flat_params, src=src, group=group)
    if flat_buffers is not None:
        dist.broadcast(flat_buffers, src=src, group=group)
