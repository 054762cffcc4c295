"""Drop-in ``UNet`` for the reference call surface:

    UNet(in_channels, heads) -> nn.Module;  forward(x) -> list of 8 NCHW f32 maps

with the same attribute names (``n_channels``, ``heads``, ``s``), the same
``state_dict`` keys/shapes/dtypes as /root/reference/src/unet.py:77-119 (and unet2.py),
``module.``-prefix tolerant loading (train.py:50,435; img2smiles2.py:43-44), and
autograd support, so that the reference training loop (train.py:94-141) and the
inference driver (img2smiles2.py:42-79) run unchanged on it.

Two ways in:
  * ``forward(x)``      -- the compatibility path: HIP forward, NCHW f32 outputs,
                           ``torch.autograd`` backward through the HIP backward plan;
  * ``train_step(...)`` -- the fast path used by bench.py / the harness: forward +
                           fused loss + backward + (gradient all-reduce) + fused Adam, all in
                           HIP kernels, optionally replayed from a hipGraph.

All parameters live in ONE flat f32 arena (gradients, Adam moments likewise): the
optimiser is a single kernel and the data-parallel all-reduce a few large buckets.
There is no CPU fallback: without the HIP library / a GPU the compute entry points raise.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from . import arch
from .engine import Engine


def _kaiming_uniform_(t, fan_in, gen=None):
    b = 1.0 / (fan_in ** 0.5)  # kaiming_uniform(a=sqrt(5)) bound == 1/sqrt(fan_in), torch default for conv/linear
    with torch.no_grad():
        t.uniform_(-b, b, generator=gen)


class _UNetFn(torch.autograd.Function):
    """forward(x) of the compatibility path; parameters are passed so autograd routes their grads"""

    @staticmethod
    def forward(ctx, model, x, flat):
        eng = model._engine_for(x, model.training)
        st = torch.cuda.current_stream().cuda_stream
        eng.img.copy_(x.reshape(eng.img.shape))
        eng.run_pack(st)
        eng.run_forward(st)
        outs = model._export_logits(eng, st)
        ctx.model, ctx.eng = model, eng
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        model, eng = ctx.model, ctx.eng
        if not eng.train:
            raise RuntimeError("backward through an eval-mode forward is not supported")
        st = torch.cuda.current_stream().cuda_stream
        lib = eng.lib
        # d(loss)/d(logits) arrives as the NCHW f32 maps the kernels consume directly: plain copies
        for i, g in enumerate(gouts):
            if g is None:
                eng.dlogits[i].zero_()
            else:
                eng.dlogits[i].copy_(g)
        eng.chan_scale.fill_(1.0)
        eng.run_backward(st)
        return None, None, model._flat_grad.clone()


class UNetBase(nn.Module):
    VARIANT = "unet"

    def __init__(self, in_channels, heads=None, dtype="fp32", dropout_p=0.2):
        super().__init__()
        heads = list(arch.DEFAULT_HEADS if heads is None else heads)
        self.n_channels = in_channels
        self.heads = heads
        self.compute_dtype = dtype  # "fp32" (parity mode, exact-f32 MFMA) or "bf16" (throughput mode)
        self.dropout_p = dropout_p
        self._table = arch.state_table(self.VARIANT, in_channels, heads)
        self._lay_p, self._np = arch.arena_layout(self._table, "param")
        self._lay_b, self._nb = arch.arena_layout(self._table, "buffer")
        self._lay_c = OrderedDict((n, i) for i, (n, s, r) in enumerate(t for t in self._table if t[2] == "counter"))
        self._flat = nn.Parameter(torch.zeros(self._np))
        self.register_buffer("_flat_buf", torch.zeros(self._nb), persistent=False)
        self.register_buffer("_counters", torch.zeros(len(self._lay_c), dtype=torch.int64), persistent=False)
        self._flat_grad = None
        self._engines = {}
        self._opt = None
        self._graphs = {}
        self.reset_parameters()

    # ------------------------------------------------------------------ parameters
    def _view(self, name):
        for n, shape, role in self._table:
            if n == name:
                break
        else:
            raise KeyError(name)
        if role == "param":
            off, cnt = self._lay_p[name]
            return self._flat.data[off:off + cnt].view(shape)
        if role == "buffer":
            off, cnt = self._lay_b[name]
            return self._flat_buf[off:off + cnt].view(shape)
        return self._counters[self._lay_c[name]]

    def reset_parameters(self, seed=None):
        """torch default initialisation of the reference modules (unet.py:82-98): kaiming-uniform(a=sqrt 5)
        conv/linear weights, uniform(+-1/sqrt(fan_in)) biases, BN gamma=1 beta=0, s = randn(10)/100."""
        gen = None
        if seed is not None:
            gen = torch.Generator().manual_seed(seed)
        fan = {}
        for name, shape, role in self._table:
            if role == "param" and name.endswith(".weight") and len(shape) >= 2:
                f = shape[1]
                for d in shape[2:]:
                    f *= d
                fan[name[:-7]] = f
        with torch.no_grad():
            for name, shape, role in self._table:
                v = self._view(name)
                if role == "param":
                    if name == "s":
                        v.copy_(torch.randn(10, generator=gen) / 100)
                    elif len(shape) >= 2:
                        _kaiming_uniform_(v, fan[name[:-7]], gen)
                    elif name.endswith(".bias") and name[:-5] in fan:
                        _kaiming_uniform_(v, fan[name[:-5]], gen)
                    elif name.endswith(".weight"):
                        v.fill_(1.0)  # BN gamma
                    else:
                        v.zero_()  # BN beta
                elif role == "buffer":
                    v.fill_(1.0 if name.endswith("running_var") else 0.0)
                else:
                    v.zero_()

    def named_reference_parameters(self):
        """(reference name, view into the arena) for every learnable tensor, reference order"""
        for name, shape, role in self._table:
            if role == "param":
                yield name, self._view(name)

    def grad_of(self, name):
        off, cnt = self._lay_p[name]
        shape = [s for n, s, r in self._table if n == name][0]
        return self._flat_grad[off:off + cnt].view(shape)

    # state_dict in the reference's layout ------------------------------------------------
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for name, shape, role in self._table:
            destination[prefix + name] = self._view(name).detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        names = set()
        with torch.no_grad():
            for name, shape, role in self._table:
                names.add(name)
                key = prefix + name
                if key not in state_dict:
                    missing_keys.append(key)
                    continue
                v = state_dict[key]
                if tuple(v.shape) != tuple(shape):
                    error_msgs.append("size mismatch for %s: %s vs %s" % (key, tuple(v.shape), tuple(shape)))
                    continue
                self._view(name).copy_(v)
        for key in state_dict.keys():
            if key.startswith(prefix) and key[len(prefix):] not in names:
                unexpected_keys.append(key)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        """also accepts checkpoints saved from nn.DataParallel (keys prefixed 'module.', train.py:435)"""
        if state_dict and all(k.startswith("module.") for k in state_dict.keys()):
            state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
        return super().load_state_dict(state_dict, strict=strict)

    def __getattr__(self, name):
        if name == "s":  # model.module.s[i] in the reference loss (train.py:127-135)
            off, cnt = self.__dict__["_lay_p"]["s"]
            return self._flat[off:off + cnt]
        return super().__getattr__(name)

    # ------------------------------------------------------------------ engines
    def _engine_for(self, x, train):
        if not x.is_cuda:
            raise L.AbcNetHipError("abcnet_amd runs on an MI355X only (got a %s tensor); there is no CPU fallback" % x.device)
        B, Cc, H, W = x.shape
        key = (B, H, W, bool(train), self.compute_dtype)
        eng = self._engines.get(key)
        if eng is None or eng.params.data_ptr() != self._flat.data.data_ptr():
            if self._flat_grad is None or self._flat_grad.device != x.device:
                self._flat_grad = torch.zeros_like(self._flat.data)
            eng = Engine(self.VARIANT, self.n_channels, self.heads, self._flat.data, self._flat_grad, self._flat_buf,
                         self._counters, (self._lay_p, self._lay_b, self._lay_c), B, H, W, self.compute_dtype, train,
                         dropout_p=self.dropout_p, device=x.device)
            self._engines[key] = eng
        return eng

    def _export_logits(self, eng, st):
        # the kernels already wrote the reference's NCHW f32 maps; hand out copies because the engine
        # reuses its buffers on the next call (the reference returns fresh tensors)
        return [t.clone() for t in eng.logits]

    def forward(self, x):
        if torch.is_grad_enabled() and self.training:
            outs = _UNetFn.apply(self, x, self._flat)
        else:
            eng = self._engine_for(x, self.training)
            st = torch.cuda.current_stream().cuda_stream
            eng.img.copy_(x.reshape(eng.img.shape))
            eng.run_pack(st)
            eng.run_forward(st)
            outs = self._export_logits(eng, st)
        return list(outs)

    # ------------------------------------------------------------------ fast path
    def forward_logits(self, x):
        """the engine's own NCHW f32 head maps (valid until the next call), no copies"""
        eng = self._engine_for(x, self.training)
        st = torch.cuda.current_stream().cuda_stream
        eng.img.copy_(x.reshape(eng.img.shape))
        eng.run_pack(st)
        eng.run_forward(st)
        return eng.logits, eng

    def nms(self, x):
        """inference prologue of img2smiles2.py:56-79: forward + peak NMS, heat-map only"""
        lg, eng = self.forward_logits(x)
        from .ops import nms_peaks
        return nms_peaks(lg[0], lg[4], lg[6], lg[7])
