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
The module tree mirrors the reference's (inc1.double_conv.0.weight, ...): its 159 (unet) /
251 (unet2) nn.Parameters and the BatchNorm buffers are views into the arenas, so
named_parameters(), per-tensor .grad after loss.backward(), torch.optim.Adam(model.parameters())
and state_dict() behave as with unet.py:78-98.
There is no CPU fallback: without the HIP library / a GPU the compute entry points raise.
"""
from __future__ import annotations

import ctypes as C
import threading
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from . import arch
from .engine import Engine


GRAD_STORE_ALIGN = 840 * 128   # lcm(1..8) x 128 elements


def _kaiming_uniform_(t, fan_in, gen=None):
    b = 1.0 / (fan_in ** 0.5)  # kaiming_uniform(a=sqrt(5)) bound == 1/sqrt(fan_in), torch default for conv/linear
    with torch.no_grad():
        t.uniform_(-b, b, generator=gen)


class _UNetFn(torch.autograd.Function):
    """forward(x) of the compatibility path; the reference-named parameters are passed so that autograd routes their
    gradients to them (loss.backward() then fills p.grad of all 159 / 251 tensors, train.py:140)"""

    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model._engine_for(x, model.training)
        st = torch.cuda.current_stream(x.device).cuda_stream
        model._load_image(eng, x)
        eng.run_pack(st)
        eng.run_forward(st)
        outs = model._export_logits(eng, st)
        ctx.model, ctx.eng, ctx.need_x = model, eng, x.requires_grad
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        model, eng = ctx.model, ctx.eng
        if not eng.train:
            raise RuntimeError("backward through an eval-mode forward is not supported")
        if ctx.need_x:
            raise RuntimeError("abcnet_amd: the gradient with respect to the input image is not produced")
        with torch.cuda.device(eng.img.device):
            st = torch.cuda.current_stream().cuda_stream
            # d(loss)/d(logits) arrives as the NCHW f32 maps the kernels consume directly: plain copies
            for i, g in enumerate(gouts):
                if g is None:
                    eng.dlogits[i].zero_()
                else:
                    eng.dlogits[i].copy_(g)
            eng.chan_scale.fill_(1.0)
            eng.run_backward(st)
            g = model._flat_grad.clone()
        model._dp_release()         # (multi-device nn.DataParallel: this replica's engine is free again)
        return (None, None) + tuple(g[off:off + cnt].view(shape) for off, cnt, shape in model._param_slices)


def _rebuild_unet(cls, in_channels, heads, dtype, dropout_p, flat, buf, counters, device, training, seed_base, req_grad):
    """unpickle / deep-copy: a fresh module tree over fresh arenas (the named tensors must stay views of ONE arena; a
    tensor-by-tensor copy, which is what nn.Module's default pickling and nn.Parameter.__deepcopy__ do, would detach them
    from the arena the engines, Adam and the all-reduce work on)"""
    m = cls(in_channels, list(heads), dtype=dtype, dropout_p=dropout_p)
    with torch.no_grad():
        m._flat.copy_(flat)
        m._flat_buf.copy_(buf)
        m._counters.copy_(counters)
    m.dropout_seed_base = m.dropout_seed = seed_base
    for p, r in zip(m.parameters(), req_grad):
        p.requires_grad_(r)
    if torch.device(device).type != "cpu":
        m = m.to(device)
    m.train(training)
    return m


class _Node(nn.Module):
    """a container of the reference's module tree (DoubleConv, Sequential, Conv2d, BatchNorm2d, ... by position): it only
    holds the reference-named parameters and buffers, which are views into the model's flat arenas"""


class UNetBase(nn.Module):
    VARIANT = "unet"

    def __init__(self, in_channels, heads=None, dtype="fp32", dropout_p=0.2):
        super().__init__()
        heads = list(arch.DEFAULT_HEADS if heads is None else heads)
        self.n_channels = in_channels
        self.heads = heads
        self.compute_dtype = dtype  # "fp32" (parity mode, exact-f32 MFMA) or "bf16" (throughput mode)
        self.dropout_p = dropout_p
        self.dropout_seed_base = 0x1234ABCD
        self.dropout_seed = self.dropout_seed_base   # a Trainer under torch.distributed mixes its rank in
        self._table = arch.state_table(self.VARIANT, in_channels, heads)
        self._lay_p, self._np = arch.arena_layout(self._table, "param")
        self._lay_b, self._nb = arch.arena_layout(self._table, "buffer")
        self._lay_c = OrderedDict((n, i) for i, (n, s, r) in enumerate(t for t in self._table if t[2] == "counter"))
        # ONE flat f32 arena per role; the reference-named tensors registered below are views into them
        self._flat = torch.zeros(self._np)
        self._flat_buf = torch.zeros(self._nb)
        self._counters = torch.zeros(len(self._lay_c), dtype=torch.int64)
        self._flat_grad = None
        self._grad_store = None
        self._engines = {}
        # multi-device nn.DataParallel (train.py:50, img2smiles2.py:43 on a multi-GPU box): the replicas torch makes every
        # forward share this object's attributes; they find the master through _master_ref and their per-device shadow
        # models (own arenas + engines) in _dp_shadows -- see _forward_replica
        self._master_ref = weakref.ref(self)
        self._dp_shadows = {}
        self._dp_lock = threading.Lock()
        self._dp_busy = False
        self._leaves = []          # (name, holder module, attribute, role)
        self._param_slices = []    # (offset, numel, shape) in parameters() order
        for name, shape, role in self._table:
            *path, attr = name.split(".")
            mod = self
            for seg in path:
                nxt = mod._modules.get(seg)
                if nxt is None:
                    nxt = _Node()
                    mod.add_module(seg, nxt)
                mod = nxt
            v = self._view(name)
            if role == "param":
                mod.register_parameter(attr, nn.Parameter(v))
                off, cnt = self._lay_p[name]
                self._param_slices.append((off, cnt, tuple(shape)))
            else:
                mod.register_buffer(attr, v)
            self._leaves.append((name, mod, attr, role))
        self.reset_parameters()

    # ------------------------------------------------------------------ copies
    def __reduce__(self):
        """torch.save(model) / pickle / copy.deepcopy: the three arenas + the constructor arguments (engines, graphs and
        the gradient arena are rebuilt on demand)"""
        return (_rebuild_unet, (type(self), self.n_channels, list(self.heads), self.compute_dtype, self.dropout_p,
                                self._flat.detach().cpu().clone(), self._flat_buf.detach().cpu().clone(), self._counters.cpu().clone(),
                                str(self._flat.device), self.training, self.dropout_seed_base,
                                [p.requires_grad for p in self.parameters()]))

    def __deepcopy__(self, memo):
        fn, args = self.__reduce__()
        new = fn(*args)
        memo[id(self)] = new
        return new

    # ------------------------------------------------------------------ parameters
    def _shape_role(self, name):
        for n, shape, role in self._table:
            if n == name:
                return shape, role
        raise KeyError(name)

    def _view(self, name):
        shape, role = self._shape_role(name)
        if role == "param":
            off, cnt = self._lay_p[name]
            return self._flat[off:off + cnt].view(shape)
        if role == "buffer":
            off, cnt = self._lay_b[name]
            return self._flat_buf[off:off + cnt].view(shape)
        return self._counters[self._lay_c[name]]

    def _rebind(self):
        """point every registered parameter / buffer at its slice of the (moved) arenas"""
        for name, mod, attr, role in self._leaves:
            v = self._view(name)
            if role == "param":
                p = mod._parameters[attr]
                p.data = v
                p.grad = None
            else:
                mod._buffers[attr] = v

    def _apply(self, fn, recurse=True):
        """model.to(device) / .cuda(rank) (train.py:48, multi_gpu_train.py:49): the ARENAS move, the named tensors are
        re-pointed at them (moving 159 tensors one by one would scatter them)"""
        flat, buf = fn(self._flat), fn(self._flat_buf)
        cnt = fn(self._counters)
        if flat.dtype != torch.float32 or buf.dtype != torch.float32 or cnt.dtype != torch.int64:
            raise L.AbcNetHipError("abcnet_amd keeps its parameters in float32 (the compute dtype is the constructor's dtype=)")
        moved = flat.device != self._flat.device
        self._flat, self._flat_buf, self._counters = flat, buf, cnt
        if moved:
            self._flat_grad = None
            self._engines = {}
        self._rebind()
        return self

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
                if v.device.type != "cpu":
                    raise L.AbcNetHipError("reset_parameters(): initialise on the host, then .to(device)")
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
        """the fast path's gradient of a reference-named tensor (a view into the flat gradient arena the Trainer's
        backward plan fills; the compatibility path fills p.grad of the named parameters instead)"""
        off, cnt = self._lay_p[name]
        return self._flat_grad[off:off + cnt].view(self._shape_role(name)[0])

    def flat_param_grads(self):
        """p.grad of all named parameters concatenated in arena order (zeros where a parameter has none)"""
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.parameters()])

    # state_dict: the registered tensors carry the reference's names, shapes and dtypes; copies go through the views
    def load_state_dict(self, state_dict, strict=True, assign=False):
        """also accepts checkpoints saved from nn.DataParallel / DDP (keys prefixed 'module.', train.py:435,
        img2smiles2.py:43-44)"""
        if assign:
            raise L.AbcNetHipError("load_state_dict(assign=True) would detach the parameters from the arena")
        if state_dict and all(k.startswith("module.") for k in state_dict.keys()):
            state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
        return super().load_state_dict(state_dict, strict=strict)

    # ------------------------------------------------------------------ engines
    def _engine_for(self, x, train, fold_bn=False, fused_heads=False, batched_heads=True, fp8=False, guards=False, heads_epilogue=False,
                    actbwd_epilogue=True, merge_reduce=True, nms_heads=False, decode=False, dual_wgrad=True, fused_convt=True):
        if not x.is_cuda:
            raise L.AbcNetHipError("abcnet_amd runs on an MI355X only (got a %s tensor); there is no CPU fallback" % x.device)
        if x.device != self._flat.device:
            # (nn.DataParallel's replicas do not come here: _forward_replica gives each device its own arenas and engines)
            raise L.AbcNetHipError("input on %s but the model lives on %s (move one of them; several GPUs: nn.DataParallel(model, "
                                   "device_ids=...) or, better, one process per GPU -- abcnet_amd.distributed.launch_ranks / torchrun)"
                                   % (x.device, self._flat.device))
        B, Cc, H, W = x.shape
        if Cc != self.n_channels:
            raise ValueError("expected %d input channels, got %d" % (self.n_channels, Cc))
        key = (B, H, W, bool(train), self.compute_dtype, self.dropout_seed, bool(fold_bn), bool(fused_heads), bool(batched_heads), bool(fp8), bool(guards), bool(heads_epilogue), bool(actbwd_epilogue), bool(merge_reduce),
               bool(nms_heads), bool(decode), bool(dual_wgrad), bool(fused_convt), L.load().abc_get_reserved_cus())     # (grid sizes and statistics rows follow abc_set_reserved_cus)
        eng = self._engines.get(key)
        if eng is None or eng.params.data_ptr() != self._flat.data_ptr():
            if self._flat_grad is None or self._flat_grad.device != x.device:
                # the gradient arena sits in a store padded to a multiple of 840 x 128 elements, so that a data-parallel
                # reduce-scatter can cut it into buckets that split evenly over any world size up to 8 (distributed.py)
                pad = -(-self._np // GRAD_STORE_ALIGN) * GRAD_STORE_ALIGN
                self._grad_store = torch.zeros(pad, dtype=torch.float32, device=x.device)
                self._flat_grad = self._grad_store[:self._np]
            with torch.cuda.device(x.device):
                eng = Engine(self.VARIANT, self.n_channels, self.heads, self._flat, self._flat_grad, self._flat_buf,
                             self._counters, (self._lay_p, self._lay_b, self._lay_c), B, H, W, self.compute_dtype, train,
                             dropout_p=self.dropout_p, device=x.device, drop_seed=self.dropout_seed, fold_bn=fold_bn,
                             fused_heads=fused_heads, batched_heads=batched_heads, fp8=fp8, guards=guards, heads_epilogue=heads_epilogue, nms_heads=nms_heads,
                             actbwd_epilogue=actbwd_epilogue, merge_reduce=merge_reduce, decode=decode, dual_wgrad=dual_wgrad, fused_convt=fused_convt)
            self._engines[key] = eng
        return eng

    def _load_image(self, eng, x):
        eng.img.copy_(x.reshape(eng.img.shape))

    def _export_logits(self, eng, st):
        # the kernels already wrote the reference's NCHW f32 maps; hand out copies because the engine
        # reuses its buffers on the next call (the reference returns fresh tensors)
        return [t.clone() for t in eng.logits]

    # ------------------------------------------------------------------ multi-device nn.DataParallel
    def _replica_tensors(self):
        """the broadcast copies torch.nn.parallel.replicate hung on this replica's module tree, in arena order"""
        params, bufs, cnts = [], [], []
        for name, shape, role in self._table:
            *path, attr = name.split(".")
            mod = self
            for seg in path:
                mod = mod._modules[seg]
            if role == "param":
                params.append(mod._former_parameters[attr])
            elif role == "buffer":
                bufs.append(mod._buffers[attr])
            else:
                cnts.append(mod._buffers[attr])
        return params, bufs, cnts

    def _dp_release(self, token=None):
        """give an engine owner back (token: only if it is still checked out by THAT forward -- a stale finaliser must not
        free a later checkout)"""
        lock = getattr(self, "_dp_lock", None)
        if lock is None:
            self._dp_busy = False
            return
        with lock:
            if token is None or getattr(self, "_dp_token", None) is token:
                self._dp_busy = False
                self._dp_token = None

    def _dp_checkout(self, master, dev):
        """an engine owner for one replica's forward (+ backward): the master itself for the first replica on its device,
        else a shadow model of that device (own arenas, engines, gradient arena), created on first use"""
        with master._dp_lock:
            if dev == master._flat.device and not master._dp_busy:
                master._dp_busy = True
                return master
            pool = master._dp_shadows.setdefault(dev, [])
            for sh in pool:
                if not sh._dp_busy:
                    sh._dp_busy = True
                    return sh
            if len(pool) >= 4:
                raise L.AbcNetHipError("nn.DataParallel: more than 4 unfinished forward passes on %s (a training forward holds its "
                                       "engine until its backward has run)" % dev)
            fn, args = master.__reduce__()
            args = list(args)
            args[8] = str(dev)
            sh = fn(*args)
            sh._dp_busy = True
            pool.append(sh)
            return sh

    def _forward_replica(self, x):
        """forward of a replica made by nn.DataParallel's replicate() (a shallow copy of the master whose module tree carries
        broadcast copies of the parameters on this replica's device).  The parameter copies are packed into the arena of the
        engine owner checked out for this device and passed to the autograd function, so that backward's gradients flow back
        through torch's Broadcast to the master's parameters, as with any module under nn.DataParallel (train.py:50,139-141).
        BatchNorm buffers: every replica normalises its own chunk; only the replica that runs on the master's own arenas
        updates the running statistics that persist -- torch's DataParallel semantics."""
        master = self._master_ref()
        if master is None:
            raise L.AbcNetHipError("nn.DataParallel replica without its master module")
        dev = x.device
        params, bufs, cnts = self._replica_tensors()
        owner = self._dp_checkout(master, dev)
        try:
            if owner is not master:
                with torch.no_grad():
                    torch.cat([p.detach().reshape(-1) for p in params], out=owner._flat)
                    torch.cat([b.reshape(-1).to(torch.float32) for b in bufs], out=owner._flat_buf)
                    owner._counters.copy_(torch.stack([c.reshape(()) for c in cnts]))
                from .distributed import rank_dropout_seed
                owner.dropout_seed = rank_dropout_seed(master.dropout_seed_base, 1 + (dev.index or 0) + 16 * master._dp_shadows[dev].index(owner))
            owner.train(self.training)
            with torch.cuda.device(dev):
                if torch.is_grad_enabled() and self.training:
                    outs = list(_UNetFn.apply(owner, x, *params))      # (backward releases the owner ...)
                    # ... and so does the END OF THE GRAPH'S LIFE: a forward whose graph is dropped without a backward (an
                    # exception in the loss, a loss only inspected) would otherwise keep its owner checked out for ever -- the
                    # master's running statistics would silently stop updating, the fifth such forward per device raise
                    token = owner._dp_token = object()
                    node = outs[0].grad_fn
                    if node is not None:
                        weakref.finalize(node, owner._dp_release, token)
                    else:
                        owner._dp_release()
                    return outs
                eng = owner._engine_for(x, self.training)
                st = torch.cuda.current_stream().cuda_stream
                owner._load_image(eng, x)
                eng.run_pack(st)
                eng.run_forward(st)
                outs = owner._export_logits(eng, st)
            owner._dp_release()
            return list(outs)
        except Exception:
            owner._dp_release()
            raise

    def forward(self, x):
        if getattr(self, "_is_replica", False):
            return self._forward_replica(x)
        if torch.is_grad_enabled() and self.training:
            outs = _UNetFn.apply(self, x, *self.parameters())
        else:
            eng = self._engine_for(x, self.training)
            with torch.cuda.device(x.device):
                st = torch.cuda.current_stream().cuda_stream
                self._load_image(eng, x)
                eng.run_pack(st)
                eng.run_forward(st)
                outs = self._export_logits(eng, st)
        return list(outs)

    # ------------------------------------------------------------------ fast path
    def forward_logits(self, x):
        """the engine's own NCHW f32 head maps (valid until the next call), no copies"""
        eng = self._engine_for(x, self.training)
        with torch.cuda.device(x.device):
            st = torch.cuda.current_stream().cuda_stream
            self._load_image(eng, x)
            eng.run_pack(st)
            eng.run_forward(st)
        return eng.logits, eng

    def nms(self, x):
        """inference prologue of img2smiles2.py:56-79: forward + peak NMS, heat-map only"""
        lg, eng = self.forward_logits(x)
        from .ops import nms_peaks
        return nms_peaks(lg[0], lg[4], lg[6], lg[7])
