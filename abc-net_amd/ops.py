"""Thin host wrappers over single C-ABI kernels used outside the engine's static plan:
fused loss (train.py:95-137), inference NMS (img2smiles2.py:61-79), fused Adam (train.py:55,141)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

HEAD_NAMES = ["atom_t", "atom_types", "atom_charges", "atom_hs", "bond_t", "bond_types", "bond_rhos", "bond_omega"]


class FusedLoss:
    """activation + 8-term loss + dlogits for fixed shapes; targets are read from the given (static) tensors"""

    def __init__(self, eng, targets, s_ptr, ds_ptr, grad_scale=1.0):
        lib = eng.lib
        self.eng, self.lib = eng, lib
        self.targets = targets  # keep alive
        self._check_targets(eng, targets)
        d = L.LossDesc()
        for i in range(8):
            d.logits[i], d.dlogits[i] = eng.logits[i].data_ptr(), eng.dlogits[i].data_ptr()
        (d.t_atom, d.t_types, d.t_charges, d.t_hs, d.t_bond, d.t_btypes, d.t_rho, d.t_omega) = (t.data_ptr() for t in targets)
        d.B, d.h, d.w = eng.B, eng.h, eng.w
        self.nblk = lib.abc_loss_blocks(C.byref(d))
        dev = eng.logits[0].device
        self.partial = torch.zeros((self.nblk, 16), dtype=torch.float64, device=dev)
        d.partial = self.partial.data_ptr()
        self.out = torch.zeros(17, dtype=torch.float64, device=dev)
        f = L.LossFinDesc()
        f.partial, f.nblk, f.s, f.ds, f.out = self.partial.data_ptr(), self.nblk, s_ptr, ds_ptr, self.out.data_ptr()
        f.chan_scale, f.nchan = eng.chan_scale.data_ptr(), eng.chan_scale.numel()
        for i in range(8):
            f.chan_off[i] = eng.head_off[i]
            f.head_c[i] = eng.heads[i]
        f.grad_scale = grad_scale
        self.d, self.f = d, f

    @staticmethod
    def _check_targets(eng, targets):
        exp = [(eng.B, 1), (eng.B, 14), (eng.B, 3), (eng.B, 2), (eng.B, 1), (eng.B, 6, 60), (eng.B, 60), (eng.B, 60)]
        dts = [torch.float32] * 6 + [torch.float64] * 2
        for t, e, dt in zip(targets, exp, dts):
            if tuple(t.shape) != tuple(e) + (eng.h, eng.w) or t.dtype != dt or not t.is_contiguous():
                raise ValueError("target %s %s does not match the contract %s %s" % (tuple(t.shape), t.dtype, e, dt))
        if eng.heads != [1, 14, 3, 2, 1, 360, 60, 60]:
            raise ValueError("the fused loss is defined for heads [1,14,3,2,1,360,60,60] (train.py:47)")

    def run(self, stream):
        L.check(self.lib.abc_loss_fwd_bwd(C.byref(self.d), stream), "loss_fwd_bwd")
        L.check(self.lib.abc_loss_finalize(C.byref(self.f), stream), "loss_finalize")

    def total_device(self):
        """the total loss of the last step as a 0-d f64 DEVICE tensor (a view of the finaliser's output: no host sync)"""
        return self.out[0]

    def result(self):
        """dict: total + weighted terms + raw terms (device sync)"""
        o = self.out.cpu()
        r = {"total": o[0].item()}
        for i, n in enumerate(HEAD_NAMES):
            r[n] = o[1 + i].item()
            r["raw_" + n] = o[9 + i].item()
        return r


class FusedHeadsLoss(FusedLoss):
    """The fused train step's heads (engine built with fused_heads=True): out_modules[i].conv2 + activation + loss +
    d(logits) + the gradient back to the heads' BatchNorm outputs in one pass (abc_heads_fused_fwd_bwd), then the same
    finalisation as FusedLoss.  Same interface."""

    def __init__(self, eng, targets, s_ptr, ds_ptr, grad_scale=1.0, keep_logits=True):
        """keep_logits=False: the logits never leave the kernel (eng.logits keep their old contents) -- for a training
        loop without the meters of train.py:145-215, the only other reader of the outputs"""
        lib = eng.lib
        self.eng, self.lib = eng, lib
        self.targets = targets
        self._check_targets(eng, targets)
        d = eng.hf
        for i in range(8):
            d.logits[i] = eng.logits[i].data_ptr() if keep_logits else None
        (d.t_atom, d.t_types, d.t_charges, d.t_hs, d.t_bond, d.t_btypes, d.t_rho, d.t_omega) = (t.data_ptr() for t in targets)
        self.nblk = eng.hf_lossblocks
        self.partial = eng.hf_losspart
        self.out = torch.zeros(17, dtype=torch.float64, device=eng.logits[0].device)
        f = L.LossFinDesc()
        f.partial, f.nblk, f.s, f.ds, f.out = self.partial.data_ptr(), self.nblk, s_ptr, ds_ptr, self.out.data_ptr()
        f.chan_scale, f.nchan = eng.chan_scale.data_ptr(), eng.chan_scale.numel()
        for i in range(8):
            f.chan_off[i] = eng.head_off[i]
            f.head_c[i] = eng.heads[i]
        f.grad_scale = grad_scale
        self.d, self.f = d, f

    def use_target_flags(self, flags):
        """flags: TargetRasterizer(sparse=True).group_flags of the rasteriser that draws THESE target tensors (or None: read every
        target plane).  A wave whose 32 pixels carry no target of a head then reads zeros from a 512-byte buffer instead of the maps
        (abc_heads_fused_desc.target_flags): the same loss and gradients bit for bit, 0.37 GB less read per step at b16 @ 384 x 384."""
        if flags is None:
            self.d.target_flags, self.d.zero_bytes = None, None
            self._tflags = None
            return
        n = self.eng.B * self.eng.h * self.eng.w // 32
        if flags.dtype not in (torch.int32, torch.uint32) or flags.numel() != n or not flags.is_cuda or not flags.is_contiguous():
            raise L.AbcNetHipError("target flags: one 32-bit word per 32 pixels of the batch (%d words) on the device" % n)
        self._tzero = torch.zeros(512, dtype=torch.uint8, device=flags.device)
        self._tflags = flags
        self.d.target_flags, self.d.zero_bytes = flags.data_ptr(), self._tzero.data_ptr()

    def run(self, stream):
        L.check(self.lib.abc_heads_fused_fwd_bwd(C.byref(self.d), stream), "heads_fused_fwd_bwd")
        L.check(self.lib.abc_loss_finalize(C.byref(self.f), stream), "loss_finalize")


METER_NAMES = [
    "atom_targets_precision", "atom_targets_precision3", "atom_targets_recall", "atom_targets_recall3",
    "atom_types_acc", "atom_charges_acc", "atom_hs_acc",
    "bond_targets_precision", "bond_targets_precision3", "bond_targets_recall", "bond_targets_recall3",
    "bond_types_acc", "bond_rhos_mae",
    "bond_omega_precision", "bond_omega_recall3", "bond_omega_recall", "bond_omega_precision3",
]


class FusedMetrics:
    """The 17 AverageMeters of train.py:145-215 as one device-resident table: `run` adds the current batch
    (logits + targets already in HBM), `result` reads it (the only host sync), `reset` is the new-epoch
    re-creation of the meters (train.py:58-81)."""

    def __init__(self, logits, targets):
        lib = L.load()
        self.lib = lib
        B, _, h, w = logits[0].shape
        exp = [(B, 1), (B, 14), (B, 3), (B, 2), (B, 1), (B, 6, 60), (B, 60), (B, 60)]
        dts = [torch.float32] * 6 + [torch.float64] * 2
        for t, e, dt in zip(targets, exp, dts):
            if tuple(t.shape) != tuple(e) + (h, w) or t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
                raise L.AbcNetHipError("metrics: target %s %s does not match the contract %s %s (device tensors; no CPU "
                                       "fallback)" % (tuple(t.shape), t.dtype, e, dt))
        for t, c in zip(logits, [1, 14, 3, 2, 1, 360, 60, 60]):
            if tuple(t.shape) != (B, c, h, w) or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise L.AbcNetHipError("metrics: logits must be the 8 contiguous NCHW f32 device maps of heads [1,14,3,2,1,360,60,60]")
        self.keep = (list(logits), list(targets))
        dev = logits[0].device
        d = L.MetricsDesc()
        for i in range(8):
            d.logits[i] = logits[i].data_ptr()
        (d.t_atom, d.t_types, d.t_charges, d.t_hs, d.t_bond, d.t_btypes, d.t_rho, d.t_omega) = (t.data_ptr() for t in targets)
        d.B, d.h, d.w = B, h, w
        self.peaks = torch.zeros((2, B, h, w), dtype=torch.uint8, device=dev)
        self.partial = torch.zeros((lib.abc_metrics_blocks(C.byref(d)), 24), dtype=torch.float64, device=dev)
        self.totals = torch.zeros((17, 2), dtype=torch.float64, device=dev)
        self.last = torch.zeros((17, 2), dtype=torch.float64, device=dev)
        d.peaks, d.partial, d.totals, d.last = self.peaks.data_ptr(), self.partial.data_ptr(), self.totals.data_ptr(), self.last.data_ptr()
        self.d = d

    def run(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        L.check(self.lib.abc_metrics_update(C.byref(self.d), stream), "metrics_update")

    def reset(self):
        self.totals.zero_()

    def result(self):
        """{name: {"sum", "count", "avg", "val"}} -- the AverageMeter fields (device sync)"""
        tot, last = self.totals.cpu(), self.last.cpu()
        out = {}
        for i, n in enumerate(METER_NAMES):
            s, c = tot[i, 0].item(), tot[i, 1].item()
            ln, ld = last[i, 0].item(), last[i, 1].item()
            out[n] = {"sum": s, "count": c, "avg": s / c if c else float("nan"), "val": ln / ld if ld else float("nan")}
        return out


def nms_peaks(atom, bond, rho, omega):
    """img2smiles2.py:61-79 on the NCHW f32 head maps: (atom_mask[B,1,h,w], bond_mask[B,1,h,w],
    |rho|[B,60,h,w], omega_mask[B,60,h,w])"""
    lib = L.load()
    B, n, h, w = omega.shape
    for t in (atom, bond, rho, omega):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise L.AbcNetHipError("nms_peaks wants contiguous f32 device tensors (no CPU fallback)")
    am, bm, r, om = torch.empty_like(atom), torch.empty_like(bond), torch.empty_like(rho), torch.empty_like(omega)
    d = L.NmsDesc()
    d.atom, d.bond, d.rho, d.omega = atom.data_ptr(), bond.data_ptr(), rho.data_ptr(), omega.data_ptr()
    d.B, d.h, d.w, d.n_omega = B, h, w, n
    d.atom_mask, d.bond_mask, d.rho_abs, d.omega_mask = am.data_ptr(), bm.data_ptr(), r.data_ptr(), om.data_ptr()
    L.check(lib.abc_nms_peaks(C.byref(d), torch.cuda.current_stream().cuda_stream), "nms_peaks")
    return am, bm, r, om


class PeakExtractor:
    """img2smiles2.py:113-191 on the device: NMS masks + raw head maps -> compact ordered candidate lists (the wire
    format into the unchanged CPU graph-assembly stage).  Static buffers, one launch, graph-capture safe; `lists()` is the
    only host sync (one small D2H per batch instead of hundreds of .item() calls per image)."""

    def __init__(self, logits, atom_mask, bond_mask, cap_atoms=512, cap_bonds=16384, btype_idx=None, rho_abs=None):
        """btype_idx / rho_abs (decode mode, InferenceRunner(decode=True)): the uint8 arg-max map of the bond-type head and the |rho| map
        the heads kernel wrote instead of the raw maps logits[5] / logits[6] (which may then be None)"""
        lib = L.load()
        self.lib = lib
        B, _, h, w = logits[0].shape
        need = [t for i, t in enumerate(logits) if not ((i == 5 and btype_idx is not None) or (i == 6 and rho_abs is not None))]
        for t in need + [atom_mask, bond_mask] + ([rho_abs] if rho_abs is not None else []):
            if t is None or not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise L.AbcNetHipError("PeakExtractor wants contiguous f32 device tensors (no CPU fallback)")
        if btype_idx is not None and not (btype_idx.is_cuda and btype_idx.is_contiguous() and btype_idx.dtype == torch.uint8 and
                                          tuple(btype_idx.shape) == (B, 60, h, w)):
            raise L.AbcNetHipError("PeakExtractor: btype_idx must be a contiguous uint8 device tensor [B, 60, h, w]")
        dev = logits[0].device
        d = L.ExtractDesc()
        d.atom_mask, d.bond_mask = atom_mask.data_ptr(), bond_mask.data_ptr()
        d.types, d.charges, d.hs = logits[1].data_ptr(), logits[2].data_ptr(), logits[3].data_ptr()
        d.btypes = None if btype_idx is not None else logits[5].data_ptr()
        d.btype_idx = None if btype_idx is None else btype_idx.data_ptr()
        d.rho = rho_abs.data_ptr() if rho_abs is not None else logits[6].data_ptr()
        d.omega = logits[7].data_ptr()
        d.B, d.h, d.w, d.cap_atoms, d.cap_bonds = B, h, w, cap_atoms, cap_bonds
        self.counts = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        self.atoms = torch.zeros((B, cap_atoms, 5), dtype=torch.int32, device=dev)
        self.bonds = torch.zeros((B, cap_bonds, 4), dtype=torch.int32, device=dev)
        self.bond_rho = torch.zeros((B, cap_bonds), dtype=torch.float32, device=dev)
        self.work = torch.zeros((lib.abc_extract_work_ints(C.byref(d)),), dtype=torch.int32, device=dev)
        self.work_masks = torch.zeros((lib.abc_extract_work_masks(C.byref(d)),), dtype=torch.int64, device=dev)
        d.counts, d.atoms, d.bonds, d.bond_rho = self.counts.data_ptr(), self.atoms.data_ptr(), self.bonds.data_ptr(), self.bond_rho.data_ptr()
        d.work, d.work_masks = self.work.data_ptr(), self.work_masks.data_ptr()
        self.d, self.keep = d, (list(logits), atom_mask, bond_mask, btype_idx, rho_abs)
        self.B, self.cap_atoms, self.cap_bonds = B, cap_atoms, cap_bonds

    def run(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        L.check(self.lib.abc_extract_peaks(C.byref(self.d), stream), "extract_peaks")

    def lists(self):
        """per image: dict(atoms int32 [n,5], bonds int32 [m,4], rho f32 [m], counts (4,), truncated bool) on the host"""
        cnt = self.counts.cpu()
        na = int(min(int(cnt[:, 1].max()), self.cap_atoms))
        nb = int(min(int(cnt[:, 3].max()), self.cap_bonds))
        atoms, bonds, rho = self.atoms[:, :na].cpu(), self.bonds[:, :nb].cpu(), self.bond_rho[:, :nb].cpu()
        out = []
        for b in range(self.B):
            a, m = int(cnt[b, 1]), int(cnt[b, 3])
            trunc = a > self.cap_atoms or m > self.cap_bonds or int(cnt[b, 0]) > self.cap_atoms or int(cnt[b, 2]) > 4096
            a, m = min(a, self.cap_atoms), min(m, self.cap_bonds)
            out.append({"atoms": atoms[b, :a], "bonds": bonds[b, :m], "rho": rho[b, :m], "counts": cnt[b].tolist(), "truncated": trunc})
        return out


class FusedAdam:
    """torch.optim.Adam(lr, weight_decay) over the flat arena as one kernel (train.py:55).  Re-create it to
    reset the moments, as the reference does at the learning-rate drop (train.py:84-85)."""

    def __init__(self, params, grads, lr=2.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8, grad_scale=1.0):
        self.lib = L.load()
        self.p, self.g = params, grads
        self.m = torch.zeros_like(params)
        self.v = torch.zeros_like(params)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=params.device)
        d = L.AdamDesc()
        d.p, d.g, d.m, d.v, d.n = params.data_ptr(), grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), params.numel()
        d.step = self.step_t.data_ptr()
        d.lr, d.beta1, d.beta2, d.eps, d.weight_decay, d.grad_scale = lr, betas[0], betas[1], eps, weight_decay, grad_scale
        self.d = d

    def step(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        L.check(self.lib.abc_adam_step(C.byref(self.d), stream), "adam_step")
