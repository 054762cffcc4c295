"""Targets from annotations on the device: the rasteriser of /root/reference/src/utils.py:83-228
(MolecularImageDataset.__getitem__) split into a tiny host half and a device half.

    host   parse_record(): the string handling, vocabulary look-ups and the float64 atan / floor of utils.py:94-163 --
           exactly the reference's Python arithmetic -- into (atoms int32 [n,5], bonds int32 [m,5], rho float64 [m]);
    device TargetRasterizer: zero + rasterise the 8 target maps in HBM (csrc/raster.hip, abc_rasterize_targets), in
           record order, into static buffers (optionally the Trainer's own target buffers).

A batch-16 step then ships a few KB of records over PCIe instead of 376 MB of mostly-zero maps.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L

ATOM_VOCAB = {'<unkonw>': 0, 'C': 1, 'N': 2, 'O': 3, 'P': 4, 'F': 5, 'Cl': 6, 'S': 7, 'Br': 8, 'B': 9,
              'Se': 10, 'I': 11, 'H': 12, 'Si': 13}        # utils.py:12-13
CHARGE_VOCAB = {0: 0, 1: 1, -1: 2}                          # utils.py:14
BOND_VOCAB = {1: 0, 2: 1, 3: 2, 4: 3}                       # utils.py:15


def parse_record(atoms_string, bonds_string, scale_x=1, scale_y=1, ddx=0, ddy=0, h=128):
    """utils.py:94-112 (atoms) and 135-163 (bonds): annotation strings -> compact records.
    atoms[i] = (x, y, type, charge, hs) with hs = -1 unless 0 or 1; bonds[i] = (x, y, type, omega bin, single)."""
    atoms = []
    for atom_string in atoms_string.split(';')[:-1]:
        atom, position = atom_string.split(':')
        if len(atom) == 1:
            atom = atom.upper()
        f = position.split(',')
        x, y, charge = int(int(f[0]) * scale_x + ddx) // 4, int(int(f[1]) * scale_y + ddy) // 4, int(f[2])
        hs = int(f[3]) if len(f) == 4 else -1
        if not (0 <= x < h and 0 <= y < h):
            raise ValueError("atom at (%d, %d) falls outside the %d x %d map" % (x, y, h, h))
        atoms.append((x, y, ATOM_VOCAB.get(atom, 0), CHARGE_VOCAB.get(charge, 0), hs if hs in (0, 1) else -1))
    bonds, rhos = [], []
    delta_omega = np.pi / 30
    for bond_string in bonds_string.split(';')[:-1]:
        bond, position = bond_string.split(':')
        type_idx = BOND_VOCAB.get(int(bond), 0)
        f = position.split(',')
        x, y = int(int(f[0]) * scale_x + ddx) // 4, int(int(f[1]) * scale_y + ddy) // 4
        delta_x, delta_y = (int(f[2]) * scale_x) / 4, (int(f[3]) * scale_y) / 4
        stereo, direction = int(f[4]), int(f[5])
        if stereo == 5 or stereo == 1:
            type_idx = 4
        elif stereo == 6:
            type_idx = 5
        if delta_x < 0:
            delta_x, delta_y = -delta_x, -delta_y
        elif delta_x == 0:
            if delta_y > 0:
                direction = 1
            delta_y = -abs(delta_y)
        rho = np.sqrt(delta_x * delta_x + delta_y * delta_y)
        omega = math.atan(delta_y / (delta_x + 1e-6))
        k = int(np.floor((omega + np.pi / 2) / delta_omega))
        single = 1 if type_idx in (4, 5) else 0
        if single and direction == 1:
            k += 30
        if not (0 <= x < h and 0 <= y < h) or not (0 <= k < (60 if single else 30)):
            raise ValueError("bond at (%d, %d), omega bin %d falls outside the maps" % (x, y, k))
        bonds.append((x, y, type_idx, k, single))
        rhos.append(float(rho))
    return (np.array(atoms, dtype=np.int32).reshape(-1, 5), np.array(bonds, dtype=np.int32).reshape(-1, 5),
            np.array(rhos, dtype=np.float64))


class TargetRasterizer:
    """device-side utils.py:83-228 for a batch; `targets` may be the Trainer's static target tensors (then `run` writes
    the loss inputs in place), otherwise fresh ones are allocated with the reference collate shapes / dtypes"""

    def __init__(self, batch, h, w=None, max_atoms=256, max_bonds=256, targets=None, device="cuda", sparse=False):
        """sparse=True: the maps are zeroed ONCE; every later run() erases the pixels the previous records drew instead of zeroing
        23 MB per image (the target tensors must not be written by anybody else in between: invalidate() after that), and
        .group_flags says which 32-pixel groups of the batch hold targets of which head -- Trainer.use_sparse_targets(rasterizer)
        lets the fused heads pass skip the all-zero planes (abc_raster_desc.group_flags)."""
        w = h if w is None else w
        if not torch.cuda.is_available():
            raise L.AbcNetHipError("TargetRasterizer needs an MI355X; abcnet_amd has no CPU fallback")
        self.lib = L.load()
        B = batch
        shapes = [(B, 1, h, w), (B, 14, h, w), (B, 3, h, w), (B, 2, h, w), (B, 1, h, w), (B, 6, 60, h, w), (B, 60, h, w), (B, 60, h, w)]
        dts = [torch.float32] * 6 + [torch.float64] * 2
        if targets is None:
            targets = [torch.zeros(s, dtype=dt, device=device) for s, dt in zip(shapes, dts)]
        for t, s, dt in zip(targets, shapes, dts):
            if tuple(t.shape) != s or t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
                raise L.AbcNetHipError("raster: target %s %s does not match the contract %s %s on the device" % (tuple(t.shape), t.dtype, s, dt))
        self.targets = list(targets)
        dev = targets[0].device
        self.B, self.h, self.w, self.max_atoms, self.max_bonds = B, h, w, max_atoms, max_bonds
        # host staging (pinned) + device record buffers
        pin = dict(pin_memory=True)
        self.h_atoms = torch.zeros((B, max_atoms, 5), dtype=torch.int32, **pin)
        self.h_bonds = torch.zeros((B, max_bonds, 5), dtype=torch.int32, **pin)
        self.h_rho = torch.zeros((B, max_bonds), dtype=torch.float64, **pin)
        self.h_cnt = torch.zeros((2, B), dtype=torch.int32, **pin)
        self.d_atoms, self.d_bonds = self.h_atoms.to(dev), self.h_bonds.to(dev)
        self.d_rho, self.d_cnt = self.h_rho.to(dev), self.h_cnt.to(dev)
        d = L.RasterDesc()
        (d.t_atom, d.t_types, d.t_charges, d.t_hs, d.t_bond, d.t_btypes, d.t_rho, d.t_omega) = (t.data_ptr() for t in self.targets)
        d.B, d.h, d.w, d.max_atoms, d.max_bonds = B, h, w, max_atoms, max_bonds
        d.atoms, d.bonds, d.rho = self.d_atoms.data_ptr(), self.d_bonds.data_ptr(), self.d_rho.data_ptr()
        d.n_atoms, d.n_bonds = self.d_cnt[0].data_ptr(), self.d_cnt[1].data_ptr()
        self.sparse = bool(sparse)
        self.group_flags = None
        self._drawn = False          # sparse: do the maps hold exactly what the prev_* records say?
        if self.sparse:
            if (h * w) % 32:
                raise L.AbcNetHipError("sparse rasteriser: h * w must be a multiple of 32")
            self.group_flags = torch.zeros(B * h * w // 32, dtype=torch.int32, device=dev)
            self.p_atoms, self.p_bonds = torch.zeros_like(self.d_atoms), torch.zeros_like(self.d_bonds)
            self.p_rho, self.p_cnt = torch.zeros_like(self.d_rho), torch.zeros_like(self.d_cnt)
            d.group_flags = self.group_flags.data_ptr()
            d.prev_atoms, d.prev_bonds, d.prev_rho, d.prev_counts = (t.data_ptr() for t in (self.p_atoms, self.p_bonds, self.p_rho, self.p_cnt))
        self.d = d

    def invalidate(self):
        """sparse: somebody else wrote the target tensors -- the next run() zeroes them completely again"""
        self._drawn = False

    def load(self, records):
        """records = list of B (atoms, bonds, rho) triples from parse_record; async H2D of a few KB"""
        if len(records) != self.B:
            raise ValueError("expected %d records" % self.B)
        # the pinned staging buffers are reused: the previous load's asynchronous copies must have left them before they are
        # overwritten (a loop that never syncs runs many steps ahead of the device -- the maps would be rasterised from a
        # LATER batch's records)
        if getattr(self, "_copied", None) is not None:
            self._copied.synchronize()
        for b, (a, q, r) in enumerate(records):
            if len(a) > self.max_atoms or len(q) > self.max_bonds:
                raise ValueError("record %d has %d atoms / %d bonds (capacity %d / %d)" % (b, len(a), len(q), self.max_atoms, self.max_bonds))
            self.h_cnt[0, b], self.h_cnt[1, b] = len(a), len(q)
            if len(a):
                self.h_atoms[b, :len(a)] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32))
            if len(q):
                self.h_bonds[b, :len(q)] = torch.from_numpy(np.ascontiguousarray(q, dtype=np.int32))
                self.h_rho[b, :len(q)] = torch.from_numpy(np.ascontiguousarray(r, dtype=np.float64))
        self.d_atoms.copy_(self.h_atoms, non_blocking=True)
        self.d_bonds.copy_(self.h_bonds, non_blocking=True)
        self.d_rho.copy_(self.h_rho, non_blocking=True)
        self.d_cnt.copy_(self.h_cnt, non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record(torch.cuda.current_stream(self.d_cnt.device))

    def run(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        if self.sparse:
            self.d.incremental = 1 if self._drawn else 0
        L.check(self.lib.abc_rasterize_targets(C.byref(self.d), stream), "rasterize_targets")
        self._drawn = True
        return self.targets
