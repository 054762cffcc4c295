"""Oracle (test infrastructure, not product): the target rasteriser of the reference dataset,
/root/reference/src/utils.py:83-228 (MolecularImageDataset.__getitem__), restated with numpy.

From the annotation strings of one molecule
    atoms_string = "C:x,y,charge[,hs];N:x,y,charge[,hs];..."      (image pixels, 512 x 512)
    bonds_string = "order:x,y,delta_x,delta_y,stereo,direction;..."
and the augmentation offsets (scale_x, scale_y, ddx, ddy) to the 8 quarter-resolution target maps, in the order of
the reference collate_fn: atom_target [1,h,h] f32, atom_type [14,h,h] f32, atom_charge [3,h,h] f32, atom_hs [2,h,h]
f32, bond_target [1,h,h] f32, bond_type [6,60,h,h] f32, bond_rho [60,h,h] f64, bond_omega_type [60,h,h] f64.

The rasterisation is ORDER DEPENDENT (a later item's 3x3 ring overwrites an earlier item's centre), so the restatement
keeps the reference's sequence of slice assignments, including numpy's slice clipping at the upper border and the
explicit clipping at 0.  Coordinates are assumed to land inside the map (0 <= x, y < h), as they do for the reference
data; the map size (128 in the reference) is a parameter.

Pinned by tests/golden/raster_128.npz (exec of the reference text by tests/golden/make_golden.py).
"""
from __future__ import annotations

import math

import numpy as np

ATOM_VOCAB = {'<unkonw>': 0, 'C': 1, 'N': 2, 'O': 3, 'P': 4, 'F': 5, 'Cl': 6, 'S': 7, 'Br': 8, 'B': 9,
              'Se': 10, 'I': 11, 'H': 12, 'Si': 13}        # utils.py:12-13
CHARGE_VOCAB = {0: 0, 1: 1, -1: 2}                          # utils.py:14
BOND_VOCAB = {1: 0, 2: 1, 3: 2, 4: 3}                       # utils.py:15


def rasterize(atoms_string, bonds_string, scale_x=1, scale_y=1, ddx=0, ddy=0, h=128):
    atom_target = np.zeros([1, h, h], dtype=np.float32)
    atom_type = np.zeros([14, h, h], dtype=np.float32)
    atom_charge = np.zeros([3, h, h], dtype=np.float32)
    atom_hs = np.zeros([2, h, h], dtype=np.float32)
    bond_target = np.zeros([1, h, h], dtype=np.float32)
    bond_type = np.zeros([6, 60, h, h], dtype=np.float32)
    delta_omega = np.pi / 30
    bond_rho = np.zeros([60, h, h])
    bond_omega_type = np.zeros([60, h, h])

    for atom_string in atoms_string.split(';')[:-1]:
        atom, position = atom_string.split(':')
        if len(atom) == 1:
            atom = atom.upper()
        idx = ATOM_VOCAB.get(atom, 0)
        f = position.split(',')
        x, y, charge = int(int(f[0]) * scale_x + ddx) // 4, int(int(f[1]) * scale_y + ddy) // 4, int(f[2])
        hs = int(f[3]) if len(f) == 4 else -1
        xb, yb = max(x - 1, 0), max(y - 1, 0)
        atom_target[0, xb:x + 2, yb:y + 2] = 0.8
        atom_target[0, x, y] = 1
        atom_type[idx, xb:x + 2, yb:y + 2] = 0.5
        atom_type[idx, x, y] = 1
        c = CHARGE_VOCAB.get(charge, 0)
        atom_charge[c, xb:x + 2, yb:y + 2] = 0.5
        atom_charge[c, x, y] = 1
        if hs == 0 or hs == 1:
            atom_hs[hs, xb:x + 2, yb:y + 2] = 0.5
            atom_hs[hs, x, y] = 1

    def put(type_idx, k, x, y, xb, yb, rho, wrap_lo, wrap_hi):
        kb = 0 if k == 0 else k - 1
        bond_rho[kb:k + 2, xb:x + 2, yb:y + 2] = rho
        bond_omega_type[kb:k + 2, xb:x + 2, yb:y + 2] = 0.8
        bond_omega_type[k, x, y] = 1
        bond_type[type_idx, kb:k + 2, xb:x + 2, yb:y + 2] = 0.5
        bond_type[type_idx, k, x, y] = 1
        if wrap_lo and k == 0:
            bond_rho[-1, xb:x + 2, yb:y + 2] = rho
            bond_omega_type[-1, xb:x + 2, yb:y + 2] = 0.8
            bond_type[type_idx, -1, xb:x + 2, yb:y + 2] = 0.5
        if wrap_hi and k == 59:
            bond_rho[0, xb:x + 2, yb:y + 2] = rho
            bond_omega_type[0, xb:x + 2, yb:y + 2] = 0.8
            bond_type[type_idx, 0, xb:x + 2, yb:y + 2] = 0.5

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
        xb, yb = max(x - 1, 0), max(y - 1, 0)
        bond_target[0, xb:x + 2, yb:y + 2] = 0.8
        bond_target[0, x, y] = 1
        if type_idx == 4 or type_idx == 5:                      # utils.py:165-185: one direction, both wrap rules
            if direction == 1:
                k += 30
            put(type_idx, k, x, y, xb, yb, rho, True, True)
        else:                                                   # utils.py:187-221: both directions
            put(type_idx, k, x, y, xb, yb, rho, True, False)
            put(type_idx, k + 30, x, y, xb, yb, rho, False, True)
    return [atom_target, atom_type, atom_charge, atom_hs, bond_target, bond_type, bond_rho, bond_omega_type]


# the seeded annotation-string generator is plain test data, not part of the restatement: it lives with the other synthetic
# inputs (abcnet_amd/synthetic.py) so that the product side never has to import oracle/; re-exported for the tests
from abcnet_amd.synthetic import random_annotations  # noqa: E402,F401
