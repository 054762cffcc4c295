"""Synthetic inputs honouring the reference tensor contract (SURVEY.md section 8 row T;
/root/reference/src/utils.py:80-92,116-228 and src/utils_for_test.py:26-39).

No real data exists offline, so the benchmark and the tests rasterise random
atoms and bonds with the reference's value sets: centre 1, 3x3 ring 0.8 (centre
maps) / 0.5 (class maps), omega spread over +-1 bin with wrap-around, rho and
omega maps in float64 (numpy default in the reference), everything else float32.
Plain torch CPU ops, seeded; independent of the oracle.
"""
from __future__ import annotations

import numpy as np
import torch


def synthetic_images(batch: int, size: int, seed: int = 7, p: float = 0.1, in_channels: int = 1):
    """Bernoulli(p) ink in {0,1}, f32 [B,C,S,S] (ink = 1)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand((batch, in_channels, size, size), generator=g) < p).float()


def synthetic_targets(batch: int, h: int, seed: int = 1, n_atoms: int = 30, n_bonds: int = 32):
    """Returns the 8 target maps in the order of the reference collate_fn
    (utils.py:300): atom_t[B,1,h,h] f32, atom_types[B,14,h,h] f32,
    atom_charges[B,3,h,h] f32, atom_hs[B,2,h,h] f32, bond_t[B,1,h,h] f32,
    bond_types[B,6,60,h,h] f32, bond_rhos[B,60,h,h] f64, bond_omega[B,60,h,h] f64."""
    g = torch.Generator().manual_seed(seed)
    ri = lambda hi, n: torch.randint(0, hi, (n,), generator=g).tolist()
    at = torch.zeros(batch, 1, h, h)
    aty = torch.zeros(batch, 14, h, h)
    ach = torch.zeros(batch, 3, h, h)
    ahs = torch.zeros(batch, 2, h, h)
    bt = torch.zeros(batch, 1, h, h)
    bty = torch.zeros(batch, 6, 60, h, h)
    rho = torch.zeros(batch, 60, h, h, dtype=torch.float64)
    om = torch.zeros(batch, 60, h, h, dtype=torch.float64)
    for b in range(batch):
        xs, ys = ri(h, n_atoms), ri(h, n_atoms)
        ty, ch, hs = ri(14, n_atoms), ri(3, n_atoms), ri(3, n_atoms)
        for x, y, t, c, k in zip(xs, ys, ty, ch, hs):
            x0, y0 = max(x - 1, 0), max(y - 1, 0)
            at[b, 0, x0:x + 2, y0:y + 2] = 0.8
            at[b, 0, x, y] = 1
            aty[b, t, x0:x + 2, y0:y + 2] = 0.5
            aty[b, t, x, y] = 1
            ach[b, c, x0:x + 2, y0:y + 2] = 0.5
            ach[b, c, x, y] = 1
            if k < 2:  # hs == -1 (unknown) leaves the map empty, utils.py:121
                ahs[b, k, x0:x + 2, y0:y + 2] = 0.5
                ahs[b, k, x, y] = 1
        xs, ys = ri(h, n_bonds), ri(h, n_bonds)
        tys, oms = ri(6, n_bonds), ri(60, n_bonds)
        rs = (torch.rand(n_bonds, generator=g, dtype=torch.float64) * 9 + 3).tolist()
        for x, y, t, o, r in zip(xs, ys, tys, oms, rs):
            x0, y0 = max(x - 1, 0), max(y - 1, 0)
            bt[b, 0, x0:x + 2, y0:y + 2] = 0.8
            bt[b, 0, x, y] = 1
            bins = [o] if t >= 4 else [o % 30, o % 30 + 30]  # non-stereo bonds mark both directions
            for ob in bins:
                for d in (-1, 0, 1):
                    k = (ob + d) % 60
                    rho[b, k, x0:x + 2, y0:y + 2] = r
                    om[b, k, x0:x + 2, y0:y + 2] = 0.8
                    bty[b, t, k, x0:x + 2, y0:y + 2] = 0.5
            for ob in bins:
                om[b, ob, x, y] = 1
                bty[b, t, ob, x, y] = 1
    return [at, aty, ach, ahs, bt, bty, rho, om]


def correlated_logits(targets, seed: int = 19, centre_noise: float = 1.2):
    """Head logits [B,{1,14,3,2,1,360,60,60},h,h] f32 that CORRELATE with the given target maps (so that the
    training meters of train.py:145-215 are non-trivial): target-shaped signal plus seeded Gaussian noise; the
    omega logits are quantised to quarter steps so that exact ties between neighbouring bins occur."""
    g = torch.Generator().manual_seed(seed)
    t_at, t_ty, t_ch, t_hs, t_bt, t_bty, t_rho, t_om = targets
    B, _, h, w = t_at.shape
    rn = lambda c: torch.randn((B, c, h, w), generator=g)
    return [
        5.0 * t_at - 2.5 + centre_noise * rn(1),
        4.0 * t_ty + rn(14),
        4.0 * t_ch + rn(3),
        4.0 * t_hs + rn(2),
        5.0 * t_bt - 2.5 + centre_noise * rn(1),
        4.0 * t_bty.reshape(B, 360, h, w) + rn(360),
        (t_rho + 0.5 * rn(60).double()).float(),
        torch.round((5.0 * t_om.float() - 2.5 + rn(60)) * 4) / 4,
    ]


def random_annotations(n_atoms, n_bonds, seed, size=512):
    """seeded annotation strings in the reference's format, covering the branches: unknown and two-letter elements,
    3- and 4-field atoms, hs in {-1,0,1,2}, all bond orders and stereo codes, both directions, vertical bonds
    (delta_x == 0), border positions (x or y == 0 / last), overlapping neighbourhoods, omega bins 0 and 29"""
    rng = np.random.RandomState(seed)
    elems = ['C', 'N', 'O', 'P', 'F', 'Cl', 'S', 'Br', 'B', 'Se', 'I', 'H', 'Si', 'Xx', 'c', 'n']
    atoms = []
    for i in range(n_atoms):
        x, y = int(rng.randint(0, size)), int(rng.randint(0, size))
        if i % 7 == 0:
            x = [0, size - 1, 3, size - 4][(i // 7) % 4]
        if i % 11 == 0:
            y = [0, size - 1][(i // 11) % 2]
        s = "%s:%d,%d,%d" % (elems[int(rng.randint(0, len(elems)))], x, y, int(rng.choice([0, 0, 0, 1, -1, 2])))
        if rng.rand() < 0.7:
            s += ",%d" % int(rng.choice([0, 1, 2, -1]))
        atoms.append(s)
    bonds = []
    for i in range(n_bonds):
        x, y = int(rng.randint(0, size)), int(rng.randint(0, size))
        if i % 9 == 0:
            x = [0, size - 1][(i // 9) % 2]
        dx, dy = int(rng.randint(-40, 41)), int(rng.randint(-40, 41))
        if i % 5 == 0:
            dx = 0
        if i % 13 == 0:
            dy = 0
        if dx == 0 and dy == 0:
            dy = 7
        if i % 17 == 0:
            dx, dy = 1, -40   # steep: omega bin 0
        if i % 19 == 0:
            dx, dy = 1, 40    # omega bin 29
        bonds.append("%d:%d,%d,%d,%d,%d,%d" % (int(rng.choice([1, 2, 3, 4, 7])), x, y, dx, dy, int(rng.choice([0, 0, 1, 5, 6])),
                                                int(rng.choice([0, 1]))))
    return ";".join(atoms) + ";", ";".join(bonds) + ";"


# ----------------------------------------------------------------------------------------------------------------------
# Drawn molecules: images whose ink DEPENDS on the annotations (the Bernoulli images above carry no information about the
# targets, so nothing can be learnt from them).  Used to train a network for a few hundred steps on the device: trained
# weights are what an accuracy statement about the bf16 / fp8 inference graphs has to be measured on (at random
# initialisation a BatchNorm + ReLU network amplifies any perturbation by ~1.2x per layer -- 1e-3 relative weight noise moves
# the output maps by 8 % of their spread -- so there every reduced-precision arithmetic, the reference's own autocast run
# included, looks 20 % wrong).

_ELEMS = ['C', 'N', 'O', 'P', 'F', 'Cl', 'S', 'Br', 'B', 'Se', 'I', 'H', 'Si']


def _glyphs():
    """one fixed 9 x 9 binary glyph per element (seeded; denser than a line so that the network can tell them apart)"""
    rng = np.random.RandomState(20240923)
    g = (rng.rand(len(_ELEMS), 9, 9) < 0.55).astype(np.float32)
    g[:, 0, :] = g[:, -1, :] = g[:, :, 0] = g[:, :, -1] = 0
    return g


_GLYPHS = _glyphs()


def _stamp(img, x, y, r):
    img[max(x - r, 0):x + r + 1, max(y - r, 0):y + r + 1] = 1.0


def _line(img, p, q, width=1, taper=None, dash=False):
    """ink along p -> q (row, col); width = half-width in pixels; taper: half-width grows from 0 to `taper` (a wedge)"""
    n = int(max(abs(q[0] - p[0]), abs(q[1] - p[1])) * 2) + 2
    t = np.linspace(0.0, 1.0, n)
    xs = np.rint(p[0] + (q[0] - p[0]) * t).astype(int)
    ys = np.rint(p[1] + (q[1] - p[1]) * t).astype(int)
    S = img.shape[0]
    for i, (x, y) in enumerate(zip(xs, ys)):
        if dash and (i // 6) % 2:
            continue
        if 0 <= x < S and 0 <= y < S:
            _stamp(img, x, y, int(round(taper * t[i])) if taper is not None else width)


def drawn_molecules(batch, size, seed, n_atoms=(8, 22), margin=20, min_dist=30, max_bond=96):
    """seeded line drawings + their annotation strings in the reference's format (utils.py:94-163: atoms
    "El:x,y,charge[,hs];", bonds "order:x,y,dx,dy,stereo,direction;" with (x, y) the bond centre and (dx, dy) the half
    vector; x indexes rows).  Returns (images f32 [B,1,S,S] in {0,1}, ink = 1; list of (atoms_string, bonds_string))."""
    rng = np.random.RandomState(seed)
    imgs = np.zeros((batch, 1, size, size), dtype=np.float32)
    notes = []
    for b in range(batch):
        img = imgs[b, 0]
        want = int(rng.randint(n_atoms[0], n_atoms[1] + 1))
        pts = []
        for _ in range(want * 30):
            if len(pts) == want:
                break
            c = rng.randint(margin, size - margin, size=2)
            c = (c // 4) * 4 + 2          # cell centres: the 1/4-resolution target cell is unambiguous
            if all((c[0] - p[0]) ** 2 + (c[1] - p[1]) ** 2 >= min_dist ** 2 for p in pts):
                pts.append((int(c[0]), int(c[1])))
        n = len(pts)
        P = np.array(pts)
        # bonds: every atom to its (up to) two nearest neighbours within max_bond pixels, whose midpoint cell is free
        pairs = set()
        d2 = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
        for i in range(n):
            for j in np.argsort(d2[i])[1:3]:
                if d2[i, j] <= max_bond ** 2:
                    pairs.add((min(i, int(j)), max(i, int(j))))
        bonds = []
        for (i, j) in sorted(pairs):
            a, c = P[i], P[j]
            order = int(rng.choice([1, 1, 1, 2, 2, 3, 4]))
            stereo = int(rng.choice([0, 0, 0, 0, 1, 6])) if order == 1 else 0
            direction = int(rng.randint(0, 2))
            mid = (a + c) // 2
            half = (c - a) // 2
            if half[0] == 0 and half[1] == 0:
                continue
            u = np.array([-(c - a)[1], (c - a)[0]], dtype=np.float64)
            u /= np.linalg.norm(u) + 1e-9
            if stereo in (1, 6):
                p, q = (a, c) if direction == 0 else (c, a)
                _line(img, p, q, taper=4, dash=(stereo == 6))
            elif order == 1:
                _line(img, a, c)
            elif order == 2:
                _line(img, a + 3 * u, c + 3 * u)
                _line(img, a - 3 * u, c - 3 * u)
            elif order == 3:
                _line(img, a, c)
                _line(img, a + 5 * u, c + 5 * u)
                _line(img, a - 5 * u, c - 5 * u)
            else:                      # aromatic: a full and a dashed line
                _line(img, a + 3 * u, c + 3 * u)
                _line(img, a - 3 * u, c - 3 * u, dash=True)
            bonds.append("%d:%d,%d,%d,%d,%d,%d" % (order, mid[0], mid[1], half[0], half[1], stereo, direction))
        atoms = []
        for (x, y) in pts:
            e = int(rng.choice(len(_ELEMS), p=[0.5] + [0.5 / (len(_ELEMS) - 1)] * (len(_ELEMS) - 1)))
            charge = int(rng.choice([0, 0, 0, 0, 1, -1]))
            hs = int(rng.choice([-1, 0, 0, 1]))
            if e != 0 or charge != 0:      # a hetero atom or a charged carbon: clear the junction, draw the glyph
                img[x - 7:x + 8, y - 7:y + 8] = 0.0
                img[x - 4:x + 5, y - 4:y + 5] = _GLYPHS[e]
                if charge == 1:
                    img[x - 7, y + 5:y + 8] = 1.0
                    img[x - 8:x - 5, y + 6] = 1.0
                elif charge == -1:
                    img[x - 7, y + 5:y + 8] = 1.0
                if hs == 1:
                    img[x + 6:x + 8, y - 1:y + 2] = 1.0
            s = "%s:%d,%d,%d" % (_ELEMS[e], x, y, charge)
            if hs >= 0:
                s += ",%d" % hs
            atoms.append(s)
        notes.append((";".join(atoms) + ";", ";".join(bonds) + ";" if bonds else ""))
    return torch.from_numpy(imgs), notes
