"""Oracle (test infrastructure, not product): the candidate extraction of the reference inference driver,
/root/reference/src/img2smiles2.py:113-191, restated as a function of one image's head maps.

What the reference does between the NMS (img2smiles2.py:61-79) and the graph assembly / RDKit stage
(img2smiles2.py:193-344, out of scope):

  bonds (128-169): for every bond-centre peak (x, y) in raster order, for every omega bin k whose RAW
      logit is non-zero (`bond_omega_img[:, x, y].nonzero()` -- the NMS'd omega map computed at 75-79 is
      not consulted), keep the bin unless the opposite direction wins:
          k <= 28 : drop if v[k] <  max(v[k+29], v[k+30])
          k == 29 : drop if v[29] < v[58] or v[29] < v[0]
          k == 30 : drop if v[30] <= v[0] or v[30] <= v[59]
          k >= 31 : drop if v[k] <= max(v[k-31], v[k-30])
      and emit (x, y, k, argmax over the 6 bond types at bin k, |rho|[k, x, y]);
  atoms (171-191): for every atom-centre peak in raster order, skip it if an ALREADY ACCEPTED atom lies
      within squared distance < 4, else emit (x, y, argmax type, argmax charge, argmax hs).

x is the row index and y the column index, as in the reference (`x, y = position`).
Pinned by tests/golden/decode_128.npz (exec of the reference text by tests/golden/make_golden.py).
"""
from __future__ import annotations

import torch


def _keep_bin(v, k):
    if k <= 28:
        return not (v[k] < max(v[k + 29], v[k + 30]))
    if k == 29:
        return not (v[29] < v[58] or v[29] < v[0])
    if k == 30:
        return not (v[30] <= v[0] or v[30] <= v[59])
    return not (v[k] <= max(v[k - 31], v[k - 30]))


def extract(atom_mask, bond_mask, types, charges, hs, btypes, rho_abs, omega):
    """one image: atom_mask/bond_mask [h,w] (NMS output), types [14,h,w], charges [3,h,w], hs [2,h,w],
    btypes [360,h,w] (channel = type*60 + bin), rho_abs [60,h,w], omega [60,h,w] raw logits.
    Returns (atoms [n,5] int64: x,y,type,charge,hs ; bonds [m,4] int64: x,y,bin,type ; rho [m] f32)."""
    h, w = atom_mask.shape
    bt = btypes.reshape(6, 60, h, w)
    bonds, rhos = [], []
    for pos in bond_mask.nonzero(as_tuple=False).tolist():
        x, y = pos
        v = omega[:, x, y].tolist()
        for k in range(60):
            if v[k] == 0.0:
                continue
            if not _keep_bin(v, k):
                continue
            bonds.append([x, y, k, int(bt[:, k, x, y].argmax().item())])
            rhos.append(rho_abs[k, x, y].item())
    atoms = []
    for pos in atom_mask.nonzero(as_tuple=False).tolist():
        x, y = pos
        if any((x - a[0]) ** 2 + (y - a[1]) ** 2 < 4 for a in atoms):
            continue
        atoms.append([x, y, int(types[:, x, y].argmax().item()), int(charges[:, x, y].argmax().item()),
                      int(hs[:, x, y].argmax().item())])
    return (torch.tensor(atoms, dtype=torch.int64).reshape(-1, 5), torch.tensor(bonds, dtype=torch.int64).reshape(-1, 4),
            torch.tensor(rhos, dtype=torch.float32))
