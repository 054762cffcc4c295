"""Oracle (test infrastructure, not product): the activation + loss block of the
reference trainer, /root/reference/src/train.py:95-137, restated as a function.

Differences from the reference text (behaviour preserved):
  * the hard-coded 128x128 in the bond-type view (train.py:101) is the map size
    of the given tensors;
  * ``s`` is passed in (reference reads model.module.s, train.py:127-135).

Head order (train.py:94): atom_t, atom_types, atom_charges, atom_hs, bond_t,
bond_types(6x60), bond_rhos(60), bond_omega(60).  Target dtypes follow
src/utils.py:83-92: rho and omega targets are float64, the rest float32, so the
rho/omega terms and the total come out float64 exactly as in the reference.

Pinned by tests/golden/loss_*.npz (exec-slice of the reference file text run in
the dev container by tests/golden/make_golden.py).
"""
from __future__ import annotations

import torch

ATOM_TYPE_WEIGHTS = [1, 0.1, 0.1, 0.1, 1, 1, 1, 1, 1, 10, 10, 10, 10, 10]  # train.py:16
LO, HI = 1e-5, 1 - 1e-5
# index into s for each term, and the factor in front of exp(-s): train.py:127-135
S_INDEX = {"atom_t": 0, "bond_t": 1, "atom_types": 2, "atom_charges": 3, "bond_types": 4,
           "bond_rhos": 6, "bond_omega": 7, "atom_hs": 9}
S_EXPFAC = {"bond_rhos": 0.5}
TERM_ORDER = ["atom_t", "bond_t", "atom_types", "atom_charges", "bond_types", "bond_rhos", "bond_omega", "atom_hs"]


def activations(preds):
    """train.py:95-105."""
    a_t, a_ty, a_ch, a_hs, b_t, b_ty, b_rho, b_om = preds
    h, w = b_ty.shape[-2], b_ty.shape[-1]
    cl = lambda v: torch.clamp(v, LO, HI)
    return (
        cl(torch.sigmoid(a_t)),
        cl(torch.softmax(a_ty, dim=1)),
        cl(torch.softmax(a_ch, dim=1)),
        cl(torch.softmax(a_hs, dim=1)),
        cl(torch.sigmoid(b_t)),
        cl(torch.softmax(b_ty.view(-1, 6, 60, h, w), dim=1)),
        torch.abs(b_rho),
        cl(torch.sigmoid(b_om)),
    )


def _center_focal(t, p):
    # train.py:107-108 / 116-117: penalty-reduced focal, normalised by #(t==1)
    pos = (t == 1).float()
    num = torch.sum(-pos * (1 - p) ** 2 * torch.log(p) - (1 - t) ** 4 * p ** 2 * torch.log(1 - p))
    return num / torch.sum(t == 1)


def _class_focal(t, p, w=None, eps=0.0):
    # train.py:109, 111, 114, 119
    body = t * (1 - p) ** 2 * torch.log(p)
    if w is not None:
        body = w * body
    return torch.sum(-body) / (torch.sum(t) + eps) if eps else torch.sum(-body) / torch.sum(t)


def loss_terms(preds, targets, device=None):
    """Unweighted terms (before the uncertainty factors)."""
    p_at, p_ty, p_ch, p_hs, p_bt, p_bty, p_rho, p_om = activations(preds)
    t_at, t_ty, t_ch, t_hs, t_bt, t_bty, t_rho, t_om = targets
    w = torch.tensor(ATOM_TYPE_WEIGHTS, dtype=torch.float32, device=t_ty.device).reshape(1, 14, 1, 1)
    terms = {}
    terms["atom_t"] = _center_focal(t_at, p_at)
    terms["atom_types"] = _class_focal(t_ty, p_ty, w)
    terms["atom_charges"] = _class_focal(t_ch, p_ch)
    terms["atom_hs"] = _class_focal(t_hs, p_hs, eps=0.1)
    terms["bond_t"] = _center_focal(t_bt, p_bt)
    terms["bond_types"] = _class_focal(t_bty, p_bty)
    # train.py:121
    terms["bond_rhos"] = torch.sum(torch.abs(p_rho - t_rho) * torch.sum(t_bty, dim=1)) / torch.sum(t_bty)
    # train.py:124-125
    wpix = torch.sum(t_om, dim=1, keepdim=True)
    inner = (t_om == 1) * ((1 - p_om) ** 2) * torch.log(p_om) + (1 - t_om) ** 4 * (p_om ** 2) * torch.log(1 - p_om)
    terms["bond_omega"] = -torch.sum(wpix * inner) / torch.sum(t_om)
    return terms


def abc_loss(preds, targets, s):
    """Total loss of train.py:137 and the dict of WEIGHTED terms."""
    terms = loss_terms(preds, targets)
    weighted = {}
    for k in TERM_ORDER:
        si = s[S_INDEX[k]]
        weighted[k] = terms[k] * (S_EXPFAC.get(k, 1.0) * torch.exp(-si) + si)
    total = (weighted["atom_t"] + weighted["bond_t"] + weighted["atom_types"] + weighted["atom_charges"]
             + weighted["bond_types"] + weighted["bond_rhos"] + weighted["bond_omega"] + weighted["atom_hs"])
    return total, weighted, terms
