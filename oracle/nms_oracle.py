"""Oracle (test infrastructure, not product): inference-time peak NMS of
/root/reference/src/img2smiles2.py:61-79, restated with the map size taken from
the tensors instead of the hard-coded 128.

  atom/bond centre mask : 3x3 local max (stride 1, -inf padding) AND logit > -1
  rho                   : |rho|
  omega mask            : circular 3-tap local max over the 60 bins AND logit > -1

Pinned by tests/golden/nms_*.npz (exec-slice of the reference text).
"""
import torch
import torch.nn.functional as F


def center_mask(logit):
    pooled = F.max_pool2d(logit, kernel_size=3, stride=1, padding=1)
    return (pooled == logit) * (logit > -1).float()


def omega_mask(om):
    n, c, h, w = om.shape
    ring = torch.cat([om[:, c - 1:], om, om[:, :1]], dim=1).permute(0, 2, 3, 1).reshape(-1, h * w, c + 2)
    m = F.max_pool1d(ring, stride=1, kernel_size=3, padding=0).reshape(-1, h, w, c).permute(0, 3, 1, 2)
    return ((m == om) * (om > -1)).float()


def nms(atom_t, bond_t, rho, omega):
    return center_mask(atom_t), center_mask(bond_t), torch.abs(rho), omega_mask(omega)
