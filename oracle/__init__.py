"""CPU oracle for the ABC-Net U-Net hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``abc-net_amd/``) never imports this package and fails loudly when its HIP
library is missing.

What it restates (reference = /root/reference, read-only, never copied):
  * ``unet_oracle``  -- src/unet.py:6-119 and src/unet2.py:6-173 as a functional
    torch-CPU program over a name-keyed state dict (same ATen ops, same order).
  * ``loss_oracle``  -- src/train.py:95-137 (activation block + 8 loss terms +
    uncertainty weighting), parametrised on the map size.
  * ``nms_oracle``   -- src/img2smiles2.py:61-79 (peak NMS masks).
  * ``adam_oracle``  -- torch.optim.Adam as configured at src/train.py:55.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4).  The oracle is pinned against outputs of the reference itself,
imported in the development container by ``tests/golden/make_golden.py``; the
resulting fixtures live in ``tests/golden/*.npz`` and are checked by
``tests/test_oracle_golden.py``.
"""
