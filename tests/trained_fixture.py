"""Test infrastructure: a unet.py TRAINED for a few hundred steps on the device, for accuracy statements about the reduced-precision
inference graphs (tests/test_gpu_trained.py).

Why: at random initialisation a BatchNorm + ReLU network amplifies perturbations by ~1.2x per layer (23 layers: ~80x; measured on
tests/golden/calibrated_unet.npz: 1e-3 relative weight noise moves the fp32 output maps by 8 % of their spread, the reference's own
bf16 autocast run by 22 %), so on random weights every bf16 / fp8 deviation is dominated by that amplification and says nothing
about the kernels.  Training tames it and produces what config 5 (img2smiles2.py:42-79) is run on in practice: peaked heat maps,
heavy-tailed activations.

The recipe is deterministic (seeded data, seeded initialisation, bit-reproducible device step): drawn molecules
(abcnet_amd.synthetic.drawn_molecules: ink that depends on the annotations) -> reference-format annotation strings -> parse_record
-> the device rasteriser into the Trainer's target buffers -> Trainer.step (train.py:83-141 on the HIP path, bf16, dropout 0.2,
Adam).  The weights it returns are INPUTS of the tests; the expected outputs come from the oracle (pinned to the reference) on the
host.
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import abcnet_amd  # noqa: E402,F401
from abcnet_amd.raster import TargetRasterizer, parse_record  # noqa: E402
from abcnet_amd.synthetic import drawn_molecules  # noqa: E402

HEADS = [1, 14, 3, 2, 1, 360, 60, 60]
DEV = "cuda"


def train_unet(steps=3000, batch=16, size=384, lr=5e-4, lr_drop=(0.7, 1e-4), pool=32, seed=1234, log_every=250, log=None):
    """returns (model on the device in eval mode, info).  lr_drop = (fraction of the steps, new learning rate): a NEW Adam at
    the drop, as train.py:84-85 does."""
    from abcnet_amd.train import Trainer
    from abcnet_amd.unet import UNet
    m = UNet(1, HEADS, dtype="bf16", dropout_p=0.2)
    m.reset_parameters(seed=seed)
    # CenterNet's prior initialisation of the three sigmoid heads (atom centres, bond centres, omega): with torch's default
    # bias every pixel starts at p = 0.5, the focal loss of the ~10^4 negatives per positive drives the whole map far
    # negative in the first steps and the positives take thousands of steps to come back; p = 0.1 at the start avoids that
    # phase.  The initial weights are the fixture's to choose -- the model, the loss and the optimiser are the reference's.
    with torch.no_grad():
        sd0 = m.state_dict()
        for i in (0, 4, 7):
            sd0["out_modules.%d.conv2.bias" % i].fill_(-2.19)
        m.load_state_dict(sd0)
    m = m.to(DEV)
    tr = Trainer(m, batch, size, size, lr=lr, use_graph=True, metrics=True)
    rz = TargetRasterizer(batch, size // 4, max_atoms=64, max_bonds=64, targets=tr.targets)
    t0 = time.time()
    imgs, recs = [], []
    for p in range(pool):
        x, notes = drawn_molecules(batch, size, seed=50000 + p)
        imgs.append(x)
        recs.append([parse_record(a, b, h=size // 4) for a, b in notes])
    imgs = torch.stack(imgs).to(DEV)
    # the whole pool's records on the device (the loop below must not reuse pinned host staging without a sync)
    drec = []
    for p in range(pool):
        rz.load(recs[p])
        torch.cuda.synchronize()
        drec.append((rz.d_atoms.clone(), rz.d_bonds.clone(), rz.d_rho.clone(), rz.d_cnt.clone()))
    info = {"gen_s": time.time() - t0, "loss": []}
    drop_at = int(lr_drop[0] * steps) if lr_drop else -1
    t0 = time.time()
    for it in range(steps):
        if it == drop_at:
            tr.reset_optimizer(lr_drop[1])
        p = it % pool
        tr.eng.img.copy_(imgs[p].reshape(tr.eng.img.shape))
        for dst, src in zip((rz.d_atoms, rz.d_bonds, rz.d_rho, rz.d_cnt), drec[p]):
            dst.copy_(src)
        rz.run()
        tr.step()
        if (it + 1) % log_every == 0 or it == 0:
            torch.cuda.synchronize()
            lv = tr.loss_value()
            info["loss"].append((it + 1, lv["total"]))
            mt = tr.metrics.result()
            if log:
                log("   terms " + " ".join("%s %.3f" % (k, v) for k, v in lv.items() if k != "total"))
                log("step %d loss %.4f  atom P/R %.3f/%.3f  bond P/R %.3f/%.3f  types acc %.3f  (%.1f s)" % (
                    it + 1, lv["total"], mt["atom_targets_precision"]["avg"], mt["atom_targets_recall"]["avg"],
                    mt["bond_targets_precision"]["avg"], mt["bond_targets_recall"]["avg"], mt["atom_types_acc"]["avg"], time.time() - t0))
            info["meters"] = {k: v["avg"] for k, v in mt.items()}      # (the meters over the last log_every steps)
            tr.metrics.reset()
    torch.cuda.synchronize()
    info["train_s"] = time.time() - t0
    if "meters" not in info:
        info["meters"] = {k: v["avg"] for k, v in tr.metrics.result().items()}
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    del tr, rz
    torch.cuda.empty_cache()
    m.eval()
    return m, sd, info
