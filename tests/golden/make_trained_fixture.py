#!/usr/bin/env python3
"""Train the FROZEN accuracy fixture once, on an MI355X:

    gpurun -- 'python tests/golden/make_trained_fixture.py'          (writes gpurun_out/trained_unet_state.npz + .json)
    cp gpurun_out/trained_unet_state.npz tests/golden/               (then: python tests/golden/make_golden.py trained, in the dev container)

unet.py trained for 3000 steps on drawn molecules by tests/trained_fixture.py (the reference's model, loss and optimiser on the
HIP path), then every floating-point PARAMETER rounded to bf16 -- the stored fixture IS the rounded network: its weights are
exactly representable in the device's compute type and in f32, so the reference, the oracle and every device graph start from
bit-identical weights -- and stored as the 16 upper bits (uint16).  BatchNorm running statistics stay f32, the step counters int64.
The weights are INPUT data of tests (tests/test_gpu_trained.py); the expected outputs are made from them by the reference itself
(make_golden.py trained -> trained_unet.npz)."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import trained_fixture as TF  # noqa: E402

STEPS = 3000


def pack_state(sd):
    """state_dict -> dict of numpy arrays: 'p:<key>' uint16 (bf16 bits of a parameter), 'b:<key>' f32 / int64 (buffers)"""
    out = {}
    for k, v in sd.items():
        v = v.detach().cpu()
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            out["b:" + k] = v.numpy()
        else:
            out["p:" + k] = v.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    return out


def unpack_state(path):
    """the frozen state as an f32 state_dict in the reference's layout (CPU tensors)"""
    z = np.load(path)
    sd = {}
    for name in z.files:
        kind, k = name.split(":", 1)
        a = z[name]
        if kind == "p":
            sd[k] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()
        else:
            sd[k] = torch.from_numpy(a.copy())
    return sd


if __name__ == "__main__":
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    m, sd, info = TF.train_unet(steps=STEPS, log=lambda s: print(s, file=sys.stderr, flush=True))
    ref_keys = list(m.state_dict().keys())
    packed = pack_state(sd)
    np.savez_compressed(os.path.join(out_dir, "trained_unet_state.npz"), **packed)
    back = unpack_state(os.path.join(out_dir, "trained_unet_state.npz"))
    assert list(back.keys()) == ref_keys, "key order"
    worst = max(((back[k].float() - sd[k].cpu().float()).abs().max() / (sd[k].cpu().float().abs().max() + 1e-30)).item() for k in ref_keys if k.endswith("weight"))
    rec = {"how": "python tests/golden/make_trained_fixture.py on an MI355X: tests/trained_fixture.train_unet(steps=%d), parameters rounded to bf16" % STEPS,
           "steps": STEPS, "loss": info["loss"], "meters": info["meters"], "train_s": info["train_s"], "worst_relative_rounding": worst,
           "bytes": os.path.getsize(os.path.join(out_dir, "trained_unet_state.npz"))}
    with open(os.path.join(out_dir, "trained_unet_state.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec), "total %.1f s" % (time.time() - t0))
