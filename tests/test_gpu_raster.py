"""GPU parity: the device-side target rasteriser (abc_rasterize_targets through the C ABI, fed by the host parser of
abcnet_amd/raster.py) against the maps the reference text itself produced (utils.py:83-228, tests/golden/raster_128.npz)
and the oracle -- bit-exact, dtypes included (order-dependent integer / constant work)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import abcnet_amd  # noqa: E402,F401
from abcnet_amd.raster import TargetRasterizer, parse_record  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402

DEV = "cuda"


def _case(gold, ci):
    na, nb, seed, sx, sy, ddx, ddy = gold["c%d_args" % ci]
    a, b = ro.random_annotations(int(na), int(nb), int(seed), size=int(512 * min(sx, sy)) - 1)
    sx = int(sx) if sx == 1 else float(sx)
    sy = int(sy) if sy == 1 else float(sy)
    return a, b, sx, sy, int(ddx), int(ddy)


def test_raster_matches_golden_and_oracle(golden_dir):
    gold = np.load(os.path.join(golden_dir, "raster_128.npz"))
    cases = [_case(gold, ci) for ci in range(3)]
    rz = TargetRasterizer(3, 128, max_atoms=64, max_bonds=80, device=DEV)
    for t in rz.targets:
        t.fill_(7)     # stale contents must be zeroed by the call
    rz.load([parse_record(*c) for c in cases])
    maps = [t.cpu().numpy() for t in rz.run()]
    torch.cuda.synchronize()
    for ci, c in enumerate(cases):
        want = ro.rasterize(*c)
        for mi in range(8):
            got = maps[mi][ci]
            assert got.dtype == want[mi].dtype and got.shape == want[mi].shape
            assert np.array_equal(got, want[mi]), (ci, mi)
            flat = got.reshape(-1)
            nz = np.flatnonzero(flat)
            assert np.array_equal(nz, gold["c%d_m%d_idx" % (ci, mi)]) and np.array_equal(flat[nz], gold["c%d_m%d_val" % (ci, mi)])


@pytest.mark.parametrize("h", [32, 96])
def test_raster_dense_overlaps_empty_image_and_trainer_buffers(h):
    """hundreds of overlapping items on a small map (order dependence everywhere), one image with no annotation at
    all, and rasterising straight into a Trainer-style target list"""
    B = 4
    recs, want = [], []
    for b in range(B):
        if b == 1:
            a, q = "", ""
        else:
            a, q = ro.random_annotations(200, 200, 500 + b, size=4 * h)
        recs.append(parse_record(a, q, h=h))
        want.append(ro.rasterize(a, q, h=h))
    shapes = [(B, 1, h, h), (B, 14, h, h), (B, 3, h, h), (B, 2, h, h), (B, 1, h, h), (B, 6, 60, h, h), (B, 60, h, h), (B, 60, h, h)]
    dts = [torch.float32] * 6 + [torch.float64] * 2
    targets = [torch.full(s, 3.0, dtype=dt, device=DEV) for s, dt in zip(shapes, dts)]
    rz = TargetRasterizer(B, h, max_atoms=200, max_bonds=200, targets=targets)
    rz.load(recs)
    rz.run()
    torch.cuda.synchronize()
    for mi in range(8):
        got = targets[mi].cpu().numpy()
        for b in range(B):
            assert np.array_equal(got[b], want[b][mi]), (b, mi)
    assert all(float(t[1].abs().sum()) == 0.0 for t in targets)


def test_raster_rejects_out_of_range_and_cpu():
    with pytest.raises(ValueError):
        parse_record("C:600,10,0;", "", h=128)
    with pytest.raises(Exception):
        TargetRasterizer(1, 32, targets=[torch.zeros(1)] * 8)


def test_sparse_rasteriser_erases_its_own_drawing_and_flags_every_target():
    """TargetRasterizer(sparse=True) over a SEQUENCE of batches (dense overlapping items, an empty image, batches of different sizes of
    molecules): after every run() the eight maps equal the oracle's for THAT batch bit for bit -- i.e. the incremental form erased
    everything the previous records had drawn, nothing else, and the first call zeroed stale contents -- and the group flags cover
    every non-zero target: a 32-pixel group whose flag bit of a head is clear holds only zeros in that head's planes (the contract
    abc_heads_fused_fwd_bwd relies on).  invalidate() makes the next run zero the maps completely again."""
    B, h = 3, 96
    shapes = [(B, 1, h, h), (B, 14, h, h), (B, 3, h, h), (B, 2, h, h), (B, 1, h, h), (B, 6, 60, h, h), (B, 60, h, h), (B, 60, h, h)]
    dts = [torch.float32] * 6 + [torch.float64] * 2
    targets = [torch.full(s, 5.0, dtype=dt, device=DEV) for s, dt in zip(shapes, dts)]      # stale contents
    rz = TargetRasterizer(B, h, max_atoms=120, max_bonds=120, targets=targets, sparse=True)
    bit_of = [0, 1, 2, 3, 4, 5, 6, 7]

    def check(batch_seed, counts):
        recs, want = [], []
        for b in range(B):
            na, nb = counts[b]
            a, q = ro.random_annotations(na, nb, batch_seed + b, size=4 * h) if na + nb else ("", "")
            recs.append(parse_record(a, q, h=h))
            want.append(ro.rasterize(a, q, h=h))
        rz.load(recs)
        rz.run()
        torch.cuda.synchronize()
        flags = rz.group_flags.cpu().numpy().astype(np.uint32).reshape(B, h * h // 32)
        for mi in range(8):
            got = targets[mi].cpu().numpy()
            for b in range(B):
                assert np.array_equal(got[b], want[b][mi]), (batch_seed, b, mi)
                # non-zero anywhere in the head's planes of a pixel group => the group's flag bit is set
                nz = (got[b].reshape(-1, h * h) != 0).any(axis=0).reshape(-1, 32).any(axis=1)
                assert not (nz & (((flags[b] >> bit_of[mi]) & 1) == 0)).any(), (batch_seed, b, mi)
        return flags

    f1 = check(100, [(100, 100), (0, 0), (30, 32)])
    assert f1[1].sum() == 0 and f1[0].any() and f1[2].any()
    check(200, [(5, 4), (60, 70), (0, 0)])          # fewer items than before: the old ones must be gone
    check(300, [(30, 32), (30, 32), (30, 32)])
    check(300, [(30, 32), (30, 32), (30, 32)])      # the same records again (what a benchmark loop does)
    for t in targets:
        t.fill_(9)                                  # somebody else wrote the maps
    rz.invalidate()
    f5 = check(400, [(10, 10), (0, 0), (1, 1)])
    assert f5[1].sum() == 0
    # sparsity the fused heads pass lives on: at 30 atoms + 32 bonds per 96 x 96 image (a 3 x 3 neighbourhood touches three 32-pixel
    # groups) about a quarter of the groups carry a target of a given head
    f = check(500, [(30, 32), (30, 32), (30, 32)])
    assert ((f >> 5) & 1).mean() < 0.4 and (f & 1).mean() < 0.4
