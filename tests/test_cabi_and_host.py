"""CPU tier: the C-ABI library loads and exports every symbol include/abcnet_hip.h declares, the ctypes
mirror structs match the library, and the host-side logic (architecture table, plan builder, bucket planner,
sampler, dropout mirror, gloo gradient reducer at world size 2) behaves.  No kernel is launched here."""
import os
import re
import sys

import pytest
import torch

import abcnet_amd  # noqa: F401
from abcnet_amd import _lib as L
from abcnet_amd import arch
from abcnet_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADS = [1, 14, 3, 2, 1, 360, 60, 60]


def test_library_exports_every_declared_symbol():
    lib = L.load()  # raises on a missing symbol or a struct-size mismatch
    hdr = open(os.path.join(ROOT, "include", "abcnet_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(abc_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export %s" % name
        assert name in L.SYMBOLS, "binding does not cover %s" % name
    assert set(L.SYMBOLS) <= declared | {"abc_sizeof"}


def test_status_and_error_text_without_gpu():
    lib = L.load()
    assert lib.abc_version() >= 100
    assert lib.abc_conv_chunk(L.BF16, 128) == 32 and lib.abc_conv_chunk(L.BF16, 16) == 16 and lib.abc_conv_chunk(L.F32, 1) == 16
    d = L.ConvDesc()
    d.ntaps = 0
    assert lib.abc_conv_stat_blocks(d) == -1  # invalid descriptor is refused on the host, nothing is launched
    assert b"ntaps" in lib.abc_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.AbcNetHipError):
        L.load()


@pytest.mark.parametrize("variant,n,nparam", [("unet", 261, 10698575), ("unet2", 353, 11177340)])
def test_arch_table_counts(variant, n, nparam):
    t = arch.state_table(variant, 1, HEADS)
    assert len(t) == n
    assert sum(arch.numel(s) for _, s, r in t if r == "param") == nparam


def test_engine_plan_builds_without_gpu():
    """the static launch plan (descriptor construction + host-side geometry queries) needs no device"""
    from abcnet_amd.engine import Engine
    from abcnet_amd.unet import UNet
    m = UNet(1, HEADS)
    m._flat_grad = torch.zeros_like(m._flat.data)
    lay = (m._lay_p, m._lay_b, m._lay_c)
    ev = Engine("unet", 1, HEADS, m._flat.data, m._flat_grad, m._flat_buf, m._counters, lay, 2, 64, 64, "bf16", False, device="cpu")
    tr = Engine("unet", 1, HEADS, m._flat.data, m._flat_grad, m._flat_buf, m._counters, lay, 2, 64, 64, "fp32", True, device="cpu")
    # (bf16: the eight heads' 1x1 convolutions are one batched launch, the fp32 parity mode launches them one by one; the
    #  training plan starts with the dropout-step counter)
    # eval: the 34 BatchNorm coefficient refreshes ride with the weight packing, not in the forward plan; bf16 also gathers
    # the eight conv1 biases of the heads for their ONE merged 128 -> 8 x 128 convolution
    assert len(ev.pack_ops) == 34 + 1 + 1 and len(tr.pack_ops) == 1
    assert sum(1 for op in ev.fwd_ops if "out_modules" in op[2] and "conv1" in op[2]) == 1
    assert sum(1 for op in tr.fwd_ops if "out_modules" in op[2] and "conv1" in op[2]) == 8
    # (train adds 34 BatchNorm finalisations, the heads' eight as one batched launch; fp32 keeps the 8 conv1 launches)
    # (bf16 runs the four phases of a ConvTranspose2d in ONE pass over the input, abc_convt_fused_fwd -- or, where that kernel does not serve
    #  the layer, as one batched launch of four convolutions, abc_conv_fwd_batch; fp32 launches the four phases one by one)
    nb = sum(1 for op in ev.fwd_ops if "(4 phases" in op[2])
    assert nb == 3 and sum(1 for op in ev.fwd_ops if "(4 phases, one pass)" in op[2]) == 3 and not any("(4 phases" in op[2] for op in tr.fwd_ops)
    assert len(ev.bwd_ops) == 0 and len(ev.fwd_ops) + 3 * nb + 7 + (34 - 7) + 7 + 1 == len(tr.fwd_ops) > 80 and len(tr.bwd_ops) > 200
    assert tr.fwd_ops[0][2] == "dropout step"
    assert sum(op[4]["flops"] for op in ev.fwd_ops) == sum(op[4]["flops"] for op in tr.fwd_ops)
    # every learnable tensor except the conv biases in front of a BatchNorm (exactly-zero gradient) and s
    # (written by the loss kernel) is finalised by some backward op
    written = set(w for op in tr.bwd_ops for w in op[3])
    missing = [n for n in m._lay_p if n not in written]
    assert missing[0] == "s" and all(n.endswith(("double_conv.0.bias", "double_conv.3.bias", "conv1.bias")) for n in missing[1:])
    # algorithmic forward flops of the plan == the survey's figure for unet @ 64x64 scaled (52.86 GF @ 384^2 per image)
    flops = sum(op[4]["flops"] for op in tr.fwd_ops)
    per_img_384 = flops / 2 * (384 / 64) ** 2
    assert abs(per_img_384 - 52.86e9) / 52.86e9 < 0.02


def test_sizes_that_are_not_multiples_of_32_build_for_unet_only():
    """unet.py:51-56: the transposed convs meet skip tensors of 2n or 2n+1 rows; the plan lowers both (per axis: crop the first
    row, or not).  unet2.py's plan builds at such sizes too (unet2.py:104-109); below 32 pixels five poolings leave nothing."""
    from abcnet_amd.engine import Engine, convT_phase_taps, convT_pack_parity
    from abcnet_amd.unet import UNet
    from abcnet_amd.unet2 import UNet as UNet2
    m = UNet(1, HEADS)
    m._flat_grad = torch.zeros_like(m._flat)
    lay = (m._lay_p, m._lay_b, m._lay_c)
    e = Engine("unet", 1, HEADS, m._flat, m._flat_grad, m._flat_buf, m._counters, lay, 1, 72, 88, "fp32", True, device="cpu")
    assert (e.h, e.w) == (18, 22) and [tuple(t.shape[2:]) for t in e.logits] == [(18, 22)] * 8
    ups = {r.cname: r for r in e.recs if r.kind == "convT"}
    # 72: 72 36 18 9 4 2 -> up1 (2 -> 4: crop), up2 (4 -> 9: no crop), up3 (9 -> 18: crop); 88: 88 44 22 11 5 2 -> no crop, no crop, crop
    assert ups["up1.up"].taps_bwd[0] == (-1, 0) and ups["up2.up"].taps_bwd[0] == (0, 0) and ups["up3.up"].taps_bwd[0] == (-1, -1)
    assert convT_phase_taps(0, 1, crop_y=False, crop_x=True) == [(0, 1), (0, 0), (-1, 1), (-1, 0)]
    assert convT_pack_parity(0, False) == 1 and convT_pack_parity(1, True) == 1
    with pytest.raises(ValueError):
        Engine("unet", 1, HEADS, m._flat, m._flat_grad, m._flat_buf, m._counters, lay, 1, 31, 96, "fp32", False, device="cpu")
    m2 = UNet2(1, HEADS)
    m2._flat_grad = torch.zeros_like(m2._flat)
    e2 = Engine("unet2", 1, HEADS, m2._flat, m2._flat_grad, m2._flat_buf, m2._counters, (m2._lay_p, m2._lay_b, m2._lay_c), 1, 72, 88,
                "fp32", True, device="cpu")
    assert (e2.h, e2.w) == (18, 22)
    with pytest.raises(ValueError):
        Engine("unet2", 1, HEADS, m2._flat, m2._flat_grad, m2._flat_buf, m2._counters, (m2._lay_p, m2._lay_b, m2._lay_c), 1, 96, 24,
               "fp32", False, device="cpu")


def test_bucket_plan_covers_arena_once_in_ready_order():
    sizes = [10, 400, 5, 300, 300, 7, 1000, 50]
    ready = [-1, 70, 70, 40, 30, 30, 10, 5]
    b = D.plan_buckets(ready, sizes, 500)
    covered = sorted((lo, hi) for lo, hi, _ in b)
    assert covered[0][0] == 0 and covered[-1][1] == sum(sizes)
    assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    assert [r for _, _, r in b] == sorted(r for _, _, r in b)
    # the arena's tail (heads, produced first by backward) forms the earliest bucket
    assert b[0][1] == sum(sizes)


def test_sampler_matches_torch_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    ds = list(range(103))
    for world in (2, 8):
        for rank in range(world):
            s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=0)
            s.set_epoch(3)
            assert list(iter(s)) == D.sampler_indices(len(ds), world, rank, epoch=3, seed=0)


def test_dropout_mirror_matches_device_hash_definition():
    """abcnet_amd.dropout.keep_mask must equal the C definition (common.hpp: abc_drop_keep), checked on a Python
    big-int transcription of the same arithmetic"""
    from abcnet_amd.dropout import keep_mask

    def ref(idx, seed, p):
        M = 0xFFFFFFFF
        h = ((idx * 0x9E3779B1) & M) ^ seed
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & M
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & M
        h ^= h >> 16
        return (h >> 8) * (1.0 / 16777216.0) >= p

    idx = torch.tensor([0, 1, 2, 12345, 2 ** 20 + 7, 150994943], dtype=torch.int64)
    got = keep_mask(idx, 0x1234ABCD, 0.2).tolist()
    assert got == [ref(int(i), 0x1234ABCD, 0.2) for i in idx]
    big = keep_mask(torch.arange(1 << 16, dtype=torch.int64), 99, 0.2).float().mean().item()
    assert abs(big - 0.8) < 0.01


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    D.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    sizes = [10, 4000, 5, 3000, 3000, 7, 10000, 50]
    ready = [-1, 70, 70, 40, 30, 30, 10, 5]
    g = torch.randn(sum(sizes))
    mine = g.clone()
    buckets = D.plan_buckets(ready, sizes, 5000)
    red = D.GradReducer(g, buckets)
    # emulate the backward plan: after op j the buckets that became final are launched (async), in ready order
    for j in range(0, 80):
        red.after_op(j)
    for lo, hi, r in buckets:
        if r < 0:
            red.bucket_ready(lo, hi)
    red.finish()
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(gathered)
    # the other exchanges on buckets aligned for an even split over the ranks (the padded gradient store of the model):
    # reduce-scatter + all-gather in place, all-to-all + local sum + all-gather; each is self-checked against all_reduce at
    # construction and falls back to it (with the reason) when the backend does not serve it
    modes = {}
    tot = sum(sizes)
    align = 128 * world
    pad = -(-tot // align) * align
    ab = D.align_buckets(buckets, sizes, ready, align, pad)
    for mode in D.GradReducer.MODES:
        store = torch.zeros(pad)
        store[:tot] = mine
        red = D.GradReducer(store, ab, mode=mode)
        for j in range(0, 80):
            red.after_op(j)
        for lo, hi, r in ab:
            if r < 0:
                red.bucket_ready(lo, hi)
        red.finish()
        modes[mode] = (red.mode, red.fallback_reason, torch.equal(store[:tot], g), bool((store[tot:] == 0).all()))
    # parameters: rank 0 wins (DDP constructor semantics)
    p = torch.full((17,), float(rank))
    D.broadcast_parameters(p)
    # scalar loss for logging (multi_gpu_train.py:116)
    lm = D.reduce_mean(torch.tensor([float(rank + 1)]), world)
    # the eval pass's meters over the ranks (multi_gpu_train.py:280-302) as one collective
    tot = torch.tensor([[3.0 + rank, 4.0], [1.0, 2.0 + 2 * rank], [0.0, 0.0], [5.0 * rank, 1.0 * rank]], dtype=torch.float64)
    glob, rmean = D.reduce_meters(tot)
    q.put((rank, torch.allclose(g, want, atol=1e-6), p.sum().item(), lm.item(), modes, ab, pad, glob.tolist(), rmean.tolist()))
    dist.destroy_process_group()


def test_gloo_world2_bucketed_allreduce_and_broadcast():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29640 + (os.getpid() % 200)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), "bucketed all-reduce != sum over ranks"
    assert all(r[2] == 0.0 for r in res), "rank 0 parameters must win the broadcast"
    assert all(abs(r[3] - 1.5) < 1e-6 for r in res)
    for r in res:
        modes, ab, pad = r[4], r[5], r[6]
        # aligned buckets tile the padded store exactly once, each splits evenly over the ranks, ready order kept
        cov = sorted((lo, hi) for lo, hi, _ in ab)
        assert cov[0][0] == 0 and cov[-1][1] == pad and all(cov[i][1] == cov[i + 1][0] for i in range(len(cov) - 1))
        assert all((hi - lo) % (128 * 2) == 0 for lo, hi, _ in ab) and [x[2] for x in ab] == sorted(x[2] for x in ab)
        for mode, (used, why, same_as_allreduce, tail_zero) in modes.items():
            # bit-identical to the all_reduce result (two summands: one rounding whatever the order), padding untouched
            assert same_as_allreduce and tail_zero, (mode, used, why)
            assert used == mode or why is not None, (mode, used, why)
        assert modes["all_reduce"][0] == "all_reduce"
    print("exchange modes used over gloo:", {m: v[:2] for m, v in res[0][4].items()})
    # reduce_meters: sums over the ranks, and the reference's mean of per-rank averages (a rank with count 0 left out; a meter
    # nobody counted is nan), identical on both ranks
    for r in res:
        glob, rmean = r[7], r[8]
        assert glob == [[7.0, 8.0], [2.0, 6.0], [0.0, 0.0], [5.0, 1.0]]
        assert abs(rmean[0] - (3 / 4 + 4 / 4) / 2) < 1e-12 and abs(rmean[1] - (1 / 2 + 1 / 4) / 2) < 1e-12
        assert rmean[2] != rmean[2] and rmean[3] == 5.0


def test_align_buckets_ready_is_max_over_overlapped_tensors():
    sizes = [10, 400, 5, 300, 300, 7, 1000, 50]
    ready = [-1, 70, 70, 40, 30, 30, 10, 5]
    b = D.plan_buckets(ready, sizes, 500)
    ab = D.align_buckets(b, sizes, ready, 256, 2304)
    offs = [0]
    for s_ in sizes:
        offs.append(offs[-1] + s_)
    for lo, hi, r in ab:
        assert lo % 256 == 0 and hi % 256 == 0
        over = [ready[i] for i in range(len(sizes)) if offs[i] < hi and offs[i + 1] > lo]
        assert r == (max(over) if over else -1)


def test_model_deepcopy_and_pickle_keep_the_named_tensors_inside_one_arena():
    """copy.deepcopy(model) / torch.save(model): nn.Parameter.__deepcopy__ clones tensor by tensor, which would detach the
    159 named parameters from the arena the engines, Adam and the all-reduce use"""
    import copy
    import io
    from abcnet_amd.unet import UNet
    m = UNet(1, HEADS)
    next(iter(m.parameters())).requires_grad_(False)
    for clone in (copy.deepcopy(m), torch.load(io.BytesIO(_saved(m)), weights_only=False)):
        assert clone._flat.data_ptr() != m._flat.data_ptr() and torch.equal(clone._flat, m._flat)
        assert torch.equal(clone._flat_buf, m._flat_buf) and clone.training == m.training
        for (n, p), (n0, p0) in zip(clone.named_parameters(), m.named_parameters()):
            off, cnt = clone._lay_p[n]
            assert n == n0 and p.data_ptr() == clone._flat.data_ptr() + 4 * off and p.requires_grad == p0.requires_grad
        with torch.no_grad():
            dict(clone.named_parameters())["inc1.double_conv.0.weight"].add_(1.0)
        assert not torch.equal(clone._flat, m._flat)        # the named tensor IS the arena


def _saved(m):
    import io
    b = io.BytesIO()
    torch.save(m, b)
    return b.getvalue()


_RANK_SCRIPT = """
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, %r)
import abcnet_amd
from abcnet_amd import distributed as D
mode = sys.argv[1]
rank, world = D.init_process_group(backend="gloo")
assert (rank, world) == (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
assert t.item() == world * (world + 1) / 2
if mode == "fail" and rank == 1:
    sys.exit(3)
if mode == "fail":
    time.sleep(600)     # must be ended by the launcher once rank 1 has failed
print("rank", rank, "ok")
"""


def test_launch_ranks_starts_fresh_processes_and_reports_failures(tmp_path):
    """multi_gpu_train.py:30-36 (mp.spawn of main_worker) as fresh child processes: every rank joins the group; when one
    rank fails the others are ended and the failure is reported (bench.py --gpus N turns that into a non-zero exit)"""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % ROOT)
    out = open(tmp_path / "rank0.out", "w")
    assert D.launch_ranks([str(script), "ok"], 2, timeout=300, rank0_stdout=out) == [0, 0]
    out.close()
    assert "rank 0 ok" in open(tmp_path / "rank0.out").read()
    import time
    t0 = time.time()
    codes = D.launch_ranks([str(script), "fail"], 2, timeout=300)
    assert codes[1] == 3 and codes[0] not in (0, None) and time.time() - t0 < 120


def test_bench_refuses_more_gpus_than_visible_and_experiment_knobs():
    """`python bench.py --gpus N` without a launcher must start N ranks or fail -- never print a 1-GPU line as N GPUs"""
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    if torch.cuda.device_count() < 2:
        assert "GPU(s) visible" in r.stderr
    env["ABC_CONV_NOWD"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "experiment switches" in r.stderr


def test_rank_dropout_seeds_differ_and_rank0_keeps_the_base():
    base = 0x1234ABCD
    seeds = [D.rank_dropout_seed(base, r) for r in range(8)]
    assert seeds[0] == base and len(set(seeds)) == 8 and all(0 <= s < 2 ** 32 for s in seeds)


def test_act_bwd_epilogue_plan_and_its_refusals():
    """Engine._actbwd_target / abc_conv_actbwd_ok (host logic only): at the benchmarked shape the bf16 training plan lets 15 of
    unet.py's 26 act_bwd passes (10 of unet2.py's 14: three of them in the 32 -> 32 kernel of its first levels) ride in the epilogue of the data gradient in front of them -- the first BatchNorm
    of a DoubleConv, a trunk layer, the trunk's last layer behind the heads' merged data gradient: layers whose activated output has
    exactly one reader (unet.py:12-17); layers
    with a skip or a pooled reader, the narrow levels (another kernel) and the 24 x 24 / 12 x 12 levels (no whole 16-pixel tile
    columns) keep the pass of their own; fp32 plans and actbwd_epilogue=False have none; the library refuses ragged shapes."""
    import ctypes as C
    from abcnet_amd import _lib as L
    from abcnet_amd.engine import Engine, taps_square
    from abcnet_amd.unet import UNet
    from abcnet_amd.unet2 import UNet as UNet2

    def plan(cls, variant, dtype, **kw):
        m = cls(1, HEADS, dtype=dtype)
        m._flat_grad = torch.zeros_like(m._flat.data)
        e = Engine(variant, 1, HEADS, m._flat.data, m._flat_grad, m._flat_buf, m._counters, (m._lay_p, m._lay_b, m._lay_c), 16, 384, 384,
                   dtype, True, device="cpu", **kw)
        names = [op[2] for op in e.bwd_ops]
        return e, [n for n in names if "+ act_bwd" in n], [n for n in names if n.startswith("act_bwd")]

    e, fused, plain = plan(UNet, "unet", "bf16", fused_heads=True)
    assert len(fused) == 15 and len(plain) == 11, (fused, plain)
    for n in fused:
        # "dgrad X.double_conv.3 + act_bwd X.double_conv.1" (inside a DoubleConv) or "dgrad Y.double_conv.0 + act_bwd X.double_conv.4"
        assert n.endswith("double_conv.1") or n.endswith("double_conv.4"), n
    assert fused[0] == "dgrad heads.conv1 + act_bwd dconv2.double_conv.4"
    tg = [r for r in e.recs if getattr(r, "fused_g", None) is not None]
    assert len(tg) == 15 and all(r.ld == r.cout and r.coff == 0 and r.grad_pool is None for r in tg)
    assert not any("down4" in n or "down5" in n or "up1" in n for n in fused)
    assert sum(("inc1" in n or "inc2" in n) and "inc3" not in n.split(" + ")[0] for n in fused) == 3      # (the 16-channel levels: conv_narrow.hip's own epilogue)
    _e, fused0, plain0 = plan(UNet, "unet", "bf16", fused_heads=True, actbwd_epilogue=False)
    assert not fused0 and len(plain0) == 26
    _e, fused32, _p = plan(UNet, "unet", "fp32")
    assert not fused32
    _e, fused2, plain2 = plan(UNet2, "unet2", "bf16")
    assert len(fused2) == 10 and len(plain2) == 4, (fused2, plain2)
    assert sum("inc1" in n.split(" + ")[1] or "inc2" in n.split(" + ")[1] for n in fused2) == 2, fused2
    # the library's own answer for single descriptors
    lib = L.load()

    def ok(B, H, W, Cin, Cout, **kw):
        d = L.ConvDesc()
        d.src.x, d.src.Hx, d.src.Wx, d.src.ldx = 4096, H, W, Cin
        d.w = d.y = d.stats = 4096
        d.stats_rows = 2
        d.dtype_in = d.dtype_c = d.dtype_out = L.BF16
        d.B, d.Hin, d.Win, d.cin_off, d.Cin = B, H, W, 0, Cin
        d.Hg, d.Wg, d.Hout, d.Wout, d.ldy, d.cout_off, d.Cout, d.Cout_pad = H, W, H, W, Cout, 0, Cout, -(-Cout // 32) * 32
        d.stride, d.om = 1, 1
        L.set_taps(d, taps_square(3))
        d.actbwd_y, d.actbwd_ld = 4096, Cout
        d.actbwd_scale = d.actbwd_shift = d.actbwd_slope = d.actbwd_mean = d.actbwd_invstd = 4096
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.abc_conv_actbwd_ok(C.byref(d)), lib.abc_conv_variant(C.byref(d))

    assert ok(16, 96, 96, 128, 128) == (1, 1) and ok(16, 192, 192, 64, 64) == (1, 1) and ok(2, 48, 48, 256, 256) == (1, 1)
    assert ok(16, 24, 24, 512, 512)[0] == 0          # 24 columns: no whole 16-pixel tiles
    assert ok(1, 30, 48, 64, 64)[0] == 0             # 30 rows: no tile height divides it
    assert ok(16, 384, 384, 16, 16) == (1, 5) and ok(3, 40, 56, 16, 16) == (1, 5)      # the narrow-level kernel: ragged shapes too
    assert ok(16, 192, 192, 32, 32) == (1, 5)        # whole 4 x 16 tiles of the 32 -> 32 kernel (conv_n32r2_kernel, R = 1)
    assert ok(3, 42, 48, 32, 32)[0] == 0             # ... ragged rows: the general narrow-level kernel's 32-channel form has no such epilogue
    assert ok(16, 96, 96, 128, 128, dtype_out=L.F32)[0] == 0 and ok(16, 96, 96, 128, 128, accumulate=1)[0] == 0
    assert ok(16, 96, 96, 128, 128, stats=None)[0] == 0 and ok(16, 96, 96, 128, 128, actbwd_ld=100)[0] == 0


def test_reserved_cus_cannot_change_while_a_plan_is_alive():
    """abc_set_reserved_cus is process-wide and plans bake it into their grids and BatchNorm partial-row buffers (the launches re-derive
    the grids from the current value): engine.set_reserved_cus refuses to move it while any plan exists, accepts the value it already
    has, and rounds to a multiple of four (workgroups per CU times CUs must stay a multiple of the eight XCDs)"""
    from abcnet_amd import engine as E
    lib = L.load()
    assert lib.abc_get_reserved_cus() == 0

    class Plan:      # (stands in for an Engine: building one needs a GPU)
        pass
    import weakref
    plan = Plan()
    saved, E.Engine._live = E.Engine._live, weakref.WeakSet([plan])      # (plans other tests of this process built stay out of it)
    try:
        assert E.set_reserved_cus(0) == 0
        with pytest.raises(L.AbcNetHipError, match="plan"):
            E.set_reserved_cus(8)
        assert lib.abc_get_reserved_cus() == 0
        E.Engine._live.discard(plan)
        assert E.set_reserved_cus(6) == 8 and lib.abc_get_reserved_cus() == 8
        assert E.set_reserved_cus(0) == 0 and lib.abc_get_reserved_cus() == 0
    finally:
        E.Engine._live = saved
        L.check(lib.abc_set_reserved_cus(0), "set")
