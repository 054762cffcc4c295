import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd import _lib as L
from abcnet_amd.engine import head_offsets
from abcnet_amd.synthetic import synthetic_targets
HEADS = [1, 14, 3, 2, 1, 360, 60, 60]
DEV = "cuda"
if os.environ.get("ABC_TOOL_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["ABC_TOOL_LIB"])
lib = L.load()
B, hw, ld = 16, 96, 1024
npix = B * hw * hw
g = torch.Generator().manual_seed(3)
feat = (torch.randn((npix, ld), generator=g) * 1.5).to(torch.bfloat16).to(DEV)
sc = (torch.rand(ld, generator=g) * 0.8 + 0.4).to(DEV); sh = (torch.randn(ld, generator=g) * 0.3).to(DEV); sl = torch.full((ld,), 0.01).to(DEV)
mean = (torch.randn(ld, generator=g) * 0.2).to(DEV); invstd = (torch.rand(ld, generator=g) + 0.5).to(DEV)
w2 = [(torch.randn((c, 128), generator=g) * 0.15).to(DEV) for c in HEADS]
b2 = [(torch.randn((c,), generator=g) * 0.5).to(DEV) for c in HEADS]
tg = [t.to(DEV) for t in synthetic_targets(B, hw, seed=1)]
d = L.HeadsFusedDesc()
d.feat, d.ld = feat.data_ptr(), ld
d.scale, d.shift, d.slope, d.mean, d.invstd = sc.data_ptr(), sh.data_ptr(), sl.data_ptr(), mean.data_ptr(), invstd.data_ptr()
d.drop_p, d.drop_seed, d.drop_salt = float(os.environ.get("HF_DROP_P", "0.2")), 0x1234567, None
pack = torch.zeros(lib.abc_heads_fused_pack_bytes(), dtype=torch.uint8, device=DEV)
logits = [torch.zeros((B, c, hw, hw), device=DEV) for c in HEADS]
for i in range(8):
    d.w2[i], d.b2[i], d.logits[i] = w2[i].data_ptr(), b2[i].data_ptr(), logits[i].data_ptr()
d.w2_pack = pack.data_ptr()
(d.t_atom, d.t_types, d.t_charges, d.t_hs, d.t_bond, d.t_btypes, d.t_rho, d.t_omega) = (t.data_ptr() for t in tg)
d.B, d.h, d.w = B, hw, hw
nchunk = lib.abc_heads_fused_chunks(C.byref(d))
dl = torch.zeros(lib.abc_heads_fused_dl_elems(C.byref(d)), dtype=torch.bfloat16, device=DEV)
gbuf = torch.zeros((npix, ld), dtype=torch.bfloat16, device=DEV)
bnp = torch.zeros((nchunk, 2, ld), device=DEV)
lp = torch.zeros((nchunk, 16), dtype=torch.float64, device=DEV)
d.dl, d.g, d.bn_partial, d.loss_partial = dl.data_ptr(), gbuf.data_ptr(), bnp.data_ptr(), lp.data_ptr()
off = head_offsets(HEADS)
cs = torch.ones(sum(HEADS), device=DEV)
d.chan_scale = cs.data_ptr()
dw2 = [torch.zeros((c, 128), device=DEV) for c in HEADS]; db2 = [torch.zeros((c,), device=DEV) for c in HEADS]
work = torch.zeros(lib.abc_heads_fused_wgrad_floats(C.byref(d)), device=DEV)
for i in range(8):
    d.chan_off[i], d.dw2[i], d.db2[i] = off[i], dw2[i].data_ptr(), db2[i].data_ptr()
d.wgrad_work = work.data_ptr()
st = torch.cuda.current_stream().cuda_stream
L.check(lib.abc_heads_fused_pack(C.byref(d), st), "pack")
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
args = sys.argv[1:]
if args and args[0] == "--flags":
    # the rasteriser's group flags, from the dense maps: bit i = some pixel of the 32-pixel group has a target of head i (rho: the bond types')
    args = args[1:]
    fl = torch.zeros(npix // 32, dtype=torch.int32, device=DEV)
    for i, tt in enumerate(tg):
        m = tt.reshape(B, -1, hw * hw)
        any_ = (m != 0).any(dim=1).reshape(B * hw * hw // 32, 32).any(dim=1)
        fl |= any_.to(torch.int32) << (5 if i == 6 else i)
    zb = torch.zeros(512, dtype=torch.uint8, device=DEV)
    d.target_flags, d.zero_bytes = fl.data_ptr(), zb.data_ptr()
    print("flagged groups per head:", [round(float(((fl >> i) & 1).float().mean()), 3) for i in range(8)])
for dbg in [int(x, 0) for x in (args or ["0"])]:
    os.environ["ABC_HF_DBG"] = str(dbg)     # bits 0-6: phase ablations of heads_fused.hip; 256 << g: work type g not run
    print("dbg %3d: fwd_bwd %.1f us" % (dbg, t(lambda: L.check(lib.abc_heads_fused_fwd_bwd(C.byref(d), st), "f"))), flush=True)
os.environ["ABC_HF_DBG"] = "0"
print("wgrad %.1f us" % t(lambda: L.check(lib.abc_heads_fused_wgrad(C.byref(d), st), "w")))
