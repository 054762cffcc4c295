import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import test_gpu_heads_fused as T
r = T._run(2, 32, 0.2)
g = r["g"].float().cpu()
print("nan count", torch.isnan(g).sum().item(), "inf", torch.isinf(g).sum().item())
for i in range(8):
    sl = g[:, 128*i:128*(i+1)]
    bad = ~torch.isfinite(sl)
    if bad.any():
        idx = bad.nonzero()
        print("head", i, "bad", bad.sum().item(), "pixels", idx[:, 0].unique()[:20].tolist(), "chans", idx[:, 1].unique()[:40].tolist())
for i in range(5):
    print("dw", i, torch.isfinite(r["dw2"][i]).all().item(), T.rel(r["dw2"][i].cpu(), r["ref"]["ws"][i].grad), T.rel(r["db2"][i].cpu(), r["ref"]["bs"][i].grad))
dl = r["dl"].float().cpu()
print("dl nonfinite", (~torch.isfinite(dl)).sum().item())
for i, t in enumerate(r["logits"]):
    print("logits", i, (~torch.isfinite(t)).sum().item(), end="; ")
print()
print("bnp nonfinite", (~torch.isfinite(r["bnp"])).sum().item())
lib = r["lib"]; nchunk = r["nchunk"]; row0 = 0
for i in range(8):
    rows = lib.abc_heads_fused_rows(i)
    blk = dl[row0 * nchunk * 128:(row0 + rows) * nchunk * 128].view(nchunk, rows, 128)
    bad = (~torch.isfinite(blk)).nonzero()
    if len(bad): print("dl head", i, bad[:10].tolist())
    row0 += rows
