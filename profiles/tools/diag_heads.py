import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import abcnet_amd
from abcnet_amd.synthetic import synthetic_images, synthetic_targets
from abcnet_amd.train import Trainer
from abcnet_amd.unet import UNet
from oracle import unet_oracle as uo
HEADS = uo.HEADS
B, S = 2, 64
x, tg = synthetic_images(B, S, seed=7), synthetic_targets(B, S // 4, seed=1)
def one():
    m = UNet(1, HEADS, dtype="bf16", dropout_p=0.2); m.load_state_dict(uo.filled_state("unet", 1, HEADS, seed=0)); m = m.to("cuda")
    tr = Trainer(m, B, S, S, lr=0.0, use_graph=False)
    tr.load_batch(x.cuda(), [t.cuda() for t in tg]); tr.step(); torch.cuda.synchronize()
    return m, m._flat_grad.clone()
m1, g1 = one()
os.environ["ABC_NO_HEADS_BATCH"] = "1"
m2, g2 = one()
rows = []
for name, (off, n) in m1._lay_p.items():
    a, b = g1[off:off + n].double(), g2[off:off + n].double()
    d = (a - b).norm().item(); r = b.norm().item()
    rows.append((d / (r + 1e-30), name, d, r))
rows.sort(reverse=True)
for rel, name, d, r in rows[:25]: print("%-50s rel %.3e  |d| %.3e |g| %.3e" % (name, rel, d, r))
print("...")
for rel, name, d, r in rows[-5:]: print("%-50s rel %.3e  |d| %.3e |g| %.3e" % (name, rel, d, r))
