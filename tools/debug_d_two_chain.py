import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tgsr_oracle as O
from tgsr_amd.miscc.config import cfg
cfg.GAN.DF_DIM = 8; cfg.TEXT.EMBEDDING_DIM = 32
from tgsr_amd import model, custom_ops as C
def rel(a, b): return float((a.detach().cpu().double() - b.detach().double()).abs().max()) / (float(b.detach().abs().max()) + 1e-30)
torch.manual_seed(11)
d = model.D_NET256()
sd = {k: (v.detach().double().clone() if v.is_floating_point() else v.clone()) for k, v in d.state_dict().items()}
d.cuda().train()
g = torch.Generator().manual_seed(5)
B = 4
xs = [torch.rand(B, 3, 256, 256, generator=g) * 2 - 1 for _ in range(2)]
Rs = [torch.randn(B, 64, 4, 4, generator=g) for _ in range(2)]
mods = (("down1", d.img_code_s16.down1), ("down2", d.img_code_s16.down2), ("down3", d.img_code_s16.down3), ("extra0", d.extra[0]), ("extra1", d.extra[1]), ("reduce0", d.reduce[0]), ("reduce1", d.reduce[1]))
acts, loss = [], 0
for x, R in zip(xs, Rs):
    h = C.conv4x4s2(x.cuda(), d.img_code_s16.conv0.weight, True); h.retain_grad(); cur = [("conv0", h)]
    for nm, m in mods:
        h = m(h); h.retain_grad(); cur.append((nm, h))
    acts.append(cur)
    loss = loss + (h * R.cuda()).sum()
loss.backward()
sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
racts, rl = [], 0
for x, R in zip(xs, Rs):
    r = O.leaky(F.conv2d(x.double(), sdr["img_code_s16.conv0.weight"], None, 2, 1)); r.retain_grad(); cur = [r]
    for p, fn in (("img_code_s16.down1.", O.down_block), ("img_code_s16.down2.", O.down_block), ("img_code_s16.down3.", O.down_block), ("extra.0.", O.down_block), ("extra.1.", O.down_block), ("reduce.0.", O.block3x3_leaky), ("reduce.1.", O.block3x3_leaky)):
        r = fn(r, sdr, p, True, {}); r.retain_grad(); cur.append(r)
    racts.append(cur); rl = rl + (r * R.double()).sum()
rl.backward()
for i in range(2):
    for (nm, a), b in zip(acts[i], racts[i]):
        print("pass %d %-8s act %.2e  grad %.2e" % (i, nm, rel(a, b), rel(a.grad, b.grad)))
