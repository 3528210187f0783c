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
x = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
# HIP chain with retained grads
acts = []
h = C.conv4x4s2(x.cuda(), d.img_code_s16.conv0.weight, True); h.retain_grad(); acts.append(("conv0", h))
for nm, m in (("down1", d.img_code_s16.down1), ("down2", d.img_code_s16.down2), ("down3", d.img_code_s16.down3), ("extra0", d.extra[0]), ("extra1", d.extra[1]), ("reduce0", d.reduce[0]), ("reduce1", d.reduce[1])):
    h = m(h); h.retain_grad(); acts.append((nm, h))
R = torch.randn(h.shape, generator=g)
(h * R.cuda()).sum().backward()
# oracle chain fp64
sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
racts = []
r = O.leaky(F.conv2d(x.double(), sdr["img_code_s16.conv0.weight"], None, 2, 1)); r.retain_grad(); racts.append(r)
for p, fn in (("img_code_s16.down1.", O.down_block), ("img_code_s16.down2.", O.down_block), ("img_code_s16.down3.", O.down_block), ("extra.0.", O.down_block), ("extra.1.", O.down_block), ("reduce.0.", O.block3x3_leaky), ("reduce.1.", O.block3x3_leaky)):
    r = fn(r, sdr, p, True, {}); r.retain_grad(); racts.append(r)
(r * R.double()).sum().backward()
for (nm, a), b in zip(acts, racts):
    print("%-8s %-20s act %.2e  grad %.2e   sum(grad) %.6e vs %.6e" % (nm, tuple(a.shape), rel(a, b), rel(a.grad, b.grad), float(a.grad.double().sum()), float(b.grad.sum())))
