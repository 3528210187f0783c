import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tgsr_oracle as O
from tgsr_amd.miscc.config import cfg
cfg.GAN.DF_DIM = 8; cfg.TEXT.EMBEDDING_DIM = 32
from tgsr_amd import model
def rel(a, b): return float((a.detach().cpu().double() - b.detach().double()).abs().max()) / (float(b.detach().abs().max()) + 1e-30)
torch.manual_seed(11)
d = model.D_NET256()
sd = {k: (v.detach().double().clone() if v.is_floating_point() else v.clone()) for k, v in d.state_dict().items()}
d.cuda().train()
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
B = 4
x1 = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
x2 = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
R1, R2 = torch.randn(B, 64, 4, 4, generator=g), torch.randn(B, 64, 4, 4, generator=g)
mode = sys.argv[1] if len(sys.argv) > 1 else "two"
def hip_loss():
    l = (d(x1.cuda()) * R1.cuda()).sum()
    if mode == "two":
        l = l + (d(x2.cuda()) * R2.cuda()).sum()
    return l
hip_loss().backward()
sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
l = (O.d_features(sdr, x1.double()) * R1.double()).sum()
if mode == "two":
    l = l + (O.d_features(sdr, x2.double()) * R2.double()).sum()
l.backward()
for k, p in d.named_parameters():
    if sdr[k].grad is None: continue
    print("%-36s rel err %.3e" % (k, rel(p.grad, sdr[k].grad)))
