#!/usr/bin/env python3
"""Per-parameter gradient error of discriminator_loss through a HIP D_NET vs the fp64 oracle (debug aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tgsr_oracle as O
from tgsr_amd.miscc.config import cfg
from tgsr_amd import model
from tgsr_amd.miscc import losses

cfg.GAN.DF_DIM = 8
cfg.TEXT.EMBEDDING_DIM = 32
name, size = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("D_NET256", 256)
torch.manual_seed(11)
d = getattr(model, name)()
sd = {k: v.detach().clone() for k, v in d.state_dict().items()}
d.cuda().train()
B = 4
g = torch.Generator().manual_seed(5)
real, fake = torch.rand(B, 3, size, size, generator=g) * 2 - 1, torch.rand(B, 3, size, size, generator=g) * 2 - 1
cond = torch.randn(B, 32, generator=g)
rl, fl = torch.ones(B), torch.zeros(B)
dt = torch.float64
sdr = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.to(dt) if v.is_floating_point() else v)) for k, v in sd.items()}
loss = O.discriminator_loss(sdr, real.to(dt), fake.to(dt), cond.to(dt), rl.to(dt), fl.to(dt))
loss.backward()
got = losses.discriminator_loss(d, real.cuda(), fake.cuda(), cond.cuda(), rl.cuda(), fl.cuda())
got.backward()
print("loss", float(got), float(loss))
for k, p in d.named_parameters():
    gr = sdr[k].grad
    print("%-40s %-22s rel err %.3e  |g|max %.3e" % (k, tuple(p.shape), float((p.grad.cpu().double() - gr).abs().max()) / (float(gr.abs().max()) + 1e-12), float(gr.abs().max())))
