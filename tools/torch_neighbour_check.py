#!/usr/bin/env python3
"""Diagnostic: are PyTorch's own elementwise / optimizer kernels exposed to the packed-fp32 hazard of profiles/HISTORY.md 3.13 when they
run beside this library's MFMA-bound kernels (as they do during a training step's backward)?  Same method as
tests/test_hip_concurrency.py: victim on one stream, neighbour looping on another, torch.equal against the solo result."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tgsr_amd import lp
dev = "cuda"
g = torch.Generator().manual_seed(0)
R = lambda *sh: torch.randn(*sh, generator=g).to(dev)
B = 16
x = lp.from_nchw(R(B, 64, 64, 64), "bf16")
wp = lp.pack_upconv_weight(R(64, 64, 3, 3) * 0.1, "bf16")
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
o = lp.new_image(B, 128, 128, 32, "bf16", dev)
neighbour = lambda: lp.upconv_glu(x, wp, 64, 64, sc, sh, out=o)
a, b, c = R(1 << 22), R(1 << 22), R(1 << 22)
params = [R(64, 64, 3, 3) for _ in range(24)]
grads = [R(64, 64, 3, 3) for _ in range(24)]
def adam():
    ps = [p.clone().requires_grad_(True) for p in params]
    for p, gr in zip(ps, grads):
        p.grad = gr
    opt = torch.optim.Adam(ps, lr=2e-4, betas=(0.5, 0.999))
    opt.step(); opt.step()
    return torch.cat([p.detach().flatten() for p in ps])
victims = {
    "a * 1.5 + b": lambda: a * 1.5 + b,
    "sigmoid(a) * b + c": lambda: torch.sigmoid(a) * b + c,
    "addcmul / lerp": lambda: torch.lerp(torch.addcmul(a, b, c, value=0.3), b, 0.999),
    "_foreach_mul_ + _foreach_add_ (EMA)": lambda: torch.cat([t.flatten() for t in torch._foreach_add(torch._foreach_mul(params, 0.999), grads, alpha=0.001)]),
    "Adam.step x2 (24 tensors)": adam,
    "mse_loss + mean": lambda: torch.stack([torch.nn.functional.mse_loss(a, b), a.mean(), (a * b).sum()]),
    "bf16 cast + tanh": lambda: torch.tanh(a.to(torch.bfloat16).float()),
}
side = torch.cuda.Stream()
for name, fn in victims.items():
    ref = fn().clone(); torch.cuda.synchronize()
    assert torch.equal(fn(), ref), name
    bad = 0
    for it in range(40):
        with torch.cuda.stream(side):
            for _ in range(3):
                neighbour()
        out = fn()
        with torch.cuda.stream(side):
            for _ in range(3):
                neighbour()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    print("%-40s: %d of 40 runs differ beside lp_upconv_glu_kernel" % (name, bad), flush=True)
