#!/usr/bin/env python3
"""Diagnostic: the generator train step comes out at 10.7-10.9 ms in most processes and 11.9-12.0 ms in some, with the same
kernel time.  Does it depend on the priority of the stream the step runs on (the weight gradients' side stream competes with it)?
Prints the mean step time of several blocks of steps, on the default stream and on a high-priority stream."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tgsr_amd.miscc.config import cfg, cfg_reset
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.train import SRTrainer
cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256
dev = torch.device("cuda")
B = 16
tr = SRTrainer(41, device=dev)
cap, lens, LR, LRb = synthetic_batch(B, seed=100)
g = torch.Generator().manual_seed(7)
hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).to(dev) for s in (64, 128, 256)]
cap, LR, LRb, lens = cap.to(dev), LR.to(dev), LRb.to(dev), lens.tolist()
def block(n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.step(cap, lens, LR, LRb, hr)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(5):
    tr.step(cap, lens, LR, LRb, hr)
print("default stream      :", " ".join("%.2f" % block() for _ in range(6)))
hp = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hp):
    for _ in range(3):
        tr.step(cap, lens, LR, LRb, hr)
    print("high-priority stream:", " ".join("%.2f" % block() for _ in range(6)))
print("default stream again:", " ".join("%.2f" % block() for _ in range(6)))
