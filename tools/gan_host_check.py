#!/usr/bin/env python3
"""Is the G/D alternation step (BASELINE configs[2]; GD=0: the generator step alone) bound by the host?  Times, per step: the host's time to ENQUEUE a step onto an
empty queue (no synchronisation inside), and the wall time of back-to-back steps.  If the two are equal the GPU is waiting for Python.
    python tools/gan_host_check.py [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.train import SRTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda"
tr = SRTrainer(41, device=dev, discriminators=(os.environ.get("GD", "1") == "1"))
cap, lens, LR, LRb = synthetic_batch(B, seed=100)
g = torch.Generator().manual_seed(7)
hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).to(dev) for s in (64, 128, 256)]
cap, LR, LRb, lens = cap.to(dev), LR.to(dev), LRb.to(dev), lens.tolist()
for _ in range(3):
    tr.step(cap, lens, LR, LRb, hr)
torch.cuda.synchronize()
enq = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(cap, lens, LR, LRb, hr)          # returns the loss as a tensor? a float forces a sync: see below
    enq.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    tr.step(cap, lens, LR, LRb, hr)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
print("split %s  enqueue one step onto an empty queue: %s ms   back-to-back: %.2f ms per step" %
      (os.environ.get("TGSR_DCONV_SPLIT", "default"), " ".join("%.2f" % (e * 1e3) for e in enq), wall * 1e3), flush=True)
