#!/usr/bin/env python3
"""Experiment: the generator training step (forward + backward + Adam + EMA) captured in one hipGraph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.train import SRTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tr = SRTrainer(41, device="cuda")
tr.opt = torch.optim.Adam(tr.params, lr=cfg.TRAIN.GENERATOR_LR, betas=(0.5, 0.999), capturable=True)
cap, lens, LR, LRb = synthetic_batch(B, seed=100)
g = torch.Generator().manual_seed(7)
hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).cuda() for s in (64, 128, 256)]
cap, LR, LRb, lens = cap.cuda(), LR.cuda(), LRb.cuda(), lens.tolist()
def bench(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("eager step: %.2f ms" % bench(lambda: tr.step(cap, lens, LR, LRb, hr)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        tr.step(cap, lens, LR, LRb, hr)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    loss = tr.step(cap, lens, LR, LRb, hr)
torch.cuda.synchronize()
print("captured", flush=True)
l0 = None
for i in range(5):
    gr.replay()
    torch.cuda.synchronize()
    print("replay", i, float(loss))
print("graphed step: %.2f ms" % bench(lambda: gr.replay()))
