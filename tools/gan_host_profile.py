#!/usr/bin/env python3
"""cProfile of the host side of the G/D alternation step (what Python spends per step while the GPU waits or works).
    python tools/gan_host_profile.py [B] > gpurun_out/gan_host_profile.txt"""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.train import SRTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda"
tr = SRTrainer(41, device=dev, discriminators=True)
cap, lens, LR, LRb = synthetic_batch(B, seed=100)
g = torch.Generator().manual_seed(7)
hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).to(dev) for s in (64, 128, 256)]
cap, LR, LRb, lens = cap.to(dev), LR.to(dev), LRb.to(dev), lens.tolist()
for _ in range(3):
    tr.step(cap, lens, LR, LRb, hr)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(cap, lens, LR, LRb, hr)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
print("== by internal time (5 steps)")
st.sort_stats("tottime").print_stats(45)
print("== by cumulative time")
st.sort_stats("cumulative").print_stats(60)
