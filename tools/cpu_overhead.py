#!/usr/bin/env python3
"""How much of a step is host enqueue time? (eager launches through ctypes)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgsr_amd.miscc.config import cfg, cfg_reset
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.trainer import SRPipeline
cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256
w = bench.load_weights()
p = SRPipeline(41, device="cuda", branch_num=4).load_state_dicts(w["E."], w["GL."], w["GH."])
cap, lens, LR, LRb = synthetic_batch(16); cap, LR, LRb = cap.cuda(), LR.cuda(), LRb.cuda(); lens = lens.tolist()
for _ in range(5): p(cap, lens, LR, LRb)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): p(cap, lens, LR, LRb)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# B=1 latency (shipped yml batch size)
cap1, lens1, LR1, LRb1 = synthetic_batch(1); cap1, LR1, LRb1 = cap1.cuda(), LR1.cuda(), LRb1.cuda(); lens1 = lens1.tolist()
for _ in range(5): p(cap1, lens1, LR1, LRb1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): p(cap1, lens1, LR1, LRb1)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("B=1: enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
