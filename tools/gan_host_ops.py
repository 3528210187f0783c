#!/usr/bin/env python3
"""Which Python lines issue the small aten launches of a G/D step (fill_, add, copy_, zero_ ...): torch.profiler with stacks.
    python tools/gan_host_ops.py > gpurun_out/gan_host_ops.txt"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.train import SRTrainer
from torch.profiler import profile, ProfilerActivity
B = 16
dev = "cuda"
tr = SRTrainer(41, device=dev, discriminators=True)
cap, lens, LR, LRb = synthetic_batch(B, seed=100)
g = torch.Generator().manual_seed(7)
hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).to(dev) for s in (64, 128, 256)]
cap, LR, LRb, lens = cap.to(dev), LR.to(dev), LRb.to(dev), lens.tolist()
for _ in range(3):
    tr.step(cap, lens, LR, LRb, hr)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(cap, lens, LR, LRb, hr)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if e.key in ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone",
                                   "aten::mul", "aten::mean", "aten::sum", "aten::cat", "aten::empty", "aten::sigmoid", "aten::log_sigmoid")]
rows.sort(key=lambda e: -e.count)
for e in rows[:70]:
    st = [s for s in e.stack if "tgsr_amd" in s or "autograd" in s][:4]
    print("%-18s x%-4d cpu %7.1f us total | %s" % (e.key, e.count, e.cpu_time_total, " <- ".join(s.split("/")[-1] for s in st)))
print()
tot = {}
for e in prof.key_averages():
    tot[e.key] = (e.count, e.cpu_time_total, e.self_cpu_time_total)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][2])[:40]:
    print("%-60s x%-5d self cpu %8.1f us  total %8.1f us" % (k[:60], v[0], v[2], v[1]))
