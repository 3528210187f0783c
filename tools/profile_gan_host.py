#!/usr/bin/env python3
"""Which Python call sites issue the small copy / add / fill kernels of one G/D alternation (torch.profiler, grouped by
stack).  python tools/profile_gan_host.py [--gan 0|1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256     # the shipped checkpoints' widths (bench.py sets the same)
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.train import SRTrainer
gan = "--gan" not in sys.argv or sys.argv[sys.argv.index("--gan") + 1] == "1"
tr = SRTrainer(41, device="cuda", discriminators=gan)
B = 16
cap, lens, LR, LRb = synthetic_batch(B)
g = torch.Generator().manual_seed(7)
hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).cuda() for s in (64, 128, 256)]
cap, LR, LRb, lens = cap.cuda(), LR.cuda(), LRb.cuda(), lens.tolist()
for _ in range(2):
    tr.step(cap, lens, LR, LRb, hr)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(cap, lens, LR, LRb, hr)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_stack_n=8)
want = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat",
        "aten::mul", "aten::sum", "aten::flip", "aten::empty", "aten::zeros")
agg = {}
for r in rows:
    if r.key in want:
        stack = [s for s in r.stack if "tgsr_amd" in s or "torch/optim" in s or "bench" in s][:3]
        k = (r.key, " <- ".join(s.split("/")[-1] for s in stack))
        agg[k] = agg.get(k, 0) + r.count
for (op, st), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print("%5d  %-18s %s" % (n, op, st))
