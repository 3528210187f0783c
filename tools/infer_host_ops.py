#!/usr/bin/env python3
"""Which Python lines issue device-to-device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) in one EAGER fp32 inference step:
torch.profiler with stacks.   python tools/infer_host_ops.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg, cfg_reset
cfg_reset()
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
cfg.TREE.BRANCH_NUM, cfg.TREE.BASE_SIZE = 4, 32
from tgsr_amd.synthetic import random_init_, synthetic_batch
from tgsr_amd.trainer import SRPipeline
from torch.profiler import profile, ProfilerActivity
import bench
pipe = SRPipeline(41, device="cuda", low="lr", overlap=False)
weights = bench.load_weights()                      # the shipped face checkpoint, as bench.py loads it
pipe.load_state_dicts(weights["E."], weights["GL."], weights["GH."])
pool = bench.batch_pool(16, 0, torch.device("cuda", 0))          # bench.py's eight resident batches, rotated step by step
for k in range(3):
    b = pool[k % len(pool)]
    pipe(b["cap"], b["lens"], b["LR"], b["LRb"])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    b = pool[3 % len(pool)]
    pipe(b["cap"], b["lens"], b["LR"], b["LRb"])
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=8)
rows = [e for e in ka if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "hipMemcpyAsync", "aten::cat", "aten::fill_", "aten::zero_")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    st = [s for s in e.stack if "tgsr_amd" in s][:5]
    print("%-18s x%-4d | %s" % (e.key, e.count, " <- ".join(s.split("/")[-1] for s in st)))
print()
dev = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        dev[e.name[:70]] = dev.get(e.name[:70], 0) + 1
for k, v in sorted(dev.items(), key=lambda kv: -kv[1])[:25]:
    print("%5d  %s" % (v, k))
