#!/usr/bin/env python3
"""List the non-tgsr device activity of one inference step (torch.profiler): which torch ops still launch copies /
elementwise kernels around the HIP path.  GPU only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgsr_amd.miscc.config import cfg, cfg_reset
from tgsr_amd.trainer import SRPipeline
from tgsr_amd.synthetic import synthetic_batch

cfg_reset()
cfg.GAN.GF_DIM = 32
cfg.TEXT.EMBEDDING_DIM = 256
cfg.TREE.BRANCH_NUM = 4
cfg.TREE.BASE_SIZE = 32
dev = torch.device("cuda:0")
pipe = SRPipeline(41, device=dev, low="lr", overlap=False, branch_num=4)
w = bench.load_weights()
if w is not None:
    pipe.load_state_dicts(w["E."], w["GL."], w["GH."])
cap, lens, LR, LRb = synthetic_batch(16, seed=100)
cap, LR, LRb = cap.to(dev), LR.to(dev), LRb.to(dev)
lens = lens.tolist()
for _ in range(3):
    pipe(cap, lens, LR, LRb)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    pipe(cap, lens, LR, LRb)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
