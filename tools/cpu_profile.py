#!/usr/bin/env python3
"""cProfile of the host side of one inference step (eager enqueue path)."""
import cProfile, os, pstats, sys, torch
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
pr = cProfile.Profile(); pr.enable()
for _ in range(50): p(cap, lens, LR, LRb)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
