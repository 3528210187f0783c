#!/usr/bin/env python3
"""Time of the DAMSM word / region similarity grid's backward (tgsr_damsm_words_bwd: damsm_pair_bwd_kernel + two reductions) at the
G/D step's size: batch 16, 17 x 17 regions, 256 features, captions of up to 18 words.   python tools/bench_damsm_bwd.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops
B, ndf, T = 16, 256, 18
g = torch.Generator().manual_seed(0)
img = torch.randn(B, ndf, 17, 17, generator=g).cuda()
words = torch.randn(B, ndf, T, generator=g).cuda()
lens = [18, 17, 15, 15, 14, 13, 12, 12, 11, 10, 10, 9, 8, 7, 6, 5]
gsim = torch.randn(B, B, generator=g).cuda()
for _ in range(3):
    out = ops.damsm_words_bwd(img, words, lens, 5.0, 5.0, gsim)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.damsm_words_bwd(img, words, lens, 5.0, 5.0, gsim)
e1.record()
torch.cuda.synchronize()
print("damsm_words_bwd: %.1f us per call (B = %d, %d regions, ndf = %d)" % (e0.elapsed_time(e1) * 1e3 / 20, B, 289, ndf))
