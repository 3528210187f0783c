#!/usr/bin/env python3
"""The forward GEMM of two discriminator layers, timed (us).  python tools/exp_dconv_fwd.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops
def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for ci, co, H in ((64, 128, 128), (256, 512, 32)):
    x = torch.randn(32, ci, H, H, device="cuda")
    w = torch.randn(co, ci, 4, 4, device="cuda") / (ci * 16) ** 0.5
    out.append("%d->%d@%d %.1f" % (ci, co, H, timeit(lambda: ops.conv4x4s2(x, w))))
print(os.environ.get("TGSR_DCONV_SPLIT", "-"), os.environ.get("TGSR_IG6_EXP", "-"), "  ".join(out), flush=True)
