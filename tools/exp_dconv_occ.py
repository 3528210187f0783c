#!/usr/bin/env python3
"""How the x6 GEMM's time grows with the number of workgroups (64 -> 128 @128^2: 32 workgroups per image): 256, 512, 768, 1024."""
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
for ci, co, H in ((64, 128, 128), (256, 512, 32)):
    for B in (4, 8, 12, 16, 24, 32):
        x = torch.randn(B, ci, H, H, device="cuda")
        w = torch.randn(co, ci, 4, 4, device="cuda") / (ci * 16) ** 0.5
        t = timeit(lambda: ops.conv4x4s2(x, w))
        fl = 2.0 * B * (H // 2) ** 2 * co * ci * 16
        print("%d->%d @%d B %2d: %7.1f us  %6.1f TF/s" % (ci, co, H, B, t, fl / t / 1e6), flush=True)
