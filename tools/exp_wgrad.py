#!/usr/bin/env python3
"""Per-layer timing of the weight-gradient kernels (tgsr_wino_wgrad / tgsr_upwino_wgrad / direct) at the generator's layer
shapes, batch 16: microseconds per call (kernel + slab reduction) and executed TFLOP/s of the Winograd-domain products.
    python3 tools/exp_wgrad.py            (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgsr_amd import ops

dev = "cuda"
B = 16
g = torch.Generator().manual_seed(0)
shapes = [("GL c1 64->128", 64, 128, False), ("GL c2 64->64", 64, 64, False), ("GH c1 32->64", 32, 64, False),
          ("GH c2 32->32", 32, 32, False), ("GL up 64->64", 64, 64, True), ("GH up 32->64", 32, 64, True)]
for name, cin, cout, up in shapes:
    for r in (32, 64, 128):
        if name.startswith("GH c") and r > 32 and cout == 64 and cin == 32 and False:
            continue
        x = torch.randn(B, cin, r, r, generator=g).to(dev)
        ro = 2 * r if up else r
        dy = torch.randn(B, cout, ro, ro, generator=g).to(dev)
        out = torch.empty(cout, cin, 3, 3, device=dev)
        for _ in range(3):
            ops.conv3x3_wgrad(dy, x, up, True, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.conv3x3_wgrad(dy, x, up, True, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        pos = 9 if up else 16
        tiles = B * r * r if up else B * (r // 2) * (r // 2)
        exe = 2.0 * pos * cout * cin * tiles
        print("%-16s %4d^2  %8.1f us   %6.1f TFLOP/s executed (%.2f of 157.3)" % (name, r, us, exe / us / 1e6, exe / us / 1e6 / 157.3), flush=True)
