#!/usr/bin/env python3
"""Timing of the six image-head launches of one forward (B=16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops
tot = 0
for K, act in ((3, False), (5, True)):
    for s in (64, 128, 256):
        x = torch.randn(16, 32, s, s, device="cuda"); w = torch.randn(3, 32, K, K, device="cuda") / (K * 32 ** 0.5)
        add = torch.randn(16, 3, s, s, device="cuda") if act else None
        for _ in range(3): ops.conv_to3(x, w, tanh_axpy=act, addend=add, alpha=0.5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): ops.conv_to3(x, w, tanh_axpy=act, addend=add, alpha=0.5)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20; tot += us
        print("K=%d %3d^2: %7.1f us  %.1f TFLOP/s  %.0f GB/s" % (K, s, us, 2.0 * 16 * s * s * 3 * 32 * K * K / us / 1e6, 4.0 * 16 * s * s * (32 + 3 + (3 if act else 0)) / us / 1e3))
print("total %.1f us" % tot)
