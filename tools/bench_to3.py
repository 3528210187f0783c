#!/usr/bin/env python3
"""us per launch of the fp32 image heads (tgsr_conv_to3_fwd) at the generator's shapes, B=16:  python tools/bench_to3.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops
B = 16
for cin, K, H, act in [(32, 5, 64, True), (32, 5, 128, True), (32, 5, 256, True), (64, 3, 64, False), (64, 3, 128, False), (64, 3, 256, False)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    w = torch.randn(3, cin, K, K, device="cuda") / (K * cin ** 0.5)
    add = torch.randn(B, 3, H, H, device="cuda") if act else None
    for _ in range(3):
        ops.conv_to3(x, w, tanh_axpy=act, addend=add, alpha=0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv_to3(x, w, tanh_axpy=act, addend=add, alpha=0.5)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    print("Cin %d K %d %dx%d: %7.1f us   %6.0f GB/s" % (cin, K, H, H, us, 4.0 * B * (cin + 3) * H * H / us / 1e3))
