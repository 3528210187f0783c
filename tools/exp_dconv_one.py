#!/usr/bin/env python3
"""One discriminator layer, forward + data gradient + weight gradient, a few launches (for the counter passes).
    python tools/exp_dconv_one.py Cin Cout H [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops
ci, co, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
x = torch.randn(B, ci, H, H, device="cuda")
w = torch.randn(co, ci, 4, 4, device="cuda") / (ci * 16) ** 0.5
dy = torch.randn(B, co, H // 2, H // 2, device="cuda")
for _ in range(5):
    ops.conv4x4s2(x, w)
    ops.conv4x4s2_dgrad(dy, w, H, H)
    ops.conv4x4s2_wgrad(dy, x)
torch.cuda.synchronize()
