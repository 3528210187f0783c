#!/usr/bin/env python3
"""wino_conv3x3_kernel per layer of the generators (batch 16 unless B=...): microseconds per launch (100 back-to-back launches
between two events), algorithmic TFLOP/s and the executed fraction of the fp32 MFMA peak (16 of 36 products, 157.3 TFLOP/s).
   python tools/exp_wino.py            (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgsr_amd import ops

B = int(os.environ.get("B", "16"))
dev = torch.device("cuda")
layers = ((64, 128, 128, 1, 0), (64, 64, 128, 0, 1), (64, 128, 64, 1, 0), (64, 64, 64, 0, 1), (64, 128, 32, 1, 0), (64, 64, 32, 0, 1),
          (32, 64, 32, 1, 0), (32, 64, 128, 1, 0))
print("Cin Cout  H  glu res |   us   alg TFLOP/s  executed frac | workgroups")
tot = 0.0
for cin, cout, h, glu, res in layers:
    x = torch.randn(B, cin, h, h, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    up = ops.pack_wino_weight(w, glu=bool(glu))
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    co = cout // 2 if glu else cout
    r = torch.randn(B, co, h, h, device=dev) if res else None
    out = torch.empty(B, co, h, h, device=dev)
    f = lambda: ops.conv3x3_wino(x, up, cout, sc, sh, bool(glu), r, out)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    flop = 2.0 * B * h * h * cout * cin * 9
    print("%3d %4d %4d  %d   %d  | %7.1f  %7.1f   %.3f" % (cin, cout, h, glu, res, us, flop / us / 1e6, flop * 16 / 36 / us / 1e6 / 157.3))
