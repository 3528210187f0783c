#!/usr/bin/env python3
"""The discriminators' convolutions layer by layer (D_NET256's shapes at the G/D step's batch: 32 = real + fake, DF_DIM 64):
forward, data gradient, weight gradient - us per launch and TFLOP/s of the 157.3 fp32 MFMA peak.  profiles/HISTORY.md 3.9.
    python tools/exp_dconv.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda"
LAYERS = [(4, 3, 64, 256), (4, 64, 128, 128), (4, 128, 256, 64), (4, 256, 512, 32), (4, 512, 1024, 16), (4, 1024, 2048, 8),
          (3, 2048, 1024, 4), (3, 1024, 512, 4)]
def timeit(fn, n=10):
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
tot = [0.0, 0.0]
print("kind Cin->Cout @H        fwd us (TF/s)      dgrad us (TF/s)     wgrad us (TF/s)")
for kind, ci, co, H in LAYERS:
    x = torch.randn(B, ci, H, H, device=dev)
    w = torch.randn(co, ci, kind, kind, device=dev) / (ci * kind * kind) ** 0.5
    Ho = H // 2 if kind == 4 else H
    dy = torch.randn(B, co, Ho, Ho, device=dev)
    fl = 2.0 * B * Ho * Ho * co * ci * kind * kind
    if kind == 4:
        f = lambda: ops.conv4x4s2(x, w)
        d = lambda: ops.conv4x4s2_dgrad(dy, w, H, H)
        g = lambda: ops.conv4x4s2_wgrad(dy, x)
    else:
        f = lambda: ops.conv3x3_gemm(x, w)
        d = lambda: ops.conv3x3_gemm_dgrad(dy, w)
        g = lambda: ops.conv3x3_gemm_wgrad(dy, x)
    tf, td, tg = timeit(f), timeit(d), timeit(g)
    tot[0] += tf + td + tg
    tot[1] += 3 * fl
    print("%dx%d %5d->%-5d @%-4d  %8.1f (%5.1f)   %8.1f (%5.1f)   %8.1f (%5.1f)" % (kind, kind, ci, co, H, tf, fl / tf / 1e6, td, fl / td / 1e6, tg, fl / tg / 1e6), flush=True)
print("sum %.1f us, %.1f TFLOP/s = %.2f of the fp32 MFMA peak" % (tot[0], tot[1] / tot[0] / 1e6, tot[1] / tot[0] / 1e6 / 157.3))
