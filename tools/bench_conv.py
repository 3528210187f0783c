#!/usr/bin/env python3
"""Per-shape timing of tgsr_conv3x3_fwd over the conv shapes of one x8 SR forward (B=16).  GPU only.
    python tools/bench_conv.py [--reps 20] [--batch 16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops  # noqa: E402

# (name, count per forward, Cin, Cout, H_in, glu, up, res)
SHAPES = [
    ("GL.im2f 3->64 glu @32", 1, 3, 64, 32, 1, 0, 0),
    ("GL.rb1 64->128 glu @32", 2, 64, 128, 32, 1, 0, 0),
    ("GL.rb2 64->64 +res @32", 2, 64, 64, 32, 0, 0, 1),
    ("GL.up 64->64 glu 32->64", 1, 64, 64, 32, 1, 1, 0),
    ("GL.rb1 64->128 glu @64", 2, 64, 128, 64, 1, 0, 0),
    ("GL.rb2 64->64 +res @64", 2, 64, 64, 64, 0, 0, 1),
    ("GL.up 64->64 glu 64->128", 1, 64, 64, 64, 1, 1, 0),
    ("GL.rb1 64->128 glu @128", 2, 64, 128, 128, 1, 0, 0),
    ("GL.rb2 64->64 +res @128", 2, 64, 64, 128, 0, 0, 1),
    ("GL.up 64->64 glu 128->256", 1, 64, 64, 128, 1, 1, 0),
    ("GH.convin 3->64 glu @32", 1, 3, 64, 32, 1, 0, 0),
    ("GH.rb1 32->64 glu @32", 6, 32, 64, 32, 1, 0, 0),
    ("GH.rb2 32->32 +res @32", 6, 32, 32, 32, 0, 0, 1),
    ("GH.up 32->64 glu 32->64", 1, 32, 64, 32, 1, 1, 0),
    ("GH.r24a 32->64 glu @64", 1, 32, 64, 64, 1, 0, 0),
    ("GH.r24b 32->32 @64", 1, 32, 32, 64, 0, 0, 0),
    ("GH.up 32->64 glu 64->128", 1, 32, 64, 64, 1, 1, 0),
    ("GH.r48a 32->64 glu @128", 1, 32, 64, 128, 1, 0, 0),
    ("GH.r48b 32->32 @128", 1, 32, 32, 128, 0, 0, 0),
    ("GH.up 32->64 glu 128->256", 1, 32, 64, 128, 1, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--only", default="")
    ap.add_argument("--ninetap", action="store_true", help="time the 9-tap folded-upsample kernel for the upBlocks")
    a = ap.parse_args()
    dev = "cuda"
    B = a.batch
    tot_us = tot_fl = 0.0
    print("%-30s %3s %10s %10s %8s" % ("shape", "n", "us/launch", "TFLOP/s", "us*n"))
    for name, n, cin, cout, h, glu, up, res in SHAPES:
        if a.only and a.only not in name:
            continue
        x = torch.randn(B, cin, h, h, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        wp = ops.pack_conv3x3_weight(w)
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.randn(cout, device=dev) * 0.1
        ho = 2 * h if up else h
        co = cout // 2 if glu else cout
        r = torch.randn(B, co, ho, ho, device=dev) if res else None
        out = torch.empty(B, co, ho, ho, device=dev)
        if up and glu and cout % 64 == 0 and not a.ninetap:
            wpu = ops.pack_upconv_weight(w)
            run = lambda: ops.upconv3x3_glu(x, wpu, cout, sc, sh, out=out)
        else:
            run = lambda: ops.conv3x3_fused(x, wp, cout, sc, sh, glu=bool(glu), upsample=bool(up), residual=r, out=out)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        fl = 2.0 * B * ho * ho * cout * cin * 9
        print("%-30s %3d %10.1f %10.1f %8.1f" % (name, n, us, fl / us / 1e6, us * n))
        tot_us += us * n
        tot_fl += fl * n
    print("TOTAL conv3x3 per forward: %.1f us, %.1f GFLOP, %.1f TFLOP/s (%.1f%% of 157.3)"
          % (tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6, tot_fl / tot_us / 1e6 / 1.573))


if __name__ == "__main__":
    main()
