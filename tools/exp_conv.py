#!/usr/bin/env python3
"""Ad-hoc conv timing: python tools/exp_conv.py B Cin Cout H glu up res [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops
def run(B, cin, cout, h, glu, up, res, reps=10):
    dev = "cuda"
    x = torch.randn(B, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    wp = ops.pack_conv3x3_weight(w); sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * 0.1
    ho = 2 * h if up else h; co = cout // 2 if glu else cout
    r = torch.randn(B, co, ho, ho, device=dev) if res else None
    out = torch.empty(B, co, ho, ho, device=dev)
    for _ in range(2): ops.conv3x3_fused(x, wp, cout, sc, sh, glu=bool(glu), upsample=bool(up), residual=r, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ops.conv3x3_fused(x, wp, cout, sc, sh, glu=bool(glu), upsample=bool(up), residual=r, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps; fl = 2.0 * B * ho * ho * cout * cin * 9
    print("B%d %d->%d @%d glu%d up%d res%d: %.1f us  %.1f TFLOP/s" % (B, cin, cout, h, glu, up, res, us, fl / us / 1e6))
if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    if a: run(*a)
    else:
        for cin in (64, 128, 256, 512): run(16, cin, 128, 128, 1, 0, 0)
        for B in (16, 32, 64): run(B, 64, 128, 128, 1, 0, 0)
        for cin in (64, 256): run(16, cin, 64, 128, 0, 0, 1)


def run_wino(B, cin, cout, h, glu, res, reps=10):
    dev = "cuda"
    x = torch.randn(B, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    up = ops.pack_wino_weight(w, glu=bool(glu)); sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * 0.1
    co = cout // 2 if glu else cout
    r = torch.randn(B, co, h, h, device=dev) if res else None
    out = torch.empty(B, co, h, h, device=dev)
    for _ in range(2): ops.conv3x3_wino(x, up, cout, sc, sh, glu=bool(glu), residual=r, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ops.conv3x3_wino(x, up, cout, sc, sh, glu=bool(glu), residual=r, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps; fl = 2.0 * B * h * h * cout * cin * 9
    print("WINO B%d %d->%d @%d glu%d res%d: %.1f us  %.1f TFLOP/s (algorithmic)" % (B, cin, cout, h, glu, res, us, fl / us / 1e6))
