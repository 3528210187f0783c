#!/usr/bin/env python3
"""Per-layer timing of tgsr_lp_conv3x3_fwd on the generator's conv shapes (B=16): us per launch, TFLOP/s of the
direct-form FLOP count, algorithmic GB/s.   python tools/bench_lp_conv.py [--dtype bf16] [--batch 16] [--reps 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import lp  # noqa: E402

# (Cin, Cout, out H, glu, up, res, launches per forward)
LAYERS = [
    (64, 128, 32, True, False, False, 2), (64, 64, 32, False, False, True, 2), (64, 64, 64, True, True, False, 1),
    (64, 128, 64, True, False, False, 2), (64, 64, 64, False, False, True, 2), (64, 64, 128, True, True, False, 1),
    (64, 128, 128, True, False, False, 2), (64, 64, 128, False, False, True, 2), (64, 64, 256, True, True, False, 1),
    (32, 64, 32, True, False, False, 6), (32, 32, 32, False, False, True, 6), (32, 64, 64, True, True, False, 1),
    (32, 64, 64, True, False, False, 1), (32, 32, 64, False, False, False, 1), (32, 64, 128, True, True, False, 1),
    (32, 64, 128, True, False, False, 1), (32, 32, 128, False, False, False, 1), (32, 64, 256, True, True, False, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--direct-up", action="store_true", help="upBlocks by the direct 9-tap form instead of sub-pixel")
    ap.add_argument("--layers", default="", help="comma-separated indices into LAYERS (default: all)")
    a = ap.parse_args()
    layers = [LAYERS[int(i)] for i in a.layers.split(",")] if a.layers else LAYERS
    dev, B = "cuda", a.batch
    g = torch.Generator().manual_seed(0)
    tot_us = tot_fl = 0.0
    print("cin cout  H   glu up res |   us/launch  TFLOP/s  GB/s(alg) | x launches")
    for cin, cout, H, glu, up, res, n in layers:
        Hi = H // 2 if up else H
        x = lp.from_nchw(torch.randn(B, cin, Hi, Hi, generator=g).to(dev), a.dtype)
        w = lp.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev), a.dtype)
        co = cout // 2 if glu else cout
        r = lp.from_nchw(torch.randn(B, co, H, H, generator=g).to(dev), a.dtype) if res else None
        out = lp.new_image(B, H, H, co, a.dtype, dev)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        sub = up and not a.direct_up
        if sub:
            w = lp.pack_upconv_weight((torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev), a.dtype)

        def run():
            if sub:
                lp.upconv_glu(x, w, cin, cout, sc, sh, out=out)
            else:
                lp.conv3x3(x, w, cin, cout, sc, sh, glu=glu, upsample=up, residual=r, out=out)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        fl = 2.0 * B * H * H * cout * cin * 9
        by = 2.0 * (B * cin * Hi * Hi + B * co * H * H * (2 if res else 1))
        print("%3d %4d %4d  %d   %d  %d  | %9.1f  %8.1f  %8.1f | x%d" % (cin, cout, H, glu, up, res, us, fl / us / 1e6,
                                                                       by / us / 1e3, n))
        tot_us += us * n
        tot_fl += fl * n
    print("conv path per forward: %.1f us, %.1f TFLOP/s (direct-form FLOPs)" % (tot_us, tot_fl / tot_us / 1e6))


if __name__ == "__main__":
    main()
