"""Timing: sub-pixel upconv kernel vs the up-sample-aware Winograd kernel on the upBlock shapes (B=16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops
dev = "cuda"
for cin, cout, h in ((64, 64, 128), (64, 64, 64), (64, 64, 32), (32, 64, 128), (32, 64, 64), (32, 64, 32)):
    B = 16
    x = torch.randn(B, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * 0.1
    out = torch.empty(B, cout // 2, 2 * h, 2 * h, device=dev)
    p0, p1 = ops.pack_upconv_weight(w), ops.pack_upwino_weight(w)
    res = []
    for name, fn in (("subpixel", lambda: ops.upconv3x3_glu(x, p0, cout, sc, sh, out=out)),
                     ("upwino", lambda: ops.upwino_glu(x, p1, cout, sc, sh, out=out))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        res.append("%s %.1f us (%.0f TFLOP/s alg)" % (name, us, 2.0 * B * 4 * h * h * cout * cin * 9 / us / 1e6))
    print("B16 %d->%d %d->%d: %s" % (cin, cout, h, 2 * h, "   ".join(res)))
