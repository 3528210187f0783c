#!/usr/bin/env python3
"""F(4x4,3x3) kernel against F(2x2,3x3) and F.conv2d: error and time per layer (batch 16 unless B=...).
   python tools/exp_wino4.py            (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tgsr_amd import ops

B = int(os.environ.get("B", "16"))
dev = torch.device("cuda")
torch.manual_seed(0)

def timeit(f, n=50):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

# small correctness cases first (odd sizes: partial tiles, one stage, several groups)
for (b, cin, cout, h, w, glu, res) in ((1, 4, 64, 8, 64, 0, 0), (2, 8, 64, 16, 64, 1, 0), (1, 12, 128, 12, 68, 0, 1), (2, 64, 128, 40, 128, 1, 0), (2, 8, 256, 10, 72, 1, 0),
                                       (1, 32, 64, 128, 128, 0, 1)):
    x = torch.randn(b, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    co = cout // 2 if glu else cout
    r = torch.randn(b, co, h, w, device=dev) if res else None
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref = ref[:, :co] * torch.sigmoid(ref[:, co:]) if glu else (ref + r.double() if res else ref)
    up4 = ops.pack_wino4_weight(wt, glu=bool(glu))
    o4 = ops.conv3x3_wino4(x, up4, cout, sc, sh, bool(glu), r)
    up2 = ops.pack_wino_weight(wt, glu=bool(glu))
    o2 = ops.conv3x3_wino(x, up2, cout, sc, sh, bool(glu), r)
    ew = -1.0
    if cin % 8 == 0:
        ow = ops.conv3x3_wino4(x, ops.pack_wino4w_weight(wt, glu=bool(glu)), cout, sc, sh, bool(glu), r, wide=True)
        ew = float((ow.double() - ref).abs().max())
    print("B%d %d->%d %dx%d glu%d res%d: |F4-f64| max %.2e  |F2-f64| max %.2e  |F4wide-f64| %.2e" % (
        b, cin, cout, h, w, glu, res, float((o4.double() - ref).abs().max()), float((o2.double() - ref).abs().max()), ew), flush=True)

for cin, cout, h, glu, res in ((64, 128, 128, 1, 0), (64, 64, 128, 0, 1), (64, 128, 64, 1, 0), (64, 64, 64, 0, 1), (32, 128, 128, 1, 0), (128, 128, 128, 1, 0)):
    x = torch.randn(B, cin, h, h, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    co = cout // 2 if glu else cout
    r = torch.randn(B, co, h, h, device=dev) if res else None
    out = torch.empty(B, co, h, h, device=dev)
    up4, up2 = ops.pack_wino4_weight(wt, glu=bool(glu)), ops.pack_wino_weight(wt, glu=bool(glu))
    t4 = timeit(lambda: ops.conv3x3_wino4(x, up4, cout, sc, sh, bool(glu), r, out))
    t2 = timeit(lambda: ops.conv3x3_wino(x, up2, cout, sc, sh, bool(glu), r, out))
    if cin % 8 == 0:
        upw = ops.pack_wino4w_weight(wt, glu=bool(glu))
        tw = timeit(lambda: ops.conv3x3_wino4(x, upw, cout, sc, sh, bool(glu), r, out, wide=True))
        print("   wide: %.1f us (%.0f TFLOP/s alg, executed frac %.3f)" % (tw, 2.0 * B * h * h * cout * cin * 9 / tw / 1e6, 2.0 * B * h * h * cout * cin * 9 / 4 / tw / 1e6 / 157.3))
    flop = 2.0 * B * h * h * cout * cin * 9
    print("%d->%d @%d glu%d res%d: F(4x4) %.1f us (%.0f TFLOP/s alg, executed frac %.3f)   F(2x2) %.1f us (%.0f, %.3f)" % (
        cin, cout, h, glu, res, t4, flop / t4 / 1e6, flop / 4 / t4 / 1e6 / 157.3, t2, flop / t2 / 1e6, flop * 16 / 36 / t2 / 1e6 / 157.3), flush=True)
