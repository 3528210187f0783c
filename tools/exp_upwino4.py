#!/usr/bin/env python3
"""Up-sample-aware F(4x4,3x3) (tgsr_upwino4.hip) against the F(2x2) form (tgsr_upwino.hip) and F.conv2d(upsample): error and
time per upBlock shape (batch 16 unless B=...).     python tools/exp_upwino4.py      (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tgsr_amd import ops
B = int(os.environ.get("B", "16"))
dev = torch.device("cuda")
torch.manual_seed(0)

def timeit(f, n=30):
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

for (b, cin, cout, h, w, glu) in ((1, 8, 64, 4, 32, 1), (2, 8, 64, 8, 32, 0), (1, 16, 128, 6, 36, 1), (2, 64, 64, 20, 64, 1), (1, 32, 64, 64, 64, 1)):
    x = torch.randn(b, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    co = cout // 2 if glu else cout
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), wt.double(), None, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref = ref[:, :co] * torch.sigmoid(ref[:, co:]) if glu else ref
    o4 = ops.upwino4_glu(x, ops.pack_upwino4_weight(wt, glu=bool(glu)), cout, sc, sh, glu=bool(glu))
    o2 = ops.upwino_glu(x, ops.pack_upwino_weight(wt, glu=bool(glu)), cout, sc, sh, glu=bool(glu))
    print("B%d %d->%d %dx%d -> x2 glu%d: |F4-f64| max %.2e  |F2-f64| max %.2e" % (
        b, cin, cout, h, w, glu, float((o4.double() - ref).abs().max()), float((o2.double() - ref).abs().max())), flush=True)

for cin, cout, h in ((64, 64, 128), (64, 64, 64), (64, 64, 32), (32, 64, 128), (32, 64, 64), (32, 64, 32)):
    x = torch.randn(B, cin, h, h, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    out = torch.empty(B, cout // 2, 2 * h, 2 * h, device=dev)
    p4, p2 = ops.pack_upwino4_weight(wt), ops.pack_upwino_weight(wt)
    t4 = timeit(lambda: ops.upwino4_glu(x, p4, cout, sc, sh, out=out))
    t2 = timeit(lambda: ops.upwino_glu(x, p2, cout, sc, sh, out=out))
    flop = 2.0 * B * 4 * h * h * cout * cin * 9
    print("%d->%d %d^2 -> %d^2: F(4x4) %.1f us (%.0f TFLOP/s alg, executed frac %.3f)   F(2x2) %.1f us (%.0f, %.3f)" % (
        cin, cout, h, 2 * h, t4, flop / t4 / 1e6, flop * 25 / 144 / t4 / 1e6 / 157.3, t2, flop / t2 / 1e6, flop / 4 / t2 / 1e6 / 157.3), flush=True)
