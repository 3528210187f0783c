#!/usr/bin/env python3
"""Training-mode BatchNorm passes (tgsr_bn.hip) at the generator's shapes, batch 16: microseconds and algorithmic GB/s per pass.
   python tools/bench_bn.py            (on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgsr_amd import ops

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = torch.device("cuda")
B = int(os.environ.get("B", "16"))
shapes = [(128, 32, 1), (128, 64, 1), (128, 128, 1), (64, 32, 0), (64, 64, 0), (64, 128, 0), (64, 64, 1), (64, 128, 1), (64, 256, 1),
          (64, 32, 1), (32, 32, 0)]
print("raw C, H, act | stats us GB/s | fwd(from stats) us GB/s | bwd reduce+apply us GB/s")
for C, H, act in shapes:
    raw = torch.randn(B, C, H, H, device=dev)
    co = C // 2 if act == 1 else C
    g, bt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    res = torch.randn(B, co, H, H, device=dev) if act == 0 else None
    out = torch.empty(B, co, H, H, device=dev)
    stats = torch.empty(4, C, device=dev)
    dout = torch.randn(B, co, H, H, device=dev)
    draw = torch.empty_like(raw)
    rb = raw.numel() * 4
    ob = out.numel() * 4
    t_all = timeit(lambda: ops.bn_train_fwd(raw, g, bt, 1e-5, 0.1, rm, rv, act, res, None, out, stats))
    # the statistics pass alone: a bn_train_fwd on a non-GLU view costs stats + apply; time stats via the difference with from-stats
    ns = 64
    sp = torch.rand(C, ns, 2, device=dev)
    t_fwd = timeit(lambda: ops.bn_train_fwd(raw, g, bt, 1e-5, 0.1, rm, rv, act, res, None, out, stats, sp))
    t_bwd = timeit(lambda: ops.bn_train_bwd(dout, raw, stats, act, draw=draw))
    fwd_bytes = rb + ob + (ob if res is not None else 0)
    bwd_bytes = 2 * (ob + rb) + rb        # reduce reads dy + raw, apply reads dy + raw and writes draw
    print("%4d %4d %d | %7.1f %6.0f | %7.1f %6.0f | %7.1f %6.0f" % (C, H, act, t_all - t_fwd, rb / (t_all - t_fwd) / 1e3 if t_all > t_fwd else 0,
          t_fwd, fwd_bytes / t_fwd / 1e3, t_bwd, bwd_bytes / t_bwd / 1e3))
