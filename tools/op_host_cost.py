#!/usr/bin/env python3
"""Host cost per call of a training operator at a tiny size (the kernel itself is ~3 us): through torch.ops.tgsr (dispatcher), through
the tgsr_amd.ops wrapper, and the bare ctypes call with prebuilt arguments.  python tools/op_host_cost.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops, _lib, custom_ops  # noqa: F401
dev = "cuda"
B, C, H = 2, 32, 8
raw = torch.randn(B, C, H, H, device=dev)
dout = torch.randn(B, C, H, H, device=dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
out, stats = ops.bn_train_fwd(raw, gamma, beta, 1e-5, 0.1, rm, rv, 0)
dg, db, dr = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty_like(raw)
def t(fn, n=3000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6
L = _lib.lib()
ws = torch.empty(C * L.tgsr_bn_train_nsplit(B, C, H * H) * 4, device=dev)
p = lambda x: ctypes.c_void_p(x.data_ptr())
args = (p(dout), p(raw), B, C, H * H, p(stats[2]), p(stats[3]), p(stats[0]), p(stats[1]), 0, p(ws), p(ws), p(dr), p(dg), p(db), ops._stream())
print("bn_train_bwd  bare ctypes call, prebuilt args : %5.1f us" % t(lambda: L.tgsr_bn_train_bwd(*args)))
print("bn_train_bwd  ctypes call + building its args : %5.1f us" % t(lambda: L.tgsr_bn_train_bwd(p(dout), p(raw), B, C, H * H, p(stats[2]), p(stats[3]), p(stats[0]), p(stats[1]), 0, p(ws), p(ws), p(dr), p(dg), p(db), ops._stream())))
print("bn_train_bwd  ops.bn_train_bwd (slots given)  : %5.1f us" % t(lambda: ops.bn_train_bwd(dout, raw, stats, 0, dg, db, dr)))
print("bn_train_bwd  ops.bn_train_bwd (allocating)   : %5.1f us" % t(lambda: ops.bn_train_bwd(dout, raw, stats, 0)))
print("bn_train_bwd  torch.ops.tgsr.bn_train_bwd     : %5.1f us" % t(lambda: torch.ops.tgsr.bn_train_bwd(dout, raw, stats, 0, dg, db, dr)))
print("torch.empty(C)                                : %5.1f us" % t(lambda: torch.empty(C, dtype=torch.float32, device=raw.device)))
print("torch.empty_like(raw)                         : %5.1f us" % t(lambda: torch.empty_like(raw)))
print("stats[2] (select)                             : %5.1f us" % t(lambda: stats[2]))
print("ops._stream()                                 : %5.1f us" % t(lambda: ops._stream()))
print("ops._need_hip(6 tensors)                      : %5.1f us" % t(lambda: ops._need_hip(dout, raw, stats, dg, db, dr)))
print("dout.contiguous()                             : %5.1f us" % t(lambda: dout.contiguous()))
print("x.add_(1) (a torch eager kernel, for scale)   : %5.1f us" % t(lambda: dg.add_(1.0)))
x = torch.randn(2, 32, 16, 16, device=dev)
w = torch.randn(64, 32, 4, 4, device=dev)
print("conv4x4s2     ops.conv4x4s2                   : %5.1f us" % t(lambda: ops.conv4x4s2(x, w)))
print("conv4x4s2     torch.ops.tgsr.conv4x4s2        : %5.1f us" % t(lambda: torch.ops.tgsr.conv4x4s2(x, w, False)))
