#!/usr/bin/env python3
"""What Winograd F(4x4,3x3) in fp32 costs END TO END on the shipped checkpoint, layer set by layer set (CPU; the fp32 oracle with
its conv3x3 replaced by an emulation of the algorithm - U = G g G^T rounded once, V = B^T d B, products and A^T M A in fp32 -
against the fp64 oracle; full-size C1 case of tests/golden).  The table in profiles/HISTORY.md 3.1e is this script's output; it decides
which layers tgsr_amd.ops.wino4_wanted routes to tgsr_winograd4.hip.     python tools/exp_wino4_numerics.py"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tgsr_oracle as O
G_ = os.path.join(ROOT, "tests", "golden")
w = np.load(os.path.join(G_, "face_S8_weights.npz")); g = np.load(os.path.join(G_, "face_S8_c1.npz"))
def sd(pre, dt=torch.float32):
    return {k[len(pre):]: (torch.from_numpy(w[k]).to(dt) if w[k].dtype.kind == 'f' else torch.from_numpy(w[k])) for k in w.files if k.startswith(pre)}
cap, lens = torch.from_numpy(g["captions"]), g["cap_lens"].tolist()
LR, LRb = torch.from_numpy(g["LR"]), torch.from_numpy(g["LRb"])

def mats(m):
    if m == 2:
        Bt = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
        G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
        At = [[1, 1, 1, 0], [0, 1, -1, -1]]
    else:
        Bt = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
        G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
        At = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    return [torch.tensor(a, dtype=torch.float64) for a in (Bt, G, At)]

def wino(x, wt, m):
    Bt, G, At = mats(m)
    n = m + 2
    B_, C, H, W = x.shape
    U = (G @ wt.double() @ G.T).float()
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, n, m).unfold(3, n, m)
    Btf, Atf = Bt.float(), At.float()
    V = Btf @ d @ Btf.T
    M = torch.einsum("ocuv,bcijuv->boijuv", U, V)
    Y = Atf @ M @ Atf.T
    th, tw = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, U.shape[0], th * m, tw * m)

MODE = {"m": 0, "pick": lambda h, ci, co: False}
CALLS = []
orig = F.conv2d
def patched(x, wt, b=None, s=1, p=0, *a, **k):
    if (MODE["m"] and x.dtype == torch.float32 and wt.shape[2:] == (3, 3) and s == 1 and p == 1 and b is None and x.shape[2] % 4 == 0
            and MODE["pick"](x.shape[2], wt.shape[1], wt.shape[0]) and CALLS.append((x.shape[2], wt.shape[1], wt.shape[0])) is None):
        return wino(x, wt, MODE["m"])
    return orig(x, wt, b, s, p, *a, **k)
O.F.conv2d = patched

r64 = O.sr_forward(sd("E.", torch.float64), sd("GL.", torch.float64), sd("GH.", torch.float64), cap, lens, LR.double(), LRb.double())
def run(m, pick, label):
    MODE["m"], MODE["pick"] = m, pick
    del CALLS[:]
    r = O.sr_forward(sd("E."), sd("GL."), sd("GH."), cap, lens, LR, LRb)
    out = []
    for k in ("fake", "fine"):
        for i in range(3):
            e = (r[k][i].double() - r64[k][i]).abs()
            out.append("%s%d %.2e/%.1e" % (k, i, e.max(), e.mean()))
    print(label.ljust(46), "  ".join(out), " convolutions on it:", len(CALLS), flush=True)
res = lambda h, ci, co: ci == 64 and co in (64, 128)          # the ResBlock convolutions of G_SR_NET_low (64 -> 128, 64 -> 64)
run(0, lambda *a: False, "direct fp32 (torch CPU)")
run(2, lambda h, ci, co: ci % 32 == 0 and co % 32 == 0, "F(2x2) on every 3x3 layer")
run(4, lambda h, ci, co: h == 128 and res(h, ci, co), "F(4x4): the 128^2 layers")
run(4, lambda h, ci, co: (h == 128 and res(h, ci, co)) or (h == 64 and ci == 64 and co == 128), "F(4x4): 128^2 + the 64 -> 128 ones at 64^2")
run(4, lambda h, ci, co: h in (64, 128) and res(h, ci, co), "F(4x4): 128^2 + 64^2")
from tgsr_amd import ops
ops.ROUTING.reset({})
# the rule the product uses, at batch 16 (a superset of what runs on the kernel: the upBlocks' convolutions have the shapes of
# ResBlock convolutions here, on the GPU they are the up-sample-aware kernel's)
run(4, lambda h, ci, co: ops.wino4_wanted(ci, co, h, h, 16), "F(4x4): as ops.wino4_wanted routes at batch 16")
run(4, lambda h, ci, co: ci % 32 == 0 and co % 32 == 0, "F(4x4) on every 3x3 layer")
