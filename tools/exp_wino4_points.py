#!/usr/bin/env python3
"""Which interpolation points should Winograd F(4x4,3x3) use in fp32?  (CPU; profiles/HISTORY.md 3.1g.)

Two measurements per candidate point set {0, p1..p4, inf} (the Toom-Cook matrices are built from the points in exact rationals):
 (1) per layer, unit-scale data, the accumulation over input channels done the way the MFMA kernel does it (sequentially, one fp32
     rounding per channel: tools/exp_wino4_numerics.py lets a BLAS einsum do it, which is why it under-estimates what the GPU shows);
 (2) end to end on the shipped checkpoint (full-size C1 case of tests/golden), the fp32 oracle with the convolutions that
     `ops.wino4_wanted` / `ops.upwino4_wanted` route at batch 16 replaced by the emulation, against the fp64 oracle.

    python tools/exp_wino4_points.py layer
    python tools/exp_wino4_points.py e2e [plain-points] [up-points]     e.g.  e2e 1,-1,1/2,-2  1,-1,2,-2
"""
import os, sys
from fractions import Fraction as Fr
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def polymul(a, b):
    r = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            r[i + j] += x * y
    return r


def toom(points, m=4, r=3):
    """Bt [n][n], G [n][r], At [m][n] for the finite points (0 included by the caller) plus infinity, exact."""
    n = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == n - 1
    Mx = [Fr(1)]
    for p in pts:
        Mx = polymul(Mx, [-p, Fr(1)])
    Bt, G, At = [], [], [[Fr(0)] * n for _ in range(m)]
    for j, p in enumerate(pts):
        q, N = [Fr(1)], Fr(1)
        for k, pk in enumerate(pts):
            if k != j:
                q = polymul(q, [-pk, Fr(1)])
                N *= p - pk
        Bt.append(q + [Fr(0)] * (n - len(q)))
        G.append([p ** k / N for k in range(r)])
        for i in range(m):
            At[i][j] = p ** i
    Bt.append(Mx + [Fr(0)] * (n - len(Mx)))
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    At[m - 1][n - 1] = Fr(1)
    return Bt, G, At


def mats(points):
    return [torch.tensor([[float(x) for x in row] for row in M], dtype=torch.float64) for M in toom([0] + list(points))]


def parse(s):
    return [Fr(x) for x in s.split(",")]


def wino_seq(x, wt, points, chunk=1):
    """F(4x4,3x3) of x [B,C,H,W] with wt [Co,C,3,3]: U in double rounded once, V = Bt d B in fp32, M accumulated over the input
    channels in fp32 `chunk` channels at a time in order, Y = At M A in fp32."""
    Bt, G, At = mats(points)
    B_, C, H, W = x.shape
    U = (G @ wt.double() @ G.T).float()                                   # [Co, C, 6, 6]
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 6, 4).unfold(3, 6, 4)            # [B, C, th, tw, 6, 6]
    Btf, Atf = Bt.float(), At.float()
    V = Btf @ d @ Btf.T
    th, tw = V.shape[2], V.shape[3]
    M = torch.zeros(B_, U.shape[0], th, tw, 6, 6)
    for c in range(0, C, chunk):
        for cc in range(c, min(C, c + chunk)):
            M += U[None, :, cc, None, None] * V[:, None, cc]
    Y = Atf @ M @ Atf.T
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, U.shape[0], th * 4, tw * 4)


def layer_table():
    H = Fr(1, 2)
    cands = [("1,-1,2,-2 (Lavin; round 4)", [1, -1, 2, -2]), ("1,-1,1/2,-1/2", [1, -1, H, -H]), ("1,-1,1/2,-2", [1, -1, H, -2]),
             ("1,-1,-1/2,2", [1, -1, -H, 2]), ("1/2,-1/2,2,-2", [H, -H, 2, -2]), ("1,-1,3/2,-3/2", [1, -1, Fr(3, 2), -Fr(3, 2)]),
             ("1,-1,2/3,-2/3", [1, -1, Fr(2, 3), -Fr(2, 3)]), ("1,-1,2/3,-3/2", [1, -1, Fr(2, 3), -Fr(3, 2)]),
             ("1,-1,3/4,-4/3", [1, -1, Fr(3, 4), -Fr(4, 3)]), ("1/2,-2,2,-1/2 ", [H, -2, 2, -H]),
             ("1,-1,1/2,-3", [1, -1, H, -3]), ("1,-1,1/3,-3", [1, -1, Fr(1, 3), -3]), ("1,-1,1/3,-2", [1, -1, Fr(1, 3), -2])]
    torch.manual_seed(0)
    for C in (32, 64, 128):
        x = torch.randn(2, C, 64, 64)
        w = torch.randn(64, C, 3, 3) / (9 * C) ** 0.5
        ref = F.conv2d(x.double(), w.double(), padding=1)
        e = (F.conv2d(x, w, padding=1).double() - ref).abs()
        print("Cin %3d   direct fp32 (torch)            max %.2e mean %.2e" % (C, e.max(), e.mean()))
        for name, p in cands:
            e = (wino_seq(x, w, p).double() - ref).abs()
            print("          %-30s max %.2e mean %.2e" % (name, e.max(), e.mean()), flush=True)


def e2e(plain_pts, up_pts):
    from oracle import tgsr_oracle as O
    from tgsr_amd import ops
    G_ = os.path.join(ROOT, "tests", "golden")
    w = np.load(os.path.join(G_, "face_S8_weights.npz")); g = np.load(os.path.join(G_, "face_S8_c1.npz"))

    def sd(pre, dt=torch.float32):
        return {k[len(pre):]: (torch.from_numpy(w[k]).to(dt) if w[k].dtype.kind == 'f' else torch.from_numpy(w[k]))
                for k in w.files if k.startswith(pre)}
    cap, lens = torch.from_numpy(g["captions"]), g["cap_lens"].tolist()
    LR, LRb = torch.from_numpy(g["LR"]), torch.from_numpy(g["LRb"])
    state = {"on": False, "up": False, "plain": plain_pts, "upp": up_pts, "calls": [], "min_up": 128, "min_plain": None}
    orig, orig_up = F.conv2d, O.up_block

    def up_block(*a, **k):
        state["up"] = True
        try:
            return orig_up(*a, **k)
        finally:
            state["up"] = False

    def patched(x, wt, b=None, s=1, p=0, *a, **k):
        if state["on"] and x.dtype == torch.float32 and wt.shape[2:] == (3, 3) and s == 1 and p == 1 and b is None:
            h, ci, co = x.shape[2], wt.shape[1], wt.shape[0]
            if state["up"]:
                take = state["upp"] is not None and h >= state["min_up"] and ops.upwino4_wanted(ci, co, h // 2, h // 2, 16)
                pts = state["upp"]
            else:
                take = state["plain"] is not None and (ops.wino4_wanted(ci, co, h, h, 16) if state["min_plain"] is None else
                                                       (h >= state["min_plain"] and ci % 8 == 0 and co % 64 == 0))
                pts = state["plain"]
            if take:
                state["calls"].append(("up" if state["up"] else "conv", h, ci, co))
                return wino_seq(x, wt, pts)
        return orig(x, wt, b, s, p, *a, **k)
    O.F.conv2d = patched
    O.up_block = up_block
    r64 = O.sr_forward(sd("E.", torch.float64), sd("GL.", torch.float64), sd("GH.", torch.float64), cap, lens, LR.double(), LRb.double())

    def run(label, on, **kw):
        state.update(on=on, **kw)
        del state["calls"][:]
        r = O.sr_forward(sd("E."), sd("GL."), sd("GH."), cap, lens, LR, LRb)
        out = []
        for k in ("fake", "fine"):
            e = (r[k][2].double() - r64[k][2]).abs()
            out.append("%s2 %.2e/%.1e" % (k, e.max(), e.mean()))
        print(label.ljust(60), "  ".join(out), " layers:", len(state["calls"]), flush=True)
    std = [1, -1, 2, -2]
    run("direct fp32 (torch CPU)", False)
    run("round-4 points everywhere routed at batch 16", True, plain=std, upp=std, min_up=128, min_plain=None)
    run("plain %s, up %s, routed as at batch 16" % (plain_pts, up_pts), True, plain=plain_pts, upp=up_pts)
    run("  + every >= 64^2 conv and upBlock", True, plain=plain_pts, upp=up_pts, min_up=64, min_plain=64)
    run("  + every >= 32^2 conv and upBlock", True, plain=plain_pts, upp=up_pts, min_up=32, min_plain=32)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    if len(sys.argv) > 1 and sys.argv[1] == "e2e":
        e2e(parse(sys.argv[2]) if len(sys.argv) > 2 else [1, -1, Fr(1, 2), -2], parse(sys.argv[3]) if len(sys.argv) > 3 else [1, -1, 2, -2])
    else:
        layer_table()
