#!/usr/bin/env python3
"""Which F(4x4) layer class costs how much of the distance to fp64?  Batch 16, shipped checkpoint, several seeds: the HIP pipeline
with layer classes switched on one at a time (diagnostic environment knobs of tgsr_amd.ops) against an fp64 run of the oracle,
next to the CPU fp32 oracle's own distance.     python tools/diag_precision_classes.py [seed ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tgsr_oracle as O
from tgsr_amd.miscc.config import cfg, cfg_reset
from tgsr_amd.trainer import SRPipeline
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
w = np.load(os.path.join(G, "face_S8_weights.npz"))
def sd(pre, dt=torch.float32):
    return {k[len(pre):]: (torch.from_numpy(w[k]).to(dt) if w[k].dtype.kind == 'f' else torch.from_numpy(w[k])) for k in w.files if k.startswith(pre)}
from tgsr_amd import ops, util
CLASSES = [("direct kernels only (util.WINOGRAD = False)", {"TGSR_WINO4": "0", "direct": "1"}),
           ("F(2x2) only", {"TGSR_WINO4": "0"}),
           ("+ 128^2 ResBlock convs (Cin 64)", {"TGSR_UPWINO4_MIN_CIN": "1000", "TGSR_WINO4_MIN_CIN": "64", "TGSR_WINO4_MIN_PIXELS": str(128 * 128)}),
           ("+ 64->128 at 64^2", {"TGSR_UPWINO4_MIN_CIN": "1000", "TGSR_WINO4_MIN_CIN": "64"}),
           ("+ NetG_highweight 32->64 at 128^2", {"TGSR_UPWINO4_MIN_CIN": "1000"}),
           ("+ upBlocks of G_SR_NET_low (Cin 64)", {"TGSR_UPWINO4_MIN_CIN": "64"}),
           ("+ upBlocks of NetG_highweight = shipped routing", {}),
           ("only the upBlocks", {"TGSR_WINO4_MIN_CIN": "100000"})]
cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256
for seed in [int(a) for a in sys.argv[1:]] or [100, 1100, 7]:
    cap, lens, LR, LRb = O.synthetic_batch(16, seed=seed)
    r32 = O.sr_forward(sd("E."), sd("GL."), sd("GH."), cap, lens.tolist(), LR, LRb)
    r64 = O.sr_forward(sd("E.", torch.float64), sd("GL.", torch.float64), sd("GH.", torch.float64), cap, lens.tolist(), LR.double(), LRb.double())
    def dist(r):
        return "  ".join("%s2 %.2e/%.1e" % (k, float((r[k][2].cpu().double() - r64[k][2]).abs().max()), float((r[k][2].cpu().double() - r64[k][2]).abs().mean())) for k in ("fake", "fine"))
    print("seed %d   CPU fp32 oracle vs fp64: %s" % (seed, dist(r32)), flush=True)
    for label, env in CLASSES:
        ops.ROUTING.reset(env)
        util.WINOGRAD = "direct" not in env
        p = SRPipeline(41, device="cuda", branch_num=4).load_state_dicts(sd("E."), sd("GL."), sd("GH."))
        r = p(cap.cuda(), lens.tolist(), LR.cuda(), LRb.cuda())
        e = (r["fine"][2].cpu().double() - r64["fine"][2]).abs()
        am = [int(v) for v in torch.nonzero(e == e.max())[0]]
        print("   %-50s vs fp64: %s   vs CPU fp32: fine2 %.2e   worst pixel %s value %.3f" % (
            label, dist(r), float((r["fine"][2].cpu() - r32["fine"][2]).abs().max()), am, float(r64["fine"][2][tuple(am)])), flush=True)
