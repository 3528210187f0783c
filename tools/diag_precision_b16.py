#!/usr/bin/env python3
"""HIP vs the fp32 CPU oracle on the batch-16 synthetic case of tests/test_hip_parity.py::test_full_size_batch16_vs_oracle: the
margin against the stated 1e-4 with the routing the product uses at batch 16 (TGSR_WINO4=0: the F(2x2) kernels only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tgsr_oracle as O
from tgsr_amd.miscc.config import cfg, cfg_reset
from tgsr_amd.trainer import SRPipeline
from tgsr_amd import ops
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
w = np.load(os.path.join(G, "face_S8_weights.npz"))
def sd(pre, dt=torch.float32):
    return {k[len(pre):]: (torch.from_numpy(w[k]).to(dt) if w[k].dtype.kind == 'f' else torch.from_numpy(w[k])) for k in w.files if k.startswith(pre)}
cap, lens, LR, LRb = O.synthetic_batch(16)
ref = O.sr_forward(sd("E."), sd("GL."), sd("GH."), cap, lens.tolist(), LR, LRb)
cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256
for mode, extra in (("1", {}), ("1", {"TGSR_UPWINO4_MIN_CIN": "64"}), ("1", {"TGSR_UPWINO4_MIN_CIN": "1000"}), ("0", {})):
    ops.ROUTING.reset(dict(extra, TGSR_WINO4=mode))
    p = SRPipeline(41, device="cuda", branch_num=4).load_state_dicts(sd("E."), sd("GL."), sd("GH."))
    r = p(cap.cuda(), lens.tolist(), LR.cuda(), LRb.cuda())
    print("TGSR_WINO4=%s %s:" % (mode, extra), "  ".join("%s%d %.2e/%.1e" % (k, i, float((r[k][i].cpu() - ref[k][i]).abs().max()), float((r[k][i].cpu() - ref[k][i]).abs().mean()))
                                            for k in ("fake", "fine") for i in range(3)), flush=True)
