#!/usr/bin/env python3
"""HIP vs fp64 oracle vs fp32 golden on the full-size C1 case (precision diagnosis)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tgsr_oracle as O
from tgsr_amd.miscc.config import cfg, cfg_reset
from tgsr_amd.trainer import SRPipeline
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
w = np.load(os.path.join(G, "face_S8_weights.npz")); g = np.load(os.path.join(G, "face_S8_c1.npz"))
def sd(pre, dt=torch.float32):
    return {k[len(pre):]: (torch.from_numpy(w[k]).to(dt) if w[k].dtype.kind == 'f' else torch.from_numpy(w[k])) for k in w.files if k.startswith(pre)}
cap, lens = torch.from_numpy(g["captions"]), g["cap_lens"].tolist()
LR, LRb = torch.from_numpy(g["LR"]), torch.from_numpy(g["LRb"])
r64 = O.sr_forward(sd("E.", torch.float64), sd("GL.", torch.float64), sd("GH.", torch.float64), cap, lens, LR.double(), LRb.double())
cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256
p = SRPipeline(41, device="cuda", branch_num=4).load_state_dicts(sd("E."), sd("GL."), sd("GH."))
r = p(cap.cuda(), lens, LR.cuda(), LRb.cuda())
for k in ("fake", "fine"):
    for i in range(3):
        hip = r[k][i].cpu().double(); ref = r64[k][i]; gold = torch.from_numpy(g["%s%d" % (k, i)]).double()
        print("%s%d: |hip-f64| max %.3e mean %.3e   |cpu32-f64| max %.3e mean %.3e   |hip-cpu32| max %.3e   range %.2f" % (
            k, i, (hip - ref).abs().max(), (hip - ref).abs().mean(), (gold - ref).abs().max(), (gold - ref).abs().mean(),
            (hip - gold).abs().max(), ref.abs().max()))
