"""Diagnostic: hipGraph replay of the lp step against the eager step, tensor by tensor, several captures in one process."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import tgsr_oracle as O
from tgsr_amd.miscc.config import cfg, cfg_reset
from conftest import load_npz, split_sd
from tgsr_amd.trainer import SRPipeline
cfg_reset(); cfg.GAN.GF_DIM=32; cfg.TEXT.EMBEDDING_DIM=256; cfg.TREE.BRANCH_NUM=4
fw = load_npz("face_S8_weights.npz")
overlap = os.environ.get("OVERLAP", "1") == "1"
dtype = os.environ.get("DTYPE", "bf16")
B = int(os.environ.get("B", "4"))
cap, lens, LR, LRb = O.synthetic_batch(B)
args = (cap.cuda(), lens.tolist(), LR.cuda(), LRb.cuda())
def snap(o):
    d = {k: [t.clone() for t in o[k]] for k in ("fake","fine","att")}
    for k in ("words_emb", "sent_emb", "mu", "mask"):
        d[k] = [o[k].clone().float()]
    return d
bad = 0
for trial in range(int(os.environ.get("TRIALS", "6"))):
    pipe = SRPipeline(41, device="cuda", dtype=dtype, overlap=overlap).load_state_dicts(split_sd(fw,"E."), split_sd(fw,"GL."), split_sd(fw,"GH."))
    a = pipe(*args)
    b = pipe(*args)
    torch.cuda.synchronize()
    a = snap(a)
    pipe.capture(*args)
    for rep in range(2):
        g = snap(pipe.replay()); torch.cuda.synchronize()
        diff = {k: [round(float((x-y).abs().max()), 5) for x, y in zip(g[k], a[k])] for k in a}
        if any(v for vs in diff.values() for v in vs):
            bad += 1
            print("trial", trial, "replay", rep, diff, flush=True)
print("mismatching replays:", bad)
