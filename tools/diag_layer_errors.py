#!/usr/bin/env python3
"""Which KERNEL of NetG_highweight has the fat error tail?  Every stage of the net is run three ways on the SAME input (the fp64
oracle's activation at that point, rounded to fp32): the HIP module, the CPU fp32 oracle function, the fp64 oracle function -
so a stage's figure is its own rounding error, not what it inherited.  Then the chained errors at the worst pixel of the final
image.     python tools/diag_layer_errors.py [seed] [direct]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tgsr_oracle as O
from tgsr_amd.miscc.config import cfg, cfg_reset
from tgsr_amd.trainer import SRPipeline
from tgsr_amd import ops, util
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
w = np.load(os.path.join(G, "face_S8_weights.npz"))
def sd(pre, dt=torch.float32):
    return {k[len(pre):]: (torch.from_numpy(w[k]).to(dt) if w[k].dtype.kind == 'f' else torch.from_numpy(w[k])) for k in w.files if k.startswith(pre)}
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6100
if "direct" in sys.argv:
    util.WINOGRAD = False
if "f22" in sys.argv:
    ops.ROUTING.wino4 = False
cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256
cap, lens, LR, LRb = O.synthetic_batch(16, seed=seed)
H32, H64 = sd("GH."), sd("GH.", torch.float64)
r64 = O.sr_forward(sd("E.", torch.float64), sd("GL.", torch.float64), H64, cap, lens.tolist(), LR.double(), LRb.double())
p = SRPipeline(41, device="cuda", branch_num=4).load_state_dicts(sd("E."), sd("GL."), sd("GH."))
gh = p.netGH
stages = [("convin", gh.convin, lambda x, s: O.conv_bn_glu(x, s, "convin."))]
for i in range(6):
    stages.append(("residual.%d" % i, gh.residual[i], lambda x, s, i=i: O.res_block(x, s, "residual.%d." % i)))
stages += [("upscale2x", gh.upscale2x, lambda x, s: O.up_block(x, s, "upscale2x.")),
           ("residual24", gh.residual24, lambda x, s: O.residual_nosum(x, s, "residual24.")),
           ("upscale4x", gh.upscale4x, lambda x, s: O.up_block(x, s, "upscale4x.")),
           ("residual48", gh.residual48, lambda x, s: O.residual_nosum(x, s, "residual48.")),
           ("upscale8x", gh.upscale8x, lambda x, s: O.up_block(x, s, "upscale8x."))]
x64 = LR.double()
x32 = LR.clone()
feats, feats32 = {}, {}
print("stage            out std   | HIP own error max / mean | CPU fp32 own error max / mean |  HIP/CPU mean")
with torch.no_grad():
    for name, mod, fn in stages:
        xin = x64.float()
        y64 = fn(xin.double(), H64)
        yc = fn(xin, H32)
        yh = mod(xin.cuda()).cpu()
        eh, ec = (yh.double() - y64).abs(), (yc.double() - y64).abs()
        print("%-14s %9.3f   | %.2e / %.2e     | %.2e / %.2e          | %.2f" % (name, float(y64.std()), eh.max(), eh.mean(), ec.max(), ec.mean(), eh.mean() / ec.mean()), flush=True)
        x64 = fn(x64, H64)
        feats[name] = x64
        x32 = fn(x32, H32)
        feats32[name] = x32
    w5 = H64["conv_output.0.weight"]
    for k, (name, sr) in enumerate(zip(("upscale2x", "upscale4x", "upscale8x"), r64["fake"])):
        xin = feats[name].float()
        pre64 = F.conv2d(xin.double(), w5, None, 1, 2)
        y64 = torch.tanh(pre64) + 0.5 * sr
        yc = torch.tanh(F.conv2d(xin, w5.float(), None, 1, 2)) + 0.5 * sr.float()
        yh = gh._head(xin.cuda(), sr.float().cuda()).cpu()
        eh, ec = (yh.double() - y64).abs(), (yc.double() - y64).abs()
        print("head %d (5x5+tanh) pre-tanh absmax %.1f | %.2e / %.2e     | %.2e / %.2e          | %.2f" % (k, float(pre64.abs().max()), eh.max(), eh.mean(), ec.max(), ec.mean(), eh.mean() / ec.mean()), flush=True)
    # the chain as the product runs it
    r = p(cap.cuda(), lens.tolist(), LR.cuda(), LRb.cuda())
    e = (r["fine"][2].cpu().double() - r64["fine"][2]).abs()
    am = [int(v) for v in torch.nonzero(e == e.max())[0]]
    print("whole pipeline: fine2 max %.2e mean %.2e at %s" % (e.max(), e.mean(), am))
    b, c, yy, xx = am
    f8 = feats["upscale8x"]
    patch = f8[b, :, max(0, yy - 2):yy + 3, max(0, xx - 2):xx + 3]
    print("fp64 activations feeding that pixel's 5x5 head: absmax %.1f (layer std %.2f); pre-tanh value %.3f; dtanh %.3f" % (
        float(patch.abs().max()), float(f8.std()), float(F.conv2d(f8[b:b + 1], w5, None, 1, 2)[0, c, yy, xx]),
        1 - float(torch.tanh(F.conv2d(f8[b:b + 1], w5, None, 1, 2)[0, c, yy, xx])) ** 2))
    # hooks: the product's own intermediate activations against fp64 at that pixel's neighbourhood
    got = {}
    hs = [m.register_forward_hook(lambda m_, i_, o_, n=n: got.__setitem__(n, o_.detach().cpu())) for n, m, _ in stages]
    p(cap.cuda(), lens.tolist(), LR.cuda(), LRb.cuda())
    for h in hs:
        h.remove()
    for n, _, _ in stages:
        if n in got:
            eh = (got[n].double() - feats[n]).abs()
            s = got[n].shape[-1] // 32
            y0, x0 = yy * s // 8, xx * s // 8
            loc = eh[b, :, max(0, y0 - 4):y0 + 5, max(0, x0 - 4):x0 + 5]
            ec = (feats32[n].double() - feats[n]).abs()
            locc = ec[b, :, max(0, y0 - 4):y0 + 5, max(0, x0 - 4):x0 + 5]
            print("chained %-12s HIP err max %.2e mean %.2e, near the pixel max %.2e | CPU fp32 err max %.2e mean %.2e, near the pixel max %.2e | activation absmax there %.1f" % (
                n, eh.max(), eh.mean(), loc.max(), ec.max(), ec.mean(), locc.max(), float(feats[n][b, :, max(0, y0 - 4):y0 + 5, max(0, x0 - 4):x0 + 5].abs().max())))
    r32 = O.sr_forward(sd("E."), sd("GL."), H32, cap, lens.tolist(), LR, LRb)
    ec = (r32["fine"][2].double() - r64["fine"][2]).abs()
    print("CPU fp32 fine2: max %.2e mean %.2e; at the HIP path's worst pixel %.2e" % (ec.max(), ec.mean(), ec[b, c, yy, xx]))
