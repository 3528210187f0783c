#!/usr/bin/env python3
"""Where does the HIP trunk's backward leave the torch modules'?  Runs both walks on the GPU (fp32) on the same weights / image and
compares the activation and the gradient at every stage boundary.   python tools/debug_trunk.py   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from inception_v3_arch import InceptionV3Arch
from tgsr_amd.miscc.config import cfg
from tgsr_amd.util import CNN_ENCODER
from tgsr_amd.inception import InceptionTrunk

cfg.TRAIN.FLAG = True
enc = CNN_ENCODER(64, inception=InceptionV3Arch(seed=2)).eval().cuda()
for p in enc.parameters():
    p.requires_grad = False
B = int(os.environ.get("B", "2"))
g = torch.Generator().manual_seed(5)
img = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).cuda()
wf, wp = torch.randn(B, 768, 17, 17, generator=g).cuda(), torch.randn(B, 2048, generator=g).cuda()
# torch walk with every boundary kept - REF=cpu64 (default): the same modules in float64 on the CPU; REF=gpu: torch's own HIP path
import copy
if os.environ.get("REF", "cpu64") == "cpu64":
    renc = copy.deepcopy(enc).double().cpu()
    cv = lambda t: t.double().cpu()
else:
    renc, cv = enc, (lambda t: t)
x = cv(img).clone().requires_grad_(True)
wf_r, wp_r = cv(wf), cv(wp)
keep = {}
def k(name, t):
    t.retain_grad(); keep[name] = t; return t
t = k("resize", F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False))
for n in ("Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3"):
    t = k(n, getattr(renc, n)(t))
t = k("pool1", F.max_pool2d(t, 3, 2))
for n in ("Conv2d_3b_1x1", "Conv2d_4a_3x3"):
    t = k(n, getattr(renc, n)(t))
t = k("pool2", F.max_pool2d(t, 3, 2))
for n in InceptionTrunk.MIXED:
    t = k(n, getattr(renc, n)(t))
feat = keep["Mixed_6e"]
pooled = F.avg_pool2d(t, 8).view(B, -1)
((feat * wf_r).sum() + (pooled * wp_r).sum()).backward()
# HIP walk
run = InceptionTrunk(enc)
run.keep_grads = True
f2, p2 = run.forward(img)
marks = dict(run.marks)
acts = {n: run.tensors[i].clone() for n, i in marks.items()}
dx = run.backward(wf, wp)
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
print("%-16s %-22s %10s %10s   worst gradient row (of H)" % ("stage", "shape", "act rel", "grad rel"))
for n in marks:
    gh, gt = run.snaps[marks[n]], keep[n].grad
    if n != "resize":                # the HIP walk keeps a tensor's gradient with its ReLU factor (y > 0) already applied
        gt = gt * (keep[n].detach() > 0)
    gh, gt = gh.double().cpu(), gt.double().cpu()
    d = (gh - gt).abs()
    rows = d.amax(dim=(0, 1, 3))
    print("%-16s %-22s %10.2e %10.2e   row %d of %d (%.2e); mean row err %.2e" % (n, tuple(gt.shape), rel(acts[n], keep[n].detach()), rel(gh, gt),
          int(rows.argmax()), gt.shape[2], float(rows.max()), float(rows.mean())))
print("image gradient", rel(dx, x.grad))
