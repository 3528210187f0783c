"""Diagnostic: the lp stem kernel replayed from a hipGraph beside another branch of kernels - does its output change?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tgsr_amd import lp, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
B = 4
x = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(dev)
w = (torch.randn(64, 3, 3, 3, generator=g) / 5).to(dev)
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
out = lp.new_image(B, 32, 32, 64, "bf16", dev)
lp.stem(x, w, sc, sh, out=out); torch.cuda.synchronize()
ref = out.clone()
# the other branch
other = os.environ.get("OTHER", "conv")
xi = lp.from_nchw(torch.randn(B, 32, 32, 32, generator=g).to(dev), "bf16", cpitch=32)
t1, t2 = lp.new_image(B, 32, 32, 32, "bf16", dev), lp.new_image(B, 32, 32, 32, "bf16", dev)
wp1 = lp.pack_conv3x3_weight((torch.randn(64, 32, 3, 3, generator=g) / 17).to(dev), "bf16")
wp2 = lp.pack_conv3x3_weight((torch.randn(32, 32, 3, 3, generator=g) / 17).to(dev), "bf16")
s1, h1 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
s2, h2 = torch.ones(32, device=dev), torch.zeros(32, device=dev)
big = torch.randn(1 << 22, device=dev)
def branch_b():
    if other == "conv":
        for _ in range(6):
            lp.conv3x3(xi, wp1, 32, 64, s1, h1, glu=True, out=t1)
            lp.conv3x3(t1, wp2, 32, 32, s2, h2, residual=xi, out=t2)
    elif other == "stem":
        o2 = lp.new_image(B, 32, 32, 32, "bf16", dev) if False else t1
        for _ in range(6):
            lp.stem(x, w[:64], sc, sh, out=t1)
    else:
        for _ in range(6):
            big.mul_(1.0001)
side = torch.cuda.Stream()
for _ in range(2):
    lp.stem(x, w, sc, sh, out=out); branch_b()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        branch_b()
    lp.stem(x, w, sc, sh, out=out)
    main.wait_stream(side)
bad = 0
for i in range(int(os.environ.get("N", "200"))):
    out.zero_()
    gr.replay(); torch.cuda.synchronize()
    d = float((out.float() - ref.float()).abs().max())
    if d:
        bad += 1
        if bad < 4:
            nz = ((out.float() - ref.float()).abs() > 0).nonzero()
            print("replay", i, "max diff", d, "n", len(nz), "first", nz[:3].tolist(), "last", nz[-1].tolist())
print("other=%s bad=%d" % (other, bad))
