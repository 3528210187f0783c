import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd import ops, util
from oracle import tgsr_oracle as O
def rel(a, b): return float((a.cpu().double() - b.double()).abs().max()) / (float(b.abs().max()) + 1e-30)
g = torch.Generator().manual_seed(0)
for (B, Cin, Cout, H) in [(4, 64, 128, 16), (4, 32, 64, 32), (4, 64, 128, 8), (2, 64, 128, 16), (4, 16, 32, 64)]:
    x = torch.randn(B, Cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 4, 4, generator=g, dtype=torch.float64) / (4 * Cin ** 0.5)).requires_grad_(True)
    dy = torch.randn(B, Cout, H // 2, H // 2, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, None, 2, 1); y.backward(dy)
    dx = ops.conv4x4s2_dgrad(dy.float().cuda(), w.detach().float().cuda(), H, H)
    dw = ops.conv4x4s2_wgrad(dy.float().cuda(), x.detach().float().cuda())
    out = ops.conv4x4s2(x.detach().float().cuda(), w.detach().float().cuda())
    print((B, Cin, Cout, H), "fwd %.2e dgrad %.2e wgrad %.2e  sum(dx) %.6f vs %.6f" % (rel(out, y.detach()), rel(dx, x.grad), rel(dw, w.grad), float(dx.double().sum()), float(x.grad.sum())))
# block level, tight
for (B, Cin, Cout, H) in [(4, 32, 64, 32), (4, 64, 128, 16), (4, 16, 32, 64)]:
    torch.manual_seed(1)
    blk = util.downBlock(Cin, Cout)
    sd = {k: v.detach().double().clone() if v.is_floating_point() else v.clone() for k, v in blk.state_dict().items()}
    blk.cuda().train()
    x = torch.randn(B, Cin, H, H, generator=g); dy = torch.randn(B, Cout, H // 2, H // 2, generator=g)
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    xr = x.double().requires_grad_(True)
    ref = O.down_block(xr, sdr, "", True, {}); ref.backward(dy.double())
    xd = x.cuda().requires_grad_(True); out = blk(xd); out.backward(dy.cuda())
    print("block", (B, Cin, Cout, H), "out %.2e dx %.2e dw %.2e dgamma %.2e dbeta %.2e" % (rel(out, ref.detach()), rel(xd.grad, xr.grad), rel(blk[0].weight.grad, sdr["0.weight"].grad), rel(blk[1].weight.grad, sdr["1.weight"].grad), rel(blk[1].bias.grad, sdr["1.bias"].grad)))
