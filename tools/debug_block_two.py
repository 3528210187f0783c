import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tgsr_oracle as O
from tgsr_amd import util
def rel(a, b): return float((a.detach().cpu().double() - b.detach().double()).abs().max()) / (float(b.detach().abs().max()) + 1e-30)
g = torch.Generator().manual_seed(0)
for (B, Cin, Cout, H) in [(4, 32, 64, 32), (4, 16, 32, 64), (4, 64, 128, 16)]:
    torch.manual_seed(1)
    blk = util.downBlock(Cin, Cout)
    sd = {k: v.detach().double().clone() if v.is_floating_point() else v.clone() for k, v in blk.state_dict().items()}
    blk.cuda().train()
    xs = [torch.randn(B, Cin, H, H, generator=g) for _ in range(2)]
    dys = [torch.randn(B, Cout, H // 2, H // 2, generator=g) for _ in range(2)]
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    xr = [x.double().requires_grad_(True) for x in xs]
    refs = [O.down_block(x, sdr, "", True, {}) for x in xr]
    sum((r * dy.double()).sum() for r, dy in zip(refs, dys)).backward()
    xd = [x.cuda().requires_grad_(True) for x in xs]
    outs = [blk(x) for x in xd]
    sum((o * dy.cuda()).sum() for o, dy in zip(outs, dys)).backward()
    print((B, Cin, Cout, H), "dx0 %.2e dx1 %.2e dw %.2e dgamma %.2e dbeta %.2e" % (rel(xd[0].grad, xr[0].grad), rel(xd[1].grad, xr[1].grad), rel(blk[0].weight.grad, sdr["0.weight"].grad), rel(blk[1].weight.grad, sdr["1.weight"].grad), rel(blk[1].bias.grad, sdr["1.bias"].grad)))
