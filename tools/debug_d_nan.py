import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_empty, _empty_like = torch.empty, torch.empty_like
def nan_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    return t
def nan_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    return t
torch.empty, torch.empty_like = nan_empty, nan_empty_like
from tgsr_amd.miscc.config import cfg
cfg.GAN.DF_DIM = 8; cfg.TEXT.EMBEDDING_DIM = 32
from tgsr_amd import model
torch.manual_seed(11)
d = model.D_NET256().cuda().train()
g = torch.Generator().manual_seed(5)
B = 4
x1 = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).cuda()
x2 = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).cuda()
R1, R2 = torch.randn(B, 64, 4, 4, generator=g).cuda(), torch.randn(B, 64, 4, 4, generator=g).cuda()
l = (d(x1) * R1).sum() + (d(x2) * R2).sum()
l.backward()
for k, p in d.named_parameters():
    print("%-36s nan=%d" % (k, int(torch.isnan(p.grad).sum())))
