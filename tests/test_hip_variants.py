"""GPU: the constructor forms of the boundary that the shipped caller does not select (SURVEY.md 8b lists them as arguments of
`NetG_highweight(weightmap, low, useAct)`) and the discriminator blocks under `.eval()` - against vectors captured from the
reference's own modules (tests/golden/gh_variants.npz, make_golden.py gen_gh_variants), through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import load_npz, split_sd

pytestmark = pytest.mark.gpu
DEV = "cuda"
ATOL = RTOL = 1e-4          # these cases' bound: tighter than the path's stated conftest.FP32_TOL (2e-4)


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


@pytest.fixture(scope="module")
def g():
    from tgsr_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return load_npz("gh_variants.npz")


@pytest.fixture()
def cfg32():
    from tgsr_amd.miscc.config import cfg, cfg_reset
    cfg_reset()
    cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 64
    yield cfg
    cfg_reset()


def test_netg_highweight_weightmap_eval_and_gradients(g, cfg32):
    """model.py:235-245, 276-297: a_k [H, W] trainable maps.  state_dict keys a1..a3 load strictly; eval images, then train-mode
    images and the gradients of the maps, of SRb, of conv_output and of the first convolution (through tgsr::axpy_map_bwd and the
    whole HIP backward) against the reference's autograd."""
    from tgsr_amd import model
    net = model.NetG_highweight(weightmap=True, low="lr")
    sd = split_sd(g, "wm.GH.")
    assert set(sd) == set(net.state_dict()) and {"a1", "a2", "a3"} <= set(sd)
    net.load_state_dict(sd, strict=True)
    net.to(DEV).eval()
    LR, SRb = T(g["wm.LR"]), [T(g["wm.SRb%d" % k]) for k in range(3)]
    with torch.no_grad():
        ims, a, one = net(LR, SRb, LR)
    for k in range(3):
        close(ims[k], g["wm.eval.fine%d" % k])
    assert a is net.a3 and float(one) == 1.0
    # the two-phase form SRPipeline uses (conv_output on the high-frequency branch's stream, the maps applied when SRb exists)
    with torch.no_grad():
        two = net.finish_heads(net.tanh_heads(net.trunk(LR, LR)), SRb)
    for k in range(3):
        assert torch.equal(two[k], ims[k])
    net.train()
    sr = [s.clone().requires_grad_(True) for s in SRb]
    ims, _a, _one = net(LR, sr, LR)
    sum((i * T(g["wm.dy%d" % k])).sum() for k, i in enumerate(ims)).backward()
    close(ims[0], g["wm.train.fine0"], atol=2e-4)
    for k in (1, 2, 3):
        close(getattr(net, "a%d" % k).grad, g["wm.train.da%d" % k])
    close(sr[0].grad, g["wm.train.dSRb0"])
    ref = g["wm.train.dconv_output"]
    close(net.conv_output[0].weight.grad, ref, atol=2e-4 * float(np.abs(ref).max()), rtol=1e-3)
    ref = g["wm.train.dconvin"]
    close(net.convin[0].weight.grad, ref, atol=2e-3 * float(np.abs(ref).max()), rtol=1e-2)
    # a map of the wrong size is refused, not broadcast
    with pytest.raises(ValueError):
        net(LR[:, :, :16, :16], [s[:, :, :s.shape[2] // 2, :s.shape[3] // 2] for s in SRb], LR[:, :, :16, :16])


def test_axpy_map_op_forward_backward():
    from tgsr_amd import custom_ops as C
    gen = torch.Generator().manual_seed(3)
    t, s = torch.randn(3, 3, 20, 12, generator=gen), torch.randn(3, 3, 20, 12, generator=gen)
    a, dy = torch.randn(20, 12, generator=gen), torch.randn(3, 3, 20, 12, generator=gen)
    tr, sr, ar = (x.clone().requires_grad_(True) for x in (t, s, a))
    (tr + ar * sr).backward(dy)
    td, sd, ad = (x.to(DEV).requires_grad_(True) for x in (t, s, a))
    out = C.axpy_map(td, sd, ad)
    out.backward(dy.to(DEV))
    close(out, t + a * s, atol=1e-6)
    close(td.grad, tr.grad, atol=0)
    close(sd.grad, sr.grad, atol=1e-6)
    close(ad.grad, ar.grad, atol=1e-5)
    with pytest.raises(Exception):
        C.axpy_map(td, sd, ad[:, :8])


def test_netg_highweight_without_tanh(g, cfg32):
    """useAct=False (model.py:223-226): conv_output is the bare conv5x5; ims_k = conv5x5(out_k) + 0.5 SRb_k."""
    from tgsr_amd import model
    net = model.NetG_highweight(weightmap=False, low="lr", useAct=False)
    assert len(net.conv_output) == 1
    # (the fixture was captured on the CPU, where the reference's `nn.Parameter(...).cuda()` keeps `a` a parameter; on a GPU - and in
    # the shipped netGH_epoch_7.pth - it is a plain tensor that is neither saved nor trained: SURVEY Q3, tgsr_amd/model.py)
    net.load_state_dict({k: v for k, v in split_sd(g, "na.GH.").items() if k != "a"}, strict=True)
    net.to(DEV).eval()
    LR, SRb = T(g["na.LR"]), [T(g["na.SRb%d" % k]) for k in range(3)]
    with torch.no_grad():
        ims, a, one = net(LR, SRb, LR)
        two = net.finish_heads(net.tanh_heads(net.trunk(LR, LR)), SRb)
    for k in range(3):
        close(ims[k], g["na.fine%d" % k])
        assert torch.equal(two[k], ims[k])
    # differentiable: d(ims)/d(SRb) = a
    sr = SRb[0].clone().requires_grad_(True)
    net.train()
    ims, _a, _one = net(LR, [sr] + SRb[1:], LR)
    ims[0].sum().backward()
    assert torch.allclose(sr.grad, torch.full_like(sr, 0.5))


def test_downblock_in_eval_mode(g, cfg32):
    """util.py:92-98 under .eval(): BatchNorm2d normalises with its running statistics (tgsr::affine_act); output and the input /
    weight gradients against the reference module's; a whole discriminator runs in eval mode."""
    from tgsr_amd import model, util
    blk = util.downBlock(16, 32)
    blk.load_state_dict({k: v for k, v in split_sd(g, "down.").items() if k[0] in "01"}, strict=True)
    blk.to(DEV).eval()
    x = T(g["down.x"]).requires_grad_(True)
    y = blk(x)
    (y * T(g["down.dy"])).sum().backward()
    close(y, g["down.out"])
    close(x.grad, g["down.dx"])
    ref = g["down.dw"]
    close(blk[0].weight.grad, ref, atol=2e-4 * float(np.abs(ref).max()), rtol=1e-3)
    rm = blk[1].running_mean.clone()
    with torch.no_grad():
        blk(x.detach())
    assert torch.equal(rm, blk[1].running_mean)            # eval mode leaves the running statistics alone
    from tgsr_amd.miscc.config import cfg
    cfg.GAN.DF_DIM = 8
    from oracle import tgsr_oracle as O
    torch.manual_seed(2)
    d = model.D_NET64()
    sd = {k: v.detach().clone() for k, v in d.state_dict().items()}
    d.to(DEV).eval()
    img = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    cond = torch.randn(3, 64, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        feat = d(img.to(DEV))
        logit = d.COND_DNET(feat, cond.to(DEV))
    ref_feat = O.d_features(sd, img, training=False)
    close(feat, ref_feat, atol=2e-4)
    close(logit, O.d_logits(sd, "COND_DNET.", ref_feat, cond, training=False), atol=2e-4)


def test_models16_weightmap_equals_the_scalar_form_for_constant_maps(cfg32):
    """models16.py:119-125, 149-178: four maps a1..a4 (32 ... 256 pixels, 16 x 16 inputs).  With every map = 0.5 the heads must be
    bit-identical to the scalar form with a = 0.5 (the same fma); state_dict carries a1..a4 and no `a`."""
    from tgsr_amd import models16
    torch.manual_seed(4)
    ref = models16.NetG_highweight(weightmap=False, low="lr")
    net = models16.NetG_highweight(weightmap=True, low="lr")
    keys = set(net.state_dict())
    assert {"a1", "a2", "a3", "a4"} <= keys and "a" not in keys
    assert [tuple(m.shape) for m in net.maps()] == [(32, 32), (64, 64), (128, 128), (256, 256)]
    net.load_state_dict({k: v for k, v in ref.state_dict().items() if k != "a"}, strict=False)
    with torch.no_grad():
        for m in net.maps():
            m.fill_(0.5)
    ref.to(DEV).eval(); net.to(DEV).eval()
    gen = torch.Generator().manual_seed(8)
    LR = (torch.rand(2, 3, 16, 16, generator=gen) * 2 - 1).to(DEV)
    SRb = [(torch.rand(2, 3, s, s, generator=gen) * 2 - 1).to(DEV) for s in (32, 64, 128, 256)]
    with torch.no_grad():
        a, aa, _ = ref(LR, SRb, LR)
        b, bb, _ = net(LR, SRb, LR)
    assert bb is net.a4
    for x, y in zip(a, b):
        assert torch.equal(x, y)
