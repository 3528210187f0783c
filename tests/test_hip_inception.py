"""GPU: CNN_ENCODER's frozen trunk on the library's own kernels (csrc/tgsr_igemm.hip, tgsr_amd/inception.py) - every layer shape of
Inception-v3 through tgsr::gconv (forward and data gradient) against torch's convolution, the pooling / resize kernels against
torch's, and the whole walk of util.py:308-362 - features, pooled code, gradient back to the image - against the SAME modules run by
torch in float64 on the CPU.  The trunk's arithmetic is third-party (torchvision; weights absent here): this pins the kernels to the
modules' arithmetic, not to the reference."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from tgsr_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()


def rel(a, b):
    b = b.double()
    return float((a.detach().cpu().double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)


GEOM = [
    # B, Cin, H, W, Cout, kh, kw, stride, ph, pw       (Inception-v3's layer kinds, plus ragged / tiny ones)
    (2, 3, 39, 39, 32, 3, 3, 2, 0, 0),         # Conv2d_1a: image layer, stride 2, K = 27 (not a multiple of 16), wide tile
    (2, 32, 19, 19, 32, 3, 3, 1, 0, 0),        # 2a: no padding
    (2, 32, 17, 17, 64, 3, 3, 1, 1, 1),        # 2b
    (3, 64, 9, 9, 80, 1, 1, 1, 0, 0),          # 3b: 1x1, M = 80
    (2, 48, 12, 12, 64, 5, 5, 1, 2, 2),        # branch5x5_2
    (2, 128, 17, 17, 128, 1, 7, 1, 0, 3),      # 1x7
    (2, 128, 17, 17, 192, 7, 1, 1, 3, 0),      # 7x1, M = 192 (two m tiles)
    (2, 192, 17, 17, 320, 3, 3, 2, 0, 0),      # branch3x3_2 of Mixed_7a: stride 2 on 17 x 17
    (4, 384, 8, 8, 384, 1, 3, 1, 0, 1),        # 1x3 on the 8 x 8 stage: split over K
    (4, 448, 8, 8, 384, 3, 3, 1, 1, 1),        # K = 4032: split over K
    (4, 1280, 8, 8, 320, 1, 1, 1, 0, 0),       # 1x1 with 1280 input channels
    (1, 5, 7, 11, 7, 3, 1, 1, 1, 0),           # tiny and ragged
    (2, 288, 35, 35, 384, 3, 3, 2, 0, 0),      # Mixed_6a branch3x3 at its real size
    (2, 96, 35, 35, 96, 3, 3, 2, 0, 0),        # Mixed_6a branch3x3dbl_3
    (1, 3, 299, 299, 32, 3, 3, 2, 0, 0),       # Conv2d_1a at its real size
    (2, 192, 35, 35, 48, 1, 1, 1, 0, 0),       # Mixed_5b branch5x5_1
]


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("B,Cin,H,W,Cout,kh,kw,st,ph,pw", GEOM)
def test_gconv_forward_and_data_gradient(B, Cin, H, W, Cout, kh, kw, st, ph, pw, split):
    """split: on the bf16 matrix pipe with exact three-piece fp32 operands where the shape qualifies (the default; the others fall
    back inside the library) | the fp32 MFMA kernel everywhere: same tolerances."""
    from tgsr_amd import ops
    was = ops.gconv_set_form(split)
    try:
        _gconv_case(B, Cin, H, W, Cout, kh, kw, st, ph, pw)
    finally:
        ops.gconv_set_form(was)


def _gconv_case(B, Cin, H, W, Cout, kh, kw, st, ph, pw):
    from tgsr_amd import custom_ops as C
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(Cin + 7 * Cout + kh)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, kh, kw, generator=g) / (Cin * kh * kw) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    xr = x.double().requires_grad_(True)
    ref = F.relu(F.conv2d(xr, w.double() * scale.double().view(-1, 1, 1, 1), shift.double(), st, (ph, pw)))
    OH, OW = ref.shape[2], ref.shape[3]
    dy = torch.randn(B, Cout, OH, OW, generator=g)
    ref.backward(dy.double())
    xd, wd = x.to(DEV), w.to(DEV)
    wf, wt = C.gconv_pack(wd, scale.to(DEV), False), C.gconv_pack(wd, scale.to(DEV), True)
    # forward into channels [5, 5 + Cout) of a wider buffer
    out = torch.full((B, Cout + 9, OH, OW), 7.0, device=DEV)
    need = ops.gconv_ws_elems(B, Cout, OH, OW, Cin * kh * kw)
    ws = torch.empty(max(need, 1), device=DEV)
    C.gconv(False, wf, xd, 0, Cin, out, 5, kh, kw, st, ph, pw, shift.to(DEV), True, False, ws, None)
    assert rel(out[:, 5:5 + Cout], ref.detach()) < 2e-5
    assert bool((out[:, :5] == 7).all()) and bool((out[:, 5 + Cout:] == 7).all())
    # data gradient of the masked dy, read from a channel slice, accumulated onto an existing gradient
    gbuf = torch.zeros(B, Cout + 9, OH, OW, device=DEV)
    gbuf[:, 5:5 + Cout] = dy.to(DEV)
    C.relu_mask_(gbuf, out, 5, Cout)
    base = torch.randn(B, Cin, H, W, generator=g)
    dx = base.to(DEV).clone()
    need = ops.gconv_ws_elems(B, Cin, H, W, Cout * kh * kw)
    ws = torch.empty(max(need, 1), device=DEV)
    C.gconv(True, wt, gbuf, 5, Cout, dx, 0, kh, kw, st, ph, pw, None, False, True, ws, None)
    assert rel(dx - base.to(DEV), xr.grad) < 5e-5
    dx2 = torch.empty(B, Cin, H, W, device=DEV)
    C.gconv(True, wt, gbuf, 5, Cout, dx2, 0, kh, kw, st, ph, pw, None, False, False, ws, None)
    assert rel(dx2, xr.grad) < 5e-5
    if Cin > 4:         # (the image layer's own kernel takes no mask: an image is not a ReLU output)
        # the ReLU factor of the tensor whose gradient this is, applied to the contribution in the epilogue (before the +=)
        m = torch.randn(B, Cin, H, W, generator=g).to(DEV)
        dx3 = base.to(DEV).clone()
        C.gconv(True, wt, gbuf, 5, Cout, dx3, 0, kh, kw, st, ph, pw, None, False, True, ws, m)
        assert torch.equal(dx3, base.to(DEV) + torch.where(m > 0, dx2, torch.zeros_like(dx2)))


def test_pool_mask_and_resize_kernels_against_torch():
    from tgsr_amd import custom_ops as C
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 6, 17, 23, generator=g)
    x[0, 0, :6, :6] = 0.0                                       # ties: the first maximum takes the gradient (torch's rule)
    xr = x.double().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    out = torch.zeros(3, 10, ref.shape[2], ref.shape[3], device=DEV)
    C.maxpool3s2(x.to(DEV), out, 2)
    assert torch.equal(out[:, 2:8].cpu(), ref.detach().float()) and bool((out[:, :2] == 0).all())
    gb = torch.zeros_like(out)
    gb[:, 2:8] = dy.to(DEV)
    dx = torch.empty(3, 6, 17, 23, device=DEV)
    C.maxpool3s2_bwd(x.to(DEV), gb, 2, dx, False, None)
    assert rel(dx, xr.grad) < 1e-6
    C.maxpool3s2_bwd(x.to(DEV), gb, 2, dx, True, None)
    assert rel(dx, 2 * xr.grad) < 1e-6
    msk = torch.randn(3, 6, 17, 23, generator=g).to(DEV)
    C.maxpool3s2_bwd(x.to(DEV), gb, 2, dx, False, msk)
    assert rel(dx, xr.grad * (msk.cpu() > 0)) < 1e-6
    # 3x3 average pool (count_include_pad) and its adjoint
    xr = x.double().requires_grad_(True)
    ref = F.avg_pool2d(xr, 3, 1, 1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    o = torch.empty(3, 6, 17, 23, device=DEV)
    C.avgpool3(x.to(DEV), o, False, None)
    assert rel(o, ref.detach()) < 1e-6
    C.avgpool3(dy.to(DEV), o, False, None)
    assert rel(o, xr.grad) < 1e-6
    C.avgpool3(dy.to(DEV), o, True, msk)
    assert rel(o, xr.grad + xr.grad * (msk.cpu() > 0)) < 1e-6
    # global mean
    xr = x.double().requires_grad_(True)
    m = xr.mean((2, 3))
    dm = torch.randn(m.shape, generator=g)
    m.backward(dm.double())
    assert rel(C.plane_mean(x.to(DEV)), m.detach()) < 1e-6
    assert rel(C.plane_mean_bwd(dm.to(DEV), 17, 23), xr.grad) < 1e-6
    # bilinear resize to 299 x 299 (util.py:310) and to a smaller size, forward and backward
    for (H, W, OH, OW) in ((64, 64, 299, 299), (40, 52, 23, 31), (256, 256, 299, 299)):
        im = torch.randn(2, 3, H, W, generator=g)
        ir = im.double().requires_grad_(True)
        ref = F.interpolate(ir, size=(OH, OW), mode="bilinear", align_corners=False)
        dy = torch.randn(ref.shape, generator=g)
        ref.backward(dy.double())
        # (the source coordinates are computed in fp32, as torch's own HIP kernel computes them: ~1e-5 against the float64 run)
        assert rel(C.bilinear(im.to(DEV), OH, OW), ref.detach()) < 3e-5, (H, W, OH, OW)
        assert rel(C.bilinear_bwd(dy.to(DEV), H, W), ir.grad) < 3e-5, (H, W, OH, OW)


def test_whole_trunk_forward_and_image_gradient_against_the_modules_in_float64():
    """CNN_ENCODER(inception=<torchvision's layout>) in eval mode takes the HIP walk; the same module, in float64 on the CPU, walks
    the torch modules: region features, pooled code, the heads' outputs, and d(loss)/d(image) for a loss that reads both outputs."""
    from inception_v3_arch import InceptionV3Arch
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.util import CNN_ENCODER
    cfg_reset()
    cfg.TRAIN.FLAG = True
    try:
        enc = CNN_ENCODER(64, inception=InceptionV3Arch(seed=2)).eval()
        for p in enc.frozen_parameters():
            p.requires_grad = False
        import copy
        ref = copy.deepcopy(enc).double()
        enc.to(DEV)
        B = 2
        g = torch.Generator().manual_seed(5)
        img = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
        wf, wp = torch.randn(B, 768, 17, 17, generator=g), torch.randn(B, 2048, generator=g)
        xr = img.double().requires_grad_(True)
        os.environ["TGSR_TRUNK"] = "torch"
        try:
            fr, pr = ref.run_trunk(xr)
        finally:
            os.environ.pop("TGSR_TRUNK")
        ((fr * wf.double()).sum() + (pr * wp.double()).sum()).backward()
        xd = img.to(DEV).requires_grad_(True)
        assert enc._hip_trunk_ok(xd)
        fd, pd = enc.run_trunk(xd)
        assert enc._hip_trunk is not None and tuple(fd.shape) == (B, 768, 17, 17) and tuple(pd.shape) == (B, 2048)
        ((fd * wf.to(DEV)).sum() + (pd * wp.to(DEV)).sum()).backward()
        assert rel(fd, fr.detach()) < 2e-4, rel(fd, fr.detach())
        assert rel(pd, pr.detach()) < 2e-4, rel(pd, pr.detach())
        # The image gradient runs through 94 ReLU masks and 4 max pools: an activation that is 1e-6 from zero in one arithmetic and
        # on the other side in the other flips a mask (likewise an arg-max), and the gradient is discontinuous there - so the
        # comparison with the float64 run is made in the L2 norm, which a handful of flipped elements out of millions do not move
        # (tools/debug_trunk.py: stage by stage the gradients agree with torch's own fp32 HIP path to 1e-6 where no flip occurred)
        gd, gr = xd.grad.cpu().double(), xr.grad
        l2 = float((gd - gr).norm() / gr.norm())
        cos = float((gd * gr).sum() / (gd.norm() * gr.norm()))
        assert l2 < 5e-2 and cos > 0.999, (l2, cos)
        # the whole encoder (heads on the HIP GEMMs) and a second forward / backward on the same runner
        xd2 = img.flip(0).to(DEV).requires_grad_(True)
        r2, c2 = enc(xd2)
        (r2.square().mean() + c2.square().mean()).backward()
        assert torch.isfinite(xd2.grad).all() and float(xd2.grad.abs().max()) > 0
        # no gradient wanted: nothing is kept
        with torch.no_grad():
            f3, p3 = enc.run_trunk(img.to(DEV))
        assert torch.equal(f3, fd) and torch.equal(p3, pd)
        # training mode (pretrain_DAMSM.py:49-50: batch-statistics BatchNorm in the trunk) stays on the torch modules
        enc.train()
        assert not enc._hip_trunk_ok(xd)
    finally:
        cfg_reset()


def test_branch_streams_equal_the_single_stream_walk_bit_for_bit():
    """The branches of a Mixed_* block on streams of their own (the default) run the same kernels with the same order of every
    accumulation as the walk on one stream: features, pooled code and d(loss)/d(image) are identical bits - eagerly, twice in a row
    (the second walk reuses the allocator's blocks of the first), and replayed from a hipGraph that captured the forked walk."""
    from inception_v3_arch import InceptionV3Arch
    from tgsr_amd import inception
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.util import CNN_ENCODER
    cfg_reset()
    cfg.TRAIN.FLAG = True
    try:
        enc = CNN_ENCODER(64, inception=InceptionV3Arch(seed=4)).eval().to(DEV)
        for p in enc.frozen_parameters():
            p.requires_grad = False
        B = 8
        g = torch.Generator().manual_seed(9)
        img = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(DEV)
        wf, wp = torch.randn(B, 768, 17, 17, generator=g).to(DEV), torch.randn(B, 2048, generator=g).to(DEV)

        def walk(nstreams, x):
            assert enc._hip_trunk_ok(x)
            enc.run_trunk(x.detach())                       # builds the runner
            enc._hip_trunk.nstreams = nstreams
            f, p = enc.run_trunk(x)
            ((f * wf).sum() + (p * wp).sum()).backward()
            return f.detach().clone(), p.detach().clone(), x.grad.detach().clone()

        one = walk(1, img.clone().requires_grad_(True))
        assert len(enc._hip_trunk._streams) == 1
        for _ in range(2):
            many = walk(4, img.clone().requires_grad_(True))
            assert len(enc._hip_trunk._streams) == 4
            for a, b in zip(one, many):
                assert torch.equal(a, b)
        # captured: the forks and joins become the graph's branches
        xs = img.clone().requires_grad_(True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            f, p = enc.run_trunk(xs)
            gx, = torch.autograd.grad((f * wf).sum() + (p * wp).sum(), xs)
        with torch.no_grad():
            xs.copy_(img.flip(0))
        gr.replay()
        torch.cuda.synchronize()
        ref = walk(1, img.flip(0).clone().requires_grad_(True))
        assert torch.equal(f, ref[0]) and torch.equal(p, ref[1]) and torch.equal(gx, ref[2])
    finally:
        cfg_reset()


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,pad", [(2, 288, 35, 35, 384, 3, 0), (2, 96, 35, 35, 96, 3, 0), (3, 192, 17, 17, 320, 3, 0),
                                                  (2, 16, 12, 14, 32, 4, 0), (1, 32, 9, 11, 16, 5, 0), (2, 8, 6, 6, 16, 2, 0),
                                                  (2, 16, 12, 14, 32, 4, 1), (2, 16, 12, 13, 32, 3, 0)])
def test_stride2_data_gradient_by_parity_classes_matches_the_direct_form(B, Cin, H, W, Cout, k, pad):
    """inception._Layer's class form of a stride-2 layer's data gradient (four stride-1 tgsr::gconv launches over the taps of each
    input-pixel parity class, woven together by tgsr::interleave2x2_ with the ReLU mask and the accumulation) against torch's
    convolution backward in float64, and against the direct stride-2 launch it replaces."""
    from tgsr_amd import custom_ops as C
    from tgsr_amd import inception, ops
    assert inception.CLASS_DGRAD
    g = torch.Generator().manual_seed(Cin + Cout + k)

    class M:                       # a conv + bn pair as _Layer reads it
        pass
    m = M()
    m.conv = torch.nn.Conv2d(Cin, Cout, k, 2, pad, bias=False).to(DEV)
    m.bn = torch.nn.BatchNorm2d(Cout).to(DEV).eval()
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
        m.bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        m.bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    L = inception._Layer(m)
    assert L.class_dgrad_ok(H, W) == (pad == 0 and (H - k) % 2 == 0 and (W - k) % 2 == 0)
    if not L.class_dgrad_ok(H, W):
        return                     # (padded layers and sizes with an unread trailing row / column keep the direct form)
    OH, OW = L.out_hw(H, W)
    dy = torch.randn(B, Cout, OH, OW, generator=g)
    scale = (m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)).double().cpu()
    x64 = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x64, m.conv.weight.double().cpu() * scale.view(-1, 1, 1, 1), None, 2, pad).backward(dy.double())
    gd = dy.to(DEV)
    base = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    mask = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    for acc, mk in ((False, None), (True, mask)):
        dx = base.clone()
        parts = [torch.empty(B, Cin, (H - py + 1) // 2, (W - px + 1) // 2, device=DEV) for py, px, *_ in L.cls]
        ws = torch.empty(max(max(ops.gconv_ws_elems(B, Cin, t.shape[2], t.shape[3], Cout * c[3] * c[4]) for t, c in zip(parts, L.cls)), 1),
                         device=DEV)
        for t, (_py, _px, A, khc, kwc, phc, pwc) in zip(parts, L.cls):
            C.gconv(True, A, gd, 0, Cout, t, 0, khc, kwc, 1, phc, pwc, None, False, False, ws, None)
        C.interleave2x2_(parts[0], parts[1], parts[2], parts[3], dx, acc, mk)
        want = x64.grad.clone()
        if mk is not None:
            want = torch.where(mk.cpu() > 0, want, torch.zeros_like(want))
        if acc:
            want = want + base.double().cpu()
        assert rel(dx, want) < 5e-5, (acc, rel(dx, want))
    # the direct stride-2 launch agrees too
    direct = torch.empty(B, Cin, H, W, device=DEV)
    wsd = torch.empty(max(ops.gconv_ws_elems(B, Cin, H, W, Cout * k * k), 1), device=DEV)
    C.gconv(True, L.wd, gd, 0, Cout, direct, 0, k, k, 2, pad, pad, None, False, False, wsd, None)
    assert rel(direct, x64.grad) < 5e-5
