"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed goldens.

The path's ONE stated fp32 tolerance is conftest.FP32_TOL: atol = rtol = 2e-4 on un-clamped outputs, every value (round 6; SURVEY 8c
proposed 1e-4 from a single batch-2 input whose fp32-vs-fp64 deviation was 1.7e-5 - over 15 full-size cases the reference's own CPU
fp32 path is up to 8.4e-5 from fp64 and two correct fp32 evaluations differ by up to 1.35e-4 at isolated outlier pixels:
tests/test_hip_parity_margin.py).  The cases of THIS file are pinned seeds and are held to the tighter 1e-4
they measure within (ATOL below); op-level checks use tighter bounds still.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import split_sd
from oracle import tgsr_oracle as O

pytestmark = pytest.mark.gpu

from conftest import FP32_TOL
ATOL = RTOL = 1e-4        # these pinned cases' measured bound: tighter than the stated FP32_TOL
# the 256^2 images under TGSR_WINOGRAD=0 (direct kernels): the stated tolerance; the default Winograd kernels meet ATOL there too
ATOL256 = FP32_TOL if os.environ.get("TGSR_WINOGRAD", "1") == "0" else ATOL
DEV = "cuda"


def T(a, dev=DEV):
    return torch.from_numpy(np.asarray(a)).to(dev)


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from tgsr_amd import _lib
    _lib.lib()          # raises if the HIP library is missing: no silent fallback
    assert torch.cuda.is_available()


@pytest.fixture()
def cfg_small():
    from tgsr_amd.miscc.config import cfg, cfg_reset
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 64
    cfg.TREE.BRANCH_NUM = 4
    yield cfg
    cfg_reset()


@pytest.fixture()
def cfg_face():
    from tgsr_amd.miscc.config import cfg, cfg_reset
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 256
    cfg.TREE.BRANCH_NUM = 4                  # cfg/eval_*SR_attn2.yml: the x8 generators (trainer_objective.py:74-87)
    yield cfg
    cfg_reset()


# ------------------------------------------------------------------------------------------------ op level
CONV_CASES = [
    # B, Cin, H,  W,  Cout, glu, up, res
    (2, 64, 32, 32, 128, True, False, False),   # GL ResBlock conv1
    (2, 64, 32, 32, 64, False, False, True),    # GL ResBlock conv2 + skip
    (2, 64, 16, 16, 64, True, True, False),     # GL upBlock
    (3, 32, 32, 32, 64, True, False, False),    # GH ResBlock conv1
    (3, 32, 32, 32, 32, False, False, True),    # GH ResBlock conv2 + skip
    (3, 32, 32, 32, 32, False, False, False),   # residual24 tail (BN only)
    (2, 32, 24, 40, 64, True, True, False),     # GH upBlock, non-square
    (2, 3, 32, 32, 64, True, False, False),     # im2f / convin (Cin = 3)
    (1, 64, 13, 20, 128, True, False, False),   # ragged: H, W not multiples of the tile
    (1, 20, 7, 5, 32, False, False, True),      # tiny + Cin not a multiple of the chunk
    (1, 64, 5, 9, 64, True, True, False),       # ragged upsample
    (16, 64, 64, 64, 128, True, False, False),  # enough tiles for the 2-rows-per-wave variant
    (16, 64, 32, 32, 64, True, True, False),
    (4, 128, 16, 16, 256, True, False, False),  # ngf = 64: two channel groups per tile
    (4, 96, 16, 16, 96, False, False, True),    # odd channel-group count
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,glu,up,res", CONV_CASES)
def test_conv3x3_fused(B, Cin, H, W, Cout, glu, up, res):
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.3 * torch.randn(Cout, generator=g)
    co = Cout // 2 if glu else Cout
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    r = torch.randn(B, co, Ho, Wo, generator=g) if res else None
    xi = x.repeat_interleave(2, 2).repeat_interleave(2, 3) if up else x
    ref = F.conv2d(xi, w, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None]
    if glu:
        ref = O.glu(ref)
    if res:
        ref = ref + r
    wp = ops.pack_conv3x3_weight(w.to(DEV))
    out = ops.conv3x3_fused(x.to(DEV), wp, Cout, scale.to(DEV), shift.to(DEV), glu=glu, upsample=up,
                            residual=None if r is None else r.to(DEV))
    close(out, ref, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 64, 16, 16, 64), (2, 32, 24, 40, 64), (1, 64, 5, 9, 64), (16, 64, 32, 32, 64),
                                             (3, 20, 13, 33, 128), (2, 64, 128, 128, 64)])
def test_upconv_subpixel_vs_upsample_conv(B, Cin, H, W, Cout):
    """upBlock by sub-pixel decomposition == Upsample(x2 nearest) -> conv3x3 -> affine -> GLU."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B * 100 + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.3 * torch.randn(Cout, generator=g)
    xi = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = O.glu(F.conv2d(xi, w, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None])
    wide = torch.full((B, Cout // 2 + 6, 2 * H, 2 * W), 3.0, device=DEV)
    out = ops.upconv3x3_glu(x.to(DEV), ops.pack_upconv_weight(w.to(DEV)), Cout, scale.to(DEV), shift.to(DEV),
                            out=wide[:, 2:2 + Cout // 2])
    close(out, ref, atol=2e-5, rtol=2e-5)
    assert (wide[:, :2] == 3).all() and (wide[:, 2 + Cout // 2:] == 3).all()


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 64, 16, 16, 64), (3, 32, 32, 32, 64), (1, 64, 9, 20, 128),
                                             (1, 4, 3, 4, 64), (16, 64, 32, 32, 64), (2, 8, 7, 36, 128)])
def test_upblock_winograd_vs_upsample_conv(B, Cin, H, W, Cout):
    """upBlock by the up-sample-aware Winograd form (9 of 16 positions) == Upsample(x2) -> conv3x3 -> affine -> GLU."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B * 100 + Cin + H + 1)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.3 * torch.randn(Cout, generator=g)
    xi = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = O.glu(F.conv2d(xi, w, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None])
    wide = torch.full((B, Cout // 2 + 6, 2 * H, 2 * W), 3.0, device=DEV)
    out = ops.upwino_glu(x.to(DEV), ops.pack_upwino_weight(w.to(DEV)), Cout, scale.to(DEV), shift.to(DEV),
                         out=wide[:, 2:2 + Cout // 2])
    close(out, ref, atol=2e-5, rtol=2e-5)
    assert (wide[:, :2] == 3).all() and (wide[:, 2 + Cout // 2:] == 3).all()
    # without the gate (the raw convolution of the training forward): affine and affine-free
    raw_ref = F.conv2d(xi, w, None, 1, 1)
    raw = ops.upwino_glu(x.to(DEV), ops.pack_upwino_weight(w.to(DEV), glu=False), Cout, None, None, glu=False)
    close(raw, raw_ref, atol=2e-5, rtol=2e-5)
    aff = ops.upwino_glu(x.to(DEV), ops.pack_upwino_weight(w.to(DEV), glu=False), Cout, scale.to(DEV), shift.to(DEV),
                         glu=False)
    close(aff, raw_ref * scale[None, :, None, None] + shift[None, :, None, None], atol=2e-5, rtol=2e-5)


WINO_CASES = [
    # B, Cin, H, W, Cout, glu, res
    (2, 64, 32, 32, 128, True, False),
    (2, 64, 32, 32, 64, False, True),
    (3, 32, 32, 64, 64, True, False),
    (1, 64, 13, 20, 128, True, False),     # ragged rows / columns (even width)
    (1, 20, 6, 12, 64, False, True),       # five stages, image smaller than the workgroup tile
    (1, 4, 8, 32, 64, False, False),       # a single stage
    (1, 8, 10, 36, 128, True, False),      # two stages, ragged
    (16, 64, 64, 64, 128, True, False),
    (4, 128, 16, 16, 256, True, False),    # two channel groups
    (2, 64, 128, 128, 64, False, True),
    (16, 32, 32, 32, 32, False, True),     # 32-channel groups (GH ResBlock second conv)
    (2, 32, 13, 20, 32, False, False),     # ... ragged
    (1, 8, 64, 64, 96, False, True),       # three 32-channel groups
    (2, 16, 24, 40, 32, True, False),      # GLU inside a 32-channel group (16 value + 16 gate)
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,glu,res", WINO_CASES)
def test_conv3x3_winograd(B, Cin, H, W, Cout, glu, res):
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.3 * torch.randn(Cout, generator=g)
    co = Cout // 2 if glu else Cout
    r = torch.randn(B, co, H, W, generator=g) if res else None
    ref = F.conv2d(x, w, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None]
    ref = O.glu(ref) if glu else ref
    ref = ref + r if res else ref
    out = ops.conv3x3_wino(x.to(DEV), ops.pack_wino_weight(w.to(DEV), glu=glu), Cout, scale.to(DEV), shift.to(DEV), glu=glu,
                           residual=None if r is None else r.to(DEV))
    close(out, ref, atol=3e-5, rtol=3e-5)


WINO4_CASES = [
    # B, Cin, H, W, Cout, glu, res
    (2, 64, 128, 128, 128, True, False),   # the 128^2 ResBlock convolutions of G_SR_NET_low
    (2, 64, 128, 128, 64, False, True),
    (1, 4, 8, 64, 64, False, False),       # one stage, one workgroup
    (2, 8, 16, 64, 64, True, False),       # two stages
    (1, 12, 12, 68, 128, False, True),     # ragged rows / columns (W % 4 == 0), two channel groups
    (1, 20, 5, 8, 64, True, False),        # image smaller than one tile row
    (3, 32, 40, 132, 192, False, True),    # three channel groups, ragged
    (1, 64, 256, 256, 64, True, False),    # the x16 generator's 256^2 stage
    (2, 8, 10, 72, 256, True, False),      # wide form: two 128-row groups, ragged
    (2, 16, 24, 64, 128, False, True),     # wide form, plain epilogue + residual
    (16, 64, 64, 64, 128, True, False),    # the 64 -> 128 convolutions at 64^2, batch 16
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,glu,res", WINO4_CASES)
def test_conv3x3_winograd4(B, Cin, H, W, Cout, glu, res):
    """F(4x4, 3x3): against F.conv2d in fp64.  Stated bound 1e-4 on unit-scale data (measured 1.6e-5 .. 3.4e-5; F(2x2) on
    the same inputs 7e-7 .. 1.8e-6 - the price of 36 instead of 64 multiplies per 16 outputs, profiles/HISTORY.md 3.1e)."""
    from tgsr_amd import ops, custom_ops as C
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.3 * torch.randn(Cout, generator=g)
    co = Cout // 2 if glu else Cout
    r = torch.randn(B, co, H, W, generator=g) if res else None
    ref = F.conv2d(x.double(), w.double(), None, 1, 1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    ref = ref[:, :co] * torch.sigmoid(ref[:, co:]) if glu else ref
    ref = ref + r.double() if res else ref
    up = ops.pack_wino4_weight(w.to(DEV), glu=glu)
    out = ops.conv3x3_wino4(x.to(DEV), up, Cout, scale.to(DEV), shift.to(DEV), glu=glu, residual=None if r is None else r.to(DEV))
    err = float((out.cpu().double() - ref).abs().max())
    assert err < 1e-4, err
    again = ops.conv3x3_wino4(x.to(DEV), up, Cout, scale.to(DEV), shift.to(DEV), glu=glu, residual=None if r is None else r.to(DEV))
    assert torch.equal(out, again)
    # the raw convolution (no affine), as the custom op
    raw = C.conv3x3_wino4(x.to(DEV), ops.pack_wino4_weight(w.to(DEV), glu=False), Cout, None, None, False, None)
    assert float((raw.cpu().double() - F.conv2d(x.double(), w.double(), None, 1, 1)).abs().max()) < 1e-4
    if Cin % 8 == 0:
        # the register-fed form (A fragments loaded from L2 into registers a stage ahead; 128-row workgroups where Cout % 128 == 0,
        # else 64-row 4-wave ones): the same arithmetic in the same order - bit-identical to the narrow form
        upw = ops.pack_wino4w_weight(w.to(DEV), glu=glu)
        ow = C.conv3x3_wino4w(x.to(DEV), upw, Cout, scale.to(DEV), shift.to(DEV), glu, None if r is None else r.to(DEV))
        assert torch.equal(ow, out)
        assert torch.equal(ow, C.conv3x3_wino4w(x.to(DEV), upw, Cout, scale.to(DEV), shift.to(DEV), glu, None if r is None else r.to(DEV)))


def test_conv3x3_winograd4_channel_slices_and_refusals():
    """Reads from a channel slice, writes into one (batch strides of the wider tensors); misaligned tensors and
    unsupported channel counts are refused, not mis-read."""
    from tgsr_amd import ops
    from tgsr_amd._lib import TgsrError
    g = torch.Generator().manual_seed(11)
    wide_in = torch.randn(2, 96, 16, 64, generator=g).to(DEV)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24
    wide_out = torch.full((2, 80, 16, 64), 7.0, device=DEV)
    up = ops.pack_wino4_weight(w.to(DEV), glu=True)
    ops.conv3x3_wino4(wide_in[:, 32:], up, 128, None, None, glu=True, out=wide_out[:, 8:72])
    ref = O.glu(F.conv2d(wide_in[:, 32:].cpu().double(), w.double(), None, 1, 1))
    assert float((wide_out[:, 8:72].cpu().double() - ref).abs().max()) < 1e-4
    assert (wide_out[:, :8] == 7).all() and (wide_out[:, 72:] == 7).all()
    with pytest.raises(TgsrError):
        ops.pack_wino4_weight(torch.randn(32, 64, 3, 3, device=DEV))                    # Cout % 64
    with pytest.raises(TgsrError):
        ops.conv3x3_wino4(wide_in[:, 32:, :, 1:63], up, 128, None, None, glu=True)     # not contiguous rows / W % 4
    with pytest.raises(TgsrError):
        ops.conv3x3_wino4(wide_in[:, :32], up, 128, None, None, glu=True)              # pack made for 64 input channels
    with pytest.raises(TgsrError):                                                     # wide form: an even number of stages
        ops.conv3x3_wino4(torch.randn(1, 12, 8, 64, device=DEV), ops.pack_wino4w_weight(torch.randn(128, 12, 3, 3, device=DEV)),
                          128, None, None, wide=True)


UPW4_CASES = [
    # B, Cin, H, W (low resolution), Cout, glu
    (2, 64, 64, 64, 64, True),       # G_SR_NET_low's second upBlock (64^2 -> 128^2)
    (1, 64, 128, 128, 64, True),     # ... third (128^2 -> 256^2)
    (2, 32, 64, 64, 64, True),       # NetG_highweight's
    (1, 8, 4, 32, 64, True),         # two stages, two workgroups
    (2, 8, 8, 32, 64, False),        # without the gate
    (1, 16, 6, 36, 128, True),       # ragged rows / columns, two channel groups
    (3, 24, 3, 4, 64, True),         # image smaller than a tile row
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,glu", UPW4_CASES)
def test_upblock_winograd4_vs_upsample_conv(B, Cin, H, W, Cout, glu):
    """Up-sample-aware F(4x4, 3x3): against F.conv2d(F.interpolate(x, 2)) in fp64, bound 1e-4 on unit-scale data (measured
    1.0e-5 .. 2.9e-5; the F(2x2) form 5e-7 .. 2.7e-6), bit-reproducible, through the custom op and into a channel slice."""
    from tgsr_amd import ops, custom_ops as C
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.3 * torch.randn(Cout, generator=g)
    co = Cout // 2 if glu else Cout
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), None, 1, 1)
    ref = ref * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    ref = ref[:, :co] * torch.sigmoid(ref[:, co:]) if glu else ref
    up = ops.pack_upwino4_weight(w.to(DEV), glu=glu)
    out = ops.upwino4_glu(x.to(DEV), up, Cout, scale.to(DEV), shift.to(DEV), glu=glu)
    err = float((out.cpu().double() - ref).abs().max())
    assert err < 1e-4, err
    assert torch.equal(out, ops.upwino4_glu(x.to(DEV), up, Cout, scale.to(DEV), shift.to(DEV), glu=glu))
    if glu:
        wide = torch.full((B, co + 8, 2 * H, 2 * W), 3.0, device=DEV)
        C.upwino4_glu_out(x.to(DEV), C.pack_upwino4_weight(w.to(DEV), True), Cout, scale.to(DEV), shift.to(DEV), wide[:, 4:4 + co])
        assert torch.equal(wide[:, 4:4 + co], out)
        assert (wide[:, :4] == 3).all() and (wide[:, 4 + co:] == 3).all()
    from tgsr_amd._lib import TgsrError
    with pytest.raises(TgsrError):                                     # an odd number of 4-channel stages is refused
        ops.upwino4_glu(torch.randn(1, 12, 4, 32, device=DEV), ops.pack_upwino4_weight(torch.randn(64, 12, 3, 3, device=DEV)), 64,
                        scale[:64].to(DEV), shift[:64].to(DEV))


def test_winograd4_training_forms():
    """What the training step uses of the F(4x4) kernel: the raw convolution whose epilogue leaves BatchNorm's batch
    statistics as per-wave partial sums, and the data-gradient pack made from the FORWARD weight."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(21)
    B, Cin, Cout, H, W = 3, 64, 128, 24, 132
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24).to(DEV)
    up = ops.pack_wino4_weight(w)
    raw, part = ops.conv3x3_wino4_stats(x, up, Cout)
    assert torch.equal(raw, ops.conv3x3_wino4(x, up, Cout, None, None))
    assert part.shape == (Cout, ops.wino4_stats_nslots(B, H, W, Cout), 2)
    s = part.double().sum(1).cpu()
    r = raw.double().cpu()
    np.testing.assert_allclose(s[:, 0].numpy(), r.sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].numpy(), (r * r).sum((0, 2, 3)).numpy(), rtol=1e-5)
    gam, bet = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    rm0, rv0 = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    rm1, rv1 = rm0.clone(), rv0.clone()
    o0, st0 = ops.bn_train_fwd(raw, gam, bet, 1e-5, 0.1, rm0, rv0, 1)
    o1, st1 = ops.bn_train_fwd(raw, gam, bet, 1e-5, 0.1, rm1, rv1, 1, stat_partial=part)
    close(o1, o0, atol=1e-5, rtol=1e-5)
    close(st1, st0, atol=1e-5, rtol=1e-5)
    close(rv1, rv0, atol=1e-6, rtol=1e-5)
    # data gradient of a 64 -> 128 convolution: 128 channels in, 64 out, from the forward weight
    dy = torch.randn(B, Cout, H, W, generator=g).to(DEV)
    add = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    dx = ops.conv3x3_wino4(dy, ops.pack_wino4_weight(w, dgrad=True), Cin, None, None, residual=add)
    ref = F.conv_transpose2d(dy.double().cpu(), w.double().cpu(), None, 1, 1) + add.double().cpu()
    assert float((dx.double().cpu() - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    # the register-fed forms of both (128-row groups for the forward, 64-row groups for the gradient): bit-identical outputs,
    # the same per-wave statistics in another slot order
    raww, partw = ops.conv3x3_wino4_stats(x, ops.pack_wino4w_weight(w), Cout, wide=True)
    assert torch.equal(raww, raw) and partw.shape == part.shape == (Cout, ops.wino4_stats_nslots(B, H, W, Cout, wide=True), 2)
    np.testing.assert_allclose(partw.double().sum(1).cpu().numpy(), s.numpy(), rtol=1e-6, atol=1e-4)
    assert torch.equal(torch.sort(partw[:, :, 0], dim=1).values, torch.sort(part[:, :, 0], dim=1).values)
    assert torch.equal(ops.conv3x3_wino4(dy, ops.pack_wino4w_weight(w, dgrad=True), Cin, None, None, residual=add, wide=True), dx)


def test_wino4_routing_follows_the_size_policy(monkeypatch):
    """util._conv_bn sends the large layers (>= 64 x 64 pixels, a full round of workgroups) to F(4x4) and everything else to
    F(2x2) / the direct kernel; ops.ROUTING.wino4 = False (TGSR_WINO4=0 at import) keeps F(2x2) everywhere."""
    from tgsr_amd import util, custom_ops as C
    calls = []
    real4, real2 = C.conv3x3_wino4w, C.conv3x3_wino
    monkeypatch.setattr(C, "conv3x3_wino4w", lambda *a: (calls.append(4), real4(*a))[1])
    monkeypatch.setattr(C, "conv3x3_wino", lambda *a: (calls.append(2), real2(*a))[1])
    conv = torch.nn.Conv2d(64, 128, 3, 1, 1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(128).to(DEV).eval()
    fp = util._FusedParams()
    big, small, few = torch.randn(4, 64, 128, 128, device=DEV), torch.randn(4, 64, 32, 64, device=DEV), torch.randn(1, 64, 128, 128, device=DEV)
    y4 = util._conv_bn(big, fp, conv, bn, glu=True)
    util._conv_bn(small, fp, conv, bn, glu=True)
    util._conv_bn(few, fp, conv, bn, glu=True)
    assert calls == [4, 2, 2]
    from tgsr_amd import ops
    monkeypatch.setattr(ops.ROUTING, "wino4", False)          # what TGSR_WINO4=0 sets at import
    y2 = util._conv_bn(big, fp, conv, bn, glu=True)
    assert calls == [4, 2, 2, 2]
    assert float((y4 - y2).abs().max()) < 1e-4


def test_conv3x3_channel_slice_io():
    """Reads from / writes into channel slices of wider buffers (how torch.cat disappears)."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(5)
    wide_in = torch.randn(3, 96, 16, 32, generator=g).to(DEV)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    wide_out = torch.full((3, 80, 16, 32), 7.0, device=DEV)
    wp = ops.pack_conv3x3_weight(w.to(DEV))
    ops.conv3x3_fused(wide_in[:, 32:], wp, 64, None, None, glu=True, out=wide_out[:, 8:40])
    ref = O.glu(F.conv2d(wide_in[:, 32:].cpu(), w, None, 1, 1))
    close(wide_out[:, 8:40], ref, atol=2e-5, rtol=2e-5)
    assert (wide_out[:, :8] == 7).all() and (wide_out[:, 40:] == 7).all()


@pytest.mark.parametrize("B,Cin,H,W,K,act", [(2, 32, 64, 64, 3, False), (2, 32, 64, 64, 5, True),
                                              (3, 32, 19, 37, 5, True), (1, 32, 16, 16, 3, False),
                                              (1, 20, 33, 70, 5, True), (4, 32, 128, 128, 5, True),
                                              # the MFMA form of the 5x5 heads (>= 512 tiles of 8 x 64): both activations,
                                              # 2 and 3 channel chunks, and the same shape below the threshold
                                              (16, 32, 128, 128, 5, True), (8, 48, 256, 256, 5, False),
                                              (16, 32, 64, 128, 5, True)])
def test_conv_to3(B, Cin, H, W, K, act):
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(K * 100 + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(3, Cin, K, K, generator=g) / (K * Cin ** 0.5)
    add = torch.randn(B, 3, H, W, generator=g) if act else None
    ref = F.conv2d(x, w, None, 1, K // 2)
    if act:
        ref = torch.tanh(ref) + 0.5 * add
    out = ops.conv_to3(x.to(DEV), w.to(DEV), tanh_axpy=act, addend=None if add is None else add.to(DEV), alpha=0.5)
    close(out, ref, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("B,Cin,H,W,K,act", [(16, 32, 256, 256, 3, False), (4, 32, 128, 128, 3, False), (2, 32, 64, 64, 3, False),
                                             (2, 32, 64, 64, 5, True), (3, 64, 36, 128, 3, True), (1, 5, 9, 64, 3, False),
                                             (2, 32, 130, 68, 5, False), (1, 33, 16, 192, 3, True)])
def test_conv_to3_with_copies_in_flight_equals_the_double_buffered_kernel_bit_for_bit(B, Cin, H, W, K, act):
    """conv_to3_pipe_kernel (LDS copies two stages ahead in a ring of three buffers behind counted waits, the filter in LDS: the
    default where W % 4 == 0) runs the FMA chains of conv_to3_kernel in the same order: identical bits, on every tile form (16-, 8-
    and 4-row tiles with 1 / 2 / 4 channel groups), ragged heights, odd channel counts - and both agree with torch."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(K * 100 + H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(3, Cin, K, K, generator=g) / (K * Cin ** 0.5)).to(DEV)
    add = torch.randn(B, 3, H, W, generator=g).to(DEV) if act else None
    outs = []
    was = ops.conv_to3_set_pipe(True)
    try:
        for pipe in (True, False):
            ops.conv_to3_set_pipe(pipe)
            outs.append(ops.conv_to3(x, w, tanh_axpy=act, addend=add, alpha=0.5))
    finally:
        ops.conv_to3_set_pipe(was)
    assert torch.equal(outs[0], outs[1])
    ref = F.conv2d(x.cpu(), w.cpu(), None, 1, K // 2)
    if act:
        ref = torch.tanh(ref) + 0.5 * add.cpu()
    close(outs[0], ref, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (3, 64, 64), (1, 128, 128)])
def test_split_head_equals_fused_head_bit_for_bit(B, H, W):
    """NetG_highweight's head `tanh(conv5x5(out)) + a * SRb` (model.py:280) computed in two launches - the convolution + tanh
    without the addend (ahead of G_SR_NET_low), then tgsr_axpy_images - is the fused epilogue's result bit for bit, for several
    scales in one axpy launch."""
    from tgsr_amd import ops
    import tgsr_amd.custom_ops  # noqa: F401   (registers torch.ops.tgsr.*: the test must not depend on an earlier one having done so)
    g = torch.Generator().manual_seed(H + W)
    xs = [torch.randn(B, 32, H >> k, W >> k, generator=g).to(DEV) for k in range(3)]
    adds = [torch.randn(B, 3, H >> k, W >> k, generator=g).to(DEV) for k in range(3)]
    w = (torch.randn(3, 32, 5, 5, generator=g) / 28).to(DEV)
    fused = [ops.conv_to3(x, w, tanh_axpy=True, addend=a, alpha=0.5) for x, a in zip(xs, adds)]
    ts = [ops.conv_to3(x, w, tanh_axpy=True, addend=None, alpha=0.5) for x in xs]
    split = torch.ops.tgsr.axpy_images(ts, adds, 0.5)
    for f, s_ in zip(fused, split):
        assert torch.equal(f, s_)
    ref = torch.tanh(F.conv2d(xs[1].cpu(), w.cpu(), None, 1, 2)) + 0.5 * adds[1].cpu()
    close(split[1], ref, atol=2e-5, rtol=2e-5)
    with pytest.raises(ops.TgsrError):
        ops.axpy_images([ts[0]], [adds[1]], 0.5)                      # shapes must agree
    with pytest.raises(ops.TgsrError):
        ops.axpy_images(ts + ts, adds + adds, 0.5)                    # at most 4 images per launch


def test_pipeline_split_heads_equal_reference_order(face_weights, cfg_face, monkeypatch):
    """SRPipeline with NetG_highweight's convolution heads on the side stream (the default) == the six stand-alone heads in the
    reference's order (trainer.SPLIT_HEADS = False), bit for bit, eager and replayed from a hipGraph."""
    from tgsr_amd import trainer
    cap, lens, LR, LRb = O.synthetic_batch(4, seed=3)
    args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    assert trainer.SPLIT_HEADS
    p = _pipeline(face_weights)
    a = [f.clone() for f in p(*args)["fine"]]
    monkeypatch.setattr(trainer, "SPLIT_HEADS", False)
    b = [f.clone() for f in _pipeline(face_weights)(*args)["fine"]]
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    monkeypatch.setattr(trainer, "SPLIT_HEADS", True)
    p.capture(*args)
    for x, y in zip(p.replay()["fine"], a):
        assert torch.equal(x, y)


def test_word_attention_golden_quirk_and_b1(ops_small):
    from tgsr_amd import ops
    g = ops_small
    out, attn = ops.word_attention(T(g["att.h"]), T(g["att.ctx"]), T(g["att.w"]), T(g["att.mask"]))
    close(out, g["att.out"], atol=2e-5)
    close(attn, g["att.attn"], atol=2e-6)
    out1, attn1 = ops.word_attention(T(g["att.h"][:1]), T(g["att.ctx"][:1]), T(g["att.w"]), None)
    close(out1, g["att1.out"], atol=2e-5)
    close(attn1, g["att1.attn"], atol=2e-6)


@pytest.mark.parametrize("B,idf,r,T_,correct", [(3, 32, 32, 18, False), (3, 32, 32, 18, True), (16, 32, 64, 14, False),
                                                (5, 32, 20, 7, False), (2, 64, 16, 32, False), (2, 128, 8, 3, True)])
def test_word_attention_vs_oracle(B, idf, r, T_, correct):
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B * 10 + r)
    h = torch.randn(B, idf, r, r, generator=g)
    words = torch.randn(B, 256, T_, generator=g)
    w = torch.randn(idf, 256, 1, 1, generator=g) / 16
    lens = torch.randint(1, T_ + 1, (B,), generator=g)
    lens[0] = T_
    mask = torch.arange(T_)[None, :] >= lens[:, None]
    ref_o, ref_a = O.word_attention(h, words, w, mask, correct_mask=correct)
    out, attn = ops.word_attention(h.to(DEV), words.to(DEV), w.to(DEV), mask.to(DEV), correct_mask=correct)
    close(attn, ref_a, atol=5e-6, rtol=1e-4)
    close(out, ref_o, atol=5e-5, rtol=1e-4)
    # property at any size: attention is a distribution over the unmasked words of the row it was masked with
    s = attn.sum(1)
    close(s, torch.ones_like(s), atol=1e-5)


@pytest.mark.parametrize("B,idf,cdf,T_,n", [(16, 32, 256, 18, 3), (3, 64, 100, 7, 1), (2, 128, 37, 32, 4)])
def test_word_project_batched(B, idf, cdf, T_, n):
    """One-launch projection for several conv_context weight sets == conv1x1 per set; attention through `src=`."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B + idf + cdf)
    words = torch.randn(B, cdf, T_, generator=g)
    ws = [torch.randn(idf, cdf, 1, 1, generator=g) / cdf ** 0.5 for _ in range(n)]
    srcs = ops.word_project(words.to(DEV), [w.to(DEV) for w in ws])
    for w, src in zip(ws, srcs):
        ref = torch.einsum("ic,bct->bit", w.reshape(idf, cdf).double(), words.double()).float()
        close(src[:, :, :T_], ref, atol=2e-5, rtol=1e-5)
        assert float(src[:, :, T_:].abs().max()) == 0.0 if T_ < 32 else True
    h = torch.randn(B, idf, 8, 16, generator=g)
    mask = torch.arange(T_)[None, :] >= torch.randint(1, T_ + 1, (B, 1), generator=g)
    o0, a0 = ops.word_attention(h.to(DEV), words.to(DEV), ws[-1].to(DEV), mask.to(DEV))
    o1, a1 = ops.word_attention(h.to(DEV), words.to(DEV), ws[-1].to(DEV), mask.to(DEV), src=srcs[-1])
    close(o1, o0.cpu(), atol=2e-5, rtol=1e-4)
    close(a1, a0.cpu(), atol=2e-6, rtol=1e-4)


def test_bilstm_golden(ops_small):
    from tgsr_amd import ops
    g = ops_small
    sd = split_sd(g, "enc.", DEV)
    st = lambda a, b: torch.stack([sd[a], sd[b]]).contiguous()
    words, sent = ops.bilstm(T(g["enc.captions"]), g["enc.cap_lens"].tolist(), sd["encoder.weight"],
                             st("rnn.weight_ih_l0", "rnn.weight_ih_l0_reverse"),
                             st("rnn.weight_hh_l0", "rnn.weight_hh_l0_reverse"),
                             st("rnn.bias_ih_l0", "rnn.bias_ih_l0_reverse"),
                             st("rnn.bias_hh_l0", "rnn.bias_hh_l0_reverse"))
    close(words, g["enc.words_emb"], atol=1e-5)
    close(sent, g["enc.sent_emb"], atol=1e-5)
    # eval-mode per-token gate table: bit-identical to the per-position projection
    table = ops.lstm_gate_table(sd["encoder.weight"], st("rnn.weight_ih_l0", "rnn.weight_ih_l0_reverse"),
                                st("rnn.bias_ih_l0", "rnn.bias_ih_l0_reverse"), st("rnn.bias_hh_l0", "rnn.bias_hh_l0_reverse"))
    w2, s2 = ops.bilstm_table(T(g["enc.captions"]), g["enc.cap_lens"].tolist(), table,
                              st("rnn.weight_hh_l0", "rnn.weight_hh_l0_reverse"))
    assert torch.equal(w2, words) and torch.equal(s2, sent)


def test_rnn_encoder_gru_branch_golden():
    """RNN_ENCODER with cfg.RNN_TYPE = 'GRU' (util.py:207-211): the drop-in module loads the reference's state_dict and reproduces
    the reference's own outputs (enc_gru.npz) through tgsr_gru_gate_table + tgsr_bigru_table_fwd; host lengths give T_max
    columns, device lengths the full caption width with zeros behind each caption (what a captured step uses): same values."""
    import os
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.util import RNN_ENCODER
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "enc_gru.npz"))
    cfg_reset()
    cfg.RNN_TYPE = "GRU"
    try:
        for tag, nhidden in (("a", 256), ("b", 64)):
            pre = "gru_%s." % tag
            sd = {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
            enc = RNN_ENCODER(41, nhidden=nhidden).to(DEV).eval()
            enc.load_state_dict({k: v for k, v in sd.items() if k.startswith(("encoder.", "rnn."))}, strict=True)
            cap, lens = sd["captions"].to(DEV), sd["cap_lens"].tolist()
            with torch.no_grad():
                we, se = enc(cap, lens, enc.init_hidden(len(lens)))
                wf, sf = enc(cap, torch.tensor(lens, dtype=torch.int32, device=DEV), None)
            close(we, sd["words_emb"], atol=1e-5)
            close(se, sd["sent_emb"], atol=1e-5)
            T_ = max(lens)
            assert tuple(wf.shape) == (len(lens), nhidden, cap.shape[1]) and torch.equal(wf[:, :, :T_], we) and torch.equal(sf, se)
            assert float(wf[:, :, T_:].abs().max()) == 0 if T_ < cap.shape[1] else True
            for b, n in enumerate(lens):                                     # zeros behind every caption
                assert float(we[b, :, n:].abs().max()) == 0 if n < T_ else True
            enc.train()                                                      # the training branch (round 5): same values without dropout,
            enc.drop.p = 0.0                                                 # and gradients reach the recurrent weights
            wt, st = enc(cap, lens, None)
            close(wt, sd["words_emb"], atol=1e-5)
            close(st, sd["sent_emb"], atol=1e-5)
            (wt.square().sum() + st.square().sum()).backward()
            assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0 for p in enc.rnn.parameters())
    finally:
        cfg_reset()


def test_func_attention_golden(ops_small):
    from tgsr_amd import ops
    g = ops_small
    wc, attn = ops.func_attention(T(g["fa.query"]), T(g["fa.context"]), float(g["fa.gamma1"]))
    close(wc, g["fa.out"], atol=2e-5)
    close(attn, g["fa.attn"], atol=2e-6)


def test_damsm_words_and_sent_loss_golden(damsm_golden, cfg_face):
    """DAMSM goldens (losses.py:21-136): B=4, caption lengths 18/15/12/9, with and without class ids."""
    from tgsr_amd.miscc import losses
    g = damsm_golden
    cfg_face.TRAIN.SMOOTH.GAMMA1, cfg_face.TRAIN.SMOOTH.GAMMA2, cfg_face.TRAIN.SMOOTH.GAMMA3 = map(float, g["gamma"])
    labels = torch.arange(4, device=DEV)
    for tag, cls in (("cls", g["class_ids"]), ("nocls", None)):
        w0, w1, att = losses.words_loss(T(g["feats"]), T(g["words"]), labels, T(g["cap_lens"], "cpu"), cls, 4)
        s0, s1 = losses.sent_loss(T(g["cnn_code"]), T(g["sent"]), labels, cls, 4)
        close(w0, g[tag + ".w0"], atol=2e-5); close(w1, g[tag + ".w1"], atol=2e-5)
        close(s0, g[tag + ".s0"], atol=2e-5); close(s1, g[tag + ".s1"], atol=2e-5)
        for i, a in enumerate(att):
            close(a, g[tag + ".att%d" % i], atol=2e-6)


def test_damsm_loss_gradients_golden(damsm_golden, cfg_face):
    """Backward of words_loss + sent_loss (HIP DAMSM backward kernel) vs the gradients captured from the reference."""
    from tgsr_amd.miscc import losses
    g = damsm_golden
    cfg_face.TRAIN.SMOOTH.GAMMA1, cfg_face.TRAIN.SMOOTH.GAMMA2, cfg_face.TRAIN.SMOOTH.GAMMA3 = map(float, g["gamma"])
    labels = torch.arange(4, device=DEV)
    for tag, cls in (("cls", g["class_ids"]), ("nocls", None)):
        feats, words = T(g["feats"]).requires_grad_(), T(g["words"]).requires_grad_()
        cnn, sent = T(g["cnn_code"]).requires_grad_(), T(g["sent"]).requires_grad_()
        w0, w1, _ = losses.words_loss(feats, words, labels, T(g["cap_lens"], "cpu"), cls, 4)
        s0, s1 = losses.sent_loss(cnn, sent, labels, cls, 4)
        (w0 + w1 + s0 + s1).backward()
        for name, t in (("g_feats", feats), ("g_words", words), ("g_cnn", cnn), ("g_sent", sent)):
            ref = torch.from_numpy(g[tag + "." + name])
            scale = float(ref.abs().max())
            close(t.grad, ref, atol=2e-5 * scale, rtol=2e-4)


@pytest.mark.parametrize("B,ndf,Tw,S", [(5, 256, 18, 289), (3, 64, 7, 25), (2, 128, 32, 320)])
def test_damsm_backward_vs_oracle_autograd(B, ndf, Tw, S):
    """tgsr_damsm_words_bwd vs torch autograd through the oracle's per-caption loop, random upstream gradient."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B * 7 + Tw)
    ih, iw = (17, 17) if S == 289 else ((5, 5) if S == 25 else (16, 20))
    feats = torch.randn(B, ndf, ih, iw, generator=g, dtype=torch.float64).requires_grad_()
    words = torch.randn(B, ndf, Tw, generator=g, dtype=torch.float64).requires_grad_()
    lens = torch.randint(1, Tw + 1, (B,), generator=g).tolist()
    lens[0] = Tw
    gs = torch.randn(B, B, generator=g, dtype=torch.float64)
    cols = []
    for i in range(B):
        L = lens[i]
        word = words[i:i + 1, :, :L].expand(B, -1, -1)
        wc, _ = O.func_attention(word, feats, 4.0)
        row = O.cosine_similarity(word.transpose(1, 2).reshape(B * L, -1), wc.transpose(1, 2).reshape(B * L, -1))
        cols.append(torch.log(torch.exp(row.reshape(B, L) * 5.0).sum(1)))
    sim = torch.stack(cols, 1)
    (sim * gs).sum().backward()
    g_img, g_words = ops.damsm_words_bwd(feats.detach().float().to(DEV), words.detach().float().to(DEV), lens, 4.0, 5.0,
                                         gs.float().to(DEV))
    for got, ref in ((g_img, feats.grad), (g_words, words.grad)):
        ref = ref.float()
        close(got, ref, atol=3e-5 * float(ref.abs().max()), rtol=1e-3)
    for i in range(B):
        assert float(g_words[i, :, lens[i]:].abs().max()) == 0.0 if lens[i] < Tw else True


@pytest.mark.parametrize("B,ndf,Tw,S", [(16, 256, 18, 289), (3, 64, 7, 25), (5, 128, 32, 320)])
def test_damsm_similarity_vs_oracle(B, ndf, Tw, S):
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B + Tw)
    ih = int(S ** 0.5) if int(S ** 0.5) ** 2 == S else None
    feats = torch.randn(B, ndf, ih or 16, ih or 20, generator=g)
    words = torch.randn(B, ndf, Tw, generator=g)
    lens = torch.randint(1, Tw + 1, (B,), generator=g).tolist()
    lens[0] = Tw
    sim, att = ops.damsm_words_similarity(feats.to(DEV), words.to(DEV), lens, 4.0, 5.0)
    for i in range(B):
        L = lens[i]
        word = words[i:i + 1, :, :L].expand(B, -1, -1)
        wc, attn = O.func_attention(word, feats, 4.0)
        row = O.cosine_similarity(word.transpose(1, 2).reshape(B * L, -1), wc.transpose(1, 2).reshape(B * L, -1))
        ref = torch.log(torch.exp(row.reshape(B, L) * 5.0).sum(1))
        close(sim[:, i], ref, atol=2e-5, rtol=1e-5)
        close(att[i, :L], attn[i], atol=2e-6)
        assert float(att[i, L:].abs().max()) == 0.0 if L < Tw else True


def test_cnn_encoder_heads(cfg_face):
    """emb_features (conv1x1 768->256 on 17x17) and emb_cnn_code (Linear 2048->256) vs torch; trunk injected."""
    from tgsr_amd import util

    class FakeTrunk(torch.nn.Module):
        def forward(self, x):
            g = torch.Generator().manual_seed(int(x.shape[0]))
            return (torch.randn(x.shape[0], 768, 17, 17, generator=g).to(x.device),
                    torch.randn(x.shape[0], 2048, generator=g).to(x.device))

    enc = util.CNN_ENCODER(256, trunk=FakeTrunk()).to(DEV).eval()
    assert sorted(k for k in enc.state_dict() if not k.startswith("trunk.")) == \
        ["emb_cnn_code.bias", "emb_cnn_code.weight", "emb_features.weight"]
    for B in (1, 5, 16):
        x = torch.zeros(B, 3, 64, 64, device=DEV)
        feats, code = enc(x)
        f768, p2048 = enc.run_trunk(x)
        close(feats, F.conv2d(f768.cpu(), enc.emb_features.weight.detach().cpu()), atol=5e-5, rtol=1e-4)
        close(code, F.linear(p2048.cpu(), enc.emb_cnn_code.weight.detach().cpu(), enc.emb_cnn_code.bias.detach().cpu()),
              atol=1e-4, rtol=1e-4)


def test_kl_mse_golden(ops_small):
    from tgsr_amd.miscc import losses
    g = ops_small
    close(losses.KL_loss(T(g["kl.mu"]), T(g["kl.logvar"])), g["kl.out"], atol=1e-6)
    close(losses.MSE([T(g["mse.a0"]), T(g["mse.a1"])], [T(g["mse.b0"]), T(g["mse.b1"])]), g["mse.out"], atol=1e-6)


# ------------------------------------------------------------------------------------------------ module level
def test_resblock_upblock_modules_golden(ops_small, cfg_small):
    from tgsr_amd import util
    g = ops_small
    x = T(g["blk.x"])
    rb = util.ResBlock(64)
    params = lambda pre: {k: v for k, v in split_sd(g, pre).items() if k[0].isdigit() or k.startswith("block.")}
    rb.load_state_dict(params("rb."))
    rb.to(DEV).eval()
    close(rb(x), g["rb.eval"], atol=2e-5)
    ub = util.upBlock(64, 32)
    ub.load_state_dict(params("ub."))
    ub.to(DEV).eval()
    close(ub(x), g["ub.eval"], atol=2e-5)
    # weights changed in place -> the packed/folded cache must follow
    with torch.no_grad():
        ub[1].weight.mul_(2.0)
    sd = {k: v.clone() for k, v in params("ub.").items()}
    sd["1.weight"] *= 2
    close(ub(x), O.up_block(T(g["blk.x"], "cpu"), sd, ""), atol=4e-5)


def _pipeline(arrs, n_words=41):
    from tgsr_amd.trainer import SRPipeline
    p = SRPipeline(n_words, device=DEV, low="lr")
    p.load_state_dicts(split_sd(arrs, "E."), split_sd(arrs, "GL."), split_sd(arrs, "GH."))
    return p


def test_generators_small_golden(nets_small, cfg_small):
    """ngf=32 / nef=64, LR 16x16, B=3 with unequal captions (mask quirk live), eval BN, whole caller path."""
    g = nets_small
    p = _pipeline(g)
    r = p(T(g["captions"]), g["cap_lens"].tolist(), T(g["LR"]), T(g["LRb"]))
    close(r["words_emb"], g["eval.words_emb"], atol=1e-5)
    close(r["sent_emb"], g["eval.sent_emb"], atol=1e-5)
    close(r["mu"], g["eval.mu"], atol=1e-5)
    close(r["logvar"], g["eval.logvar"], atol=1e-5)
    for i in range(3):
        close(r["att"][i], g["eval.att%d" % i], atol=2e-5)
        close(r["fake"][i], g["eval.fake%d" % i])
        close(r["fine"][i], g["eval.fine%d" % i])


def test_x16_generator_golden(cfg_small):
    """models16.G_SR_NET_low (weight-tied stages, tanh heads, 4th attention stage) vs the reference golden."""
    from conftest import load_npz
    from tgsr_amd import models16, util
    g = load_npz("nets16_small.npz")
    enc = util.RNN_ENCODER(41, nhidden=64)
    enc.load_state_dict(split_sd(g, "E."))
    gl = models16.G_SR_NET_low()
    gl.load_state_dict(split_sd(g, "GL."), strict=False)      # tied aliases are stored once in the fixture
    enc.to(DEV).eval(); gl.to(DEV).eval()
    cap = T(g["captions"])
    words, sent = enc(cap, g["cap_lens"].tolist(), enc.init_hidden(2))
    mask = (cap == 0)[:, :words.size(2)]
    imgs, atts, mu, lv = gl(T(g["LR"]), sent, words, mask)
    assert len(imgs) == 4
    for i in range(4):
        close(imgs[i], g["fake%d" % i])
        close(atts[i], g["att%d" % i], atol=2e-5)
    close(mu, g["mu"], atol=1e-5)
    # x16 high-frequency net: runs (the reference's cannot, see models16.py docstring) and matches the x8 oracle
    # arithmetic stage by stage when fed the same weights for the shared modules
    gh = models16.NetG_highweight(weightmap=False, low="lr").to(DEV).eval()
    fine, a, one = gh(T(g["LR"]), imgs, T(g["LR"]))
    assert [tuple(f.shape[-2:]) for f in fine] == [(16, 16), (32, 32), (64, 64), (128, 128)] and float(a) == 0.5


def test_full_size_face_checkpoint_c1(face_c1, face_weights, cfg_face):
    """BASELINE config 1: the shipped x8 face checkpoints, B=2, 32->256."""
    from tgsr_amd.trainer import to_uint8
    g = face_c1
    p = _pipeline(face_weights)
    r = p(T(g["captions"]), g["cap_lens"].tolist(), T(g["LR"]), T(g["LRb"]))
    close(r["words_emb"], g["words_emb"], atol=1e-5)
    for i in range(3):
        # stated tolerance 1e-4 everywhere with the default (Winograd) kernels: 3.1e-5 max from an fp64 run, the
        # reference's own CPU fp32 path 3.7e-5 (tools/diag_precision.py).  TGSR_WINOGRAD=0 (direct kernels, a pure
        # 576-step fp32 FMA chain per output: 5.6e-5 .. 8.3e-5) keeps 2e-4 on the 256^2 images
        close(r["fake"][i], g["fake%d" % i], atol=ATOL256 if i == 2 else ATOL)
        close(r["fine"][i], g["fine%d" % i], atol=ATOL256 if i == 2 else ATOL)
    close(r["att"][0], g["att0"], atol=2e-5)
    close(r["att"][1], g["att1"], atol=2e-5)
    a2 = r["att"][2].cpu().numpy()
    close(a2[:, :, ::8, ::8], g["att2.sub8"], atol=2e-5)
    close(a2[:, :, 40:72, 40:72], g["att2.crop"], atol=2e-5)
    u8 = to_uint8(r["fine"][-1])
    assert (np.abs(u8.astype(int) - g["sr_uint8"].astype(int)) <= 1).all()
    assert (u8 != g["sr_uint8"]).mean() < 1e-3


def test_full_size_batch16_vs_oracle(face_weights, cfg_face):
    """BASELINE config 2 (B=16, 32->256, synthetic inputs of SURVEY 8d) against the CPU oracle."""
    cap, lens, LR, LRb = O.synthetic_batch(16)
    ref = O.sr_forward(split_sd(face_weights, "E."), split_sd(face_weights, "GL."), split_sd(face_weights, "GH."),
                       cap, lens.tolist(), LR, LRb)
    p = _pipeline(face_weights)
    r = p(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    for i in range(3):
        close(r["fake"][i], ref["fake"][i], atol=ATOL256 if i == 2 else ATOL)
        close(r["fine"][i], ref["fine"][i], atol=ATOL256 if i == 2 else ATOL)
        close(r["att"][i], ref["att"][i], atol=2e-5)
    # size-independent properties: per-sample independence (eval BN, per-sample mask mode) and determinism
    p.netGL.h_net1.att.correct_mask = p.netGL.h_net2.att.correct_mask = p.netGL.h_net3.att.correct_mask = True
    full = p(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))["fine"][2]
    half = p(cap[:8].to(DEV), lens[:8].tolist(), LR[:8].to(DEV), LRb[:8].to(DEV))["fine"][2]
    # (not bitwise: a few kernels pick their tile shape by workgroup count, i.e. by batch - different summation orders, 1e-7 -
    # and the F(4x4) convolutions of the 128^2 stage turn any such difference into one of the size of their own rounding noise,
    # 1.5e-5 measured; a leak between samples - batch statistics, the mask quirk - would be 1e-2)
    assert torch.equal(full[:8], half) or float((full[:8] - half).abs().max()) < 5e-5
    again = p(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))["fine"][2]
    assert torch.equal(full, again)


def test_pipeline_hipgraph_replay_matches_eager():
    """BASELINE config 5: the whole step (two streams, ~60 launches) captured into a hipGraph replays bit-identically,
    also on new images copied into its static inputs."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.synthetic import random_init_, synthetic_batch
    from tgsr_amd.trainer import SRPipeline
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 256
    cfg.TREE.BRANCH_NUM = 4
    cfg.TREE.BASE_SIZE = 32
    pipe = SRPipeline(41, device=DEV, low="lr", overlap=True)
    for i, m in enumerate((pipe.netGL, pipe.netGH)):
        random_init_(m, seed=i)
    cap, lens, LR, LRb = synthetic_batch(3, seed=7)
    cap, LR, LRb, lens = cap.to(DEV), LR.to(DEV), LRb.to(DEV), lens.tolist()
    eager = [f.clone() for f in pipe(cap, lens, LR, LRb)["fine"]]
    pipe.capture(cap, lens, LR, LRb)
    out = pipe.replay(cap, lens, LR, LRb)
    for a, b in zip(out["fine"], eager):
        assert torch.equal(a, b)
    _, _, LR2, LRb2 = synthetic_batch(3, seed=8)
    out2 = [f.clone() for f in pipe.replay(None, None, LR2.to(DEV), LRb2.to(DEV))["fine"]]     # captions / lengths unchanged
    with pytest.raises(ValueError):
        pipe.replay(cap, None, LR2.to(DEV), LRb2.to(DEV))                     # new captions without their lengths
    with pytest.raises(ValueError):
        pipe.replay(cap, LR2.to(DEV), LRb2.to(DEV))                           # the old three-argument positional form
    eager2 = pipe(cap, lens, LR2.to(DEV), LRb2.to(DEV))["fine"]
    for a, b in zip(out2, eager2):
        assert torch.equal(a, b)
    cfg_reset()


def test_captured_step_is_independent_of_caption_lengths():
    """Q7 (util.py:250-253, trainer_objective.py:136-140): T_max and the mask follow each batch's captions.  ONE capture,
    then six batches with six different length vectors (new captions, lengths, images) through the same hipGraph: every
    output in the reference's T_max-sized shape, within the stated fp32 tolerance of the CPU oracle on that batch
    (which runs the reference's per-batch T_max formulation, mask quirk Q1 included) and bit-equal to the eager step."""
    from oracle import tgsr_oracle as O
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.trainer import SRPipeline
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 256
    cfg.TREE.BRANCH_NUM = 4
    cfg.TREE.BASE_SIZE = 32
    B = 3
    sdE, sdL, sdH = O.random_state(seed=3)
    pipe = SRPipeline(41, device=DEV, branch_num=4).load_state_dicts(sdE, sdL, sdH)
    cap, lens, LR, LRb = O.synthetic_batch(B, lr=16)
    pipe.capture(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    seen = set()
    for seed, forced in ((21, None), (22, None), (23, None), (24, [18, 18, 18]), (25, [18, 2, 1]), (26, [1, 1, 1])):
        cap, lens, LR, LRb = O.synthetic_batch(B, seed=seed, lr=16)
        if forced is not None:
            lens = torch.tensor(forced)
            cap = torch.zeros(B, 18, dtype=torch.int64)
            for i, n in enumerate(forced):
                cap[i, :n] = torch.randint(1, 41, (n,), generator=torch.Generator().manual_seed(seed * 7 + i))
        seen.add(tuple(lens.tolist()))
        a = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
        got = pipe.replay(*a)
        torch.cuda.synchronize()
        got = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in got.items()}
        eager = pipe(*a)
        ref = O.sr_forward(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)
        T = max(lens.tolist())
        assert got["words_emb"].shape == (B, 256, T) and got["mask"].shape == (B, T)
        np.testing.assert_allclose(got["words_emb"].cpu().numpy(), ref["words_emb"].numpy(), atol=1e-5, rtol=1e-5)
        for i in range(3):
            assert got["att"][i].shape == ref["att"][i].shape
            np.testing.assert_allclose(got["att"][i].cpu().numpy(), ref["att"][i].numpy(), atol=1e-4, rtol=1e-4)
            np.testing.assert_allclose(got["fake"][i].cpu().numpy(), ref["fake"][i].numpy(), atol=1e-4, rtol=1e-4)
            np.testing.assert_allclose(got["fine"][i].cpu().numpy(), ref["fine"][i].numpy(), atol=1e-4, rtol=1e-4)
            assert torch.equal(got["fine"][i], eager["fine"][i]) and torch.equal(got["att"][i], eager["att"][i])
        assert torch.equal(got["words_emb"], eager["words_emb"]) and torch.equal(got["mask"], eager["mask"])
    assert len(seen) == 6
    cfg_reset()


def test_fp32_batch16_graph_lanes_run_the_f4_kernels_and_equal_eager(face_weights, cfg_face, monkeypatch):
    """The default bench configuration in small: batch 16 (the size at which ops.wino4_wanted / upwino4_wanted route the 128^2 and
    64^2 layers and the upBlocks to the F(4x4) kernels - register-fed fragments, counted vmcnt waits), two batches as parallel
    lanes of ONE hipGraph: every lane bit-identical to its eager step, on replays with new inputs too."""
    from tgsr_amd import custom_ops as C
    from tgsr_amd.trainer import GraphedStep
    seen = set()
    for name in ("conv3x3_wino4w", "conv3x3_wino4w_out", "upwino4_glu", "upwino4_glu_out"):
        real = getattr(C, name)
        monkeypatch.setattr(C, name, (lambda real, name: lambda *a: (seen.add(name.replace("_out", "")), real(*a))[1])(real, name))
    p = _pipeline(face_weights)
    B = 16
    batches = [O.synthetic_batch(B, seed=40 + k) for k in range(4)]
    dev = [(b[0].to(DEV), b[1].tolist(), b[2].to(DEV), b[3].to(DEV)) for b in batches]
    eager = [[t.clone() for t in p(*d)["fine"]] for d in dev]
    assert seen == {"conv3x3_wino4w", "upwino4_glu"}, seen
    step = GraphedStep(p, *dev[0], lanes=2)
    for first in (0, 2):
        out = step.replay([dev[first][0], dev[first + 1][0]], [dev[first][1], dev[first + 1][1]],
                          [dev[first][2], dev[first + 1][2]], [dev[first][3], dev[first + 1][3]])
        torch.cuda.synchronize()
        for k in range(2):
            for i in range(3):
                assert torch.equal(out[k]["fine"][i], eager[first + k][i]), (first, k, i)


@pytest.mark.parametrize("shape", [(2, 3, 256, 256), (1, 3, 7, 5), (5,)])
def test_to_uint8_bytes_match_numpy(shape):
    """trainer_objective.py:153-155 on the device: identical bytes, including .5 ties, clipping and out-of-range."""
    import numpy as np
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(shape, generator=g) * 1.2
    flat = x.view(-1)
    ties = (torch.arange(0, 256, dtype=torch.float32) + 0.5) / 127.5 - 1.0      # lands near k + 0.5
    flat[: min(flat.numel(), ties.numel())] = ties[: flat.numel()]
    a = x.numpy()
    ref = np.round(np.maximum(0, np.minimum(255, (a + 1.0) * 127.5))).astype(np.uint8)
    out = ops.to_uint8(x.to(DEV)).cpu().numpy()
    assert out.dtype == np.uint8 and out.shape == ref.shape
    assert np.array_equal(out, ref)


def test_config4_lr64_input_x4_readout(face_weights, cfg_face):
    """BASELINE config 4 as SURVEY 8 reads it (there is no x4 generator in the reference): the same fully convolutional
    networks on a 64x64 LR input, the 64->256 result is read at index 1 (h_net3 still runs, to 512^2)."""
    cap, lens, LR, LRb = O.synthetic_batch(2, lr=64)
    ref = O.sr_forward(split_sd(face_weights, "E."), split_sd(face_weights, "GL."), split_sd(face_weights, "GH."),
                       cap, lens.tolist(), LR, LRb)
    p = _pipeline(face_weights)
    r = p(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    assert tuple(r["fine"][1].shape) == (2, 3, 256, 256) and tuple(r["fine"][2].shape) == (2, 3, 512, 512)
    for i in range(3):
        close(r["fake"][i], ref["fake"][i], atol=ATOL256 if i == 2 else ATOL)
        close(r["fine"][i], ref["fine"][i], atol=ATOL256 if i == 2 else ATOL)
        close(r["att"][i], ref["att"][i], atol=2e-5)


def test_pipeline_stream_lanes_match_single_stream(nets_small, cfg_small):
    """Consecutive independent batches alternating over three stream lanes (what bench.py does) give exactly the
    results of running them one at a time."""
    from tgsr_amd.synthetic import synthetic_batch
    p = _pipeline(nets_small)
    batches = []
    for k in range(6):
        cap, lens, LR, LRb = synthetic_batch(3, seed=20 + k, lr=16)
        batches.append((cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV)))
    ref = [[f.clone() for f in p(*b)["fine"]] for b in batches]
    torch.cuda.synchronize()
    lanes = p.lanes(3)
    outs = []
    for k, b in enumerate(batches):
        with torch.cuda.stream(lanes[k % 3]):
            outs.append(p(*b)["fine"])
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        for a, b_ in zip(o, r):
            assert torch.equal(a, b_)


def test_winograd_kernels_random_shape_sweep():
    """Seeded sweep over ragged shapes (every Cin % 4, Cout % 32 / % 64 class, H from 1, W % 4 == 0) of both Winograd
    kernels against the direct convolution - tile-edge, single-stage and many-stage cases the layer shapes never hit."""
    from tgsr_amd import ops
    rng = np.random.RandomState(1234)
    for it in range(24):
        B = int(rng.randint(1, 4))
        Cin = 4 * int(rng.randint(1, 10))
        H = int(rng.randint(1, 23))
        W = 4 * int(rng.randint(1, 13))
        glu = bool(rng.randint(0, 2))
        Cout = int(rng.choice([32, 64, 96, 128])) if not glu else int(rng.choice([32, 64, 128]))
        res = (not glu) and bool(rng.randint(0, 2))
        g = torch.Generator().manual_seed(1000 + it)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
        scale, shift = 0.5 + torch.rand(Cout, generator=g), 0.3 * torch.randn(Cout, generator=g)
        co = Cout // 2 if glu else Cout
        r = torch.randn(B, co, H, W, generator=g) if res else None
        ref = F.conv2d(x, w, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None]
        ref = O.glu(ref) if glu else ref
        ref = ref + r if res else ref
        out = ops.conv3x3_wino(x.to(DEV), ops.pack_wino_weight(w.to(DEV), glu=glu), Cout, scale.to(DEV), shift.to(DEV),
                               glu=glu, residual=None if r is None else r.to(DEV))
        close(out, ref, atol=3e-5, rtol=3e-5)
        if Cout % 64 == 0:            # the up-sample-aware form on the same operands
            xi = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
            uref = F.conv2d(xi, w, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None]
            up = ops.upwino_glu(x.to(DEV), ops.pack_upwino_weight(w.to(DEV), glu=glu), Cout, scale.to(DEV),
                                shift.to(DEV), glu=glu)
            close(up, O.glu(uref) if glu else uref, atol=3e-5, rtol=3e-5)


def test_ca_net_fused_inference_kernel():
    """CA_NET.forward in eval mode under no_grad is one HIP launch (tgsr_ca_net_fwd): same mu / logvar / c_code as the
    torch formulation of util.py:372-400 (which training keeps), drawing the same normals from torch's generator."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd import util
    cfg_reset()
    try:
        torch.manual_seed(5)
        ca = util.CA_NET().to(DEV)
        x = torch.randn(7, cfg.TEXT.EMBEDDING_DIM, device=DEV)
        torch.manual_seed(11)
        with torch.enable_grad():
            c_ref, mu_ref, lv_ref = ca.eval()(x)                 # grad enabled -> the torch path
        torch.manual_seed(11)
        with torch.no_grad():
            c, mu, lv = ca(x)
        close(mu, mu_ref, atol=2e-6)
        close(lv, lv_ref, atol=2e-6)
        close(c, c_ref, atol=5e-6)
        assert mu.shape == (7, cfg.GAN.CONDITION_DIM) and c.shape == mu.shape
    finally:
        cfg_reset()


@pytest.mark.parametrize("B,T_,tdim,ncf,nsets", [(16, 18, 256, 100, 3), (3, 5, 64, 6, 2), (20, 32, 48, 9, 1), (1, 1, 16, 1, 4)])
def test_text_tail_equals_its_parts(B, T_, tdim, ncf, nsets):
    """tgsr_text_tail_fwd = word_project + CA_NET (mu, logvar) + the caption mask in one launch: bit-identical to the
    stand-alone launches (same device code), and CA_NET's MFMA form agrees with the torch formulation of util.py:372-400
    for every K split (tdim % 64 == 0: float4 operands; % 16: scalar) and ragged sample / channel blocks."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B * 100 + T_)
    cdf, idf = 40, 32
    words = torch.randn(B, cdf, T_, generator=g).to(DEV)
    ws = [torch.randn(idf, cdf, generator=g).to(DEV) for _ in range(nsets)]
    sent = torch.randn(B, tdim, generator=g).to(DEV)
    w, b = (torch.randn(4 * ncf, tdim, generator=g) * 0.1).to(DEV), torch.randn(4 * ncf, generator=g).to(DEV)
    cap = torch.randint(0, 4, (B, T_ + 3), generator=g).to(DEV)
    src, mu, lv, m8 = ops.text_tail(words, ws, sent, w, b, ncf, cap)
    ref = ops.word_project(words, ws)
    for k in range(nsets):
        assert torch.equal(src[k], ref[k])
    _, mu1, lv1 = ops.ca_net(sent, w, b, ncf, None)
    assert torch.equal(mu, mu1) and torch.equal(lv, lv1)
    y = torch.nn.functional.linear(sent.double().cpu(), w.double().cpu(), b.double().cpu())
    y = y[:, :2 * ncf] * torch.sigmoid(y[:, 2 * ncf:])
    close(mu.cpu().double(), y[:, :ncf], atol=1e-5, rtol=1e-5)
    close(lv.cpu().double(), y[:, ncf:], atol=1e-5, rtol=1e-5)
    assert m8.dtype == torch.uint8 and torch.equal(m8.view(torch.bool), cap[:, :T_] == 0)
    eps = torch.randn(B, ncf, generator=g).to(DEV)
    c, mu2, lv2 = ops.ca_net(sent, w, b, ncf, eps)
    assert torch.equal(mu2, mu1)
    close(c, eps * torch.exp(0.5 * lv2) + mu2, atol=1e-5, rtol=1e-5)


def test_multi_copy_one_launch():
    """tgsr_multi_copy: dense buffers of different types and sizes (16-byte and 4-byte paths, one segment per blockIdx.y)."""
    from tgsr_amd import ops
    from tgsr_amd._lib import TgsrError
    g = torch.Generator().manual_seed(1)
    srcs = [torch.randn(16, 3, 32, 32, generator=g).to(DEV), torch.randint(0, 99, (16, 18), generator=g).to(DEV),
            torch.randn(1000003, generator=g).to(DEV)[1:8], torch.randn(5, generator=g).to(DEV),
            torch.randn(300001, generator=g).to(DEV)[1:]]
    dsts = [torch.zeros_like(s_) for s_ in srcs]
    dsts[4] = torch.zeros(300004, device=DEV)[3:-1]
    ops.multi_copy(dsts, srcs)
    for d, s_ in zip(dsts, srcs):
        assert torch.equal(d, s_)
    with pytest.raises(TgsrError):
        ops.multi_copy([torch.zeros(3, device=DEV)], [torch.zeros(4, device=DEV)])


def test_damsm_kernels_are_bitwise_reproducible():
    """No float atomics anywhere in the library: the DAMSM similarity and its backward give bit-identical results run
    after run (the four waves' partial sums of a workgroup meet in a fixed order)."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(4)
    B, ndf, T_ = 6, 256, 12
    feats = torch.randn(B, ndf, 17, 17, generator=g).to(DEV)
    words = torch.randn(B, ndf, T_, generator=g).to(DEV)
    lens = [12, 11, 9, 7, 5, 3]
    gs = torch.randn(B, B, generator=g).to(DEV)
    sims = [ops.damsm_words_similarity(feats, words, lens, 4.0, 5.0)[0].clone() for _ in range(3)]
    grads = [tuple(t.clone() for t in ops.damsm_words_bwd(feats, words, lens, 4.0, 5.0, gs)) for _ in range(3)]
    for s in sims[1:]:
        assert torch.equal(s, sims[0])
    for gi, gw in grads[1:]:
        assert torch.equal(gi, grads[0][0]) and torch.equal(gw, grads[0][1])


def test_config3_birds_vocab_5450_batch4_vs_oracle(cfg_face):
    """BASELINE configs[3] (eval_birdSR_attn2.yml: BRANCH_NUM 4, BASE_SIZE 32, GF_DIM 32, EMBEDDING_DIM 256): the CUB
    vocabulary (n_words 5450, datasets.py:725) at the per-GPU batch of B=32 over 8 GPUs (4), full size, seeded random
    weights of that architecture, against the CPU oracle at the stated 1e-4.  Exercises the 5450-row per-token gate table
    of the text encoder (util.RNN_ENCODER -> ops.lstm_gate_table) and the small-batch kernel selection."""
    from tgsr_amd.trainer import SRPipeline
    n_words = 5450
    sdE, sdL, sdH = O.random_state(n_words=n_words, seed=11)
    cap, lens, LR, LRb = O.synthetic_batch(4, n_words=n_words, seed=31)
    assert int(cap.max()) > 2000                       # tokens from the far end of the table, not just its head
    ref = O.sr_forward(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)
    p = SRPipeline(n_words, device=DEV).load_state_dicts(sdE, sdL, sdH)
    assert tuple(p.text_encoder.encoder.weight.shape) == (n_words, 300)
    r = p(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    assert tuple(p.text_encoder._table.shape) == (n_words, 2, 512)
    close(r["words_emb"], ref["words_emb"], atol=1e-5)
    close(r["sent_emb"], ref["sent_emb"], atol=1e-5)
    for i in range(3):
        close(r["fake"][i], ref["fake"][i], atol=ATOL256 if i == 2 else ATOL)
        close(r["fine"][i], ref["fine"][i], atol=ATOL256 if i == 2 else ATOL)
        close(r["att"][i], ref["att"][i], atol=2e-5)
    # every token of the vocabulary: the table against the oracle's own input projection on a batch that walks the table
    ids = torch.arange(1, 1 + 302 * 18).reshape(302, 18)[::19][:16].contiguous()        # 16 captions of 18 tokens, ids up to 5149
    words, sent = O.rnn_encoder(sdE, ids, [18] * ids.shape[0])
    w2, s2 = p.text_encoder(ids.to(DEV), [18] * ids.shape[0])
    close(w2, words, atol=1e-5)
    close(s2, sent, atol=1e-5)


class _StubInception(torch.nn.Module):
    """An object with the sixteen Inception-v3 block attributes CNN_ENCODER.define_module copies (util.py:282-298), each a
    small conv with the stride / padding that gives the real blocks' spatial sizes (299 -> 149 -> 147 -> 147 | 73 -> 73 ->
    71 | 35 ... -> 17 x 17 x 768 ... -> 8 x 8 x 2048).  torchvision is absent here and its arithmetic is third-party: the
    stub is what lets the WALK (util.py:308-362) execute."""

    def __init__(self):
        super().__init__()
        c = torch.nn.Conv2d
        self.Conv2d_1a_3x3, self.Conv2d_2a_3x3, self.Conv2d_2b_3x3 = c(3, 4, 3, 2), c(4, 4, 3), c(4, 6, 3, 1, 1)
        self.Conv2d_3b_1x1, self.Conv2d_4a_3x3 = c(6, 8, 1), c(8, 12, 3)
        self.Mixed_5b, self.Mixed_5c, self.Mixed_5d = c(12, 16, 1), c(16, 16, 1), c(16, 16, 1)
        self.Mixed_6a = c(16, 768, 3, 2)
        self.Mixed_6b, self.Mixed_6c, self.Mixed_6d, self.Mixed_6e = (c(768, 768, 1, groups=768) for _ in range(4))
        self.Mixed_7a, self.Mixed_7b, self.Mixed_7c = c(768, 64, 3, 2), c(64, 2048, 1), c(2048, 2048, 1, groups=2048)


def test_cnn_encoder_trunk_walk_on_stub_inception(cfg_face):
    """CNN_ENCODER built the reference's way (define_module over an Inception-like object): same state_dict key names,
    frozen trunk, and forward == the reference's walk (util.py:308-368) restated here op by op on the CPU copy."""
    import copy
    from tgsr_amd import util
    torch.manual_seed(5)
    stub = _StubInception()
    m = copy.deepcopy(stub)                                     # CPU copy for the restated walk
    enc = util.CNN_ENCODER(256, inception=stub).to(DEV).eval()
    keys = set(enc.state_dict())
    assert {"Conv2d_1a_3x3.weight", "Mixed_6e.weight", "Mixed_7c.bias", "emb_features.weight", "emb_cnn_code.weight",
            "emb_cnn_code.bias"} <= keys and not any(k.startswith("trunk.") for k in keys)
    assert all(not p.requires_grad for p in enc.frozen_parameters()) and enc.emb_features.weight.requires_grad
    g = torch.Generator().manual_seed(2)
    x = torch.rand(3, 3, 256, 256, generator=g) * 2 - 1
    with torch.no_grad():
        feats, code = enc(x.to(DEV))
        y = torch.nn.Upsample(size=(299, 299), mode='bilinear')(x)              # util.py:311
        y = m.Conv2d_1a_3x3(y); y = m.Conv2d_2a_3x3(y); y = m.Conv2d_2b_3x3(y)  # :313-317
        y = F.max_pool2d(y, kernel_size=3, stride=2)                            # :319
        y = m.Conv2d_3b_1x1(y); y = m.Conv2d_4a_3x3(y)                          # :321-323
        y = F.max_pool2d(y, kernel_size=3, stride=2)                            # :326
        y = m.Mixed_5b(y); y = m.Mixed_5c(y); y = m.Mixed_5d(y)                 # :328-332
        y = m.Mixed_6a(y); y = m.Mixed_6b(y); y = m.Mixed_6c(y); y = m.Mixed_6d(y); y = m.Mixed_6e(y)   # :335-343
        features = y                                                            # :347
        y = m.Mixed_7a(y); y = m.Mixed_7b(y); y = m.Mixed_7c(y)                 # :350-354
        y = F.avg_pool2d(y, kernel_size=8).view(y.size(0), -1)                  # :356-360
        ref_code = F.linear(y, enc.emb_cnn_code.weight.cpu(), enc.emb_cnn_code.bias.cpu())     # :363
        ref_feats = F.conv2d(features, enc.emb_features.weight.cpu())            # :366
    assert tuple(feats.shape) == (3, 256, 17, 17) and tuple(code.shape) == (3, 256)
    close(feats, ref_feats, atol=2e-4, rtol=1e-3)
    close(code, ref_code, atol=2e-4, rtol=1e-3)
    with pytest.raises(ImportError):
        util.CNN_ENCODER(256)                                   # no torchvision here: a loud error, not a silent stand-in


@pytest.mark.parametrize("shape", [(4, 400), (2, 64, 9, 7), (3, 6, 5)])
def test_glu_standalone_module_forward_backward(shape):
    """util.GLU called on its own (the reference's CA_NET does, util.py:381): x[:, :C/2] * sigmoid(x[:, C/2:]) and its
    gradient, vs torch."""
    from tgsr_amd import util
    g = torch.Generator().manual_seed(len(shape))
    x = torch.randn(shape, generator=g, requires_grad=True)
    nc = shape[1] // 2
    ref = x[:, :nc] * torch.sigmoid(x[:, nc:])
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = x.detach().to(DEV).requires_grad_(True)
    out = util.GLU()(xd)
    out.backward(dy.to(DEV))
    close(out, ref, atol=1e-6, rtol=1e-6)
    close(xd.grad, x.grad, atol=1e-6, rtol=1e-5)
    with pytest.raises(AssertionError):
        util.GLU()(torch.zeros(2, 3, device=DEV))
