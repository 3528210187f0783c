"""GPU parity of the reduced-precision (bf16 / f16) inference path against its CPU model (oracle/tgsr_oracle_lp.py,
same rounding points, fp32 accumulation) and against the fp32 oracle (PSNR bounds stated per dtype).

Elementwise tolerance of an lp kernel against the CPU model: both round the same fp32 value to `dtype`, but the fp32
sums differ in order, so a stored element may land on the neighbouring representable value: |a - b| <= ULP * |b| + tiny,
ULP = 2^-7 (bf16: 8 significant bits) / 2^-10 (f16: 11 bits).
"""
import numpy as np
import pytest
import torch

from oracle import tgsr_oracle as O
from oracle import tgsr_oracle_lp as OL

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [("bf16", torch.bfloat16, 2.0 ** -7), ("f16", torch.float16, 2.0 ** -10)]


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from tgsr_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()


def lp_close(a, b, ulp, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    err = (a - b).abs()
    tol = 1.01 * ulp * b.abs() + 1e-5       # + fp32 summation-order noise near zero
    bad = err > tol
    assert not bool(bad.any()), "%s: %d of %d elements off by more than one %g-ulp (worst %g at value %g)" % (
        what, int(bad.sum()), bad.numel(), ulp, float(err.max()), float(b.flatten()[err.argmax()]))


def test_lp_image_roundtrip():
    from tgsr_amd import lp
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 8, 32, generator=g)
    for name, td, _ in DTYPES:
        img = lp.from_nchw(x.to(DEV), name, cpitch=8, coff=2)
        assert img.shape == (3, 10, 34, 8) and img.dtype == td
        back = lp.to_nchw(img, 5, 2).cpu()
        assert torch.equal(back, OL.rnd(x, td))
        assert float(img[:, 0].abs().max()) == 0 and float(img[:, :, 0].abs().max()) == 0    # border untouched
        assert float(img[..., :2].abs().max()) == 0 and float(img[..., 7:].abs().max()) == 0


CONV_CASES = [
    # B, Cin, Cout, H, W (output), glu, upsample, residual
    (2, 64, 128, 8, 32, True, False, False),
    (1, 64, 64, 16, 32, False, False, True),
    (2, 64, 64, 8, 64, True, True, False),
    (1, 32, 64, 12, 32, True, False, False),
    (2, 32, 32, 8, 32, False, False, True),
    (1, 32, 32, 4, 64, False, False, False),
    (2, 32, 64, 16, 64, True, True, False),
    (3, 64, 128, 32, 64, True, False, False),      # 8-row tiles only from 512 workgroups on; still the 4-row kernel
    (33, 64, 128, 32, 128, True, False, False),    # 33*4*4 = 528 >= 512 tiles of 8 rows: the 8-row kernel
    (33, 32, 32, 32, 128, False, False, True),
    (17, 64, 64, 64, 128, True, True, False),
]


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_lp_conv3x3(case, name, td, ulp):
    from tgsr_amd import lp, ops
    B, Cin, Cout, H, W, glu, up, res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    Hi, Wi = (H // 2, W // 2) if up else (H, W)
    x = OL.rnd(torch.randn(B, Cin, Hi, Wi, generator=g), td)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    scale = 1 + 0.1 * torch.randn(Cout, generator=g)
    shift = 0.1 * torch.randn(Cout, generator=g)
    co = Cout // 2 if glu else Cout
    r = OL.rnd(torch.randn(B, co, H, W, generator=g), td) if res else None
    ref = OL.conv_block(x, w, scale, shift, td, glu=glu, upsample=up, residual=r)
    # input in a wider buffer (cpitch 64 when Cin 32 would also work: the kernel reads channels [0, Cin)); output into a
    # channel slice of a wider image, residual read from a channel offset
    xi = lp.from_nchw(x.to(DEV), name, cpitch=Cin)
    ri = lp.from_nchw(r.to(DEV), name, cpitch=co + 8, coff=8) if res else None
    wp = lp.pack_conv3x3_weight(w.to(DEV), name)
    out = lp.new_image(B, H, W, co + 32, name, DEV)
    lp.conv3x3(xi, wp, Cin, Cout, scale.to(DEV), shift.to(DEV), glu=glu, upsample=up, residual=ri, res_coff=8,
               out=out, out_coff=32)
    torch.cuda.synchronize()
    got = lp.to_nchw(out, co, 32)
    lp_close(got, ref, ulp, "lp conv %s %s" % (name, case))
    assert float(out[..., :32].abs().max()) == 0, "channels outside the slice were written"
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, -1].abs().max()) == 0, "border written"
    assert float(out[:, :, 0].abs().max()) == 0 and float(out[:, :, -1].abs().max()) == 0, "border written"


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("B,Cin,Hi,Wi", [(2, 64, 4, 32), (1, 32, 8, 64), (3, 64, 16, 32), (17, 32, 32, 32), (9, 64, 32, 64)])
def test_lp_upconv_subpixel(B, Cin, Hi, Wi, name, td, ulp):
    """upBlock by sub-pixel decomposition against the CPU model of the same pre-summed, once-rounded taps, and (loosely)
    against the direct form it is mathematically equal to."""
    from tgsr_amd import lp
    g = torch.Generator().manual_seed(B * 7 + Cin)
    x = OL.rnd(torch.randn(B, Cin, Hi, Wi, generator=g), td)
    w = torch.randn(64, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    scale, shift = 1 + 0.1 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    ref = OL.conv_block(x, w, scale, shift, td, glu=True, upsample=True, subpixel=True)
    xi = lp.from_nchw(x.to(DEV), name, cpitch=Cin)
    out = lp.new_image(B, 2 * Hi, 2 * Wi, 64, name, DEV)
    lp.upconv_glu(xi, lp.pack_upconv_weight(w.to(DEV), name), Cin, 64, scale.to(DEV), shift.to(DEV), out=out, out_coff=0)
    got = lp.to_nchw(out, 32, 0)
    lp_close(got, ref, ulp, "lp upconv %s" % name)
    assert float(out[..., 32:].abs().max()) == 0 and float(out[:, 0].abs().max()) == 0 and float(out[:, :, -1].abs().max()) == 0
    direct = OL.conv_block(x, w, scale, shift, td, glu=True, upsample=True)
    assert OL.psnr(got.cpu(), direct, peak=float(direct.abs().max())) > (40 if name == "bf16" else 58)


@pytest.mark.parametrize("name,td,ulp", DTYPES)
def test_lp_stem(name, td, ulp):
    from tgsr_amd import lp
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 3, 32, 12, 32
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    w = torch.randn(2 * C, 3, 3, 3, generator=g) / 5.0
    scale, shift = 1 + 0.1 * torch.randn(2 * C, generator=g), 0.1 * torch.randn(2 * C, generator=g)
    ref = OL.conv_block(x, w, scale, shift, td, glu=True, round_w=False)
    out = lp.new_image(B, H, W, 64, name, DEV)
    lp.stem(x.to(DEV), w.to(DEV), scale.to(DEV), shift.to(DEV), out=out, out_coff=0)
    lp_close(lp.to_nchw(out, C, 0), ref, ulp, "lp stem " + name)
    assert float(out[..., 32:].abs().max()) == 0 and float(out[:, 0].abs().max()) == 0


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("K,act,B,H,W,cp", [(3, False, 2, 8, 32, 64), (5, True, 3, 16, 64, 32), (5, True, 1, 8, 32, 32),
                                            (3, False, 17, 64, 64, 32)])
def test_lp_conv_to3(K, act, B, H, W, cp, name, td, ulp):
    import torch.nn.functional as F
    from tgsr_amd import lp
    g = torch.Generator().manual_seed(K * 100 + H)
    x = OL.rnd(torch.randn(B, 32, H, W, generator=g), td)
    w = torch.randn(3, 32, K, K, generator=g) / (K * 32 ** 0.5)
    add = torch.randn(B, 3, H, W, generator=g) if act else None
    ref = F.conv2d(x, OL.rnd(w, td), None, 1, K // 2)
    if act:
        ref = torch.tanh(ref) + 0.5 * add
    xi = lp.from_nchw(x.to(DEV), name, cpitch=cp)
    got = lp.conv_to3(xi, lp.pack_to3_weight(w.to(DEV), name), K, tanh_axpy=act, addend=None if add is None else add.to(DEV),
                      alpha=0.5)
    # fp32 outputs of a K*K*32-term sum of low-precision products accumulated in fp32: only summation order differs
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("B,H,W,T,correct", [(3, 8, 32, 7, False), (2, 32, 32, 18, False), (5, 16, 64, 12, True), (1, 4, 32, 1, False)])
def test_lp_word_attention(B, H, W, T, correct, name, td, ulp):
    from tgsr_amd import lp, ops
    g = torch.Generator().manual_seed(B * 10 + T)
    h = OL.rnd(torch.randn(B, 32, H, W, generator=g), td)
    words = torch.randn(B, 48, T, generator=g)
    wctx = torch.randn(32, 48, 1, 1, generator=g) / 48 ** 0.5
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    mask = torch.arange(T)[None, :] >= lens[:, None]
    c_ref, a_ref = OL.word_attention(h, words, wctx, mask, td, correct_mask=correct)
    img = lp.from_nchw(h.to(DEV), name, cpitch=64)
    src = ops.word_project(words.to(DEV), [wctx.to(DEV)])[0]
    attn = lp.word_attention(img, src, mask.to(DEV), T, correct_mask=correct)
    # the projected words are rounded to `dtype` on both sides, from fp32 values that differ in their last bits (MFMA
    # vs CPU summation order): an element on a rounding boundary lands one ulp apart and moves a score by
    # ulp * |src| * |h| - a few ulp relative on the softmax
    np.testing.assert_allclose(attn.cpu().numpy(), a_ref.numpy(), atol=2 * ulp, rtol=8 * ulp)
    # c_code = sum_t src_T * P_T: P is rounded to `dtype` before the product, so a P that straddles a rounding boundary
    # moves c by up to ulp * |src|: bound by 2 ulp of the largest |src| of the sample + one ulp of the value itself
    got = lp.to_nchw(img, 32, 32).cpu()
    smax = float(torch.einsum("ic,bct->bit", wctx.reshape(32, 48), words).abs().max())
    assert float((got - c_ref).abs().max()) <= 2 * ulp * smax + 1e-5
    assert torch.equal(lp.to_nchw(img, 32, 0).cpu(), h), "h channels must be untouched"


def _attention_inputs(B, T, seed, nsets=2, ca_dim=64, ncf=10):
    """words, two conv_context weights, a sentence code + CA_NET's Linear, zero-padded captions of random lengths."""
    g = torch.Generator().manual_seed(seed)
    words = torch.randn(B, 48, T, generator=g)
    ws = [torch.randn(32, 48, 1, 1, generator=g) / 48 ** 0.5 for _ in range(nsets)]
    sent, caw, cab = torch.randn(B, ca_dim, generator=g), torch.randn(4 * ncf, ca_dim, generator=g) / 8, torch.randn(4 * ncf, generator=g)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    cap = torch.zeros(B, T + 2, dtype=torch.int64)
    for b in range(B):
        cap[b, :int(lens[b])] = torch.randint(1, 40, (int(lens[b]),), generator=g)
    return words, ws, sent, caw, cab, ncf, cap


@pytest.mark.parametrize("name,td,ulp", DTYPES)
def test_text_tail_lp_pack(name, td, ulp):
    """tgsr_text_tail_lp_fwd: the four outputs of tgsr_text_tail_fwd bit for bit + `att_pack` = the projections rounded once
    to the storage type in the MFMA fragment order the attention consumes (fragments 0, 1: row = word, k = channel;
    2, 3: row = channel, k = word in the accumulator's order) + the packed mask rows."""
    from tgsr_amd import ops
    B, T = 5, 11
    words, ws, sent, caw, cab, ncf, cap = _attention_inputs(B, T, 3)
    d = lambda t: t.to(DEV)                                                                     # noqa: E731
    base = ops.text_tail(d(words), [d(w) for w in ws], d(sent), d(caw), d(cab), ncf, d(cap))
    got = ops.text_tail(d(words), [d(w) for w in ws], d(sent), d(caw), d(cab), ncf, d(cap), lp_dtype=td)
    for a, b in zip(base, got[:4]):
        assert torch.equal(a, b)
    pack = got[4].cpu()
    src = got[0].cpu()                                                                           # [nsets, B, 32, 32]
    frag = pack[:2 * B * 4096].view(td).reshape(2, B, 4, 64, 8).float()
    q = src.to(td).float()
    f, l, j = torch.meshgrid(torch.arange(4), torch.arange(64), torch.arange(8), indexing="ij")
    lr, lh = l % 32, l // 32
    i1, t1 = 16 * f + 8 * lh + j, lr                                                             # fragments 0, 1 (f < 2)
    i2, t2 = lr, 16 * (f - 2) + 8 * (j // 4) + 4 * lh + (j % 4)                                  # fragments 2, 3
    ii, tt = torch.where(f < 2, i1, i2), torch.where(f < 2, t1, t2)
    assert torch.equal(frag, q[:, :, ii, tt])
    bits = pack[2 * B * 4096:].view(torch.int32)
    want = ((cap[:, :T] == 0).long() << torch.arange(T)).sum(1).to(torch.int32)
    assert torch.equal(bits, want)


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("B,H,W,T,correct,use_mask", [(3, 8, 32, 7, False, True), (2, 32, 32, 18, False, True),
                                                      (5, 4, 64, 12, True, True), (16, 32, 32, 14, False, True),
                                                      (2, 8, 32, 5, False, False)])
def test_lp_stem_with_fused_attention(B, H, W, T, correct, use_mask, name, td, ulp):
    """tgsr_lp_stem_att_fwd == tgsr_lp_stem_fwd followed by tgsr_lp_word_attention_fwd, bit for bit (h, c_code, attention
    maps), incl. the reference's mask quirk (row b*Q + q is masked with mask[(b*Q + q) % B]) at B = 16."""
    from tgsr_amd import lp, ops
    g = torch.Generator().manual_seed(B + T)
    x = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).to(DEV)
    w = (torch.randn(64, 3, 3, 3, generator=g) / 5.0).to(DEV)
    scale, shift = (1 + 0.1 * torch.randn(64, generator=g)).to(DEV), (0.1 * torch.randn(64, generator=g)).to(DEV)
    words, ws, sent, caw, cab, ncf, cap = _attention_inputs(B, T, 7 + T)
    d = lambda t: t.to(DEV)                                                                     # noqa: E731
    src, _mu, _lv, m8, pack = ops.text_tail(d(words), [d(v) for v in ws], d(sent), d(caw), d(cab), ncf, d(cap), lp_dtype=td)
    mask = m8.view(torch.bool) if use_mask else None
    ref = lp.new_image(B, H, W, 64, name, DEV)
    lp.stem(x, w, scale, shift, out=ref, out_coff=0)
    a_ref = lp.word_attention(ref, src[1].contiguous(), mask, T, correct_mask=correct)
    out = lp.new_image(B, H, W, 64, name, DEV)
    attn = torch.full((B, T, H, W), -1.0, device=DEV)
    lp.stem(x, w, scale, shift, out=out, out_coff=0, att=lp.AttFuse(pack, 2, 1, T, use_mask, correct, 32, attn))
    assert torch.equal(out, ref), "h / c_code differ from stem + stand-alone attention"
    assert torch.equal(attn, a_ref)
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, :, 0].abs().max()) == 0          # the border stays zero


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("head", [True, False])
@pytest.mark.parametrize("B,Hi,Wi,T,correct", [(2, 4, 32, 9, False), (3, 16, 32, 18, False), (16, 32, 32, 13, False),
                                               (2, 8, 64, 6, True)])
def test_lp_upconv_with_fused_attention(B, Hi, Wi, T, correct, head, name, td, ulp):
    """tgsr_lp_upconv_glu_att_fwd (with and without the fused 3x3 head) == the upBlock followed by the stand-alone attention
    on its output, bit for bit: feature channels, c_code channels, attention maps, head partial sums."""
    from tgsr_amd import lp, ops
    g = torch.Generator().manual_seed(B * 7 + T)
    x = lp.from_nchw(torch.randn(B, 64, Hi, Wi, generator=g).to(DEV), name, cpitch=64)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(DEV)
    scale, shift = (1 + 0.1 * torch.randn(64, generator=g)).to(DEV), (0.1 * torch.randn(64, generator=g)).to(DEV)
    wsub = lp.pack_upconv_weight(w, name)
    hw = lp.pack_to3_weight((torch.randn(3, 32, 3, 3, generator=g) / 17.0).to(DEV), name)
    words, ws, sent, caw, cab, ncf, cap = _attention_inputs(B, T, 11 + T)
    d = lambda t: t.to(DEV)                                                                     # noqa: E731
    src, _mu, _lv, m8, pack = ops.text_tail(d(words), [d(v) for v in ws], d(sent), d(caw), d(cab), ncf, d(cap), lp_dtype=td)
    mask = m8.view(torch.bool)
    Ho, Wo = 2 * Hi, 2 * Wi
    ref = lp.new_image(B, Ho, Wo, 64, name, DEV)
    out = lp.new_image(B, Ho, Wo, 64, name, DEV)
    attn = torch.full((B, T, Ho, Wo), -1.0, device=DEV)
    att = lp.AttFuse(pack, 2, 0, T, True, correct, 32, attn)
    if head:
        _, p_ref = lp.upconv_glu_head(x, wsub, 64, 64, scale, shift, hw, 3, out=ref)
        _, p_got = lp.upconv_glu_head(x, wsub, 64, 64, scale, shift, hw, 3, out=out, att=att)
        assert torch.equal(p_got, p_ref)
    else:
        lp.upconv_glu(x, wsub, 64, 64, scale, shift, out=ref)
        lp.upconv_glu(x, wsub, 64, 64, scale, shift, out=out, att=att)
    a_ref = lp.word_attention(ref, src[0].contiguous(), mask, T, correct_mask=correct)
    assert torch.equal(out, ref), "feature / c_code channels differ from upBlock + stand-alone attention"
    assert torch.equal(attn, a_ref)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_lp_pipeline_fused_attention_equals_standalone(name, cfg_face, face_weights):
    """The whole reduced-precision step with the attention of every stage inside the kernel that produces its h
    (TGSR_LP_FUSE_ATT, the default) against the stand-alone attention launches: every output bit-identical, eager and
    replayed from a hipGraph."""
    B = 3
    cap, lens, LR, LRb = O.synthetic_batch(B, seed=31)
    args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    fused = _pipe(cfg_face, face_weights, name)
    plain = _pipe(cfg_face, face_weights, name)
    plain._lp.fuse_attention = False
    assert fused._lp.fuse_attention
    a, b = fused(*args), plain(*args)
    torch.cuda.synchronize()
    for k in ("fake", "fine", "att"):
        for i in range(3):
            assert torch.equal(a[k][i], b[k][i]), "%s[%d]: fused attention differs from the stand-alone launches" % (k, i)
    fused.capture(*args)
    r = fused.replay(*args)
    torch.cuda.synchronize()
    for k in ("fake", "fine", "att"):
        for i in range(3):
            assert torch.equal(r[k][i], b[k][i])


def _pipe(cfg_face, face_weights, dtype, overlap=True):
    from conftest import split_sd
    from tgsr_amd.trainer import SRPipeline
    p = SRPipeline(41, device=DEV, low="lr", overlap=overlap, dtype=dtype)
    return p.load_state_dicts(split_sd(face_weights, "E."), split_sd(face_weights, "GL."), split_sd(face_weights, "GH."))


@pytest.fixture()
def cfg_face():
    from tgsr_amd.miscc.config import cfg, cfg_reset
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 256
    cfg.TREE.BRANCH_NUM = 4                  # cfg/eval_*SR_attn2.yml: the x8 generators (trainer_objective.py:74-87)
    yield cfg
    cfg_reset()


# PSNR (peak 2.0) of the finest SR image against the fp32 oracle, shipped face checkpoint.  SURVEY.md 8c states >= 50 dB for
# the bf16 configuration.  The CPU model of the same rounding points (oracle/tgsr_oracle_lp.py) gives 53.6-53.8 dB for the
# bf16 configuration as shipped (NetG_highweight's 32^2 trunk with f16 operands, everything else bf16; uniform bf16: 46.9)
# and 64.7-65.5 dB for f16; the kernels must land within 1 dB of the model and above these floors.
PSNR_FLOOR = {"bf16": 50.0, "f16": 60.0}


def test_lp_convert_between_the_two_types():
    """tgsr_lp_convert: f16 image -> bf16 (one round-to-nearest-even per element, = torch's conversion) and back (exact),
    zero border included."""
    from tgsr_amd import lp
    g = torch.Generator().manual_seed(4)
    x = lp.from_nchw((torch.randn(3, 32, 8, 32, generator=g) * 3).to(DEV), torch.float16)
    out = lp.new_image(3, 8, 32, 32, torch.bfloat16, DEV)
    lp.convert(x, out)
    assert torch.equal(out, x.to(torch.bfloat16))
    back = lp.new_image(3, 8, 32, 32, torch.float16, DEV)
    lp.convert(out, back)
    assert torch.equal(back, out.to(torch.float16))
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, :, 0].abs().max()) == 0        # the border stays zero
    with pytest.raises(Exception):
        lp.convert(x, lp.new_image(3, 8, 32, 32, torch.float16, DEV))                            # same type: refused


def test_bf16_configuration_runs_the_32x32_trunk_in_f16(cfg_face, face_weights, monkeypatch):
    """The shipped bf16 configuration (f16 operands in NetG_highweight's 32^2 trunk) against the uniform-bf16 one
    (TGSR_LP_BF16_TRUNK=bf16): each within 1 dB of ITS CPU model, and the shipped one >= 50 dB from the fp32 oracle where
    uniform bf16 stays below (what SURVEY.md 8c's bar is about)."""
    from conftest import split_sd
    from tgsr_amd import lp_pipeline
    sdE, sdL, sdH = (split_sd(face_weights, k) for k in ("E.", "GL.", "GH."))
    cap, lens, LR, LRb = O.synthetic_batch(4)
    with torch.no_grad():
        ref32 = O.sr_forward(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)["fine"][2]
        m_mixed = OL.sr_forward(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb, torch.bfloat16)["fine"][2]
        m_plain = OL.sr_forward(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb, torch.bfloat16, trunk_dtype=torch.bfloat16)["fine"][2]
    args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    mixed = _pipe(cfg_face, face_weights, "bf16")(*args)["fine"][2].cpu()
    monkeypatch.setattr(lp_pipeline, "F16_TRUNK", False)
    plain = _pipe(cfg_face, face_weights, "bf16")(*args)["fine"][2].cpu()
    p_mixed, p_plain = OL.psnr(mixed, ref32), OL.psnr(plain, ref32)
    assert abs(p_mixed - OL.psnr(m_mixed, ref32)) < 1.0 and abs(p_plain - OL.psnr(m_plain, ref32)) < 1.0
    assert p_mixed >= 50.0 and p_plain < 49.0 and p_mixed > p_plain + 4.0, (p_mixed, p_plain)
    assert OL.psnr(mixed, m_mixed) > OL.psnr(m_mixed, ref32) + 3.0


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("B", [2, 16])
def test_lp_pipeline_full_size(B, name, td, ulp, cfg_face, face_weights):
    from conftest import split_sd
    sdE, sdL, sdH = (split_sd(face_weights, k) for k in ("E.", "GL.", "GH."))
    cap, lens, LR, LRb = O.synthetic_batch(B)
    with torch.no_grad():
        ref32 = O.sr_forward(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)
        model = OL.sr_forward(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb, td)
    pipe = _pipe(cfg_face, face_weights, name)
    out = pipe(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    torch.cuda.synchronize()
    for k in ("fake", "fine"):
        for i in range(3):
            got = out[k][i].cpu()
            p32, pm, pmodel = OL.psnr(got, ref32[k][i]), OL.psnr(got, model[k][i]), OL.psnr(model[k][i], ref32[k][i])
            assert abs(p32 - pmodel) < 1.0, "%s %s[%d]: %.2f dB vs fp32, the CPU model predicts %.2f" % (name, k, i, p32, pmodel)
            # closer to the model than the model is to fp32: by 3 dB for f16; by 2 dB for the bf16 configuration, whose f16
            # trunk moved the MODEL 7 dB closer to fp32 while the distance between two implementations of the same roundings
            # stays what the bf16 layers' rounding ties make it (~58 dB)
            assert pm > pmodel + (3.0 if name == "f16" else 2.0), \
                "%s %s[%d]: only %.2f dB against the CPU model of the same roundings" % (name, k, i, pm)
    assert OL.psnr(out["fine"][2].cpu(), ref32["fine"][2]) >= PSNR_FLOOR[name]
    for i in range(3):   # attention maps are the fp32 softmax of lp scores
        assert OL.psnr(out["att"][i].cpu(), model["att"][i], peak=1.0) > 45.0
    np.testing.assert_allclose(out["mu"].cpu().numpy(), ref32["mu"].numpy(), atol=1e-5)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_lp_pipeline_hipgraph_and_determinism(name, cfg_face, face_weights):
    B = 4
    cap, lens, LR, LRb = O.synthetic_batch(B)
    pipe = _pipe(cfg_face, face_weights, name)
    args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    a = pipe(*args)
    b = pipe(*args)
    torch.cuda.synchronize()
    assert torch.equal(a["fine"][2], b["fine"][2]), "two eager runs differ"
    serial = _pipe(cfg_face, face_weights, name, overlap=False)(*args)
    assert torch.equal(a["fine"][2], serial["fine"][2]), "two-stream run differs from the single-stream one"
    for trial in range(4):             # several captures x replays: the two branches of the graph really overlap on replay,
        pipe.capture(*args)            # which is where a kernel that misbehaves beside another one shows (lp_stem_kernel did)
        for rep in range(3):
            g = pipe.replay()
            torch.cuda.synchronize()
            for k in ("fake", "fine", "att"):
                for i in range(3):
                    assert torch.equal(g[k][i], a[k][i]), "hipGraph replay %d/%d differs from eager (%s[%d])" % (trial, rep, k, i)
    # NEW BATCHES through the captured step - other captions, other LENGTHS (Q7: T_max and the mask follow each batch,
    # util.py:250-253, trainer_objective.py:136-140), other images: every output bit-equal to the eager step on that batch,
    # in the reference's T_max-sized shapes
    seen = set()
    for seed in (7, 8, 9, 10, 11, 12):
        cap2, lens2, LR2, LRb2 = O.synthetic_batch(B, seed=seed)
        if seed == 12:                                            # the extremes: one-word captions ... the full width
            lens2 = torch.tensor([18, 9, 2, 1])
            cap2 = torch.zeros(B, 18, dtype=torch.int64)
            for i, n in enumerate(lens2.tolist()):
                cap2[i, :n] = torch.randint(1, 41, (n,), generator=torch.Generator().manual_seed(i))
        seen.add(tuple(lens2.tolist()))
        a2 = (cap2.to(DEV), lens2.tolist(), LR2.to(DEV), LRb2.to(DEV))
        g2 = pipe.replay(*a2)
        torch.cuda.synchronize()
        g2 = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in g2.items()}
        e2 = pipe(*a2)
        torch.cuda.synchronize()
        T = max(lens2.tolist())
        assert g2["words_emb"].shape == e2["words_emb"].shape == (B, 256, T) and g2["mask"].shape == (B, T)
        for k in ("words_emb", "sent_emb", "mask", "mu", "logvar"):
            assert torch.equal(g2[k], e2[k]), "replay on batch %d: %s differs from eager" % (seed, k)
        for k in ("fake", "fine", "att"):
            for i in range(3):
                assert g2[k][i].shape == e2[k][i].shape
                assert torch.equal(g2[k][i], e2[k][i]), "replay on batch %d: %s[%d] differs from eager" % (seed, k, i)
    assert len(seen) >= 5, "the synthetic batches should differ in their caption lengths"


@pytest.mark.parametrize("name", ["bf16", "fp32"])
def test_hipgraph_with_parallel_lanes(name, cfg_face, face_weights):
    """Three independent batches captured as parallel branches of ONE hipGraph (GraphedStep(lanes=3)): every lane's
    outputs are bit-identical to the eager step on that lane's inputs, also after new inputs are copied in."""
    from tgsr_amd.trainer import GraphedStep
    B = 2
    cap, lens, LR, LRb = O.synthetic_batch(B)
    pipe = _pipe(cfg_face, face_weights, name)
    capd, lens = cap.to(DEV), lens.tolist()
    g = torch.Generator().manual_seed(3)
    LRs = [(torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(DEV) for _ in range(3)]
    LRbs = [(torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(DEV) for _ in range(3)]
    eager = [pipe(capd, lens, LRs[k], LRbs[k])["fine"][2].clone() for k in range(3)]
    step = GraphedStep(pipe, capd, lens, LRs[0], LRbs[0], lanes=3)
    out = step.replay([capd] * 3, [lens] * 3, LRs, LRbs)
    torch.cuda.synchronize()
    for k in range(3):
        assert torch.equal(out[k]["fine"][2], eager[k]), "lane %d differs from the eager step" % k
    out = step.replay(None, None, LRs[::-1], LRbs[::-1])         # lanes swapped: same graph, new inputs
    torch.cuda.synchronize()
    for k in range(3):
        assert torch.equal(out[k]["fine"][2], eager[2 - k])
    # every lane a batch of its own: other captions and other caption lengths per lane (device-side lengths + num_words, the
    # form bench.py's timed loop uses: nothing crosses PCIe)
    batches = [O.synthetic_batch(B, seed=20 + k) for k in range(3)]
    assert len({tuple(b[1].tolist()) for b in batches}) > 1
    dev = [(b[0].to(DEV), b[1].to(torch.int32).to(DEV), b[2].to(DEV), b[3].to(DEV)) for b in batches]
    out = step.replay([d[0] for d in dev], [d[1] for d in dev], [d[2] for d in dev], [d[3] for d in dev],
                      num_words=[int(b[1].max()) for b in batches])
    torch.cuda.synchronize()
    out = [{k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in o.items()} for o in out]
    for k, b in enumerate(batches):
        e = pipe(b[0].to(DEV), b[1].tolist(), b[2].to(DEV), b[3].to(DEV))
        torch.cuda.synchronize()
        assert torch.equal(out[k]["words_emb"], e["words_emb"]) and torch.equal(out[k]["mask"], e["mask"])
        for i in range(3):
            assert torch.equal(out[k]["fine"][i], e["fine"][i]) and torch.equal(out[k]["att"][i], e["att"][i]), (k, i)
    assert pipe.overlap                                          # the capture restored the two-stream setting


def _x16_state(seed=5):
    """Seeded x16 generator parameters with the reference's key names (models16 state_dict: tied tensors listed under every
    alias, NetG_highweight with its trainable `a`), BatchNorm statistics randomised so eval-mode BN does something."""
    from tgsr_amd import models16
    from tgsr_amd.synthetic import random_init_
    gl, gh = models16.G_SR_NET_low(), models16.NetG_highweight(weightmap=False, low="lr")
    random_init_(gl, seed), random_init_(gh, seed + 1)
    g = torch.Generator().manual_seed(seed)
    for m in list(gl.modules()) + list(gh.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.bias.shape, generator=g))
    with torch.no_grad():
        gh.a.fill_(0.4)
    return ({k: v.detach().clone() for k, v in gl.state_dict().items()},
            {k: v.detach().clone() for k, v in gh.state_dict().items()})


def test_x16_pipeline_fp32_golden_and_full_size(cfg_face):
    """SRPipeline(branch_num=5) selects the models16 generators exactly as trainer_objective.py:74-87 does.  (1) the
    reference-captured x16 golden (nets16_small.npz: G_SR_NET_low, LR 8x8 -> 128^2) through the pipeline; (2) full size
    (LR 32 -> 512^2, Q = 65 536 pixels in the 4th attention) against the fp32 oracle at the stated 1e-4, eager and
    replayed from a hipGraph (the `.item()` of the trainable `a` stays off the captured path)."""
    from conftest import load_npz, split_sd
    from tgsr_amd.miscc.config import cfg
    from tgsr_amd.trainer import SRPipeline
    g = load_npz("nets16_small.npz")
    cfg.TEXT.EMBEDDING_DIM = 64
    cfg.TREE.BRANCH_NUM = 5
    pipe = SRPipeline(41, device=DEV)                         # branch_num from cfg.TREE.BRANCH_NUM, like the reference
    assert type(pipe.netGL).__module__.endswith("models16") and type(pipe.netGH).__module__.endswith("models16")
    pipe.text_encoder.load_state_dict(split_sd(g, "E."))
    pipe.netGL.load_state_dict(split_sd(g, "GL."), strict=False)       # tied aliases are stored once in the fixture
    cap = torch.from_numpy(g["captions"]).to(DEV)
    LR = torch.from_numpy(g["LR"]).to(DEV)
    r = pipe(cap, g["cap_lens"].tolist(), LR, LR)
    assert len(r["fake"]) == 4 and len(r["fine"]) == 4 and tuple(r["fine"][3].shape[-2:]) == (128, 128)
    for i in range(4):
        np.testing.assert_allclose(r["fake"][i].cpu().numpy(), g["fake%d" % i], atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(r["att"][i].cpu().numpy(), g["att%d" % i], atol=2e-5, rtol=1e-4)
    # ---- full size vs the oracle
    cfg.TEXT.EMBEDDING_DIM = 256
    sdE, _, _ = O.random_state(seed=2)
    sdL, sdH = _x16_state()
    cap, lens, LR, LRb = O.synthetic_batch(2, seed=9)
    ref = OL.sr_forward16(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)
    pipe = SRPipeline(41, device=DEV, branch_num=5).load_state_dicts(sdE, sdL, sdH)
    args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    out = pipe(*args)
    assert tuple(out["fine"][3].shape) == (2, 3, 512, 512) and tuple(out["att"][3].shape[-2:]) == (256, 256)
    for i in range(4):
        np.testing.assert_allclose(out["fake"][i].cpu().numpy(), ref["fake"][i].numpy(), atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(out["fine"][i].cpu().numpy(), ref["fine"][i].numpy(), atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(out["att"][i].cpu().numpy(), ref["att"][i].numpy(), atol=2e-5, rtol=1e-4)
    eager = [f.clone() for f in out["fine"]]
    pipe.capture(*args)
    rep = pipe.replay()
    torch.cuda.synchronize()
    for a, b in zip(rep["fine"], eager):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name,td,ulp", DTYPES)
def test_x16_lp_pipeline_full_size(name, td, ulp, cfg_face):
    """The reduced-precision executor over the weight-tied x16 stages (4th attention at 256^2, tanh heads, the 16x stage
    through residual48 / upscale8x again): within 1 dB of what the CPU model of the same rounding points predicts against
    fp32, far closer to that model than the model is to fp32, and bit-identical when replayed from a hipGraph."""
    from tgsr_amd.trainer import SRPipeline
    sdE, _, _ = O.random_state(seed=2)
    sdL, sdH = _x16_state()
    cap, lens, LR, LRb = O.synthetic_batch(2, seed=9)
    with torch.no_grad():
        ref32 = OL.sr_forward16(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)
        model = OL.sr_forward16(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb, td)
    pipe = SRPipeline(41, device=DEV, dtype=name, branch_num=5).load_state_dicts(sdE, sdL, sdH)
    args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    out = pipe(*args)
    torch.cuda.synchronize()
    for k in ("fake", "fine"):
        for i in range(4):
            got = out[k][i].cpu()
            p32, pm, pmodel = OL.psnr(got, ref32[k][i]), OL.psnr(got, model[k][i]), OL.psnr(model[k][i], ref32[k][i])
            assert abs(p32 - pmodel) < 1.0, "%s %s[%d]: %.2f dB vs fp32, the CPU model predicts %.2f" % (name, k, i, p32, pmodel)
            # closer to the model than the model is to fp32 - by 1.5 dB here (the x8 test asks 3): through four weight-tied
            # stages two implementations of the same roundings pick different neighbours of a rounding tie more often
            # (above ~85 dB both distances are fp32 summation-order noise on these small-valued images)
            assert pm > min(pmodel + (1.5 if name == "f16" else 1.0), 85.0), \
                "%s %s[%d]: only %.2f dB against the CPU model of the same roundings" % (name, k, i, pm)
    for i in range(4):
        assert OL.psnr(out["att"][i].cpu(), model["att"][i], peak=1.0) > 45.0
    eager = [f.clone() for f in out["fine"]]
    pipe.capture(*args)
    rep = pipe.replay()
    torch.cuda.synchronize()
    for a, b in zip(rep["fine"], eager):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("B,Cin,Hi,Wi", [(2, 64, 4, 32), (1, 32, 8, 64), (3, 64, 16, 32), (5, 32, 12, 96)])
def test_lp_upconv_with_fused_heads(B, Cin, Hi, Wi, name, td, ulp):
    """tgsr_lp_upconv_glu_head_fwd + tgsr_lp_head_combine (the image heads computed inside the producing upBlock, per-tile
    partial sums, one combine launch for both generators) against the unfused kernels on the same values: the feature
    image bit for bit, the head images to fp32 summation order - single tiles, several tile rows / columns, with and
    without writing the feature image, GL's head with and without tanh (x16 / x8), reproducible bit for bit."""
    import torch.nn.functional as F
    from tgsr_amd import lp
    g = torch.Generator().manual_seed(B * 11 + Cin + Wi)
    x = OL.rnd(torch.randn(B, Cin, Hi, Wi, generator=g), td)
    w = torch.randn(64, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    scale, shift = 1 + 0.1 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    w3 = torch.randn(3, 32, 3, 3, generator=g) / (3 * 32 ** 0.5)
    w5 = torch.randn(3, 32, 5, 5, generator=g) / (5 * 32 ** 0.5)
    xi = lp.from_nchw(x.to(DEV), name, cpitch=Cin)
    wp = lp.pack_upconv_weight(w.to(DEV), name)
    sc, sh = scale.to(DEV), shift.to(DEV)
    p3, p5 = lp.pack_to3_weight(w3.to(DEV), name), lp.pack_to3_weight(w5.to(DEV), name)
    Ho, Wo = 2 * Hi, 2 * Wi
    # ---- unfused: the feature image, then the two heads on it (the high one adds alpha * low)
    h = lp.new_image(B, Ho, Wo, 32, name, DEV)
    lp.upconv_glu(xi, wp, Cin, 64, sc, sh, out=h)
    for low_tanh in (False, True):
        low_ref = lp.conv_to3(h, p3, 3, tanh_axpy=low_tanh)
        high_ref = lp.conv_to3(h, p5, 5, tanh_axpy=True, addend=low_ref, alpha=0.4)
        # ---- fused: one launch per "generator" (here both read the same upBlock), feature image written once / never
        h2, part3 = lp.upconv_glu_head(xi, wp, Cin, 64, sc, sh, p3, 3)
        none, part5 = lp.upconv_glu_head(xi, wp, Cin, 64, sc, sh, p5, 5, write_out=False)
        assert none is None and torch.equal(h2, h), "the feature image of the fused launch differs"
        low = torch.full((B, 3, Ho, Wo), float("nan"), device=DEV)
        high = torch.full((B, 3, Ho, Wo), float("nan"), device=DEV)
        lp.head_combine(B, [(Ho, Wo)], [part3], [part5], [low], [high], low_tanh, 0.4)
        np.testing.assert_allclose(low.cpu().numpy(), low_ref.cpu().numpy(), atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(high.cpu().numpy(), high_ref.cpu().numpy(), atol=2e-5, rtol=1e-5)
        # an already computed low image as the combine's input (partial_low = None), and against the CPU arithmetic
        high2 = torch.empty_like(high)
        lp.head_combine(B, [(Ho, Wo)], [None], [part5], [low], [high2], low_tanh, 0.4)
        assert torch.equal(high2, high)
        hc = lp.to_nchw(h, 32, 0).cpu()
        ref3 = F.conv2d(hc, OL.rnd(w3, td), None, 1, 1)
        np.testing.assert_allclose(low.cpu().numpy(), (torch.tanh(ref3) if low_tanh else ref3).numpy(), atol=2e-5, rtol=1e-5)
        # reproducible: same partial sums, same images, bit for bit
        _, part3b = lp.upconv_glu_head(xi, wp, Cin, 64, sc, sh, p3, 3)
        lowb, highb = torch.empty_like(low), torch.empty_like(high)
        lp.head_combine(B, [(Ho, Wo)], [part3b], [part5], [lowb], [highb], low_tanh, 0.4)
        assert torch.equal(part3b, part3) and torch.equal(lowb, low) and torch.equal(highb, high)


def test_lp_head_combine_several_scales_in_one_launch():
    """Three scales of both generators finished by ONE tgsr_lp_head_combine launch == scale by scale."""
    from tgsr_amd import lp
    g = torch.Generator().manual_seed(8)
    B, name = 2, "bf16"
    sizes, pl, ph = [], [], []
    for Hi, Wi in ((4, 32), (8, 64), (16, 128)):
        x = lp.from_nchw(torch.randn(B, 32, Hi, Wi, generator=g).to(DEV), name, cpitch=32)
        wp = lp.pack_upconv_weight((torch.randn(64, 32, 3, 3, generator=g) / 17.0).to(DEV), name)
        p3 = lp.pack_to3_weight((torch.randn(3, 32, 3, 3, generator=g) / 17.0).to(DEV), name)
        p5 = lp.pack_to3_weight((torch.randn(3, 32, 5, 5, generator=g) / 28.0).to(DEV), name)
        pl.append(lp.upconv_glu_head(x, wp, 32, 64, None, None, p3, 3, write_out=False)[1])
        ph.append(lp.upconv_glu_head(x, wp, 32, 64, None, None, p5, 5, write_out=False)[1])
        sizes.append((2 * Hi, 2 * Wi))
    mk = lambda: [torch.empty(B, 3, H, W, device=DEV) for H, W in sizes]      # noqa: E731
    low, high, low1, high1 = mk(), mk(), mk(), mk()
    lp.head_combine(B, sizes, pl, ph, low, high, False, 0.5)
    for k in range(3):
        lp.head_combine(B, [sizes[k]], [pl[k]], [ph[k]], [low1[k]], [high1[k]], False, 0.5)
        assert torch.equal(low[k], low1[k]) and torch.equal(high[k], high1[k])


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_lp_pipeline_fused_heads_equal_standalone_heads(name, cfg_face, face_weights, monkeypatch):
    """The whole pipeline with the heads fused into their upBlocks (default) against TGSR_LP_FUSE_HEADS=0 (six stand-alone
    head launches): same stored activations, so the images differ only by fp32 summation order."""
    from tgsr_amd import lp_pipeline
    cap, lens, LR, LRb = O.synthetic_batch(3)
    args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    assert lp_pipeline.FUSE_HEADS
    fused = _pipe(cfg_face, face_weights, name)(*args)
    monkeypatch.setattr(lp_pipeline, "FUSE_HEADS", False)
    plain = _pipe(cfg_face, face_weights, name)(*args)
    torch.cuda.synchronize()
    for k in ("fake", "fine"):
        for i in range(3):
            np.testing.assert_allclose(fused[k][i].cpu().numpy(), plain[k][i].cpu().numpy(), atol=3e-5, rtol=1e-5)
    for i in range(3):
        assert torch.equal(fused["att"][i], plain["att"][i])


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("B,H,W", [(3, 32, 32), (2, 64, 64), (16, 32, 32), (1, 12, 96), (5, 4, 32)])
def test_lp_resblocks_chain_equals_four_launches(B, H, W, name, td, ulp):
    """tgsr_lp_resblocks_fwd - the two ResBlocks of a generator stage (four dependent convolutions, util.py:110-130) in ONE launch,
    neighbouring tiles synchronised through device flags - against four tgsr_lp_conv3x3_fwd launches: bit-identical images, also on
    the second and third launch over the same flag buffer (the flags are never reset), the error word untouched; one-tile images and
    several tile columns included."""
    from tgsr_amd import custom_ops, lp, ops          # noqa: F401  (custom_ops registers torch.ops.tgsr.*)
    g = torch.Generator().manual_seed(B * 100 + H + W)
    x = lp.from_nchw(OL.rnd(torch.randn(B, 64, H, W, generator=g), td).to(DEV), name, cpitch=64)
    ws = [torch.randn(co, 64, 3, 3, generator=g) / 24.0 for co in (128, 64, 128, 64)]
    packs = [lp.pack_conv3x3_weight(w.to(DEV), name) for w in ws]
    scales = [(1 + 0.1 * torch.randn(w.shape[0], generator=g)).to(DEV) for w in ws]
    shifts = [(0.1 * torch.randn(w.shape[0], generator=g)).to(DEV) for w in ws]
    img = lambda: lp.new_image(B, H, W, 64, name, DEV)                        # noqa: E731
    # reference: the four launches the executor issued before
    tmp, a, b = img(), img(), img()
    lp.conv3x3(x, packs[0], 64, 128, scales[0], shifts[0], glu=True, out=tmp)
    lp.conv3x3(tmp, packs[1], 64, 64, scales[1], shifts[1], residual=x, out=a)
    lp.conv3x3(a, packs[2], 64, 128, scales[2], shifts[2], glu=True, out=tmp)
    lp.conv3x3(tmp, packs[3], 64, 64, scales[3], shifts[3], residual=a, out=b)
    torch.cuda.synchronize()
    flags = lp.resblocks_flags(B, H, W, DEV)
    tmp2, a2, b2 = img(), img(), img()
    for rep in range(3):
        if rep:
            a2.zero_(), b2.zero_(), tmp2.zero_()
        torch.ops.tgsr.lp_resblocks(x, packs, scales, shifts, tmp2, a2, b2, flags)
        torch.cuda.synchronize()
        assert int(flags[-1]) == 0, "a flag wait timed out"
        assert int(flags[0]) == rep + 1 and bool((flags[:-1] == rep + 1).all())
        assert torch.equal(a2, a) and torch.equal(b2, b) and torch.equal(tmp2, tmp), "launch %d differs from the four launches" % rep
    with pytest.raises(ops.TgsrError):
        lp.resblocks(x, packs, scales, shifts, tmp2, a2, b2, flags[:-1])     # not this size's flag buffer


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_lp_pipeline_resblock_chain_equals_separate_launches(name, cfg_face, face_weights, monkeypatch):
    """The whole reduced-precision step with the 32^2 and 64^2 stages' ResBlocks as one launch each (TGSR_LP_CHAIN=1) == the four
    launches per stage, bit for bit: eager, replayed from a hipGraph, and as three parallel lanes of one graph (three instances of
    the flag-synchronised kernel in flight at once, each with its own flag buffer)."""
    from tgsr_amd import lp_pipeline
    from tgsr_amd.trainer import GraphedStep
    B = 4
    cap, lens, LR, LRb = O.synthetic_batch(B, seed=12)
    args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
    monkeypatch.setattr(lp_pipeline, "CHAIN", True)        # off by default: measured slower than four launches (profiles/HISTORY.md 3.17)
    chained = _pipe(cfg_face, face_weights, name)
    a = chained(*args)
    assert any(st.get("flags") is not None for bufs in chained._lp.bufs.values() for st in bufs["gl"])   # the chain kernel did run
    monkeypatch.setattr(lp_pipeline, "CHAIN", False)
    b = _pipe(cfg_face, face_weights, name)(*args)
    torch.cuda.synchronize()
    for k in range(3):
        assert torch.equal(a["fine"][k], b["fine"][k]) and torch.equal(a["fake"][k], b["fake"][k]) and torch.equal(a["att"][k], b["att"][k])
    monkeypatch.setattr(lp_pipeline, "CHAIN", True)
    want = [f.clone() for f in a["fine"]]
    chained.capture(*args)
    for _ in range(3):
        r = chained.replay()
        torch.cuda.synchronize()
        for x, y in zip(r["fine"], want):
            assert torch.equal(x, y)
    step = GraphedStep(chained, *args, lanes=3)
    out = step.replay()
    torch.cuda.synchronize()
    for lane in out:
        for x, y in zip(lane["fine"], want):
            assert torch.equal(x, y)
    for bufs in step.bufs:
        for st in bufs["gl"]:
            if st.get("flags") is not None:
                assert int(st["flags"][-1]) == 0
