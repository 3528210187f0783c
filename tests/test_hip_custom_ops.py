"""GPU: the `torch.ops.tgsr.*` custom operators (tgsr_amd/custom_ops.py) - torch.library.opcheck on the two
differentiable ones, agreement with the ctypes wrappers they sit on, and that the drop-in modules dispatch through them."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from tgsr_amd import _lib, custom_ops  # noqa: F401
    _lib.lib()
    assert torch.cuda.is_available()


def test_opcheck_conv_to3_and_conv4x4s2():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 32, 16, 16, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(3, 32, 5, 5, generator=g) / 28.0).to(DEV).requires_grad_(True)
    add = torch.randn(2, 3, 16, 16, generator=g).to(DEV)
    utils = ("test_schema", "test_autograd_registration", "test_faketensor")
    torch.library.opcheck(torch.ops.tgsr.conv_to3.default, (x, w, True, add, 0.5), test_utils=utils)
    torch.library.opcheck(torch.ops.tgsr.conv_to3.default, (x, w[:, :, 1:4, 1:4].contiguous(), False, None, 0.0),
                          test_utils=utils)
    xd = torch.randn(2, 8, 16, 16, generator=g).to(DEV).requires_grad_(True)
    wd = (torch.randn(16, 8, 4, 4, generator=g) / 11.0).to(DEV).requires_grad_(True)
    torch.library.opcheck(torch.ops.tgsr.conv4x4s2.default, (xd, wd, True), test_utils=utils)
    torch.library.opcheck(torch.ops.tgsr.glu.default, (torch.randn(3, 8, 5, generator=g).to(DEV).requires_grad_(True),),
                          test_utils=utils)
    torch.library.opcheck(torch.ops.tgsr.conv3x3_fused_out.default,
                          (torch.randn(1, 32, 8, 32, device=DEV), torch.zeros(32 * 32 * 9, device=DEV), 32, None, None,
                           False, False, None, torch.empty(1, 32, 8, 32, device=DEV)), test_utils=("test_schema", "test_faketensor"))


def test_custom_op_autograd_matches_torch():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 32, 8, 16, generator=g)
    w = torch.randn(3, 32, 3, 3, generator=g) / 17.0
    add = torch.randn(3, 3, 8, 16, generator=g)
    dy = torch.randn(3, 3, 8, 16, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = torch.tanh(F.conv2d(xr, wr, None, 1, 1)) + 0.5 * add
    ref.backward(dy)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    out = torch.ops.tgsr.conv_to3(xd, wd, True, add.to(DEV), 0.5)
    out.backward(dy.to(DEV))
    assert torch.allclose(out.cpu(), ref, atol=1e-5) and torch.allclose(xd.grad.cpu(), xr.grad, atol=1e-5)
    assert torch.allclose(wd.grad.cpu(), wr.grad, atol=1e-4, rtol=1e-4)


def test_modules_dispatch_through_torch_ops(monkeypatch):
    """util's fused blocks call torch.ops.tgsr.* (not the ctypes wrappers directly): count the dispatches of one forward."""
    from tgsr_amd import custom_ops as C, util
    from tgsr_amd.miscc.config import cfg, cfg_reset
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    calls = {"n": 0}
    real = C.conv3x3_wino

    def counting(*a):
        calls["n"] += 1
        return real(*a)
    monkeypatch.setattr(C, "conv3x3_wino", counting)
    rb = util.ResBlock(64).to(DEV).eval()
    y = rb(torch.randn(2, 64, 16, 32, device=DEV))
    torch.cuda.synchronize()
    assert y.shape == (2, 64, 16, 32) and calls["n"] == 2
    assert "tgsr::conv3x3_wino" in str(torch.ops.tgsr.conv3x3_wino.default._schema)
    cfg_reset()


def test_filter_packed_for_another_channel_count_raises():
    """A weight [Cout, Cin', 3, 3] applied to a Cin-channel input is a shape error in torch (the reference raises it for
    NetG_highweight at GF_DIM != 32: ResBlock(channel_num=32) is hard-coded, model.py:258-262).  The HIP kernels walk
    the packed filter by the INPUT's channel count, so the wrappers must refuse instead of reading past the pack."""
    from tgsr_amd import lp, ops
    from tgsr_amd._lib import TgsrError
    x = torch.randn(2, 128, 32, 32, device=DEV)
    w = torch.randn(64, 32, 3, 3, device=DEV)
    with pytest.raises(TgsrError):
        ops.conv3x3_wino(x, ops.pack_wino_weight(w), 64, None, None)
    with pytest.raises(TgsrError):
        ops.conv3x3_fused(x, ops.pack_conv3x3_weight(w), 64, None, None)
    with pytest.raises(TgsrError):
        ops.upwino_glu(x, ops.pack_upwino_weight(w, glu=False), 64, None, None, glu=False)
    xi = lp.from_nchw(x[:, :64].contiguous(), "bf16")
    with pytest.raises(TgsrError):
        lp.conv3x3(xi, lp.pack_conv3x3_weight(w, "bf16"), 64, 64, None, None)


def test_opcheck_training_text_damsm_and_lp_operators():
    """Every `torch.ops.tgsr.*` entry point of the training / text / DAMSM / reduced-precision paths: schema (declared
    mutations are the only ones) and fake kernel (shapes, dtypes) through torch.library.opcheck; the differentiable ones
    also their autograd registration."""
    from tgsr_amd import lp
    T = torch.ops.tgsr
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g).to(DEV)                 # noqa: E731
    basic = ("test_schema", "test_faketensor")
    full = basic + ("test_autograd_registration",)
    B, Cc, H, W = 2, 64, 8, 16
    raw = R(B, Cc, H, W)
    gam, bet, rm, rv = R(Cc), R(Cc), torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    chk = torch.library.opcheck
    chk(T.bn_train_fwd.default, (raw, gam, bet, 1e-5, 0.1, rm, rv, 1, None, nbt), test_utils=basic)
    out, stats = T.bn_train_fwd(raw, gam, bet, 1e-5, 0.1, rm, rv, 1, None, nbt)
    upw = T.pack_wino_weight(R(Cc, Cc, 3, 3), False, False)
    chk(T.conv3x3_wino_stats.default, (raw, upw, Cc), test_utils=basic)
    _, sp = T.conv3x3_wino_stats(raw, upw, Cc)
    chk(T.bn_train_fwd_from_stats.default, (raw, gam, bet, 1e-5, 0.1, rm, rv, 1, None, nbt, sp), test_utils=basic)
    chk(T.bn_train_fwd_out.default, (raw, gam, bet, 1e-5, 0.1, rm, rv, 2, nbt, torch.empty_like(raw), torch.empty(4, Cc, device=DEV)),
        test_utils=basic)
    chk(T.bn_train_bwd.default, (R(B, Cc // 2, H, W), raw, stats, 1, torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV), None),
        test_utils=basic)
    x32 = R(B, 32, H, W)
    for up in (False, True):
        dr = R(B, Cc, H * (2 if up else 1), W * (2 if up else 1))
        chk(T.conv3x3_wgrad.default, (dr, x32, up, True, torch.empty(Cc, 32, 3, 3, device=DEV)), test_utils=basic)
    chk(T.conv3x3_wgrad.default, (R(B, 32, H, W), R(B, 3, H, W), False, True, torch.empty(32, 3, 3, 3, device=DEV)), test_utils=basic)
    chk(T.sumpool2x2.default, (raw,), test_utils=basic)
    upu4 = T.pack_upwino4_weight(R(Cc, 32, 3, 3), True)
    chk(T.pack_upwino4_weight.default, (R(Cc, 32, 3, 3), True), test_utils=basic)
    xu4 = R(B, 32, 4, 32)
    chk(T.upwino4_glu.default, (xu4, upu4, Cc, gam, bet), test_utils=basic)
    chk(T.upwino4_glu_out.default, (xu4, upu4, Cc, gam, bet, torch.empty(B, Cc // 2, 8, 64, device=DEV)), test_utils=basic)
    up4 = T.pack_wino4_weight(R(Cc, Cc, 3, 3), False, False)
    chk(T.pack_wino4_weight.default, (R(Cc, Cc, 3, 3), True, False), test_utils=basic)
    chk(T.pack_wino4_weight.default, (R(Cc, 128, 3, 3), False, True), test_utils=basic)
    chk(T.conv3x3_wino4_stats.default, (x4 if False else R(B, Cc, 8, 64), up4, Cc), test_utils=basic)
    x4 = R(B, Cc, 8, 64)
    chk(T.conv3x3_wino4.default, (x4, up4, Cc, gam, bet, False, R(B, Cc, 8, 64)), test_utils=basic)
    upw4 = T.pack_wino4w_weight(R(128, Cc, 3, 3), True, False)
    chk(T.pack_wino4w_weight.default, (R(128, Cc, 3, 3), False, False), test_utils=basic)
    chk(T.pack_wino4w_weight.default, (R(Cc, 128, 3, 3), False, True), test_utils=basic)
    chk(T.conv3x3_wino4w_stats.default, (R(B, Cc, 8, 64), T.pack_wino4w_weight(R(128, Cc, 3, 3), False, False), 128), test_utils=basic)
    chk(T.conv3x3_wino4w.default, (x4, upw4, 128, None, None, True, None), test_utils=basic)
    chk(T.conv3x3_wino4w_out.default, (x4, upw4, 128, None, None, True, None, torch.empty(B, Cc, 8, 64, device=DEV)), test_utils=basic)
    chk(T.conv3x3_wino4_out.default, (x4, up4, Cc, None, None, False, None, torch.empty(B, Cc, 8, 64, device=DEV)), test_utils=basic)
    w33 = R(Cc, 32, 3, 3) / 17.0
    chk(T.pack_conv3x3_weight.default, (w33, False), test_utils=basic)
    chk(T.pack_conv3x3_weight.default, (w33, True), test_utils=basic)
    chk(T.pack_wino_weight.default, (w33, True, False), test_utils=basic)
    chk(T.pack_wino_weight.default, (w33, False, True), test_utils=basic)
    chk(T.pack_upwino_weight.default, (w33, False), test_utils=basic)
    chk(T.upwino.default, (x32, T.pack_upwino_weight(w33, False), Cc, None, None, False), test_utils=basic)
    xg, wg = R(2, 256, 4, 4), R(256, 256, 3, 3) / 48.0
    chk(T.conv3x3_gemm.default, (xg, wg), test_utils=basic)
    chk(T.conv3x3_gemm_dgrad.default, (xg, wg), test_utils=basic)
    chk(T.conv3x3_gemm_wgrad_out.default, (xg, xg, torch.empty_like(wg)), test_utils=basic)
    chk(T.conv4x4s2_wgrad_out.default, (R(2, 16, 4, 4), R(2, 8, 8, 8), torch.empty(16, 8, 4, 4, device=DEV)), test_utils=basic)
    # word attention backward, projections
    h, words, wc = R(2, 32, 8, 8), R(2, 48, 7), R(32, 48, 1, 1)
    src = T.word_project(words, [wc])[0]
    mask = torch.zeros(2, 7, dtype=torch.bool, device=DEV)
    chk(T.word_project.default, (words, [wc, wc]), test_utils=basic)
    chk(T.word_attention_bwd.default, (h, src, mask, False, 7, R(2, 32, 8, 8)), test_utils=basic)
    # differentiable GEMM / dot heads
    chk(T.rowdot.default, (R(3, 512).requires_grad_(True), R(512).requires_grad_(True), R(1).requires_grad_(True)), test_utils=full)
    chk(T.rowdot_bwd.default, (R(3), R(3, 512), R(512), True, True), test_utils=basic)
    chk(T.linear.default, (R(4, 64).requires_grad_(True), R(32, 64).requires_grad_(True), R(32).requires_grad_(True)), test_utils=full)
    chk(T.conv1x1.default, (R(2, 64, 5, 5).requires_grad_(True), R(32, 64, 1, 1).requires_grad_(True)), test_utils=full)
    # text encoder
    Hh, ntok = 32, 20
    emb, w_ih, w_hh = R(ntok, 24) * 0.1, R(2, 4 * Hh, 24) * 0.2, R(2, 4 * Hh, Hh) * 0.2
    b_ih, b_hh = R(2, 4 * Hh) * 0.1, R(2, 4 * Hh) * 0.1
    cap = torch.randint(1, ntok, (3, 6), generator=g).to(DEV)
    lens = [6, 4, 2]
    chk(T.lstm_gate_table.default, (emb, w_ih, b_ih, b_hh), test_utils=basic)
    chk(T.bilstm_table.default, (cap, lens, T.lstm_gate_table(emb, w_ih, b_ih, b_hh), w_hh), test_utils=basic)
    chk(T.bilstm_table_static.default, (cap, torch.tensor(lens, dtype=torch.int32, device=DEV),
                                        T.lstm_gate_table(emb, w_ih, b_ih, b_hh), w_hh), test_utils=basic)
    gw_ih, gw_hh, gb = R(2, 3 * Hh, 24) * 0.2, R(2, 3 * Hh, Hh) * 0.2, R(2, 3 * Hh) * 0.1
    chk(T.gru_gate_table.default, (emb, gw_ih, gb, gb), test_utils=basic)
    gtab, gbn = T.gru_gate_table(emb, gw_ih, gb, gb)
    chk(T.bigru_table.default, (cap, lens, gtab, gw_hh, gbn), test_utils=basic)
    chk(T.bigru_table_static.default, (cap, torch.tensor(lens, dtype=torch.int32, device=DEV), gtab, gw_hh, gbn), test_utils=basic)
    xe = R(3, 6, 24)
    chk(T.bilstm_train.default, (xe, w_ih, w_hh, b_ih, b_hh, lens), test_utils=basic)
    wds, sent, acts = T.bilstm_train(xe, w_ih, w_hh, b_ih, b_hh, lens)
    chk(T.bilstm_bwd.default, (lens, w_hh, acts, wds, torch.randn_like(wds), torch.randn_like(sent)), test_utils=basic)
    gw_ih, gw_hh, gb = R(2, 96, 24) * 0.2, R(2, 96, 32) * 0.2, R(2, 96) * 0.1
    chk(T.bigru_train.default, (xe, gw_ih, gw_hh, gb, gb.clone(), lens), test_utils=basic)
    gwds, gsent, gacts = T.bigru_train(xe, gw_ih, gw_hh, gb, gb.clone(), lens)
    chk(T.bigru_bwd.default, (lens, gw_hh, gacts, gwds, torch.randn_like(gwds), torch.randn_like(gsent)), test_utils=basic)
    # DAMSM
    feats, wemb = R(3, 64, 5, 5), R(3, 64, 6)
    chk(T.damsm_words.default, (feats, wemb, lens, 4.0, 5.0), test_utils=basic)
    chk(T.damsm_words_bwd.default, (feats, wemb, lens, 4.0, 5.0, R(3, 3)), test_utils=basic)
    chk(T.func_attention.default, (wemb, feats, 4.0), test_utils=basic)
    chk(T.ca_net.default, (R(3, 64), R(40, 64), R(40), 10, R(3, 10)), test_utils=basic)
    chk(T.ca_net.default, (R(3, 64), R(40, 64), R(40), 10, None), test_utils=basic)
    cap = torch.randint(0, 9, (3, 12), generator=g).to(DEV)
    chk(T.text_tail.default, (R(3, 48, 9), [R(32, 48), R(32, 48)], R(3, 64), R(40, 64), R(40), 10, cap), test_utils=basic)
    chk(T.multi_copy.default, ([torch.empty(5, 7, device=DEV), torch.empty(3, dtype=torch.int64, device=DEV)],
                               [R(5, 7), cap[0, :3].contiguous()]), test_utils=basic)
    chk(T.to_uint8.default, (R(2, 3, 8, 8),), test_utils=basic)
    chk(T.axpy_images.default, ([R(2, 3, 8, 8), R(2, 3, 4, 4)], [R(2, 3, 8, 8), R(2, 3, 4, 4)], 0.5), test_utils=basic)
    # reduced-precision path (lp images)
    xi = lp.from_nchw(R(2, 64, 8, 32), "bf16", cpitch=64)
    wp = lp.pack_conv3x3_weight(R(64, 64, 3, 3) / 24.0, "bf16")
    sc, sh = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    chk(T.lp_conv3x3.default, (xi, wp, 64, 64, sc, sh, True, False, None, 0, lp.new_image(2, 8, 32, 64, "bf16", DEV), 32), test_utils=basic)
    chk(T.lp_conv3x3.default, (xi, wp, 64, 64, sc, sh, False, False, xi, 0, lp.new_image(2, 8, 32, 64, "bf16", DEV), 0), test_utils=basic)
    rb_packs = [lp.pack_conv3x3_weight(R(co, 64, 3, 3) / 24.0, "bf16") for co in (128, 64, 128, 64)]
    rb_aff = [torch.ones(co, device=DEV) for co in (128, 64, 128, 64)], [torch.zeros(co, device=DEV) for co in (128, 64, 128, 64)]
    img64 = lambda: lp.new_image(2, 8, 32, 64, "bf16", DEV)                                   # noqa: E731
    chk(T.lp_resblocks.default, (xi, rb_packs, rb_aff[0], rb_aff[1], img64(), img64(), img64(), lp.resblocks_flags(2, 8, 32, DEV)),
        test_utils=basic)
    wu = lp.pack_upconv_weight(R(64, 64, 3, 3) / 24.0, "bf16")
    chk(T.lp_upconv_glu.default, (xi, wu, 64, 64, sc, sh, lp.new_image(2, 16, 64, 32, "bf16", DEV), 0), test_utils=basic)
    p3 = lp.pack_to3_weight(R(3, 32, 3, 3) / 17.0, "bf16")
    p5 = lp.pack_to3_weight(R(3, 32, 5, 5) / 28.0, "bf16")
    part3 = torch.empty(lp.head_partial_elems(2, 16, 64, 3), device=DEV)
    part5 = torch.empty(lp.head_partial_elems(2, 16, 64, 5), device=DEV)
    chk(T.lp_upconv_glu_head.default, (xi, wu, 64, 64, sc, sh, p3, 3, part3, lp.new_image(2, 16, 64, 32, "bf16", DEV), 0), test_utils=basic)
    chk(T.lp_upconv_glu_head.default, (xi, wu, 64, 64, sc, sh, p5, 5, part5, None, 0), test_utils=basic)
    T.lp_upconv_glu_head(xi, wu, 64, 64, sc, sh, p3, 3, part3, None, 0)
    T.lp_upconv_glu_head(xi, wu, 64, 64, sc, sh, p5, 5, part5, None, 0)
    low, high = torch.empty(2, 3, 16, 64, device=DEV), torch.empty(2, 3, 16, 64, device=DEV)
    chk(T.lp_head_combine.default, ([16], [64], [part3], [part5], [low], [high], False, 0.5), test_utils=basic)
    chk(T.lp_stem.default, (R(2, 3, 8, 32), R(64, 3, 3, 3) / 5.0, sc, sh, lp.new_image(2, 8, 32, 64, "bf16", DEV), 0), test_utils=basic)
    h32 = lp.from_nchw(R(2, 32, 8, 32), "bf16", cpitch=64)
    chk(T.lp_conv_to3.default, (h32, p5, 5, True, R(2, 3, 8, 32), 0.5), test_utils=basic)
    chk(T.lp_word_attention.default, (h32, R(2, 32, 32), None, 7, False, 32), test_utils=basic)
    # the attention pack of the text tail and the producers of h that attend in their epilogue; the f16 <-> bf16 hand-over
    tail = (R(2, 48, 9), [R(32, 48), R(32, 48)], R(2, 64), R(40, 64), R(40), 10, cap[:2].contiguous(), True)
    chk(T.text_tail_lp.default, tail, test_utils=basic)
    pack = T.text_tail_lp(*tail)[4]
    att = (pack, 2, 1, 9, True, False, 32)
    chk(T.lp_stem_att.default, (R(2, 3, 8, 32), R(64, 3, 3, 3) / 5.0, sc, sh, lp.new_image(2, 8, 32, 64, "bf16", DEV), 0) + att +
        (torch.empty(2, 9, 8, 32, device=DEV),), test_utils=basic)
    chk(T.lp_upconv_glu_att.default, (xi, wu, 64, 64, sc, sh, lp.new_image(2, 16, 64, 64, "bf16", DEV), 0) + att +
        (torch.empty(2, 9, 16, 64, device=DEV),), test_utils=basic)
    chk(T.lp_upconv_glu_head_att.default, (xi, wu, 64, 64, sc, sh, p3, 3, part3, lp.new_image(2, 16, 64, 64, "bf16", DEV), 0) + att +
        (None,), test_utils=basic)
    chk(T.lp_convert.default, (lp.from_nchw(R(2, 32, 8, 32), "f16"), lp.new_image(2, 8, 32, 32, "bf16", DEV)), test_utils=basic)


def test_only_the_tensor_wrappers_touch_the_c_abi():
    """`_lib.lib()` (the ctypes handle of libtgsr_hip.so) is used by tgsr_amd/ops.py and tgsr_amd/lp.py only: modules,
    autograd formulas, losses, trainers and the lp executor go through torch.ops.tgsr.* (which sit on those wrappers)."""
    import glob
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tgsr_amd")
    for path in glob.glob(os.path.join(root, "**", "*.py"), recursive=True):
        if os.path.basename(path) in ("ops.py", "lp.py", "_lib.py"):
            continue
        src = open(path).read()
        assert "_lib.lib()" not in src and "lib()." not in src, path


def test_opcheck_round6_operators():
    """The operators added in round 6: the weight-map heads (`axpy_map`), eval-mode BatchNorm + activation (`affine_act`), the GAN
    losses' BCE terms (`weighted_bce`) and the kernels of CNN_ENCODER's frozen trunk (`gconv` ... `bilinear`): schema (declared
    mutations are the only ones), fake kernels, and the autograd registration of the differentiable ones."""
    T = torch.ops.tgsr
    g = torch.Generator().manual_seed(1)
    R = lambda *s: torch.randn(*s, generator=g).to(DEV)                 # noqa: E731
    basic = ("test_schema", "test_faketensor")
    full = basic + ("test_autograd_registration",)
    chk = torch.library.opcheck
    chk(T.axpy_map.default, (R(2, 3, 8, 12).requires_grad_(True), R(2, 3, 8, 12).requires_grad_(True), R(8, 12).requires_grad_(True)),
        test_utils=full)
    chk(T.axpy_map_bwd.default, (R(2, 3, 8, 12), R(2, 3, 8, 12), R(8, 12), True, True), test_utils=basic)
    raw, sc, sh = R(2, 6, 4, 8), R(6), R(6)
    chk(T.affine_act.default, (raw, sc, sh, 2), test_utils=basic)
    chk(T.affine_act_bwd.default, (R(2, 6, 4, 8), T.affine_act(raw, sc, sh, 2), sc, 2), test_utils=basic)
    a, b = R(11).requires_grad_(True), R(6).requires_grad_(True)
    t, w = torch.rand(17, generator=g).to(DEV), torch.rand(17, generator=g).to(DEV)
    chk(T.weighted_bce.default, (a, b, t, w), test_utils=full)
    chk(T.weighted_bce.default, (a, None, t[:11].contiguous(), w[:11].contiguous()), test_utils=full)
    chk(T.weighted_bce_bwd.default, (R(1).reshape(()), a.detach(), b.detach(), t, w), test_utils=basic)
    wt = R(8, 5, 3, 3)
    chk(T.gconv_pack.default, (wt, R(8), False), test_utils=basic)
    chk(T.gconv_pack.default, (wt, None, True), test_utils=basic)
    x, out = R(2, 5, 9, 9), torch.zeros(2, 10, 9, 9, device=DEV)
    chk(T.gconv.default, (False, T.gconv_pack(wt, None, False), x, 0, 5, out, 1, 3, 3, 1, 1, 1, R(8), True, False, None, None), test_utils=basic)
    chk(T.gconv.default, (True, T.gconv_pack(wt, None, True), out, 1, 8, torch.zeros(2, 5, 9, 9, device=DEV), 0, 3, 3, 1, 1, 1, None, False,
                          True, None, R(2, 5, 9, 9)), test_utils=basic)
    chk(T.maxpool3s2.default, (x, torch.zeros(2, 7, 4, 4, device=DEV), 2), test_utils=basic)
    chk(T.maxpool3s2_bwd.default, (x, R(2, 7, 4, 4), 2, torch.zeros(2, 5, 9, 9, device=DEV), True, R(2, 5, 9, 9)), test_utils=basic)
    chk(T.avgpool3.default, (x, torch.zeros(2, 5, 9, 9, device=DEV), False, None), test_utils=basic)
    chk(T.plane_mean.default, (x,), test_utils=basic)
    chk(T.plane_mean_bwd.default, (R(2, 5), 9, 9), test_utils=basic)
    chk(T.relu_mask_.default, (R(2, 5, 9, 9), x, 1, 3), test_utils=basic)
    chk(T.bilinear.default, (x, 13, 17), test_utils=basic)
    chk(T.bilinear_bwd.default, (R(2, 5, 13, 17), 9, 9), test_utils=basic)
    # the flat Adam update: four in-place buffers and the device-side step state, all declared
    n = 1000
    chk(T.adam_flat_.default, (R(n), R(n), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV),
                               torch.tensor([0.0, 1.0, 1.0], device=DEV), 1e-3, 0.5, 0.999, 1e-8, 0.0, True), test_utils=basic)
    chk(T.interleave2x2_.default, (R(2, 3, 3, 4), R(2, 3, 3, 3), R(2, 3, 2, 4), R(2, 3, 2, 3), torch.zeros(2, 3, 5, 7, device=DEV), True,
                                   R(2, 3, 5, 7)), test_utils=basic)
    chk(T.sum_stack.default, (R(3, 2, 4, 6), 3, torch.zeros(2, 4, 6, device=DEV)), test_utils=basic)
