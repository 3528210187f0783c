"""GPU: a kernel's result must not depend on what runs beside it.

Every inference kernel of the path is launched on one stream while an MFMA-heavy neighbour loops on a second stream, and
its output is compared bit for bit with the one it gives alone.  The case this pins: packed fp32 instructions
(v_pk_fma_f32 ...) whose registers are reloaded right behind them read the NEW contents when the matrix pipe is busy with
another wave's MFMAs (profiles/HISTORY.md section 3.13, tools/lds_neighbour_check.py) - the LSTM recurrence and the lp stem gave
different results under a hipGraph's concurrency until their packed instructions were removed."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _neighbours(g, B=16):
    from tgsr_amd import lp, ops
    R = lambda *sh: torch.randn(*sh, generator=g).to(DEV)            # noqa: E731
    dt = "bf16"
    out = {}
    x = lp.from_nchw(R(B, 64, 64, 64), dt)
    wp = lp.pack_upconv_weight(R(64, 64, 3, 3) * 0.1, dt)
    sc, sh = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    o = lp.new_image(B, 128, 128, 32, dt, DEV)
    out["lp upBlock 64 @64->128"] = lambda: lp.upconv_glu(x, wp, 64, 64, sc, sh, out=o)
    x2 = lp.from_nchw(R(B, 32, 32, 32), dt)
    wp2 = lp.pack_upconv_weight(R(64, 32, 3, 3) * 0.1, dt)
    o2 = lp.new_image(B, 64, 64, 32, dt, DEV)
    out["lp upBlock 32 @32->64"] = lambda: lp.upconv_glu(x2, wp2, 32, 64, sc, sh, out=o2)
    x3 = lp.from_nchw(R(B, 64, 128, 128), dt)
    wp3 = lp.pack_conv3x3_weight(R(128, 64, 3, 3) * 0.1, dt)
    sc3, sh3 = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    o3 = lp.new_image(B, 128, 128, 64, dt, DEV)
    out["lp conv 64->128 glu @128^2"] = lambda: lp.conv3x3(x3, wp3, 64, 128, sc3, sh3, glu=True, out=o3)
    xf = R(B, 64, 64, 64)
    upf = ops.pack_wino_weight(R(128, 64, 3, 3) * 0.1, True, False)
    out["fp32 wino 64->128 glu @64^2"] = lambda: ops.conv3x3_wino(xf, upf, 128, sc3, sh3, True, None)
    xf4 = R(B, 64, 128, 128)
    upf4 = ops.pack_wino4_weight(R(128, 64, 3, 3) * 0.1, True)
    out["fp32 wino4 64->128 glu @128^2"] = lambda: ops.conv3x3_wino4(xf4, upf4, 128, sc3, sh3, True, None)
    xu4, pu4 = R(B, 64, 64, 64), ops.pack_upwino4_weight(R(64, 64, 3, 3) * 0.1)
    sc64n, sh64n = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    out["fp32 upwino4 64 @64->128"] = lambda: ops.upwino4_glu(xu4, pu4, 64, sc64n, sh64n)
    upfw = ops.pack_wino4w_weight(R(128, 64, 3, 3) * 0.1, True)
    out["fp32 wino4 wide 64->128 glu @128^2"] = lambda: ops.conv3x3_wino4(xf4, upfw, 128, sc3, sh3, True, None, wide=True)
    return out


def _victims(g, B=16):
    from tgsr_amd import lp, ops
    R = lambda *sh: torch.randn(*sh, generator=g).to(DEV)            # noqa: E731
    dt = "bf16"
    v = {}
    T, H, ntok = 18, 128, 41
    cap = torch.randint(1, ntok, (B, T), generator=g).to(DEV)
    table, w_hh = R(ntok, 2, 4 * H) * 0.5, R(2, 4 * H, H) * 0.08
    v["bilstm_table (H = 128)"] = lambda: torch.cat([t.flatten() for t in ops.bilstm_table(cap, [T] * B, table, w_hh)])
    gtab, gbn = ops.gru_gate_table(R(ntok, 300) * 0.1, R(2, 3 * H, 300) * 0.05, R(2, 3 * H) * 0.1, R(2, 3 * H) * 0.1)
    gw_hh = R(2, 3 * H, H) * 0.08
    v["bigru_table (H = 128)"] = lambda: torch.cat([t.flatten() for t in ops.bigru_table(cap, [T] * B, gtab, gw_hh, gbn)])
    words, sent = R(B, 256, T), R(B, 256)
    ws = [R(32, 256) for _ in range(3)]
    caw, cab = R(400, 256) * 0.1, R(400)
    v["text_tail"] = lambda: torch.cat([t.flatten().float() for t in ops.text_tail(words, ws, sent, caw, cab, 100, cap)])
    img = R(B, 3, 32, 32)
    wst, scs, shs = R(64, 3, 3, 3) * 0.2, torch.rand(64, generator=g).to(DEV) + 0.5, R(64) * 0.1
    v["lp stem"] = lambda: lp.stem(img, wst, scs, shs, dtype=dt).float().flatten()
    himg = lp.from_nchw(R(B, 32, 32, 32), dt, cpitch=64)
    src = R(B, 32, 32)
    v["lp word attention @32^2"] = lambda: torch.cat([lp.word_attention(himg, src, None, T).flatten(), himg.float().flatten()])
    xs = lp.from_nchw(R(B, 32, 32, 32), dt)
    wps = lp.pack_conv3x3_weight(R(64, 32, 3, 3) * 0.1, dt)
    sc64, sh64 = torch.rand(64, generator=g).to(DEV) + 0.5, R(64) * 0.1
    v["lp conv 32->64 glu @32^2"] = lambda: lp.conv3x3(xs, wps, 32, 64, sc64, sh64, glu=True).float().flatten()
    xw = R(B, 32, 32, 32)
    upw = ops.pack_wino_weight(R(64, 32, 3, 3) * 0.1, True, False)
    v["fp32 wino 32->64 glu @32^2"] = lambda: ops.conv3x3_wino(xw, upw, 64, sc64, sh64, True, None).flatten()
    xw4 = R(2, 32, 64, 64)
    upw4 = ops.pack_wino4_weight(R(64, 32, 3, 3) * 0.1, True)
    v["fp32 wino4 32->64 glu @64^2"] = lambda: ops.conv3x3_wino4(xw4, upw4, 64, sc64, sh64, True, None).flatten()
    xww, upww = R(2, 32, 64, 64), ops.pack_wino4w_weight(R(128, 32, 3, 3) * 0.1, True)
    sc128, sh128 = torch.rand(128, generator=g).to(DEV) + 0.5, R(128) * 0.1
    v["fp32 wino4 wide 32->128 glu @64^2"] = lambda: ops.conv3x3_wino4(xww, upww, 128, sc128, sh128, True, None, wide=True).flatten()
    xuv, puv = R(2, 32, 32, 32), ops.pack_upwino4_weight(R(64, 32, 3, 3) * 0.1)
    v["fp32 upwino4 32 @32->64"] = lambda: ops.upwino4_glu(xuv, puv, 64, sc64, sh64).flatten()
    hf, wf, wc = R(B, 32, 32, 32), R(B, 256, T), R(32, 256)
    v["fp32 word attention @32^2"] = lambda: torch.cat([t.flatten() for t in ops.word_attention(hf, wf, wc, None)])
    xh, wh = R(B, 32, 64, 64), R(3, 32, 3, 3) * 0.1
    v["fp32 conv_to3 3x3 @64^2"] = lambda: ops.conv_to3(xh, wh).flatten()
    wh5, addend = R(3, 32, 5, 5) * 0.1, R(B, 3, 64, 64)
    v["fp32 conv_to3 5x5 tanh @64^2"] = lambda: ops.conv_to3(xh, wh5, tanh_axpy=True, addend=addend, alpha=0.5).flatten()
    xu = R(B, 64, 32, 32)
    upu = ops.pack_upwino_weight(R(128, 64, 3, 3) * 0.1, True)
    sc128, sh128 = torch.rand(128, generator=g).to(DEV) + 0.5, R(128) * 0.1
    v["fp32 upBlock (upwino) 64 @32->64"] = lambda: ops.upwino_glu(xu, upu, 128, sc128, sh128).flatten()
    # lp upBlock with its fused 3x3 head + the combine of the partial sums
    xl = lp.from_nchw(R(B, 64, 32, 32), dt)
    wpu = lp.pack_upconv_weight(R(64, 64, 3, 3) * 0.1, dt)
    hwp = lp.pack_to3_weight(R(3, 32, 3, 3) * 0.1, dt)

    def up_head():
        out, part = lp.upconv_glu_head(xl, wpu, 64, 64, sc64, sh64, hwp, 3)
        low = torch.empty(B, 3, 64, 64, device=DEV)
        lp.head_combine(B, [(64, 64)], [part], [None], [low], [None], True, 0.5)
        return torch.cat([out.float().flatten(), low.flatten()])
    v["lp upBlock + fused head + combine @32->64"] = up_head
    # the producers that attend to the words in their epilogue (text tail with the attention pack, stem, upBlock), and the
    # f16 -> bf16 hand-over of NetG_highweight's trunk
    capz = cap.clone()
    capz[:, 9:] = 0
    tail = ops.text_tail(words, ws, sent, caw, cab, 100, capz, lp_dtype=torch.bfloat16)
    v["text_tail + attention pack"] = lambda: torch.cat([t.flatten().float() for t in
                                                         ops.text_tail(words, ws, sent, caw, cab, 100, capz, lp_dtype=torch.bfloat16)])
    pack = tail[4]

    def stem_att():
        out = lp.new_image(B, 32, 32, 64, dt, DEV)
        attn = torch.empty(B, T, 32, 32, device=DEV)
        lp.stem(img, wst, scs, shs, out=out, att=lp.AttFuse(pack, 3, 0, T, True, False, 32, attn))
        return torch.cat([out.float().flatten(), attn.flatten()])
    v["lp stem + fused attention"] = stem_att

    def up_att():
        out = lp.new_image(B, 64, 64, 64, dt, DEV)
        attn = torch.empty(B, T, 64, 64, device=DEV)
        _, part = lp.upconv_glu_head(xl, wpu, 64, 64, sc64, sh64, hwp, 3, out=out, att=lp.AttFuse(pack, 3, 1, T, True, False, 32, attn))
        return torch.cat([out.float().flatten(), attn.flatten(), part.flatten()])
    v["lp upBlock + fused head + fused attention @32->64"] = up_att
    x16 = lp.from_nchw(R(B, 32, 32, 32), "f16")

    def conv_():
        o = lp.new_image(B, 32, 32, 32, dt, DEV)
        lp.convert(x16, o)
        return o.float().flatten()
    v["lp convert f16 -> bf16"] = conv_
    xt = lp.from_nchw(R(B, 32, 64, 64), dt)
    wt3 = lp.pack_to3_weight(R(3, 32, 5, 5) * 0.1, dt)
    v["lp conv_to3 5x5 tanh @64^2"] = lambda: lp.conv_to3(xt, wt3, 5, True, addend, 0.5).flatten()
    return v


def test_results_do_not_depend_on_the_neighbour_kernel():
    g = torch.Generator().manual_seed(11)
    victims, neighbours = _victims(g), _neighbours(g)
    side = torch.cuda.Stream()
    bad = []
    for vn, vf in victims.items():
        ref = vf().clone()
        torch.cuda.synchronize()
        assert torch.equal(vf(), ref), "%s is not reproducible on its own" % vn
        for nn, nf in neighbours.items():
            torch.cuda.synchronize()
            for it in range(12):
                with torch.cuda.stream(side):
                    for _ in range(3):
                        nf()
                out = vf()
                with torch.cuda.stream(side):
                    for _ in range(3):
                        nf()
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    bad.append((vn, nn, it, float((out - ref).abs().max())))
                    break
    assert not bad, "results changed beside another kernel: %s" % bad


def test_training_kernels_beside_mfma_neighbours():
    """The training kernels run beside the MFMA-bound weight gradients of the side stream (and the discriminators' streams).
    Same check for them: bit-identical results with a matrix-pipe-bound neighbour on the other stream."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(5)
    R = lambda *sh: torch.randn(*sh, generator=g).to(DEV)            # noqa: E731
    B, C, HW = 8, 64, 64
    raw, gam, bet = R(B, C, HW, HW), torch.rand(C, generator=g).to(DEV) + 0.5, R(C) * 0.1
    dout = R(B, C // 2, HW, HW)

    def bn():
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        y, st = ops.bn_train_fwd(raw, gam, bet, 1e-5, 0.1, rm, rv, 1, None, None)
        draw, dg, db = ops.bn_train_bwd(dout, raw, st, 1)
        return torch.cat([t.flatten() for t in (y, st, rm, rv, draw, dg, db)])
    x3, w3, dy3 = R(B, 32, 64, 64), R(3, 32, 5, 5) * 0.1, R(B, 3, 64, 64)
    out3 = ops.conv_to3(x3, w3, tanh_axpy=True, addend=R(B, 3, 64, 64), alpha=0.5)

    def to3_bwd():
        dx, dw = ops.conv_to3_bwd(dy3, out3, None, 0.5, x3, w3, True, True, True)
        return torch.cat([dx.flatten(), dw.flatten()])
    xs_, dys_ = R(B, 64, 64, 64), R(B, 64, 64, 64)

    def wgrad():
        dw = torch.empty(64, 64, 3, 3, device=DEV)
        ops.conv3x3_wgrad(dys_, xs_, False, True, out=dw)
        return dw.flatten()
    victims = {"BatchNorm train fwd + bwd": bn, "conv_to3 backward": to3_bwd, "Winograd weight gradient": wgrad}
    neighbours = _neighbours(g, B=16)
    side = torch.cuda.Stream()
    bad = []
    for vn, vf in victims.items():
        ref = vf().clone()
        torch.cuda.synchronize()
        assert torch.equal(vf(), ref), "%s is not reproducible on its own" % vn
        for nn, nf in neighbours.items():
            for it in range(10):
                with torch.cuda.stream(side):
                    for _ in range(3):
                        nf()
                out = vf()
                with torch.cuda.stream(side):
                    for _ in range(3):
                        nf()
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    bad.append((vn, nn, it, float((out - ref).abs().max())))
                    break
    assert not bad, "results changed beside another kernel: %s" % bad
