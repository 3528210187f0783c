"""GPU: training path (batch-statistics BN, backward kernels) against torch autograd over the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tgsr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, atol, rtol=1e-4):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


CASES = [
    # B, Cin, H, W, Cout, glu, up, res
    (3, 64, 16, 32, 128, True, False, False),
    (3, 64, 16, 32, 64, False, False, True),
    (2, 64, 16, 16, 64, True, True, False),
    (3, 32, 9, 20, 64, True, True, False),      # upBlock wgrad in the Winograd domain: ragged, 32-channel input
    (2, 64, 24, 40, 128, True, True, False),    # ... two cout groups, several chunks per row
    (4, 32, 32, 32, 64, True, False, False),
    (4, 32, 32, 32, 32, False, False, True),
    (2, 32, 16, 32, 32, False, False, False),
    (2, 3, 32, 32, 64, True, False, False),
    (16, 64, 32, 32, 128, True, False, False),
    (2, 32, 13, 20, 64, True, False, False),     # Winograd-domain wgrad: odd height (partial tiles), 32-channel input
    (1, 64, 5, 40, 64, False, False, True),      # ... several chunks per tile row
    (3, 32, 11, 24, 32, False, False, True),     # Winograd-domain wgrad of a 32 -> 32 layer (NetG_highweight's ResBlocks), ragged
    (2, 64, 8, 40, 32, False, False, False),     # ... 64 -> 32: two ci blocks, one co block
    (2, 32, 12, 20, 32, True, True, False),      # upBlock(32, 16) / GF_DIM = 16: Cout = 32 has no up-sample-aware wgrad -> direct
    (2, 32, 8, 16, 96, True, True, False),       # ... Cout % 64 == 32 likewise
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,glu,up,res", CASES)
def test_conv_bn_act_train_fwd_bwd(B, Cin, H, W, Cout, glu, up, res):
    from tgsr_amd.autograd import ConvBnAct
    g = torch.Generator().manual_seed(B + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    gamma = 1 + 0.2 * torch.randn(Cout, generator=g)
    beta = 0.1 * torch.randn(Cout, generator=g)
    rm = 0.1 * torch.randn(Cout, generator=g)
    rv = 0.5 + torch.rand(Cout, generator=g)
    co = Cout // 2 if glu else Cout
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    r = torch.randn(B, co, Ho, Wo, generator=g) if res else None
    dy = torch.randn(B, co, Ho, Wo, generator=g)

    # reference: torch autograd over the oracle's functions (CPU)
    xr, wr, gr, br = (t.clone().requires_grad_() for t in (x, w, gamma, beta))
    rr = r.clone().requires_grad_() if res else None
    sd = {"weight": gr, "bias": br, "running_mean": rm.clone(), "running_var": rv.clone()}
    upd = {}
    xi = xr.repeat_interleave(2, 2).repeat_interleave(2, 3) if up else xr
    y = O.batch_norm(F.conv2d(xi, wr, None, 1, 1), sd, "", training=True, update=upd)
    y = O.glu(y) if glu else y
    y = y + rr if res else y
    y.backward(dy)

    xd, wd, gd, bd = (t.to(DEV).requires_grad_() for t in (x, w, gamma, beta))
    rd = r.to(DEV).requires_grad_() if res else None
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    out = ConvBnAct.apply(xd, wd, gd, bd, rmd, rvd, rd, glu, up, 0.1, 1e-5, nbt)
    assert int(nbt) == 1
    out.backward(dy.to(DEV))
    close(out, y, atol=5e-5)
    close(rmd, upd["running_mean"], atol=1e-5)
    close(rvd, upd["running_var"], atol=1e-5)
    scale = float(dy.numel()) ** 0.5
    if Cin >= 32:                       # the stem convs take the data image: no input gradient needed, but check anyway
        close(xd.grad, xr.grad, atol=2e-4, rtol=1e-3)
    close(wd.grad, wr.grad, atol=2e-5 * scale, rtol=2e-3)
    close(gd.grad, gr.grad, atol=2e-5 * scale, rtol=2e-3)
    close(bd.grad, br.grad, atol=2e-5 * scale, rtol=2e-3)
    if res:
        close(rd.grad, rr.grad, atol=1e-6)


@pytest.mark.parametrize("B,Cin,H,W,K,act", [(2, 32, 32, 64, 3, False), (2, 32, 32, 64, 5, True), (3, 32, 19, 70, 5, True),
                                              (1, 32, 64, 64, 3, False), (2, 20, 16, 16, 5, True),
                                              # MFMA weight gradient: ragged tile (W = 48, H % 4 != 0), Cin 16 / 64, and
                                              # the taller wave tiles big launches pick (2 and 4 rows per wave)
                                              (2, 64, 41, 48, 3, False), (2, 16, 24, 32, 5, True), (3, 48, 18, 80, 5, False),
                                              (8, 32, 256, 256, 5, True), (16, 32, 256, 256, 3, False)])
def test_conv_to3_backward(B, Cin, H, W, K, act):
    from tgsr_amd.autograd import ConvTo3
    g = torch.Generator().manual_seed(K * 10 + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(3, Cin, K, K, generator=g) / (K * Cin ** 0.5)
    add = torch.randn(B, 3, H, W, generator=g) if act else None
    dy = torch.randn(B, 3, H, W, generator=g)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    ar = add.clone().requires_grad_() if act else None
    y = F.conv2d(xr, wr, None, 1, K // 2)
    y = torch.tanh(y) + 0.5 * ar if act else y
    y.backward(dy)
    xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    ad = add.to(DEV).requires_grad_() if act else None
    out = ConvTo3.apply(xd, wd, ad, act, 0.5)
    out.backward(dy.to(DEV))
    close(out, y, atol=2e-5)
    close(xd.grad, xr.grad, atol=2e-5, rtol=1e-3)
    close(wd.grad, wr.grad, atol=3e-5 * float(B * H * W) ** 0.5, rtol=2e-3)
    if act:
        close(ad.grad, ar.grad, atol=1e-6)


@pytest.mark.parametrize("B,idf,r,T_,correct", [(3, 32, 16, 6, False), (3, 32, 32, 18, True), (2, 64, 16, 9, False),
                                                (5, 32, 20, 7, False), (16, 32, 64, 14, False)])
def test_word_attention_backward(B, idf, r, T_, correct):
    from tgsr_amd.autograd import WordAttention
    g = torch.Generator().manual_seed(B * 7 + r)
    h = torch.randn(B, idf, r, r, generator=g)
    words = torch.randn(B, 64, T_, generator=g)
    w = torch.randn(idf, 64, 1, 1, generator=g) / 8
    lens = torch.randint(1, T_ + 1, (B,), generator=g)
    lens[0] = T_
    mask = torch.arange(T_)[None, :] >= lens[:, None]
    dc = torch.randn(B, idf, r, r, generator=g)
    hr, wr_, wdr = h.clone().requires_grad_(), w.clone().requires_grad_(), words.clone().requires_grad_()
    c_ref, _ = O.word_attention(hr, wdr, wr_, mask, correct_mask=correct)
    c_ref.backward(dc)
    hd, wd, wdd = h.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), words.to(DEV).requires_grad_()
    c, attn = WordAttention.apply(hd, wdd, wd, mask.to(DEV), correct)
    c.backward(dc.to(DEV))
    close(c, c_ref, atol=5e-5)
    close(hd.grad, hr.grad, atol=5e-5, rtol=1e-3)
    close(wd.grad, wr_.grad, atol=2e-4 * r, rtol=2e-3)
    close(wdd.grad, wdr.grad, atol=2e-4 * r, rtol=2e-3)


def test_generators_train_mode_golden_and_gradients(nets_small):
    """Whole x8 generators in train mode (batch-stat BN): forward vs the reference golden, every parameter gradient
    of MSE + KL vs torch autograd over the oracle."""
    from conftest import split_sd
    from tgsr_amd import model
    from tgsr_amd.miscc import losses
    from tgsr_amd.miscc.config import cfg, cfg_reset
    g = nets_small
    cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 64
    try:
        T = lambda a, d=DEV: torch.from_numpy(np.asarray(a)).to(d)
        sdL, sdH = split_sd(g, "GL."), {k: v for k, v in split_sd(g, "GH.").items() if k != "a"}   # `a`: model.py:246-248
        gl, gh = model.G_SR_NET_low(), model.NetG_highweight(weightmap=False, low="lr")
        gl.load_state_dict(sdL); gh.load_state_dict(sdH)
        gl.to(DEV).train(); gh.to(DEV).train()
        words, sent, mask = T(g["train.words_emb"]), T(g["train.sent_emb"]), T(g["train.mask"])
        LR, LRb = T(g["LR"]), T(g["LRb"])
        gen = torch.Generator().manual_seed(5)
        hr = [torch.rand(3, 3, s, s, generator=gen) * 2 - 1 for s in (32, 64, 128)]
        imgs, atts, mu, lv = gl(LR, sent, words, mask)
        fine, a, one = gh(LR, imgs, LRb)
        for i in range(3):
            close(imgs[i], g["train.fake%d" % i], atol=2e-4)
            close(fine[i], g["train.fine%d" % i], atol=2e-4)
        loss = losses.MSE(imgs, [h.to(DEV) for h in hr]) + losses.MSE(fine, [h.to(DEV) for h in hr]) + losses.KL_loss(mu, lv)
        loss.backward()
        # oracle + torch autograd on CPU
        pL = {k: (v.clone().requires_grad_() if v.dtype.is_floating_point and "running" not in k else v.clone())
              for k, v in sdL.items()}
        pH = {k: (v.clone().requires_grad_() if v.dtype.is_floating_point and "running" not in k else v.clone())
              for k, v in sdH.items()}
        ri, ra, rmu, rlv = O.g_sr_net_low(pL, T(g["LR"], "cpu"), T(g["train.sent_emb"], "cpu"), T(g["train.words_emb"], "cpu"),
                                          T(g["train.mask"], "cpu"), training=True)
        rf, _, _ = O.netg_highweight(pH, T(g["LR"], "cpu"), ri, T(g["LRb"], "cpu"), "lr", training=True)
        rloss = O.mse(ri, hr) + O.mse(rf, hr) + O.kl_loss(rmu, rlv)
        rloss.backward()
        close(loss, rloss, atol=1e-4)
        worst = 0.0
        for net, ref in ((gl, pL), (gh, pH)):
            for k, p in net.named_parameters():
                r = ref[k].grad
                assert p.grad is not None, k
                denom = float(r.abs().max()) + 1e-6
                err = float((p.grad.cpu() - r).abs().max()) / denom
                worst = max(worst, err)
                assert err < 2e-4, (k, err, denom)       # measured worst over all 225 tensors: 1.6e-5 (round 4); bound ~10x
        print("worst relative parameter-gradient error", worst)
    finally:
        cfg_reset()


def test_side_stream_weight_gradients_change_nothing(monkeypatch):
    """SRTrainer issues the generators' weight gradients on a side stream while a step's backward is in flight
    (autograd.on_wgrad_stream) and joins it before anything reads them.  Same weights, same batch: one backward with the
    side stream must give bit-identical gradients to one backward on a single stream (same kernels, same order per
    stream; a missing join or a gradient accumulated on the wrong stream would show up as a stale or torn value)."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.synthetic import synthetic_batch
    from tgsr_amd.train import SRTrainer
    cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256
    try:
        B = 4
        cap, lens, LR, LRb = synthetic_batch(B, seed=3)
        gen = torch.Generator().manual_seed(21)
        hr = [(torch.rand(B, 3, s, s, generator=gen) * 2 - 1).to(DEV) for s in (64, 128, 256)]
        args = (cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV), hr)
        a = SRTrainer(41, device=DEV)
        assert a._wside is not None
        monkeypatch.setenv("TGSR_WGRAD_SIDE", "0")
        b = SRTrainer(41, device=DEV)
        assert b._wside is None
        for m, n in ((a.text_encoder, b.text_encoder), (a.netGL, b.netGL), (a.netGH, b.netGH)):
            n.load_state_dict(m.state_dict())
        grads = []
        for t in (a, b):
            torch.manual_seed(5)                      # CA_NET draws its normals from the global generator
            t._zero(t.bucket)
            loss, _, _ = t.loss(*args)
            with t._wgrad_side():
                loss.backward()
            t.bucket.end_step()
            torch.cuda.synchronize()
            grads.append({n: p.grad.clone() for n, p in list(t.netGL.named_parameters()) + list(t.netGH.named_parameters())})
        for n, ga in grads[0].items():
            assert torch.equal(ga, grads[1][n]), n
    finally:
        cfg_reset()


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 32, 16, 32, 64), (3, 64, 12, 40, 128), (1, 32, 32, 32, 32), (2, 8, 6, 4, 96)])
def test_batch_statistics_in_the_conv_epilogue(B, Cin, H, W, Cout):
    """tgsr_wino_conv3x3_stats_fwd: the same raw output as the plain Winograd convolution (bit for bit) plus one
    (sum, sum of squares) pair per channel and wave tile - ragged tiles contribute only their in-image pixels;
    tgsr_bn_train_fwd_from_stats on them = tgsr_bn_train_fwd with its own statistics pass (tolerance: the order of the
    fp32 partial sums differs), running statistics and num_batches_tracked included."""
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B * 7 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).to(DEV)
    up = ops.pack_wino_weight(w, False, False)
    ref = ops.conv3x3_wino(x, up, Cout, None, None, False, None)
    out, part = ops.conv3x3_wino_stats(x, up, Cout)
    assert torch.equal(out, ref) and part.shape[0] == Cout and part.shape[2] == 2
    s = part.double().sum(1)
    close(s[:, 0], ref.double().sum((0, 2, 3)), atol=1e-3, rtol=1e-5)
    close(s[:, 1], (ref.double() ** 2).sum((0, 2, 3)), atol=1e-3, rtol=1e-5)
    if H * W % 4 == 0:
        for act in (0, 1):
            gam, bet = torch.rand(Cout, generator=g).to(DEV) + 0.5, torch.randn(Cout, generator=g).to(DEV)
            res = []
            for sp in (None, part):
                rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
                nbt = torch.zeros((), dtype=torch.int64, device=DEV)
                y, st = ops.bn_train_fwd(ref, gam, bet, 1e-5, 0.1, rm, rv, act, None, nbt, stat_partial=sp)
                res.append((y, st, rm, rv, nbt))
            for a_, b_ in zip(res[0][:4], res[1][:4]):
                close(a_, b_, atol=2e-5, rtol=2e-5)
            assert int(res[1][4]) == 1


def test_pack_cache_changes_nothing_over_steps(monkeypatch):
    """SRTrainer re-packs the generators' conv weights behind the optimizer on a stream of its own (autograd.PackCache)
    instead of in front of every convolution.  Four optimisation steps with and without the cache, same initial weights
    and batches: bit-identical losses and parameters (a pack served one step late, or rewritten while a kernel of the
    previous backward still reads it, would show up here); an in-place write outside the optimizer re-packs on the spot."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.synthetic import synthetic_batch
    from tgsr_amd.train import SRTrainer
    cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256
    try:
        B = 4
        a = SRTrainer(41, device=DEV)
        assert a._packs is not None
        monkeypatch.setenv("TGSR_PACK_CACHE", "0")
        b = SRTrainer(41, device=DEV)
        assert b._packs is None
        for m, n in ((a.text_encoder, b.text_encoder), (a.netGL, b.netGL), (a.netGH, b.netGH)):
            n.load_state_dict(m.state_dict())
        losses_ = [[], []]
        for k, t in enumerate((a, b)):
            for it in range(4):
                cap, lens, LR, LRb = synthetic_batch(B, seed=3 + it)
                gen = torch.Generator().manual_seed(21 + it)
                hr = [(torch.rand(B, 3, s, s, generator=gen) * 2 - 1).to(DEV) for s in (64, 128, 256)]
                torch.manual_seed(5 + it)
                if it == 2:                              # a write behind the cache's back: version counter moves
                    with torch.no_grad():
                        t.netGH.convin[0].weight.mul_(1.01)
                        t.netGL.h_net1.upsample[1].weight.add_(0.001)
                losses_[k].append(float(t.step(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV), hr)))
        torch.cuda.synchronize()
        assert losses_[0] == losses_[1], losses_
        assert len(a._packs.entries) > 40
        for (n, p), (_, q) in zip(list(a.netGL.named_parameters()) + list(a.netGH.named_parameters()),
                                  list(b.netGL.named_parameters()) + list(b.netGH.named_parameters())):
            assert torch.equal(p, q), n
    finally:
        cfg_reset()


def test_graph_replayed_generator_update_equals_eager_one():
    """train.SRTrainer replays the generators' update (zero the bucket, forward, MSE + KL, backward with the weight gradients on
    their side branch, fused Adam, the re-pack of every cached weight pack, EMA) from a hipGraph after GRAPH_G_WARMUP eager steps,
    one capture per batch shape.  Two trainers from one initialisation - one replaying, one eager (same capturable Adam) - take
    the same eight steps on changing batches of two caption widths: losses, parameters, running statistics and the EMA copies
    must be bit-identical, the replaying trainer must really be replaying, and a replayed update must move the parameters'
    version counters (the caches of derived tensors key on them)."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd import train
    cfg_reset()
    cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
    try:
        B = 4
        trs = []
        for graphs in (True, False):
            torch.manual_seed(5)
            tr = train.SRTrainer(41, device=DEV)
            assert tr._graph_capable and tr._packs is not None   # capturable Adam on both
            tr._graph_g = graphs                                 # (the default policy keeps the generator-only step eager: train.py)
            trs.append(tr)
        out = [[], []]
        for step in range(8):
            cap, lens, _LR, LRb = O.synthetic_batch(B, seed=40 + step % 2)      # two caption widths in turn
            g = torch.Generator().manual_seed(step)
            LR = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
            hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).to(DEV) for s in (64, 128, 256)]
            for k, tr in enumerate(trs):
                torch.manual_seed(100 + step)                    # CA_NET's noise
                v0 = tr.params[0]._version
                out[k].append(float(tr.step(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV), hr)))
                assert tr.params[0]._version > v0
        torch.cuda.synchronize()
        caps = list(trs[0]._ggraphs.values())
        assert caps and all(isinstance(c, dict) for c in caps), "the update was not captured: %r" % (caps,)
        assert not trs[1]._ggraphs
        assert out[0] == out[1], (out[0], out[1])
        assert out[0][-1] < out[0][0]
        for a, b in zip((trs[0].netGL, trs[0].netGH), (trs[1].netGL, trs[1].netGH)):
            for (ka, va), (_kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
                assert torch.equal(va, vb), ka
        for a, b in zip(trs[0].avg_param_G, trs[1].avg_param_G):
            assert torch.equal(a, b)
        # the eval-mode forward after replayed updates sees the updated weights (version-keyed caches were told)
        cap, lens, LR, LRb = O.synthetic_batch(B, seed=77)
        ims = []
        for tr in trs:
            tr.netGL.eval(); tr.netGH.eval()
            with torch.no_grad():
                w, s_, m = tr._text(cap.to(DEV), lens.tolist())
                ims.append(tr._forward_nets(LR.to(DEV), LRb.to(DEV), w, s_, m)[1][2])
        assert torch.equal(ims[0], ims[1])
    finally:
        cfg_reset()


def test_measured_graph_policy_settles_and_changes_no_bit():
    """TGSR_GRAPH_G=auto (the default): the trainer times GRAPH_G_TRIALS eager steps, captures, times as many replayed ones and
    keeps the faster form - whichever it keeps, every step along the way (eager, capturing, replayed, chosen) leaves the same
    bits as a trainer pinned to the eager form; the report says what was measured."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd import train
    cfg_reset()
    cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
    try:
        B = 4
        trs = []
        for pinned in (False, True):
            torch.manual_seed(5)
            tr = train.SRTrainer(41, device=DEV)
            assert tr._auto is not None and tr.graph_policy["mode"] == "auto"
            if pinned:
                tr._graph_g = False
            trs.append(tr)
        out = [[], []]
        forms = []
        cap, lens, _LR, LRb = O.synthetic_batch(B, seed=40)
        for step in range(train.GRAPH_G_SETTLED + 2):
            g = torch.Generator().manual_seed(step)
            LR = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
            hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).to(DEV) for s in (64, 128, 256)]
            for k, tr in enumerate(trs):
                torch.manual_seed(100 + step)
                out[k].append(float(tr.step(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV), hr)))
            forms.append(None if trs[0]._auto is None else trs[0]._auto["form"])
        a = trs[0]
        assert forms[:train.GRAPH_G_WARMUP + train.GRAPH_G_TRIALS] == ["eager"] * (train.GRAPH_G_WARMUP + train.GRAPH_G_TRIALS)
        assert forms[train.GRAPH_G_SETTLED - 1] is None and a._auto is None, forms          # settled
        pol = a.graph_policy
        assert pol["chosen"] in ("eager", "replay") and pol["eager_ms"] > 0 and pol["replay_ms"] > 0, pol
        assert a._graph_g == (pol["chosen"] == "replay") and a._ggraphs
        assert out[0] == out[1], (out[0], out[1])
        for x, y in zip(a.params, trs[1].params):
            assert torch.equal(x, y)
    finally:
        cfg_reset()


def test_train_step_decreases_loss_and_updates_running_stats(nets_small):
    from conftest import split_sd
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.train import SRTrainer
    g = nets_small
    cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 64
    try:
        tr = SRTrainer(41, device=DEV, lr=1e-3)
        tr.text_encoder.load_state_dict(split_sd(g, "E."))
        T = lambda a: torch.from_numpy(np.asarray(a)).to(DEV)
        gen = torch.Generator().manual_seed(9)
        hr = [(torch.rand(3, 3, s, s, generator=gen) * 2 - 1).to(DEV) for s in (32, 64, 128)]
        rm0 = tr.netGH.convin[1].running_mean.clone()
        ls = [float(tr.step(T(g["captions"]), g["cap_lens"].tolist(), T(g["LR"]), T(g["LRb"]), hr)) for _ in range(6)]
        assert ls[-1] < ls[0], ls
        assert not torch.equal(rm0, tr.netGH.convin[1].running_mean)
        assert int(tr.netGH.convin[1].num_batches_tracked) == 6
    finally:
        cfg_reset()


@pytest.mark.parametrize("B,T,ntoken,ninput,H", [(5, 9, 30, 40, 32), (16, 18, 41, 300, 128), (2, 3, 10, 7, 64)])
def test_rnn_encoder_train_backward_vs_oracle(B, T, ntoken, ninput, H):
    """RNN_ENCODER.train(): embedding -> (dropout p=0) -> HIP LSTM forward + BPTT; every parameter gradient against
    torch autograd through the oracle's explicit packed-sequence recurrence (fp64)."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.util import RNN_ENCODER
    cfg_reset()
    cfg.TEXT.WORDS_NUM = T
    g = torch.Generator().manual_seed(B * 100 + T)
    enc = RNN_ENCODER(ntoken, ninput=ninput, drop_prob=0.0, nhidden=2 * H).to(DEV)
    enc.train()
    lens = sorted(torch.randint(1, T + 1, (B,), generator=g).tolist(), reverse=True)
    lens[0] = T
    cap = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lens):
        cap[b, :n] = torch.randint(1, ntoken, (n,), generator=g)
    gw = torch.randn(B, 2 * H, T, generator=g, dtype=torch.float64)
    gs = torch.randn(B, 2 * H, generator=g, dtype=torch.float64)
    # oracle, fp64
    sd = {k: v.detach().cpu().double().requires_grad_() for k, v in enc.state_dict().items()}
    ow, os_ = O.rnn_encoder(sd, cap, lens)
    ((ow * gw).sum() + (os_ * gs).sum()).backward()
    # HIP
    words, sent = enc(cap.to(DEV), lens, enc.init_hidden(B))
    close(words, ow.float(), atol=1e-5)
    close(sent, os_.float(), atol=1e-5)
    ((words * gw.float().to(DEV)).sum() + (sent * gs.float().to(DEV)).sum()).backward()
    for name, p in enc.named_parameters():
        ref = sd[name].grad.float()
        close(p.grad, ref, atol=2e-5 * max(1.0, float(ref.abs().max())), rtol=1e-3)
    cfg_reset()


@pytest.mark.parametrize("B,T,ntoken,ninput,H", [(4, 9, 41, 300, 128), (3, 6, 30, 40, 64), (5, 12, 60, 24, 32), (1, 1, 7, 16, 32)])
def test_rnn_encoder_gru_train_backward_vs_oracle(B, T, ntoken, ninput, H):
    """RNN_ENCODER with cfg.RNN_TYPE == 'GRU' in training mode (util.py:207-211, 233-260: the reference trains whichever cell the
    config names): embedding -> (dropout p=0) -> HIP GRU forward (tgsr_bigru_train_fwd) + BPTT (tgsr_bigru_bwd); outputs and every
    parameter gradient against torch autograd through the oracle's explicit packed-sequence GRU recurrence in fp64."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.util import RNN_ENCODER
    cfg_reset()
    cfg.TEXT.WORDS_NUM = T
    cfg.RNN_TYPE = 'GRU'
    try:
        g = torch.Generator().manual_seed(B * 100 + T)
        enc = RNN_ENCODER(ntoken, ninput=ninput, drop_prob=0.0, nhidden=2 * H).to(DEV)
        enc.train()
        lens = sorted(torch.randint(1, T + 1, (B,), generator=g).tolist(), reverse=True)
        lens[0] = T
        cap = torch.zeros(B, T, dtype=torch.int64)
        for b, n in enumerate(lens):
            cap[b, :n] = torch.randint(1, ntoken, (n,), generator=g)
        gw = torch.randn(B, 2 * H, T, generator=g, dtype=torch.float64)
        gs = torch.randn(B, 2 * H, generator=g, dtype=torch.float64)
        sd = {k: v.detach().cpu().double().requires_grad_() for k, v in enc.state_dict().items()}
        ow, os_ = O.rnn_encoder_gru(sd, cap, lens)
        ((ow * gw).sum() + (os_ * gs).sum()).backward()
        words, sent = enc(cap.to(DEV), lens, enc.init_hidden(B))
        close(words, ow.float(), atol=1e-5)
        close(sent, os_.float(), atol=1e-5)
        ((words * gw.float().to(DEV)).sum() + (sent * gs.float().to(DEV)).sum()).backward()
        for name, p in enc.named_parameters():
            ref = sd[name].grad.float()
            close(p.grad, ref, atol=2e-5 * max(1.0, float(ref.abs().max())), rtol=1e-3)
        # eval mode on the same weights still goes through the per-token table and agrees with the training forward
        enc.eval()
        with torch.no_grad():
            we, se = enc(cap.to(DEV), lens, enc.init_hidden(B))
        close(we, words.detach(), atol=1e-5)
        close(se, sent.detach(), atol=1e-5)
    finally:
        cfg_reset()


def test_damsm_pretrain_step_decreases_loss():
    """One pretrain_DAMSM-style step chain (pretrain_DAMSM.py:60-100): RNN_ENCODER.train() + CNN_ENCODER heads on
    synthetic trunk features, words_loss + sent_loss, Adam, grad-clip 0.25 - all gradients from HIP kernels; the
    loss must go down on a fixed batch."""
    from tgsr_amd.miscc import losses
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.util import CNN_ENCODER, RNN_ENCODER
    cfg_reset()
    cfg.TEXT.WORDS_NUM = 12
    cfg.TRAIN.FLAG = True
    B, T, nef = 6, 12, 256
    g = torch.Generator().manual_seed(11)
    text = RNN_ENCODER(50, nhidden=nef).to(DEV).train()
    image = CNN_ENCODER(nef, trunk=torch.nn.Identity()).to(DEV).train()
    lens = [12, 10, 9, 7, 4, 2]
    cap = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lens):
        cap[b, :n] = torch.randint(1, 50, (n,), generator=g)
    feats = torch.randn(B, 768, 17, 17, generator=g).to(DEV)
    pooled = torch.randn(B, 2048, generator=g).to(DEV)
    labels = torch.arange(B, device=DEV)
    params = [p for p in list(text.parameters()) + [image.emb_features.weight, image.emb_cnn_code.weight,
                                                     image.emb_cnn_code.bias]]
    opt = torch.optim.Adam(params, lr=2e-3, betas=(0.5, 0.999))
    torch.manual_seed(0)
    hist = []
    for _ in range(8):
        opt.zero_grad()
        words_features, sent_code = image.heads(feats, pooled)
        words_emb, sent_emb = text(cap.to(DEV), lens, text.init_hidden(B))
        w0, w1, _ = losses.words_loss(words_features, words_emb, labels, lens, None, B)
        s0, s1 = losses.sent_loss(sent_code, sent_emb, labels, None, B)
        loss = w0 + w1 + s0 + s1
        loss.backward()
        torch.nn.utils.clip_grad_norm_(text.parameters(), 0.25)      # pretrain_DAMSM.py:96-97
        opt.step()
        hist.append(float(loss.detach()))
    assert all(np.isfinite(hist)) and hist[-1] < hist[0] - 0.05, hist
    cfg_reset()


def test_damsm_trainer_epoch_protocol():
    """DAMSMTrainer: per-epoch Adam reset, lr x 0.98, clipped text-encoder gradients, loss goes down."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.train import DAMSMTrainer
    cfg_reset()
    cfg.TEXT.WORDS_NUM = 10
    cfg.TRAIN.FLAG = True
    tr = DAMSMTrainer(40, device=DEV, lr=2e-3)
    g = torch.Generator().manual_seed(5)
    B = 5
    lens = [10, 8, 6, 3, 1]
    cap = torch.zeros(B, 10, dtype=torch.int64)
    for b, n in enumerate(lens):
        cap[b, :n] = torch.randint(1, 40, (n,), generator=g)
    feats, pooled = torch.randn(B, 768, 17, 17, generator=g).to(DEV), torch.randn(B, 2048, generator=g).to(DEV)
    torch.manual_seed(1)
    first = float(tr.step_features(feats, pooled, cap.to(DEV), lens))
    for _ in range(5):
        last = float(tr.step_features(feats, pooled, cap.to(DEV), lens))
    opt0 = tr.opt
    tr.end_epoch(); tr.start_epoch()
    assert tr.opt is not opt0 and abs(tr.lr - 2e-3 * 0.98) < 1e-12
    assert np.isfinite(last) and last < first
    total = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in tr.text_encoder.parameters()))
    assert float(total) <= cfg.TRAIN.RNN_GRAD_CLIP * 1.001
    cfg_reset()


def test_damsm_trainer_evaluate_snapshot_resume(tmp_path):
    """pretrain_DAMSM.py:133-163 (evaluate: eval mode, <= 51 batches, sums divided by the last step index), :286-291
    (snapshots) and :172-186 (resume: both encoders + the epoch parsed from the file name) against the oracle's losses."""
    from oracle import tgsr_oracle as O
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.train import DAMSMTrainer
    cfg_reset()
    cfg.TEXT.WORDS_NUM = 10
    cfg.TRAIN.FLAG = True
    cfg.TRAIN.SNAPSHOT_INTERVAL = 5
    tr = DAMSMTrainer(40, device=DEV, lr=2e-3)
    g = torch.Generator().manual_seed(11)
    B, lens = 4, [10, 7, 4, 2]

    def batch():
        cap = torch.zeros(B, 10, dtype=torch.int64)
        for b, n in enumerate(lens):
            cap[b, :n] = torch.randint(1, 40, (n,), generator=g)
        return (torch.randn(B, 768, 17, 17, generator=g).to(DEV), torch.randn(B, 2048, generator=g).to(DEV), cap.to(DEV),
                lens, None)
    val = [batch() for _ in range(3)]
    tr.step_features(*val[0][:4])
    s_loss, w_loss = tr.evaluate_features(val)
    assert not tr.text_encoder.training and not tr.image_encoder.training          # left in eval mode, like the reference
    # the same three batches through the oracle's restatement of the encoders' heads and the two losses
    sdE = {k: v.detach().cpu() for k, v in tr.text_encoder.state_dict().items()}
    ie = tr.image_encoder
    s_ref = w_ref = 0.0
    for feats, pooled, cap, ln, _cls in val:
        wf = F.conv2d(feats.cpu(), ie.emb_features.weight.detach().cpu())
        sc = F.linear(pooled.cpu(), ie.emb_cnn_code.weight.detach().cpu(), ie.emb_cnn_code.bias.detach().cpu())
        we, se = O.rnn_encoder(sdE, cap.cpu(), ln)
        sm = cfg.TRAIN.SMOOTH
        w0, w1, _ = O.words_loss(wf, we, torch.arange(B), ln, None, B, sm.GAMMA1, sm.GAMMA2, sm.GAMMA3)
        s0, s1 = O.sent_loss(sc, se, torch.arange(B), None, B, sm.GAMMA3)
        w_ref += float(w0 + w1)
        s_ref += float(s0 + s1)
    assert abs(s_loss - s_ref / 2) < 2e-4 * max(1.0, abs(s_ref)) and abs(w_loss - w_ref / 2) < 2e-4 * max(1.0, abs(w_ref))
    assert tr.evaluate_features(val[:1]) == (float("inf"), float("inf"))          # one batch: divided by step index 0
    assert tr.snapshot_due(10, 600) and tr.snapshot_due(600, 600) and not tr.snapshot_due(7, 600)
    pi, pt = tr.snapshot(str(tmp_path), 10)
    assert pt.endswith("text_encoder10.pth") and pi.endswith("image_encoder10.pth")
    tr2 = DAMSMTrainer(40, device=DEV, lr=2e-3)
    assert tr2.resume(pt) == 11 and tr2.text_encoder.training
    for a, b in zip(tr.text_encoder.state_dict().values(), tr2.text_encoder.state_dict().values()):
        assert torch.equal(a, b)
    for a, b in zip(tr.image_encoder.state_dict().values(), tr2.image_encoder.state_dict().values()):
        assert torch.equal(a, b)
    s2, w2 = tr2.evaluate_features(val)
    assert s2 == s_loss and w2 == w_loss                                          # same parameters, same kernels: same bits
    assert tr2.resume('') == 0
    cfg_reset()


def test_sr_trainer_with_damsm_term():
    """SRTrainer with an image encoder: the DAMSM ranking term (generator_loss, losses.py:375-386) reaches the
    generators through the HIP DAMSM backward; one step runs and changes the result of the pixel-only loss."""
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.synthetic import synthetic_batch
    from tgsr_amd.train import SRTrainer
    from tgsr_amd.util import CNN_ENCODER
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 256
    cfg.TREE.BRANCH_NUM = 4
    cfg.TREE.BASE_SIZE = 32

    class Trunk(torch.nn.Module):        # stand-in for the frozen Inception trunk: any differentiable torch module
        def __init__(self):
            super().__init__()
            self.f = torch.nn.Conv2d(3, 768, 1)
            self.p = torch.nn.Linear(3, 2048)

        def forward(self, x):
            f = self.f(F.adaptive_avg_pool2d(x, 17))
            return f, self.p(x.mean((2, 3)))

    torch.manual_seed(3)
    enc = CNN_ENCODER(256, trunk=Trunk()).to(DEV).eval()
    for p in enc.parameters():
        p.requires_grad = False
    cap, lens, LR, LRb = synthetic_batch(2, seed=5)
    cap, LR, LRb, lens = cap.to(DEV), LR.to(DEV), LRb.to(DEV), lens.tolist()
    g = torch.Generator().manual_seed(1)
    hr = [(torch.rand(2, 3, s, s, generator=g) * 2 - 1).to(DEV) for s in (64, 128, 256)]
    losses_ = []
    for with_enc in (False, True):
        torch.manual_seed(7)
        tr = SRTrainer(41, device=DEV, image_encoder=enc if with_enc else None)
        torch.manual_seed(9)
        losses_.append(float(tr.step(cap, lens, LR, LRb, hr)))
        assert all(torch.isfinite(p.grad).all() for p in tr.params)
    assert np.isfinite(losses_).all() and losses_[1] > losses_[0] + 1e-3      # the ranking term is positive
    cfg_reset()


def test_models16_train_mode_gradients():
    """x16 generators (models16.py): weight-tied stages accumulate their gradients over the three uses, the tanh heads
    and NetG_highweight's trainable `a` train on HIP; every parameter gradient vs torch autograd over the oracle."""
    from conftest import load_npz, split_sd
    from tgsr_amd import models16
    from tgsr_amd.miscc import losses
    from tgsr_amd.miscc.config import cfg, cfg_reset
    g = load_npz("nets16_small.npz")
    cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 64; cfg.TREE.BRANCH_NUM = 5
    try:
        sdL = split_sd(g, "GL.")
        gl = models16.G_SR_NET_low()
        gl.load_state_dict(sdL, strict=False)
        torch.manual_seed(4)
        gh = models16.NetG_highweight(weightmap=False, low="lr")
        sdH = {k: v.detach().clone() for k, v in gh.state_dict().items()}
        gl.to(DEV).train(); gh.to(DEV).train()
        gen = torch.Generator().manual_seed(9)
        B = 2
        LR = torch.rand(B, 3, 8, 8, generator=gen) * 2 - 1
        words = torch.randn(B, 64, 6, generator=gen)
        sent = torch.randn(B, 64, generator=gen)
        mask = torch.arange(6)[None, :] >= torch.tensor([[6], [4]])
        hr = [torch.rand(B, 3, s, s, generator=gen) * 2 - 1 for s in (16, 32, 64, 128)]
        imgs, atts, mu, lv = gl(LR.to(DEV), sent.to(DEV), words.to(DEV), mask.to(DEV))
        fine, a, one = gh(LR.to(DEV), imgs, LR.to(DEV))
        hd = [h.to(DEV) for h in hr]
        loss = losses.MSE(imgs, hd) + losses.MSE(fine, hd) + losses.KL_loss(mu, lv)
        loss.backward()
        # oracle: the fixture stores tied tensors once (under the first alias); the oracle reads h_net2 / img_net1
        pL = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
              if v.dtype.is_floating_point else v.cpu() for k, v in sdL.items()}
        pH = {k: (v.clone().requires_grad_() if v.dtype.is_floating_point and "running" not in k else v.clone())
              for k, v in sdH.items()}
        ri, ra, rmu, rlv = O.g_sr_net_low16(pL, LR, sent, words, mask, training=True)
        rf, _, _ = O.netg_highweight16(pH, LR, ri, LR, "lr", training=True)
        rloss = O.mse(ri, hr) + O.mse(rf, hr) + O.kl_loss(rmu, rlv)
        rloss.backward()
        close(loss, rloss, atol=2e-4)
        seen = set()

        def ref_key(k, ref):                         # a tied tensor is listed once, under any of its aliases
            for grp in (("h_net2.", "h_net3.", "h_net4."), ("img_net1.", "img_net2.", "img_net3.", "img_net4.")):
                for a_ in grp:
                    if k.startswith(a_):
                        for b_ in grp:
                            if b_ + k[len(a_):] in ref and ref[b_ + k[len(a_):]].grad is not None:
                                return b_ + k[len(a_):]
            return k

        worst16 = 0.0
        for net, ref in ((gl, pL), (gh, pH)):
            for k0, p in net.named_parameters():
                k = ref_key(k0, ref)
                if k not in ref or ref[k].grad is None:
                    assert any(u in k0 for u in ("upscale16x", "residual816")), k0     # never-called modules only
                    continue
                r = ref[k].grad
                assert p.grad is not None, k
                err = float((p.grad.cpu() - r).abs().max()) / (float(r.abs().max()) + 1e-6)
                worst16 = max(worst16, err)
                assert err < 5e-3, (k, err)
                seen.add(k)
        print("x16: worst relative parameter-gradient error", worst16)
        assert "a" in seen and any(k.startswith("h_net2.") for k in seen) and "img_net1.img.0.weight" in seen
    finally:
        cfg_reset()


@pytest.mark.parametrize("B,C,H,W,act,res", [(16, 512, 16, 16, 2, False), (16, 2048, 4, 4, 2, False), (4, 7, 2, 2, 0, False),
                                              (3, 33, 8, 8, 0, True), (16, 128, 16, 16, 2, False), (2, 5, 60, 32, 2, False)])
def test_small_batchnorm_layers_in_one_launch_equal_the_two_pass_form_bit_for_bit(B, C, H, W, act, res):
    """A layer whose channel fits one workgroup's walk (B * HW < 8192, no GLU: the discriminators' small maps) runs statistics +
    normalise, and reduce + apply, as ONE launch each (ops.bn_set_fuse_small, the default): every output - activations, batch and
    running statistics, the three gradients - carries the bits of the two-launch form."""
    from tgsr_amd import _lib, ops
    assert _lib.lib().tgsr_bn_train_nsplit(B, C, H * W) == 1
    g = torch.Generator().manual_seed(B * 1000 + C)
    raw = (torch.randn(B, C, H, W, generator=g) * 1.7 + 0.3).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    resid = torch.randn(B, C, H, W, generator=g).to(DEV) if res else None
    dout = torch.randn(B, C, H, W, generator=g).to(DEV)
    outs = []
    was = ops.bn_set_fuse_small(True)
    try:
        for fused in (True, False):
            ops.bn_set_fuse_small(fused)
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            nbt = torch.zeros((), dtype=torch.int64, device=DEV)
            out, stats = ops.bn_train_fwd(raw, gamma, beta, 1e-5, 0.1, rm, rv, act=act, residual=resid, nbt=nbt)
            draw, dgamma, dbeta = ops.bn_train_bwd(dout, raw, stats, act)
            outs.append((out, stats, rm, rv, nbt, draw, dgamma, dbeta))
    finally:
        ops.bn_set_fuse_small(was)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # and against torch's own BatchNorm (fp64) for good measure
    x64 = raw.double().cpu().requires_grad_(True)
    y = torch.nn.functional.batch_norm(x64, None, None, gamma.double().cpu(), beta.double().cpu(), True, 0.1, 1e-5)
    if res:
        y = y + resid.double().cpu()
    if act == 2:
        y = torch.nn.functional.leaky_relu(y, 0.2)
    y.backward(dout.double().cpu())
    assert torch.allclose(outs[0][0].cpu().double(), y.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(outs[0][5].cpu().double(), x64.grad, rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(16, 768, 256, 17, 17), (4, 64, 48, 16, 16), (2, 768, 256, 17, 17)])
def test_conv1x1_forward_and_gradients_on_the_implicit_gemm_kernel(B, Cin, Cout, H, W):
    """CNN_ENCODER.emb_features (util.py:300, 367) through torch.ops.tgsr.conv1x1: where the shape qualifies the forward, the data
    gradient AND the weight gradient (K = B H W as the channels of a 1x1 convolution over Cin "pixels") run on the trunk's implicit
    GEMM (three-piece bf16 operands); batch 2 (K % 16 != 0) keeps the plain GEMM kernel.  All against torch in float64."""
    import tgsr_amd.custom_ops as C
    from tgsr_amd import ops
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    dy = torch.randn(B, Cout, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr).backward(dy.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    out = C.conv1x1(xd, wd)
    out.backward(dy.to(DEV))
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max()) / float(b.abs().max())      # noqa: E731
    assert rel(out, torch.nn.functional.conv2d(x.double(), w.double())) < 2e-5
    assert rel(xd.grad, xr.grad) < 2e-5 and rel(wd.grad, wr.grad) < 5e-5, (rel(xd.grad, xr.grad), rel(wd.grad, wr.grad))
    took = ops.conv1x1_wgrad(dy.to(DEV), x.to(DEV)) is not None
    assert took == ((B * H * W) % 16 == 0 and B * H * W >= 1024 and Cin >= 64)
