"""Pins the CPU oracle (oracle/tgsr_oracle.py) to vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from conftest import split_sd
from oracle import tgsr_oracle as O

ATOL = 1e-5  # restatement vs reference on CPU, same fp32 arithmetic up to summation order


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol=ATOL, rtol=1e-5):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def test_glu(ops_small):
    close(O.glu(T(ops_small["blk.x"])), ops_small["glu.out"])


def test_word_attention_mask_quirk(ops_small):
    g = ops_small
    out, attn = O.word_attention(T(g["att.h"]), T(g["att.ctx"]), T(g["att.w"]), T(g["att.mask"]))
    close(out, g["att.out"])
    close(attn, g["att.attn"])
    # the per-sample ("correct") masking is NOT what the reference computes for B>1 with unequal lengths
    out2, _ = O.word_attention(T(g["att.h"]), T(g["att.ctx"]), T(g["att.w"]), T(g["att.mask"]), correct_mask=True)
    assert np.abs(out2.numpy() - g["att.out"]).max() > 1e-3


def test_word_attention_b1_nomask(ops_small):
    g = ops_small
    out, attn = O.word_attention(T(g["att.h"][:1]), T(g["att.ctx"][:1]), T(g["att.w"]), None)
    close(out, g["att1.out"])
    close(attn, g["att1.attn"])


def test_func_attention(ops_small):
    g = ops_small
    out, attn = O.func_attention(T(g["fa.query"]), T(g["fa.context"]), float(g["fa.gamma1"]))
    close(out, g["fa.out"])
    close(attn, g["fa.attn"])


def test_resblock_eval_train(ops_small):
    g = ops_small
    sd = split_sd(g, "rb.")
    x = T(g["blk.x"])
    close(O.res_block(x, sd, ""), g["rb.eval"])
    upd = {}
    close(O.res_block(x, sd, "", training=True, update=upd), g["rb.train"], atol=2e-5)
    for k, v in upd.items():
        close(v, g["rb.after." + k])


def test_upblock_eval_train(ops_small):
    g = ops_small
    sd = split_sd(g, "ub.")
    x = T(g["blk.x"])
    close(O.up_block(x, sd, ""), g["ub.eval"])
    close(O.up_block(x, sd, "", training=True), g["ub.train"], atol=2e-5)


def test_rnn_encoder(ops_small):
    g = ops_small
    sd = split_sd(g, "enc.")
    w, s = O.rnn_encoder(sd, T(g["enc.captions"]), g["enc.cap_lens"].tolist())
    close(w, g["enc.words_emb"], atol=1e-6)
    close(s, g["enc.sent_emb"], atol=1e-6)


def test_rnn_encoder_gru_branch():
    """util.py:207-211 (cfg.RNN_TYPE == 'GRU'): the oracle's explicit packed-sequence GRU against the reference's own
    RNN_ENCODER run with that flag (tests/golden/enc_gru.npz, make_golden.py gen_gru): H = 128 with ragged captions incl. a
    one-word one, H = 32 at batch 1."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "enc_gru.npz"))
    for tag in ("a", "b"):
        sd = {k[len("gru_%s." % tag):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("gru_%s." % tag)}
        we, se = O.rnn_encoder_gru(sd, sd["captions"], sd["cap_lens"])
        assert tuple(we.shape) == tuple(sd["words_emb"].shape)
        np.testing.assert_allclose(we.numpy(), sd["words_emb"].numpy(), atol=1e-6)
        np.testing.assert_allclose(se.numpy(), sd["sent_emb"].numpy(), atol=1e-6)


def test_ca_net_kl_mse(ops_small):
    g = ops_small
    mu, lv = O.ca_net(split_sd(g, "ca."), T(g["enc.sent_emb"]), p="")
    close(mu, g["ca.mu"], atol=1e-6)
    close(lv, g["ca.logvar"], atol=1e-6)
    close(O.kl_loss(T(g["kl.mu"]), T(g["kl.logvar"])), g["kl.out"], atol=1e-6)
    close(O.mse([T(g["mse.a0"]), T(g["mse.a1"])], [T(g["mse.b0"]), T(g["mse.b1"])]), g["mse.out"], atol=1e-6)


def _run_nets(g, mode):
    sdE, sdL, sdH = split_sd(g, "E."), split_sd(g, "GL."), split_sd(g, "GH.")
    cap, lens = T(g["captions"]), g["cap_lens"].tolist()
    words, sent = O.rnn_encoder(sdE, cap, lens)
    mask = (cap == 0)[:, :words.shape[2]]
    tr = mode == "train"
    imgs, atts, mu, lv = O.g_sr_net_low(sdL, T(g["LR"]), sent, words, mask, training=tr)
    fine, a, one = O.netg_highweight(sdH, T(g["LR"]), imgs, T(g["LRb"]), "lr", training=tr)
    return words, sent, mask, imgs, atts, mu, lv, fine, a, one


def test_generators_small_eval(nets_small):
    g = nets_small
    words, sent, mask, imgs, atts, mu, lv, fine, a, one = _run_nets(g, "eval")
    close(words, g["eval.words_emb"], atol=1e-6)
    assert (mask.numpy() == g["eval.mask"]).all()
    close(mu, g["eval.mu"]); close(lv, g["eval.logvar"])
    for i in range(3):
        close(imgs[i], g["eval.fake%d" % i], atol=5e-5)
        close(atts[i], g["eval.att%d" % i], atol=1e-5)
        close(fine[i], g["eval.fine%d" % i], atol=5e-5)
    close(a, g["eval.a"]); close(one, g["eval.one"])


def test_generators_small_train_bn(nets_small):
    g = nets_small
    _, _, _, imgs, atts, _, _, fine, _, _ = _run_nets(g, "train")
    for i in range(3):
        close(imgs[i], g["train.fake%d" % i], atol=2e-4, rtol=1e-4)
        close(fine[i], g["train.fine%d" % i], atol=2e-4, rtol=1e-4)


def test_damsm_losses_and_grads(damsm_golden):
    g = damsm_golden
    gam = g["gamma"]
    for tag, cls in (("cls", g["class_ids"]), ("nocls", None)):
        feats = T(g["feats"]).requires_grad_()
        words = T(g["words"]).requires_grad_()
        cnn = T(g["cnn_code"]).requires_grad_()
        sent = T(g["sent"]).requires_grad_()
        labels = torch.arange(4)
        w0, w1, att = O.words_loss(feats, words, labels, g["cap_lens"].tolist(), cls, 4, *map(float, gam))
        s0, s1 = O.sent_loss(cnn, sent, labels, cls, 4, float(gam[2]))
        close(w0, g[tag + ".w0"]); close(w1, g[tag + ".w1"])
        close(s0, g[tag + ".s0"]); close(s1, g[tag + ".s1"])
        for i, a in enumerate(att):
            close(a, g[tag + ".att%d" % i], atol=1e-6)
        gr = torch.autograd.grad(w0 + w1 + s0 + s1, [feats, words, cnn, sent])
        for t, k in zip(gr, ("g_feats", "g_words", "g_cnn", "g_sent")):
            close(t, g[tag + "." + k], atol=1e-6, rtol=1e-4)


def test_full_size_face_checkpoint(face_c1, face_weights):
    """C1: shipped x8 face checkpoints, B=2, 32->256, through the whole caller-counterpart path."""
    g, w = face_c1, face_weights
    r = O.sr_forward(split_sd(w, "E."), split_sd(w, "GL."), split_sd(w, "GH."), T(g["captions"]),
                     g["cap_lens"].tolist(), T(g["LR"]), T(g["LRb"]))
    close(r["words_emb"], g["words_emb"], atol=1e-6)
    close(r["mu"], g["mu"]); close(r["logvar"], g["logvar"])
    for i in range(3):
        close(r["fake"][i], g["fake%d" % i], atol=1e-4, rtol=1e-4)
        close(r["fine"][i], g["fine%d" % i], atol=1e-4, rtol=1e-4)
    close(r["att"][0], g["att0"], atol=1e-5); close(r["att"][1], g["att1"], atol=1e-5)
    a2 = r["att"][2].numpy()
    close(a2[:, :, ::8, ::8], g["att2.sub8"], atol=1e-5)
    close(a2[:, :, 40:72, 40:72], g["att2.crop"], atol=1e-5)
    u8 = O.to_uint8(r["fine"][-1]).numpy()
    assert (np.abs(u8.astype(int) - g["sr_uint8"].astype(int)) <= 1).all()
    assert (u8 != g["sr_uint8"]).mean() < 1e-3


def test_checkpoint_manifest_matches_random_state():
    """state_dict contract: oracle.random_state produces exactly the shipped key/shape set."""
    import json, os
    from conftest import GOLDEN
    man = json.load(open(os.path.join(GOLDEN, "ckpt_manifest.json")))
    _, GL, GH = O.random_state()
    assert {k: list(v.shape) for k, v in GL.items()} == {k: v[0] for k, v in man["netG_epoch_7"].items()}
    assert {k: list(v.shape) for k, v in GH.items()} == {k: v[0] for k, v in man["netGH_epoch_7"].items()}


def test_x16_generator_weight_tied_stages():
    """models16.G_SR_NET_low: stages 2-4 and the four tanh heads share one module each."""
    from conftest import load_npz
    g = load_npz("nets16_small.npz")
    sdE, sdL = split_sd(g, "E."), split_sd(g, "GL.")
    cap, lens = T(g["captions"]), g["cap_lens"].tolist()
    words, sent = O.rnn_encoder(sdE, cap, lens)
    mask = (cap == 0)[:, :words.shape[2]]
    imgs, atts, mu, lv = O.g_sr_net_low16(sdL, T(g["LR"]), sent, words, mask)
    assert len(imgs) == 4 and imgs[3].shape[-1] == 128
    for i in range(4):
        close(imgs[i], g["fake%d" % i], atol=5e-5)
        close(atts[i], g["att%d" % i], atol=1e-5)
    assert int(g["gh16_runs"]) == 0   # the shipped x16 NetG_highweight.forward raises (models16.py:178)


def test_gan_loss_formulas_golden():
    """oracle.discriminator_loss / generator_loss against the REFERENCE's own functions (losses.py:290-316, 351-391) run by
    tests/golden/make_golden.py::gen_gan on plain-torch discriminators of the build-declared topology: loss values, every
    discriminator parameter gradient, and the gradients reaching the fake images / sentence / word embeddings (the DAMSM
    ranking term with a class mask rides along through a stub image encoder)."""
    import torch.nn.functional as F
    from conftest import load_npz
    g = load_npz("gan_losses.npz")
    B = g["sent"].shape[0]
    img = lambda k: T(g[k]).float() / 127.5 - 1.0
    real = [img("real0.u8"), img("real1.u8")]
    sent, words = T(g["sent"]), T(g["words"])
    lens, cls = g["cap_lens"].tolist(), g["class_ids"]
    rl, fl, ml = torch.ones(B), torch.zeros(B), torch.arange(B)
    g1, g2, g3 = (float(v) for v in g["gamma"])
    sds = []
    for k in range(2):
        sd = split_sd(g, "D%d." % k)
        sds.append({n: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in n else v)
                    for n, v in sd.items()})
    # ---- discriminator_loss: value + every parameter gradient
    for k in range(2):
        fake = img("fake%d.u8" % k)
        err = O.discriminator_loss(sds[k], real[k], fake, sent, rl, fl)
        close(err, g["errD%d" % k], atol=2e-6)
        names = [n for n, v in sds[k].items() if v.requires_grad]
        grads = torch.autograd.grad(err, [sds[k][n] for n in names])
        assert len(names) == sum(1 for n in g if n.startswith("gD%d." % k))
        for n, gr in zip(names, grads):
            close(gr, g["gD%d.%s" % (k, n)], atol=2e-6, rtol=1e-4)
    cond_only = {n: v for n, v in sds[0].items() if not n.startswith("UNCOND_DNET.")}
    close(O.discriminator_loss(cond_only, real[0], img("fake0.u8"), sent, rl, fl), g["errD0.cond_only"], atol=2e-6)
    # ---- generator_loss (adversarial + ranking term) and what it sends back
    fakes = [img("fake%d.u8" % k).requires_grad_(True) for k in range(2)]
    sent_r, words_r = sent.clone().requires_grad_(True), words.clone().requires_grad_(True)
    ew, eb, pw, pb = (T(g["enc." + n]) for n in ("f.weight", "f.bias", "p.weight", "p.bias"))
    enc = lambda x: (F.conv2d(F.adaptive_avg_pool2d(x, 17), ew, eb), F.linear(x.mean((2, 3)), pw, pb))
    sd_eval = [{n: v.detach() for n, v in sd.items()} for sd in sds]
    errG = O.generator_loss(sd_eval, enc, fakes, rl, words_r, sent_r, ml, lens, cls, g1, g2, g3, float(g["lambda"]))
    close(errG, g["errG"], atol=2e-5, rtol=1e-5)
    gr = torch.autograd.grad(errG, fakes + [sent_r, words_r])
    close(gr[0], g["gG.fake0"], atol=1e-7, rtol=1e-3)
    close(gr[1][:, :, ::4, ::4], g["gG.fake1.sub4"], atol=1e-7, rtol=1e-3)
    close(gr[2], g["gG.sent"], atol=2e-6, rtol=1e-4)
    close(gr[3], g["gG.words"], atol=2e-6, rtol=1e-4)
    err2 = O.generator_loss(sd_eval, enc, [f.detach() for f in fakes], rl, words, sent, ml, lens, None, g1, g2, g3,
                            float(g["lambda"]), w=0.5, s=2.0, g=3.0)
    close(err2, g["errG.nocls.w05.s2.g3"], atol=2e-5, rtol=1e-5)
    assert str(g["logs"]).startswith("g_loss0: ")


def test_netg_highweight_weightmap_noact_and_eval_downblock():
    """gh_variants.npz (the reference's own modules, tests/golden/make_golden.py gen_gh_variants): NetG_highweight(weightmap=True)
    eval images + train-mode gradients of the maps, NetG_highweight(useAct=False) eval images, downBlock under .eval()."""
    from conftest import load_npz
    g = load_npz("gh_variants.npz")
    sd = split_sd(g, "wm.GH.")
    SRb = [T(g["wm.SRb%d" % k]) for k in range(3)]
    ims, a, one = O.netg_highweight(sd, T(g["wm.LR"]), SRb, T(g["wm.LR"]), "lr")
    for k in range(3):
        close(ims[k], g["wm.eval.fine%d" % k], atol=2e-5)
    close(a, g["wm.eval.a"])
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    sr = [s_.clone().requires_grad_(True) for s_ in SRb]
    ims, _a, _one = O.netg_highweight(sdr, T(g["wm.LR"]), sr, T(g["wm.LR"]), "lr", training=True, update={})
    sum((i * T(g["wm.dy%d" % k])).sum() for k, i in enumerate(ims)).backward()
    close(ims[0], g["wm.train.fine0"], atol=2e-5)
    for k in (1, 2, 3):
        close(sdr["a%d" % k].grad, g["wm.train.da%d" % k], atol=2e-5)
    close(sr[0].grad, g["wm.train.dSRb0"], atol=2e-5)
    close(sdr["conv_output.0.weight"].grad, g["wm.train.dconv_output"], atol=2e-3, rtol=1e-4)
    sd = split_sd(g, "na.GH.")
    ims, a, one = O.netg_highweight(sd, T(g["na.LR"]), [T(g["na.SRb%d" % k]) for k in range(3)], T(g["na.LR"]), "lr", use_act=False)
    for k in range(3):
        close(ims[k], g["na.fine%d" % k], atol=2e-5)
    sd = split_sd(g, "down.")
    x = T(g["down.x"]).requires_grad_(True)
    y = O.down_block(x, sd, "", training=False)
    (y * T(g["down.dy"])).sum().backward()
    close(y, g["down.out"])
    close(x.grad, g["down.dx"])
