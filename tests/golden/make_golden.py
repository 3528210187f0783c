#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference (CPU, this container only).

TEST INFRASTRUCTURE - runs only where /root/reference exists (the build container). Nothing here is
imported by the product (`tgsr_amd/`), by `-m gpu` tests, by `bench.py` or by `smoke()`: those read
only the committed `.npz`/`.json` fixtures this script writes.

What it does: puts /root/reference on sys.path, injects attr-dict / torchvision
stubs, makes `.cuda()` an identity and `cfg.CUDA = False`, switches the bool-mask path on
(`server = 1`), builds the reference modules, feeds seeded inputs and dumps inputs + parameters +
outputs.  The reference's files are never copied, only imported and called.

    python tests/golden/make_golden.py            # regenerates every fixture (about a minute)
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("TGSR_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- harness-side patches
def _install_stubs():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)

    class EasyDict(dict):
        """attr-dict stand-in for the (absent) easydict package."""

        def __init__(self, d=None, **kw):
            super().__init__()
            d = dict(d or {}, **kw)
            for k, v in d.items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            dict.__setitem__(self, k, v)

        __setitem__ = __setattr__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    m = types.ModuleType("easydict")
    m.EasyDict = EasyDict
    sys.modules["easydict"] = m
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    # the reference calls .cuda() unconditionally (model.py:246-248): identity on this CPU box
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.set_device = lambda *a, **k: None


def _load_ref(ngf=32, nef=256):
    _install_stubs()
    from miscc.config import cfg
    cfg.CUDA = False
    cfg.GAN.GF_DIM = ngf
    cfg.TEXT.EMBEDDING_DIM = nef
    cfg.TREE.BRANCH_NUM = 4
    cfg.TREE.BASE_SIZE = 32
    cfg.GAN.R_NUM = 2
    cfg.TRAIN.SMOOTH.GAMMA1 = 4.0
    cfg.TRAIN.SMOOTH.GAMMA2 = 5.0
    cfg.TRAIN.SMOOTH.GAMMA3 = 10.0
    import GlobalAttention
    import model
    import util
    from miscc import losses
    GlobalAttention.server = 1
    losses.server = 1
    return cfg, GlobalAttention, util, model, losses


def _np(t):
    return t.detach().cpu().numpy().copy()


def _sd_np(module, prefix):
    return {prefix + k: _np(v) for k, v in module.state_dict().items()}


def _randomize_bn(module, gen):
    """Give every BatchNorm non-trivial affine + running statistics so eval-mode BN is exercised."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            n = m.num_features
            m.weight.data = 1.0 + 0.2 * torch.randn(n, generator=gen)
            m.bias.data = 0.1 * torch.randn(n, generator=gen)
            m.running_mean = 0.1 * torch.randn(n, generator=gen)
            m.running_var = 0.5 + torch.rand(n, generator=gen)


def _captions(gen, lens, n_words, width=18):
    cap = torch.zeros(len(lens), width, dtype=torch.int64)
    for i, n in enumerate(lens):
        cap[i, :n] = torch.randint(1, n_words, (n,), generator=gen)
    return cap, torch.tensor(lens, dtype=torch.int64)


# --------------------------------------------------------------------------- fixtures
def gen_ops():
    """Op/module-level goldens on small shapes (ngf=32 so channel counts match the HIP tiles)."""
    cfg, GA, util, model, losses = _load_ref(ngf=32, nef=64)
    g = torch.Generator().manual_seed(1234)
    torch.manual_seed(1234)          # module constructors draw their initial weights from the GLOBAL generator
    out = {}

    # GlobalAttentionGeneral incl. the mask.repeat quirk (GlobalAttention.py:109-116): B=3, unequal lengths
    att = GA.GlobalAttentionGeneral(32, 64)
    h = torch.randn(3, 32, 8, 8, generator=g)
    ctx = torch.randn(3, 64, 6, generator=g)
    mask = torch.tensor([[0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 1, 1], [0, 0, 0, 1, 1, 1]], dtype=torch.bool)
    att.applyMask(mask)
    with torch.no_grad():
        wc, am = att(h, ctx)
    out.update({"att.h": _np(h), "att.ctx": _np(ctx), "att.mask": _np(mask),
                "att.w": _np(att.conv_context.weight), "att.out": _np(wc), "att.attn": _np(am)})
    # same without a mask, B=1 (shipped yml batch size)
    att.applyMask(None)
    with torch.no_grad():
        wc1, am1 = att(h[:1], ctx[:1])
    out.update({"att1.out": _np(wc1), "att1.attn": _np(am1)})

    # func_attention (GlobalAttention.py:33-74)
    q = torch.randn(4, 32, 5, generator=g)
    c = torch.randn(4, 32, 5, 5, generator=g)
    with torch.no_grad():
        fw, fa = GA.func_attention(q, c, 4.0)
    out.update({"fa.query": _np(q), "fa.context": _np(c), "fa.gamma1": np.float32(4.0),
                "fa.out": _np(fw), "fa.attn": _np(fa)})

    # ResBlock(64) / upBlock(64,32), eval and train mode (util.py:74-80, 110-130)
    x = torch.randn(2, 64, 8, 8, generator=g)
    out["blk.x"] = _np(x)
    rb = util.ResBlock(64)
    _randomize_bn(rb, g)
    out.update(_sd_np(rb, "rb."))
    rb.eval()
    with torch.no_grad():
        out["rb.eval"] = _np(rb(x.clone()))
    rb.train()
    with torch.no_grad():
        out["rb.train"] = _np(rb(x.clone()))
    out.update({"rb.after." + k: _np(v) for k, v in rb.state_dict().items() if "running" in k})
    ub = util.upBlock(64, 32)
    _randomize_bn(ub, g)
    out.update(_sd_np(ub, "ub."))
    ub.eval()
    with torch.no_grad():
        out["ub.eval"] = _np(ub(x))
    ub.train()
    with torch.no_grad():
        out["ub.train"] = _np(ub(x))

    # GLU, KL, MSE
    with torch.no_grad():
        out["glu.out"] = _np(util.GLU()(x))
        mu = torch.randn(3, 100, generator=g)
        lv = 0.3 * torch.randn(3, 100, generator=g)
        out.update({"kl.mu": _np(mu), "kl.logvar": _np(lv), "kl.out": _np(losses.KL_loss(mu.clone(), lv.clone()))})
        a = [torch.randn(2, 3, 8, 8, generator=g), torch.randn(2, 3, 16, 16, generator=g)]
        b = [torch.randn(2, 3, 8, 8, generator=g), torch.randn(2, 3, 16, 16, generator=g)]
        out.update({"mse.a0": _np(a[0]), "mse.a1": _np(a[1]), "mse.b0": _np(b[0]), "mse.b1": _np(b[1]),
                    "mse.out": _np(losses.MSE(a, b))})

    # RNN_ENCODER (util.py:175-260): vocab 41, nhidden 64, B=3, lengths 6/4/3, eval (no dropout)
    enc = util.RNN_ENCODER(41, nhidden=64)
    enc.eval()
    cap, lens = _captions(g, [6, 4, 3], 41)
    with torch.no_grad():
        we, se = enc(cap, lens, enc.init_hidden(3))
    out.update(_sd_np(enc, "enc."))
    out.update({"enc.captions": _np(cap), "enc.cap_lens": _np(lens), "enc.words_emb": _np(we), "enc.sent_emb": _np(se)})

    # CA_NET mu/logvar (util.py:372-400)
    ca = util.CA_NET()
    with torch.no_grad():
        _, mu, lv = ca(se)
    out.update(_sd_np(ca, "ca."))
    out.update({"ca.mu": _np(mu), "ca.logvar": _np(lv)})
    np.savez_compressed(os.path.join(OUT, "ops_small.npz"), **out)
    print("ops_small.npz", len(out), "arrays")


def gen_nets_small():
    """Whole-generator goldens, ngf=32 / nef=64, LR 16x16, B=3 with unequal captions (Q1), eval + train BN."""
    cfg, GA, util, model, losses = _load_ref(ngf=32, nef=64)
    g = torch.Generator().manual_seed(4321)
    torch.manual_seed(7)
    out = {}
    enc = util.RNN_ENCODER(41, nhidden=64)
    enc.eval()
    netGL = model.G_SR_NET_low()
    netGH = model.NetG_highweight(weightmap=False, low="lr")
    _randomize_bn(netGL, g)
    _randomize_bn(netGH, g)
    cap, lens = _captions(g, [7, 5, 4], 41)
    LR = torch.rand(3, 3, 16, 16, generator=g) * 2 - 1
    LRb = torch.rand(3, 3, 16, 16, generator=g) * 2 - 1
    out.update(_sd_np(enc, "E."))
    out.update(_sd_np(netGL, "GL."))
    out.update(_sd_np(netGH, "GH."))
    out.update({"captions": _np(cap), "cap_lens": _np(lens), "LR": _np(LR), "LRb": _np(LRb)})
    for mode in ("eval", "train"):
        netGL.train(mode == "train")
        netGH.train(mode == "train")
        with torch.no_grad():
            we, se = enc(cap, lens, enc.init_hidden(3))
            mask = (cap == 0)[:, :we.size(2)]
            imgs, atts, mu, lv = netGL(LR, se, we, mask)
            fine, a, one = netGH(LR, imgs, LRb)
        p = mode + "."
        out.update({p + "words_emb": _np(we), p + "sent_emb": _np(se), p + "mask": _np(mask),
                    p + "mu": _np(mu), p + "logvar": _np(lv), p + "a": _np(a), p + "one": _np(one)})
        for i in range(3):
            out[p + "fake%d" % i] = _np(imgs[i])
            out[p + "att%d" % i] = _np(atts[i])
            out[p + "fine%d" % i] = _np(fine[i])
    np.savez_compressed(os.path.join(OUT, "nets_small.npz"), **out)
    print("nets_small.npz", len(out), "arrays")


def gen_nets16_small():
    """x16 variants (models16.py): weight-tied stages; ngf=32 / nef=64, LR 8x8 -> 128x128, B=2, eval BN."""
    cfg, GA, util, model, losses = _load_ref(ngf=32, nef=64)
    import models16
    g = torch.Generator().manual_seed(1616)
    torch.manual_seed(16)
    out = {}
    enc = util.RNN_ENCODER(41, nhidden=64)
    enc.eval()
    netGL = models16.G_SR_NET_low()
    _randomize_bn(netGL, g)
    netGL.eval()
    cap, lens = _captions(g, [6, 4], 41)
    LR = torch.rand(2, 3, 8, 8, generator=g) * 2 - 1
    out.update(_sd_np(enc, "E."))
    tied = ("h_net3.", "h_net4.", "img_net2.", "img_net3.", "img_net4.")   # aliases of h_net2 / img_net1
    out["GL.keys"] = np.array(sorted(netGL.state_dict().keys()))
    out.update({k: v for k, v in _sd_np(netGL, "GL.").items() if not k[3:].startswith(tied)})
    out.update({"captions": _np(cap), "cap_lens": _np(lens), "LR": _np(LR)})
    with torch.no_grad():
        we, se = enc(cap, lens, enc.init_hidden(2))
        mask = (cap == 0)[:, :we.size(2)]
        imgs, atts, mu, lv = netGL(LR, se, we, mask)
    for i in range(4):
        out["fake%d" % i] = _np(imgs[i])
        out["att%d" % i] = _np(atts[i])
    out["mu"] = _np(mu)
    # the x16 NetG_highweight cannot run as shipped: models16.py:178 adds the 8x image SRb8 to a 16x tensor
    netGH = models16.NetG_highweight(weightmap=False, low="lr")
    netGH.eval()
    try:
        with torch.no_grad():
            netGH(LR, imgs, LR)
        out["gh16_runs"] = np.array(1)
    except RuntimeError as e:
        out["gh16_runs"] = np.array(0)
        print("models16.NetG_highweight.forward raises as shipped:", str(e).splitlines()[0])
    out["gh16_keys"] = np.array(sorted(netGH.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "nets16_small.npz"), **out)
    print("nets16_small.npz", len(out), "arrays")


def gen_damsm():
    """DAMSM words_loss / sent_loss goldens (losses.py:21-136): B=4, lens 18/15/12/9, gammas 4/5/10."""
    cfg, GA, util, model, losses = _load_ref(ngf=32, nef=256)
    g = torch.Generator().manual_seed(99)
    B = 4
    feats = torch.randn(B, 256, 17, 17, generator=g, requires_grad=True)
    words = torch.randn(B, 256, 18, generator=g, requires_grad=True)
    cnn_code = torch.randn(B, 256, generator=g, requires_grad=True)
    sent = torch.randn(B, 256, generator=g, requires_grad=True)
    lens = torch.tensor([18, 15, 12, 9])
    labels = torch.arange(B)
    out = {"feats": _np(feats), "words": _np(words), "cnn_code": _np(cnn_code), "sent": _np(sent),
           "cap_lens": _np(lens), "gamma": np.array([4.0, 5.0, 10.0], np.float32)}
    for tag, cls in (("cls", np.array([0, 1, 1, 3])), ("nocls", None)):
        w0, w1, att = losses.words_loss(feats, words, labels, lens, cls, B)
        s0, s1 = losses.sent_loss(cnn_code, sent, labels, cls, B)
        total = w0 + w1 + s0 + s1
        gr = torch.autograd.grad(total, [feats, words, cnn_code, sent])
        out.update({tag + ".w0": _np(w0), tag + ".w1": _np(w1), tag + ".s0": _np(s0), tag + ".s1": _np(s1),
                    tag + ".g_feats": _np(gr[0]), tag + ".g_words": _np(gr[1]),
                    tag + ".g_cnn": _np(gr[2]), tag + ".g_sent": _np(gr[3])})
        for i, a in enumerate(att):
            out[tag + ".att%d" % i] = _np(a)
        if cls is not None:
            out["class_ids"] = cls
    np.savez_compressed(os.path.join(OUT, "damsm.npz"), **out)
    print("damsm.npz", len(out), "arrays")


def gen_face_s8():
    """Full-size C1 golden with the shipped x8 face checkpoints (B=2, lens 14/9, seed 100 = test1.py:170)."""
    cfg, GA, util, model, losses = _load_ref(ngf=32, nef=256)
    torch.manual_seed(100)
    g = torch.Generator().manual_seed(100)
    ck = os.path.join(REF, "Checkpoint", "face_S8")
    sdL = torch.load(os.path.join(ck, "netG_epoch_7.pth"), map_location="cpu", weights_only=True)
    sdH = torch.load(os.path.join(ck, "netGH_epoch_7.pth"), map_location="cpu", weights_only=True)
    netGL = model.G_SR_NET_low()
    netGH = model.NetG_highweight(weightmap=False, low="lr")
    netGL.load_state_dict(sdL, strict=True)
    missing = netGH.load_state_dict(sdH, strict=False)
    assert set(missing.missing_keys) <= {"a"} and not missing.unexpected_keys, missing
    enc = util.RNN_ENCODER(41, nhidden=256)  # text_encoder200.pth is not shipped: seeded random init
    for m in (enc, netGL, netGH):
        m.eval()
    cap, lens = _captions(g, [14, 9], 41)
    LR = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    LRb = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    with torch.no_grad():
        we, se = enc(cap, lens, enc.init_hidden(2))
        mask = (cap == 0)[:, :we.size(2)]
        imgs, atts, mu, lv = netGL(LR, se, we, mask)
        fine, a, one = netGH(LR, imgs, LRb)
    # weights are data: commit them so the GPU box (no /root/reference there) can run the real checkpoint
    w = {}
    w.update({"GL." + k: _np(v) for k, v in sdL.items()})
    w.update({"GH." + k: _np(v) for k, v in sdH.items()})
    w.update(_sd_np(enc, "E."))
    np.savez_compressed(os.path.join(OUT, "face_S8_weights.npz"), **w)
    out = {"captions": _np(cap), "cap_lens": _np(lens), "LR": _np(LR), "LRb": _np(LRb),
           "words_emb": _np(we), "sent_emb": _np(se), "mask": _np(mask), "mu": _np(mu), "logvar": _np(lv)}
    for i in range(3):
        out["fake%d" % i] = _np(imgs[i])
        out["fine%d" % i] = _np(fine[i])
    out["att0"] = _np(atts[0])
    out["att1"] = _np(atts[1])
    a2 = _np(atts[2])  # [2,14,128,128] = 1.8 MB: keep statistics + a strided sample + one crop
    out["att2.mean"] = a2.mean(axis=(2, 3))
    out["att2.sub8"] = a2[:, :, ::8, ::8]
    out["att2.crop"] = a2[:, :, 40:72, 40:72]
    u8 = np.round(np.maximum(0, np.minimum(255, (_np(fine[-1]) + 1.0) * 127.5))).astype(np.uint8)
    out["sr_uint8"] = u8  # trainer_objective.py:153-155
    np.savez_compressed(os.path.join(OUT, "face_S8_c1.npz"), **out)
    manifest = {"netG_epoch_7": {k: [list(v.shape), str(v.dtype)] for k, v in sdL.items()},
                "netGH_epoch_7": {k: [list(v.shape), str(v.dtype)] for k, v in sdH.items()}}
    with open(os.path.join(OUT, "ckpt_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("face_S8_c1.npz", len(out), "arrays; weights", len(w), "tensors")


def _plain_d_net(ndf, nef, extra_down, gen):
    """The build-declared discriminator topology (tgsr_amd/model.py::_D_NET: AttnGAN's D_NET64/128/256 with logit heads)
    as PLAIN torch.nn modules with the same state_dict key names - harness-side, CPU, stock torch arithmetic.  The
    reference ships no discriminator class (grep D_NET hits only miscc/losses.py), so this is what its loss functions
    are called on here; any module exposing COND_DNET / UNCOND_DNET would do (losses.py:290-316, 351-366)."""
    nn = torch.nn

    def block(conv, c):
        return nn.Sequential(conv, nn.BatchNorm2d(c), nn.LeakyReLU(0.2, inplace=True))

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv0 = nn.Conv2d(3, ndf, 4, 2, 1, bias=False)
            self.down1 = block(nn.Conv2d(ndf, ndf * 2, 4, 2, 1, bias=False), ndf * 2)
            self.down2 = block(nn.Conv2d(ndf * 2, ndf * 4, 4, 2, 1, bias=False), ndf * 4)
            self.down3 = block(nn.Conv2d(ndf * 4, ndf * 8, 4, 2, 1, bias=False), ndf * 8)

        def forward(self, x):
            return self.down3(self.down2(self.down1(torch.nn.functional.leaky_relu(self.conv0(x), 0.2))))

    class Logits(nn.Module):
        def __init__(self, bcondition):
            super().__init__()
            self.bcondition = bcondition
            if bcondition:
                self.jointConv = block(nn.Conv2d(ndf * 8 + nef, ndf * 8, 3, 1, 1, bias=False), ndf * 8)
            self.outlogits = nn.Sequential(nn.Conv2d(ndf * 8, 1, kernel_size=4, stride=4))

        def forward(self, h_code, c_code=None):
            if self.bcondition and c_code is not None:
                c = c_code.view(-1, nef, 1, 1).repeat(1, 1, 4, 4)
                h_code = self.jointConv(torch.cat((h_code, c), 1))
            return self.outlogits(h_code).view(-1)

    class D(nn.Module):
        def __init__(self):
            super().__init__()
            self.img_code_s16 = Enc()
            ch = ndf * 8
            self.extra = nn.ModuleList()
            for _ in range(extra_down):
                self.extra.append(block(nn.Conv2d(ch, ch * 2, 4, 2, 1, bias=False), ch * 2))
                ch *= 2
            self.reduce = nn.ModuleList()
            while ch > ndf * 8:
                self.reduce.append(block(nn.Conv2d(ch, ch // 2, 3, 1, 1, bias=False), ch // 2))
                ch //= 2
            self.UNCOND_DNET = Logits(False)
            self.COND_DNET = Logits(True)

        def forward(self, x):
            x = self.img_code_s16(x)
            for m in self.extra:
                x = m(x)
            for m in self.reduce:
                x = m(x)
            return x

    d = D()
    for m in d.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data = 1.0 + 0.1 * torch.randn(m.num_features, generator=gen)
            m.bias.data = 0.1 * torch.randn(m.num_features, generator=gen)
    return d.train()                                   # the reference's losses run the discriminators in training mode


def gen_gan():
    """The two GAN loss FORMULAS run by the reference itself: `losses.discriminator_loss` (losses.py:290-316) and
    `losses.generator_loss` (:351-391, with a stub image encoder so words_loss / sent_loss ride along), on plain-torch
    discriminators of the build-declared topology (64^2 and 128^2 scales, ndf 4).  Stores inputs (images as the uint8
    codes k of x = k / 127.5 - 1: exactly representable, 4x smaller), parameters, outputs and gradients."""
    cfg, GA, util, model, losses = _load_ref(ngf=32, nef=32)
    cfg.TRAIN.SMOOTH.LAMBDA = 5.0
    g = torch.Generator().manual_seed(2024)
    torch.manual_seed(2024)
    ndf, nef, B, T = 4, 32, 3, 7
    sizes = (64, 128)
    out = {"ndf": np.array(ndf), "nef": np.array(nef), "gamma": np.array([4.0, 5.0, 10.0], np.float32),
           "lambda": np.float32(cfg.TRAIN.SMOOTH.LAMBDA)}
    ds = [_plain_d_net(ndf, nef, k, g) for k in range(len(sizes))]
    for k, d in enumerate(ds):
        out.update(_sd_np(d, "D%d." % k))

    def img(s):
        k = torch.randint(0, 256, (B, 3, s, s), generator=g, dtype=torch.uint8)
        return k, k.float() / 127.5 - 1.0

    real, fake = [], []
    for i, s_ in enumerate(sizes):
        kr, r = img(s_)
        kf, f = img(s_)
        out["real%d.u8" % i], out["fake%d.u8" % i] = _np(kr), _np(kf)
        real.append(r)
        fake.append(f.requires_grad_(True))
    sent = torch.randn(B, nef, generator=g, requires_grad=True)
    words = torch.randn(B, nef, T, generator=g, requires_grad=True)
    lens = torch.tensor([7, 5, 3])
    class_ids = np.array([0, 1, 1])
    rl, fl, ml = torch.ones(B), torch.zeros(B), torch.arange(B)
    out.update({"sent": _np(sent), "words": _np(words), "cap_lens": _np(lens), "class_ids": class_ids})
    # ---- discriminator_loss per scale: value + the gradient of every parameter
    for k, d in enumerate(ds):
        errD = losses.discriminator_loss(d, real[k], fake[k], sent.detach(), rl, fl)
        grads = torch.autograd.grad(errD, list(d.parameters()))
        out["errD%d" % k] = _np(errD)
        for (name, _), gr in zip(d.named_parameters(), grads):
            out["gD%d.%s" % (k, name)] = _np(gr)
    # the unconditional head missing (TRAIN.B_NET_D-style netD.UNCOND_DNET is None branch, losses.py:313-314)
    unc, ds[0].UNCOND_DNET = ds[0].UNCOND_DNET, None
    out["errD0.cond_only"] = _np(losses.discriminator_loss(ds[0], real[0], fake[0], sent.detach(), rl, fl))
    ds[0].UNCOND_DNET = unc

    # ---- generator_loss: stub image encoder (any module image -> (regions [B,nef,17,17], code [B,nef]))
    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.f = torch.nn.Conv2d(3, nef, 1)
            self.p = torch.nn.Linear(3, nef)

        def forward(self, x):
            return self.f(torch.nn.functional.adaptive_avg_pool2d(x, 17)), self.p(x.mean((2, 3)))

    enc = Enc()
    out.update(_sd_np(enc, "enc."))
    errG, logs = losses.generator_loss(ds, enc, fake, rl, words, sent, ml, lens, class_ids)
    gr = torch.autograd.grad(errG, fake + [sent, words])
    out["errG"] = _np(errG)
    out["logs"] = np.array(logs)
    out["gG.fake0"] = _np(gr[0])
    out["gG.fake1.sub4"] = _np(gr[1])[:, :, ::4, ::4]
    out["gG.sent"], out["gG.words"] = _np(gr[2]), _np(gr[3])
    # without the class mask and at scale weights (w, s, g) != 1
    errG2, _ = losses.generator_loss(ds, enc, fake, rl, words, sent, ml, lens, None, w=0.5, s=2.0, g=3.0)
    out["errG.nocls.w05.s2.g3"] = _np(errG2)
    np.savez_compressed(os.path.join(OUT, "gan_losses.npz"), **out)
    print("gan_losses.npz", len(out), "arrays;", logs)


def gen_gru():
    """RNN_ENCODER with cfg.RNN_TYPE = 'GRU' (util.py:207-211, 244-258): the reference's own module, eval mode, two sizes -
    the shipped hidden size (nhidden 256 -> H = 128) on ragged captions incl. a one-word one, and H = 32 at batch 1."""
    cfg, GA, util, model, losses = _load_ref(ngf=32, nef=256)
    cfg.RNN_TYPE = "GRU"
    out = {}
    try:
        for tag, nhidden, lens, seed in (("a", 256, [9, 6, 6, 2, 1], 11), ("b", 64, [5], 12)):
            torch.manual_seed(seed)
            g = torch.Generator().manual_seed(seed)
            enc = util.RNN_ENCODER(41, nhidden=nhidden)
            assert isinstance(enc.rnn, torch.nn.GRU)
            enc.eval()
            cap, lens_t = _captions(g, lens, 41)
            with torch.no_grad():
                we, se = enc(cap, lens_t, enc.init_hidden(len(lens)))
            out.update(_sd_np(enc, "gru_%s." % tag))
            out.update({"gru_%s.captions" % tag: _np(cap), "gru_%s.cap_lens" % tag: _np(lens_t),
                        "gru_%s.words_emb" % tag: _np(we), "gru_%s.sent_emb" % tag: _np(se)})
    finally:
        cfg.RNN_TYPE = "LSTM"
    np.savez_compressed(os.path.join(OUT, "enc_gru.npz"), **out)
    print("enc_gru.npz", len(out), "arrays")


def gen_gh_variants():
    """NetG_highweight's two non-shipped constructor forms (model.py:214, 223-226, 235-245, 276-297) and downBlock under .eval()
    (util.py:92-98), all on the reference's own modules:
      wm.*    weightmap=True, ngf 32, B = 1, LR 32 x 32 (the maps are hard-wired to 64 / 128 / 256 pixels): randomised a1..a3, eval-mode
              images; train-mode BatchNorm: images, d(a_k), d(conv_output weight), d(SRb) for sum_k <ims_k, dy_k>;
      na.*    useAct=False (no Tanh behind conv5x5), LR 16 x 16, B = 2, eval mode: images;
      down.*  downBlock(16, 32) in eval mode on [3, 16, 16, 16]: output, and the input gradient of <out, dy>."""
    cfg, GA, util, model, losses = _load_ref(ngf=32, nef=64)
    out = {}
    g = torch.Generator().manual_seed(2468)
    torch.manual_seed(11)
    net = model.NetG_highweight(weightmap=True, low="lr")
    _randomize_bn(net, g)
    for k, n in ((1, 64), (2, 128), (3, 256)):
        getattr(net, "a%d" % k).data = 1.0 + 0.3 * torch.randn(n, n, generator=g)
    LR = torch.rand(1, 3, 32, 32, generator=g) * 2 - 1
    SRb = [torch.rand(1, 3, s, s, generator=g) * 2 - 1 for s in (64, 128, 256)]
    dy = [torch.randn(1, 3, s, s, generator=g) for s in (64, 128, 256)]
    out.update(_sd_np(net, "wm.GH."))
    out["wm.LR"] = _np(LR)
    for k in range(3):
        out["wm.SRb%d" % k], out["wm.dy%d" % k] = _np(SRb[k]), _np(dy[k])
    net.eval()
    with torch.no_grad():
        ims, a, one = net(LR, SRb, LR)
    for k in range(3):
        out["wm.eval.fine%d" % k] = _np(ims[k])
    out["wm.eval.a"], out["wm.eval.one"] = _np(a), _np(one)
    net.train()
    sr = [s.clone().requires_grad_(True) for s in SRb]
    ims, a, one = net(LR, sr, LR)
    sum((i * d).sum() for i, d in zip(ims, dy)).backward()
    out["wm.train.fine0"] = _np(ims[0])
    for k in range(3):
        out["wm.train.da%d" % (k + 1)] = _np(getattr(net, "a%d" % (k + 1)).grad)
    out["wm.train.dSRb0"] = _np(sr[0].grad)
    out["wm.train.dconv_output"] = _np(net.conv_output[0].weight.grad)
    out["wm.train.dconvin"] = _np(net.convin[0].weight.grad)

    torch.manual_seed(12)
    net = model.NetG_highweight(weightmap=False, low="lr", useAct=False)
    _randomize_bn(net, g)
    net.eval()
    LR = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    SRb = [torch.rand(2, 3, s, s, generator=g) * 2 - 1 for s in (32, 64, 128)]
    out.update(_sd_np(net, "na.GH."))
    out["na.LR"] = _np(LR)
    with torch.no_grad():
        ims, a, one = net(LR, SRb, LR)
    for k in range(3):
        out["na.SRb%d" % k], out["na.fine%d" % k] = _np(SRb[k]), _np(ims[k])

    torch.manual_seed(13)
    blk = util.downBlock(16, 32)
    _randomize_bn(blk, g)
    blk.eval()
    x = torch.randn(3, 16, 16, 16, generator=g).requires_grad_(True)
    dyb = torch.randn(3, 32, 8, 8, generator=g)
    y = blk(x)
    (y * dyb).sum().backward()
    out.update(_sd_np(blk, "down."))
    out.update({"down.x": _np(x), "down.dy": _np(dyb), "down.out": _np(y), "down.dx": _np(x.grad),
                "down.dw": _np(blk[0].weight.grad)})
    np.savez_compressed(os.path.join(OUT, "gh_variants.npz"), **out)
    print("gh_variants.npz", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "nets", "nets16", "damsm", "face", "gan", "gru", "ghv"]
    if "ghv" in which:
        gen_gh_variants()
    if "gru" in which:
        gen_gru()
    if "gan" in which:
        gen_gan()
    if "nets16" in which:
        gen_nets16_small()
    if "ops" in which:
        gen_ops()
    if "nets" in which:
        gen_nets_small()
    if "damsm" in which:
        gen_damsm()
    if "face" in which:
        gen_face_s8()
