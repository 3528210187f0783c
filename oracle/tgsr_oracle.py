"""CPU oracle for the TGSR text-conditioned SR hot path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file; the
product (`tgsr_amd/`) never does, and fails loudly when its HIP library is missing instead of falling
back to anything here.

This is a *restatement* of the reference's arithmetic (cxm12/TGSR, pure PyTorch) as plain functions
over a flat ``{state_dict key: tensor}`` mapping - fp32, NCHW, single process, no nn.Module, no global
cfg.  Every function cites the reference lines it follows (paths relative to /root/reference).

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function below against vectors
captured by importing the reference itself in the build container (`tests/golden/make_golden.py`
-> `tests/golden/*.npz`), including the shipped x8 face checkpoints at full size.
Unpinned pieces (no reference arithmetic to pin against): the Inception-v3 trunk
of CNN_ENCODER (third-party torchvision, not vendored) and every discriminator (absent upstream).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

BN_EPS = 1e-5        # nn.BatchNorm2d default, util.py:77
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default


# --------------------------------------------------------------------------------------- blocks
def glu(x: Tensor) -> Tensor:
    """util.py:45-53  GLU: first half of the channels times sigmoid of the second half."""
    nc = x.shape[1] // 2
    return x[:, :nc] * torch.sigmoid(x[:, nc:])


def conv3x3(x: Tensor, w: Tensor) -> Tensor:
    """util.py:62-65  3x3, stride 1, zero pad 1, no bias."""
    return F.conv2d(x, w, None, 1, 1)


def batch_norm(x: Tensor, sd: SD, p: str, training: bool = False, update: Optional[SD] = None) -> Tensor:
    """nn.BatchNorm2d as used at util.py:77,116,119.

    eval: y = (x - running_mean) / sqrt(running_var + eps) * weight + bias.
    train: the same with the biased batch variance over (N,H,W); if `update` is given the new running
    statistics (momentum 0.1, *unbiased* variance) are written into it under the same keys.
    """
    g, b = sd[p + "weight"], sd[p + "bias"]
    if not training:
        m, v = sd[p + "running_mean"], sd[p + "running_var"]
    else:
        m = x.mean(dim=(0, 2, 3))
        v = x.var(dim=(0, 2, 3), unbiased=False)
        if update is not None:
            n = x.numel() // x.shape[1]
            update[p + "running_mean"] = (1 - BN_MOMENTUM) * sd[p + "running_mean"] + BN_MOMENTUM * m.detach()
            update[p + "running_var"] = (1 - BN_MOMENTUM) * sd[p + "running_var"] + \
                BN_MOMENTUM * v.detach() * (n / max(n - 1, 1))
    inv = torch.rsqrt(v + BN_EPS)
    return (x - m[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]


def res_block(x: Tensor, sd: SD, p: str, training: bool = False, update: Optional[SD] = None) -> Tensor:
    """util.py:110-130  x + BN(conv(GLU(BN(conv(x)))))."""
    y = conv3x3(x, sd[p + "block.0.weight"])
    y = glu(batch_norm(y, sd, p + "block.1.", training, update))
    y = conv3x3(y, sd[p + "block.3.weight"])
    y = batch_norm(y, sd, p + "block.4.", training, update)
    return y + x


def up_block(x: Tensor, sd: SD, p: str, training: bool = False, update: Optional[SD] = None) -> Tensor:
    """util.py:74-80  nearest x2 -> conv3x3 -> BN -> GLU (Sequential indices 0..3)."""
    y = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    y = conv3x3(y, sd[p + "1.weight"])
    return glu(batch_norm(y, sd, p + "2.", training, update))


def conv_bn_glu(x: Tensor, sd: SD, p: str, training: bool = False, update: Optional[SD] = None) -> Tensor:
    """conv3x3 -> BN -> GLU Sequential (`im2f` util.py:741-744, `convin` model.py:228)."""
    return glu(batch_norm(conv3x3(x, sd[p + "0.weight"]), sd, p + "1.", training, update))


def residual_nosum(x: Tensor, sd: SD, p: str, training: bool = False, update: Optional[SD] = None) -> Tensor:
    """model.py:229-232  `residual24/48`: conv-BN-GLU-conv-BN, a plain Sequential with NO skip add."""
    y = glu(batch_norm(conv3x3(x, sd[p + "0.weight"]), sd, p + "1.", training, update))
    return batch_norm(conv3x3(y, sd[p + "3.weight"]), sd, p + "4.", training, update)


# --------------------------------------------------------------------------------------- attention
def word_attention(h: Tensor, words: Tensor, w_ctx: Tensor, mask: Optional[Tensor],
                   correct_mask: bool = False) -> Tuple[Tensor, Tensor]:
    """GlobalAttention.py:87-130  GlobalAttentionGeneral.forward.

    h [B,idf,ih,iw], words [B,cdf,T], w_ctx [idf,cdf,1,1], mask bool [B,T] (True = padded word).
    Returns (weightedContext [B,idf,ih,iw], attn [B,T,ih,iw]).
    Reference quirk (GlobalAttention.py:109-116): the mask is tiled with `mask.repeat(queryL, 1)` while
    the score rows are ordered b*Q+q, so score row r is masked with mask[r % B] (not mask[r // Q]).
    `correct_mask=True` gives the per-sample masking instead (opt-in, not the reference's behaviour).
    """
    B, idf, ih, iw = h.shape
    T = words.shape[2]
    Q = ih * iw
    src = torch.einsum("ic,bct->bit", w_ctx.reshape(idf, -1), words)          # :100-102  [B,idf,T]
    s = torch.einsum("biq,bit->bqt", h.reshape(B, idf, Q), src)                 # :107      [B,Q,T]
    if mask is not None:
        if correct_mask:
            m = mask[:, None, :].expand(B, Q, T)
        else:
            rows = torch.arange(B * Q, device=h.device) % B                     # :111
            m = mask[rows].reshape(B, Q, T)
        s = s.masked_fill(m, float("-inf"))
    p = torch.softmax(s, dim=2)                                                 # :118
    wc = torch.einsum("bit,bqt->biq", src, p)                                   # :126
    return wc.reshape(B, idf, ih, iw), p.transpose(1, 2).reshape(B, T, ih, iw)


def func_attention(query: Tensor, context: Tensor, gamma1: float) -> Tuple[Tensor, Tensor]:
    """GlobalAttention.py:33-74  DAMSM attention.

    query [B,ndf,L] (words), context [B,ndf,ih,iw] (regions).  Softmax over words per region, then
    x gamma1 and softmax over regions per word.  Returns (weightedContext [B,ndf,L], attn [B,L,ih,iw]).
    """
    B, ndf, L = query.shape
    ih, iw = context.shape[2], context.shape[3]
    S = ih * iw
    ctx = context.reshape(B, ndf, S)
    a = torch.einsum("bds,bdl->bsl", ctx, query)            # :53
    a = torch.softmax(a, dim=2)                              # :56 over words
    a = torch.softmax(a.transpose(1, 2) * gamma1, dim=2)     # :60-65 over regions  [B,L,S]
    wc = torch.einsum("bds,bls->bdl", ctx, a)               # :72
    return wc, a.reshape(B, L, ih, iw)


# --------------------------------------------------------------------------------------- text encoder
def rnn_encoder(sd: SD, captions: Tensor, cap_lens: Sequence[int], p: str = "") -> Tuple[Tensor, Tensor]:
    """util.py:233-260  RNN_ENCODER.forward in eval mode (dropout off), 1-layer bidirectional LSTM.

    captions int64 [B,W] sorted by length descending, cap_lens [B].  The packed-sequence semantics are
    restated explicitly: each direction runs over the sample's own first `len` tokens only; outputs
    beyond `len` are zero; the sentence code is [h_fwd(len-1), h_bwd(0)].
    Returns (words_emb [B, 2H, T_max], sent_emb [B, 2H]).  Gate order i,f,g,o (torch.nn.LSTM).
    """
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    B, T = captions.shape[0], max(lens)
    emb = sd[p + "encoder.weight"][captions[:, :T]]                       # [B,T,ninput]
    H = sd[p + "rnn.weight_hh_l0"].shape[1]
    out = emb.new_zeros(B, T, 2 * H)
    sent = emb.new_zeros(B, 2 * H)
    for d, suf in enumerate(("", "_reverse")):
        w_ih, w_hh = sd[p + "rnn.weight_ih_l0" + suf], sd[p + "rnn.weight_hh_l0" + suf]
        bias = sd[p + "rnn.bias_ih_l0" + suf] + sd[p + "rnn.bias_hh_l0" + suf]
        for b in range(B):
            hcur = emb.new_zeros(H)
            ccur = emb.new_zeros(H)
            steps = range(lens[b]) if d == 0 else range(lens[b] - 1, -1, -1)
            for t in steps:
                gts = w_ih @ emb[b, t] + w_hh @ hcur + bias
                i, f, g, o = gts[:H], gts[H:2 * H], gts[2 * H:3 * H], gts[3 * H:]
                ccur = torch.sigmoid(f) * ccur + torch.sigmoid(i) * torch.tanh(g)
                hcur = torch.sigmoid(o) * torch.tanh(ccur)
                out[b, t, d * H:(d + 1) * H] = hcur
            sent[b, d * H:(d + 1) * H] = hcur
    return out.transpose(1, 2), sent


def rnn_encoder_gru(sd: SD, captions: Tensor, cap_lens: Sequence[int], p: str = "") -> Tuple[Tensor, Tensor]:
    """util.py:207-211, 233-260 with cfg.RNN_TYPE == 'GRU': the same packed-sequence semantics as `rnn_encoder` over a 1-layer
    bidirectional torch.nn.GRU (gate order r, z, n):
        r = sigmoid(W_ir x + b_ir + W_hr h + b_hr),  z = sigmoid(W_iz x + b_iz + W_hz h + b_hz),
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (1 - z) * n + z * h;
    sent_emb = the final hidden states of the two directions (util.py:257: `hidden.transpose(0, 1)`)."""
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    B, T = captions.shape[0], max(lens)
    emb = sd[p + "encoder.weight"][captions[:, :T]]
    H = sd[p + "rnn.weight_hh_l0"].shape[1]
    out = emb.new_zeros(B, T, 2 * H)
    sent = emb.new_zeros(B, 2 * H)
    for d, suf in enumerate(("", "_reverse")):
        w_ih, w_hh = sd[p + "rnn.weight_ih_l0" + suf], sd[p + "rnn.weight_hh_l0" + suf]
        b_ih, b_hh = sd[p + "rnn.bias_ih_l0" + suf], sd[p + "rnn.bias_hh_l0" + suf]
        for b in range(B):
            hcur = emb.new_zeros(H)
            steps = range(lens[b]) if d == 0 else range(lens[b] - 1, -1, -1)
            for t in steps:
                gi, gh = w_ih @ emb[b, t] + b_ih, w_hh @ hcur + b_hh
                r = torch.sigmoid(gi[:H] + gh[:H])
                z = torch.sigmoid(gi[H:2 * H] + gh[H:2 * H])
                n = torch.tanh(gi[2 * H:] + r * gh[2 * H:])
                hcur = (1 - z) * n + z * hcur
                out[b, t, d * H:(d + 1) * H] = hcur
            sent[b, d * H:(d + 1) * H] = hcur
    return out.transpose(1, 2), sent


def ca_net(sd: SD, sent_emb: Tensor, p: str = "ca_net.") -> Tuple[Tensor, Tensor]:
    """util.py:383-387  CA_NET.encode: Linear -> GLU -> split into (mu, logvar).  The sampled c_code
    (util.py:389-396) is discarded by G_SR_NET_low (model.py:51-52), so it is not restated."""
    x = glu(F.linear(sent_emb, sd[p + "fc.weight"], sd[p + "fc.bias"]))
    c = x.shape[1] // 2
    return x[:, :c], x[:, c:]


# --------------------------------------------------------------------------------------- generators
def init_stage(sd: SD, p: str, LR: Tensor, words: Tensor, mask: Optional[Tensor], training=False,
               update=None, correct_mask=False) -> Tuple[Tensor, Tensor]:
    """util.py:763-777  INIT_STAGE_GImgup.forward: im2f -> attention -> cat -> ResBlocks -> upBlock."""
    h = conv_bn_glu(LR, sd, p + "im2f.", training, update)
    c, att = word_attention(h, words, sd[p + "att.conv_context.weight"], mask, correct_mask)
    x = torch.cat((h, c), 1)
    r = 0
    while (p + "residual.%d.block.0.weight" % r) in sd:
        x = res_block(x, sd, p + "residual.%d." % r, training, update)
        r += 1
    return up_block(x, sd, p + "upsample.", training, update), att


def next_stage(sd: SD, p: str, h: Tensor, words: Tensor, mask: Optional[Tensor], training=False,
               update=None, correct_mask=False) -> Tuple[Tensor, Tensor]:
    """util.py:807-823  NEXT_STAGE_G.forward: attention -> cat -> ResBlocks -> upBlock."""
    c, att = word_attention(h, words, sd[p + "att.conv_context.weight"], mask, correct_mask)
    x = torch.cat((h, c), 1)
    r = 0
    while (p + "residual.%d.block.0.weight" % r) in sd:
        x = res_block(x, sd, p + "residual.%d." % r, training, update)
        r += 1
    return up_block(x, sd, p + "upsample.", training, update), att


def g_sr_net_low(sd: SD, LR: Tensor, sent_emb: Tensor, words: Tensor, mask: Optional[Tensor],
                 training=False, update=None, correct_mask=False, p: str = ""):
    """model.py:48-78  G_SR_NET_low.forward -> (fake_imgs[3], att_maps[3], mu, logvar)."""
    mu, logvar = ca_net(sd, sent_emb, p + "ca_net.")
    imgs, atts = [], []
    h, a = init_stage(sd, p + "h_net1.", LR, words, mask, training, update, correct_mask)
    imgs.append(conv3x3(h, sd[p + "img_net1.img.0.weight"]))      # GET_IMAGE_G_noAct util.py:909-919
    atts.append(a)
    for k in (2, 3):
        h, a = next_stage(sd, p + "h_net%d." % k, h, words, mask, training, update, correct_mask)
        imgs.append(conv3x3(h, sd[p + "img_net%d.img.0.weight" % k]))
        atts.append(a)
    return imgs, atts, mu, logvar


def g_sr_net_low16(sd: SD, LR: Tensor, sent_emb: Tensor, words: Tensor, mask: Optional[Tensor], training=False,
                   update=None, correct_mask=False, p: str = ""):
    """models16.py:5-39  x16 G_SR_NET_low: ONE NEXT_STAGE_G object serves stages 2-4 (`h_net4 = h_net3 = h_net2`,
    :13) and ONE GET_IMAGE_G (conv3x3 + Tanh, util.py:894-905) serves the four heads (:14)."""
    mu, logvar = ca_net(sd, sent_emb, p + "ca_net.")
    w_img = sd[p + "img_net1.img.0.weight"]
    imgs, atts = [], []
    h, a = init_stage(sd, p + "h_net1.", LR, words, mask, training, update, correct_mask)
    imgs.append(torch.tanh(conv3x3(h, w_img)))
    atts.append(a)
    for _ in range(3):
        h, a = next_stage(sd, p + "h_net2.", h, words, mask, training, update, correct_mask)
        imgs.append(torch.tanh(conv3x3(h, w_img)))
        atts.append(a)
    return imgs, atts, mu, logvar


def netg_highweight(sd: SD, LR: Tensor, SRb: Sequence[Tensor], LRb: Tensor, low: str = "lr",
                    training=False, update=None, p: str = "", use_act: bool = True):
    """model.py:264-298  NetG_highweight.forward.

    weightmap=False: `a` is the constant 0.5 and `one` the constant 1 (model.py:246-248: `.cuda()` on the Parameter
    leaves a plain tensor, never trained nor saved), so ims_k = conv_output(out_k) + 0.5 * SRb_k.
    weightmap=True (recognised by the keys `a1..a3` in `sd`, model.py:235-245): ims_k = conv_output(out_k) + a_k * SRb_k with
    a_k an [H, W] map broadcast over batch and channels (:277, 286, 294); returns (ims, a3, one) (:293-295).
    use_act=False (model.py:223-226): conv_output is the bare conv5x5, no Tanh.
    """
    if low == "lrblur":
        x = LRb
    elif low == "lr-lrblur":
        x = LR - LRb
    else:
        x = LR
    maps = [sd[p + "a%d" % k] for k in (1, 2, 3)] if (p + "a1") in sd else None
    a = LR.new_tensor([0.5])
    one = LR.new_ones(1)
    out = conv_bn_glu(x, sd, p + "convin.", training, update)
    r = 0
    while (p + "residual.%d.block.0.weight" % r) in sd:
        out = res_block(out, sd, p + "residual.%d." % r, training, update)
        r += 1
    w5 = sd[p + "conv_output.0.weight"]

    def head(o, sr, k):
        c = F.conv2d(o, w5, None, 1, 2)
        return one * (torch.tanh(c) if use_act else c) + (a if maps is None else maps[k]) * sr     # model.py:224, 280

    out = up_block(out, sd, p + "upscale2x.", training, update)
    ims2 = head(out, SRb[0], 0)
    out = residual_nosum(out, sd, p + "residual24.", training, update)
    out = up_block(out, sd, p + "upscale4x.", training, update)
    ims4 = head(out, SRb[1], 1)
    out = residual_nosum(out, sd, p + "residual48.", training, update)
    out = up_block(out, sd, p + "upscale8x.", training, update)
    ims8 = head(out, SRb[2], 2)
    return [ims2, ims4, ims8], (a if maps is None else maps[2]), one


def netg_highweight16(sd: SD, LR: Tensor, SRb: Sequence[Tensor], LRb: Tensor, low: str = "lr",
                      training=False, update=None, p: str = ""):
    """models16.py:97-179  x16 NetG_highweight.forward with weightmap=False: like the x8 one plus a fourth head whose
    stage re-uses `residual48` / `upscale8x` (:172-173), and `a` is a registered parameter here (sd[p + "a"], :126).
    The shipped line :178 adds the 8x image to the 16x tensor and cannot execute (shape error, pinned by
    tests/golden/nets16_small.npz `gh16_runs == 0`); as in tgsr_amd.models16 the 16x head adds SRb[3]."""
    if low == "lrblur":
        x = LRb
    elif low == "lr-lrblur":
        x = LR - LRb
    else:
        x = LR
    a = sd[p + "a"]
    one = LR.new_ones(1)
    out = conv_bn_glu(x, sd, p + "convin.", training, update)
    r = 0
    while (p + "residual.%d.block.0.weight" % r) in sd:
        out = res_block(out, sd, p + "residual.%d." % r, training, update)
        r += 1
    w5 = sd[p + "conv_output.0.weight"]

    def head(o, sr):
        return one * torch.tanh(F.conv2d(o, w5, None, 1, 2)) + a * sr

    out = up_block(out, sd, p + "upscale2x.", training, update)
    ims = [head(out, SRb[0])]
    out = up_block(residual_nosum(out, sd, p + "residual24.", training, update), sd, p + "upscale4x.", training, update)
    ims.append(head(out, SRb[1]))
    for k in (2, 3):                                     # 8x, then 16x through the same modules
        out = up_block(residual_nosum(out, sd, p + "residual48.", training, update), sd, p + "upscale8x.", training,
                       update)
        ims.append(head(out, SRb[k]))
    return ims, a, one


def sr_forward(sd_E: SD, sd_GL: SD, sd_GH: SD, captions: Tensor, cap_lens, LR: Tensor, LRb: Tensor,
               low: str = "lr", correct_mask: bool = False):
    """Caller counterpart of trainer_objective.py:134-146 (eval mode): text encoder -> mask ->
    G_SR_NET_low -> NetG_highweight.  Returns a dict of every tensor the reference loop produces."""
    words, sent = rnn_encoder(sd_E, captions, cap_lens)
    mask = (captions == 0)[:, :words.shape[2]]                       # :136-140
    imgs, atts, mu, logvar = g_sr_net_low(sd_GL, LR, sent, words, mask, correct_mask=correct_mask)
    fine, a, one = netg_highweight(sd_GH, LR, imgs, LRb, low)
    return {"words_emb": words, "sent_emb": sent, "mask": mask, "fake": imgs, "att": atts,
            "mu": mu, "logvar": logvar, "fine": fine}


def to_uint8(img: Tensor) -> Tensor:
    """trainer_objective.py:153-155  round(clip((x+1)*127.5, 0, 255)) as uint8 (np.round = half-to-even)."""
    return torch.round(torch.clamp((img + 1.0) * 127.5, 0, 255)).to(torch.uint8)


# --------------------------------------------------------------------------------------- losses
def cosine_similarity(x1: Tensor, x2: Tensor, dim: int = 1, eps: float = 1e-8) -> Tensor:
    """losses.py:12-18."""
    w12 = (x1 * x2).sum(dim)
    return w12 / (x1.norm(2, dim) * x2.norm(2, dim)).clamp(min=eps)


def _class_mask(class_ids, B, device):
    """losses.py:25-35 / 75-80: mask[i][j] = class_ids[j] == class_ids[i], i != j."""
    if class_ids is None:
        return None
    ids = torch.as_tensor(class_ids, device=device)
    m = ids[None, :] == ids[:, None]
    m.fill_diagonal_(False)
    return m


def words_loss(img_features: Tensor, words_emb: Tensor, labels: Optional[Tensor], cap_lens, class_ids,
               batch_size: int, gamma1: float, gamma2: float, gamma3: float):
    """losses.py:65-136.  similarities[j][i] = log sum_w exp(gamma2 * cos(word_w of caption i,
    region-context of image j for that word)); x gamma3; class mask -> -inf; CE both ways.
    Returns (loss0, loss1, att_maps) with att_maps[i] = attention of image i on caption i [1,L_i,ih,iw]."""
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    B = batch_size
    sims, att_maps = [], []
    for i in range(B):
        L = lens[i]
        word = words_emb[i:i + 1, :, :L].expand(B, -1, -1)                  # :83-85
        wc, attn = func_attention(word, img_features, gamma1)               # :92
        att_maps.append(attn[i:i + 1])
        row = cosine_similarity(word.transpose(1, 2).reshape(B * L, -1),
                                wc.transpose(1, 2).reshape(B * L, -1)).reshape(B, L)
        sims.append(torch.log(torch.exp(row * gamma2).sum(1, keepdim=True)))   # :106-109
    sim = torch.cat(sims, 1) * gamma3                                       # :116-124
    m = _class_mask(class_ids, B, sim.device)
    if m is not None:
        sim = sim.masked_fill(m, float("-inf"))
    if labels is None:
        return None, None, att_maps
    return F.cross_entropy(sim, labels), F.cross_entropy(sim.t(), labels), att_maps


def sent_loss(cnn_code: Tensor, rnn_code: Tensor, labels: Optional[Tensor], class_ids, batch_size: int,
              gamma3: float, eps: float = 1e-8):
    """losses.py:21-62.  scores[i][j] = gamma3 * cos(cnn_code_i, rnn_code_j) (norm product clamped)."""
    n0 = cnn_code.norm(2, dim=1, keepdim=True)
    n1 = rnn_code.norm(2, dim=1, keepdim=True)
    s = cnn_code @ rnn_code.t() / (n0 @ n1.t()).clamp(min=eps) * gamma3
    m = _class_mask(class_ids, batch_size, s.device)
    if m is not None:
        s = s.masked_fill(m, float("-inf"))
    if labels is None:
        return None, None
    return F.cross_entropy(s, labels), F.cross_entropy(s.t(), labels)


def kl_loss(mu: Tensor, logvar: Tensor) -> Tensor:
    """losses.py:806-810  -0.5 * mean(1 + logvar - mu^2 - exp(logvar))."""
    return -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())


def mse(fake: Sequence[Tensor], label: Sequence[Tensor]) -> Tensor:
    """losses.py:779-784  sum over scales of the mean squared error."""
    return sum(F.mse_loss(f, l) for f, l in zip(fake, label))


# --------------------------------------------------------------------------------------- discriminators + GAN losses
# The reference calls `netD(img)`, `netD.COND_DNET(features, sent_emb)` and `netD.UNCOND_DNET(features)` (losses.py:
# 290-316, 351-366) but defines no discriminator class: the ARCHITECTURE below is the build's declaration (AttnGAN
# topology, tgsr_amd/model.py::_D_NET) restated with stock torch ops so the HIP kernels can be checked - parity
# UNPINNED for the architecture.  `downBlock` (util.py:92-98) and the two loss functions ARE the reference's.
def leaky(x: Tensor) -> Tensor:
    return F.leaky_relu(x, 0.2)


def down_block(x: Tensor, sd: SD, p: str, training: bool = True, update: Optional[SD] = None) -> Tensor:
    """util.py:92-98  Conv2d(in, out, 4, 2, 1, bias=False) -> BatchNorm2d -> LeakyReLU(0.2) (Sequential 0..2)."""
    return leaky(batch_norm(F.conv2d(x, sd[p + "0.weight"], None, 2, 1), sd, p + "1.", training, update))


def block3x3_leaky(x: Tensor, sd: SD, p: str, training: bool = True, update: Optional[SD] = None) -> Tensor:
    """conv3x3 -> BatchNorm2d -> LeakyReLU(0.2)."""
    return leaky(batch_norm(conv3x3(x, sd[p + "0.weight"]), sd, p + "1.", training, update))


def d_features(sd: SD, x: Tensor, training: bool = True, update: Optional[SD] = None, p: str = "") -> Tensor:
    """_D_NET.forward: image -> [B, 8 ndf, 4, 4]."""
    h = leaky(F.conv2d(x, sd[p + "img_code_s16.conv0.weight"], None, 2, 1))
    for k in (1, 2, 3):
        h = down_block(h, sd, p + "img_code_s16.down%d." % k, training, update)
    k = 0
    while (p + "extra.%d.0.weight" % k) in sd:
        h = down_block(h, sd, p + "extra.%d." % k, training, update)
        k += 1
    k = 0
    while (p + "reduce.%d.0.weight" % k) in sd:
        h = block3x3_leaky(h, sd, p + "reduce.%d." % k, training, update)
        k += 1
    return h


def d_logits(sd: SD, p: str, h: Tensor, c: Optional[Tensor] = None, training: bool = True,
             update: Optional[SD] = None) -> Tensor:
    """D_GET_LOGITS.forward: [conditional: tile the sentence code over 4x4, cat, conv3x3-BN-LeakyReLU] -> 4x4/stride-4
    conv to one logit per sample."""
    if c is not None and (p + "jointConv.0.weight") in sd:
        cc = c.view(c.shape[0], -1, 1, 1).repeat(1, 1, 4, 4)
        h = block3x3_leaky(torch.cat((h, cc), 1), sd, p + "jointConv.", training, update)
    return F.conv2d(h, sd[p + "outlogits.0.weight"], sd[p + "outlogits.0.bias"], 4).view(-1)


def discriminator_loss(sd: SD, real: Tensor, fake: Tensor, cond: Tensor, real_labels: Tensor, fake_labels: Tensor,
                       training: bool = True) -> Tensor:
    """losses.py:290-316.  real / fake / wrong-caption (batch shifted by one, :302) BCE-with-logits terms."""
    bce = F.binary_cross_entropy_with_logits
    fr, ff = d_features(sd, real, training), d_features(sd, fake.detach(), training)
    n = fr.shape[0]
    c_real = bce(d_logits(sd, "COND_DNET.", fr, cond, training), real_labels)
    c_fake = bce(d_logits(sd, "COND_DNET.", ff, cond, training), fake_labels)
    c_wrong = bce(d_logits(sd, "COND_DNET.", fr[:n - 1], cond[1:n], training), fake_labels[1:n])
    if "UNCOND_DNET.outlogits.0.weight" in sd:
        u_real = bce(d_logits(sd, "UNCOND_DNET.", fr, None, training), real_labels)
        u_fake = bce(d_logits(sd, "UNCOND_DNET.", ff, None, training), fake_labels)
        return (u_real + c_real) / 2. + (u_fake + c_fake + c_wrong) / 3.
    return c_real + (c_fake + c_wrong) / 2.


def generator_adv_loss(sds: Sequence[SD], fakes: Sequence[Tensor], sent_emb: Tensor, real_labels: Tensor,
                       training: bool = True, g: float = 1.0) -> Tensor:
    """The adversarial half of generator_loss (losses.py:358-371): per scale, BCE-with-logits of the conditional
    (+ unconditional) logits of the fake image against the REAL labels, times g."""
    bce = F.binary_cross_entropy_with_logits
    total = 0
    for sd, img in zip(sds, fakes):
        f = d_features(sd, img, training)
        e = bce(d_logits(sd, "COND_DNET.", f, sent_emb, training), real_labels)
        if "UNCOND_DNET.outlogits.0.weight" in sd:
            e = e + bce(d_logits(sd, "UNCOND_DNET.", f, None, training), real_labels)
        total = total + g * e
    return total


def generator_loss(sds: Sequence[SD], image_encoder, fakes: Sequence[Tensor], real_labels: Tensor, words_embs: Tensor,
                   sent_emb: Tensor, match_labels: Tensor, cap_lens, class_ids, gamma1: float, gamma2: float,
                   gamma3: float, lam: float, w: float = 1.0, s: float = 1.0, g: float = 1.0,
                   training: bool = True) -> Tensor:
    """losses.py:351-391 in full: the per-scale adversarial terms plus, on the LAST scale's image, the DAMSM ranking
    term w (w_loss0 + w_loss1) LAMBDA + s (s_loss0 + s_loss1) LAMBDA through `image_encoder` (image -> (region features
    [B,nef,17,17], cnn_code [B,nef])).  Pinned by tests/golden/gan_losses.npz (the reference's own function, run on
    plain-torch discriminators)."""
    total = generator_adv_loss(sds, fakes, sent_emb, real_labels, training, g)
    B = real_labels.shape[0]
    regions, code = image_encoder(fakes[len(sds) - 1])
    w0, w1, _ = words_loss(regions, words_embs, match_labels, cap_lens, class_ids, B, gamma1, gamma2, gamma3)
    s0, s1 = sent_loss(code, sent_emb, match_labels, class_ids, B, gamma3)
    return total + w * (w0 + w1) * lam + s * (s0 + s1) * lam


# --------------------------------------------------------------------------------------- synthetic workload
def synthetic_batch(B: int, n_words: int = 41, seed: int = 100, lr: int = 32, width: int = 18,
                    fixed_len: Optional[int] = None):
    """The synthetic input SURVEY.md section 8d / BASELINE.md section 4 prescribe (seed 100 = test1.py:170):
    LR, LRb ~ U(-1,1) [B,3,lr,lr]; captions int64 [B,18] with lengths from {4..14} sorted descending."""
    g = torch.Generator().manual_seed(seed)
    if fixed_len is None:
        lens = torch.randint(4, 15, (B,), generator=g)
    else:
        lens = torch.full((B,), fixed_len)
    lens = torch.sort(lens, descending=True)[0]
    cap = torch.zeros(B, width, dtype=torch.int64)
    for i in range(B):
        cap[i, :int(lens[i])] = torch.randint(1, n_words, (int(lens[i]),), generator=g)
    LR = torch.rand(B, 3, lr, lr, generator=g) * 2 - 1
    LRb = torch.rand(B, 3, lr, lr, generator=g) * 2 - 1
    return cap, lens, LR, LRb


def random_state(ngf: int = 32, nef: int = 256, ncf: int = 100, n_words: int = 41, r_num: int = 2,
                 seed: int = 0) -> Tuple[SD, SD, SD]:
    """Seeded random parameters with the reference's key names and shapes (for benches without a
    checkpoint).  Conv weights ~ N(0, 1/fan_in) (variance-preserving), BN weight ~ N(1,0.02), running
    stats (0,1); LSTM / embedding uniform like torch's defaults.  Returns (sd_E, sd_GL, sd_GH)."""
    g = torch.Generator().manual_seed(seed)

    def conv(co, ci, k=3):
        return torch.randn(co, ci, k, k, generator=g) * (1.0 / math.sqrt(ci * k * k))

    def bn(sd, p, c):
        sd[p + "weight"] = 1 + 0.02 * torch.randn(c, generator=g)
        sd[p + "bias"] = torch.zeros(c)
        sd[p + "running_mean"] = torch.zeros(c)
        sd[p + "running_var"] = torch.ones(c)
        sd[p + "num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    def resblock(sd, p, c):
        sd[p + "block.0.weight"] = conv(2 * c, c)
        bn(sd, p + "block.1.", 2 * c)
        sd[p + "block.3.weight"] = conv(c, c)
        bn(sd, p + "block.4.", c)

    def upblock(sd, p, ci, co):
        sd[p + "1.weight"] = conv(2 * co, ci)
        bn(sd, p + "2.", 2 * co)

    GL: SD = {}
    GL["ca_net.fc.weight"] = torch.randn(4 * ncf, nef, generator=g) / math.sqrt(nef)
    GL["ca_net.fc.bias"] = torch.zeros(4 * ncf)
    for k in (1, 2, 3):
        p = "h_net%d." % k
        GL[p + "att.conv_context.weight"] = torch.randn(ngf, nef, 1, 1, generator=g) / math.sqrt(nef)
        if k == 1:
            GL[p + "im2f.0.weight"] = conv(2 * ngf, 3)
            bn(GL, p + "im2f.1.", 2 * ngf)
        for r in range(r_num):
            resblock(GL, p + "residual.%d." % r, 2 * ngf)
        upblock(GL, p + "upsample.", 2 * ngf, ngf)
        GL["img_net%d.img.0.weight" % k] = conv(3, ngf)
    GH: SD = {}
    GH["convin.0.weight"] = conv(2 * ngf, 3)
    bn(GH, "convin.1.", 2 * ngf)
    for r in range(6):
        resblock(GH, "residual.%d." % r, ngf)
    for s in ("2", "4", "8"):
        upblock(GH, "upscale%sx." % s, ngf, ngf)
    GH["conv_output.0.weight"] = conv(3, ngf, 5)
    for s in ("24", "48"):
        p = "residual%s." % s
        GH[p + "0.weight"] = conv(2 * ngf, ngf)
        bn(GH, p + "1.", 2 * ngf)
        GH[p + "3.weight"] = conv(ngf, ngf)
        bn(GH, p + "4.", ngf)
    H = nef // 2
    k = 1.0 / math.sqrt(H)
    E: SD = {"encoder.weight": (torch.rand(n_words, 300, generator=g) * 2 - 1) * 0.1}
    for suf in ("", "_reverse"):
        E["rnn.weight_ih_l0" + suf] = (torch.rand(4 * H, 300, generator=g) * 2 - 1) * k
        E["rnn.weight_hh_l0" + suf] = (torch.rand(4 * H, H, generator=g) * 2 - 1) * k
        E["rnn.bias_ih_l0" + suf] = (torch.rand(4 * H, generator=g) * 2 - 1) * k
        E["rnn.bias_hh_l0" + suf] = (torch.rand(4 * H, generator=g) * 2 - 1) * k
    return E, GL, GH
