"""CPU model of the reduced-precision (bf16 / f16) inference path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

The reference (cxm12/TGSR) computes in fp32 only; BASELINE.json configs[4] asks for a bf16 path.  This file restates
`oracle.tgsr_oracle` with the rounding points of the HIP lp path made explicit, so that the kernels can be checked
tightly (same roundings, fp32 accumulation order aside) AND the cost of the reduced precision against the fp32 oracle
can be stated independently of any kernel:

  * every activation the generators store is rounded once to `dtype` (round-to-nearest-even); conv / attention MFMA
    operands are those stored values and `dtype`-rounded weights; accumulation, BatchNorm affine, GLU, residual add,
    softmax and tanh are fp32;
  * the two 3-channel stems (im2f util.py:741-744, convin model.py:228) read the fp32 LR image with fp32 weights;
  * the text encoder, CA_NET and the word projection `conv_context` (GlobalAttention.py:100-102) stay fp32; the
    projected words are rounded to `dtype` once; the attention maps returned to the caller are the fp32 softmax;
  * the six image heads return fp32 images (no rounding of the outputs).

Measured with this model on the shipped face checkpoint (B=2..16, seeds 100 / 7), PSNR of the finest SR image against
the fp32 oracle (peak 2.0): uniform bf16 46.9-47.4 dB, f16 64.7-65.5 dB.  Where the bf16 loss comes from (layer groups
switched to f16 one at a time, everything else bf16): all of G_SR_NET_low in f16 gains 0.4 dB; NetG_highweight's 64^2 ..
256^2 layers and heads in f16 gain 0.2 dB; NetG_highweight's 32^2 TRUNK alone (convin + six chained ResBlocks, 1.2 % of
the MACs) in f16 gives 53.6-53.8 dB - six residual additions rounded to 8-bit mantissas in a row are the loss.  The bf16
configuration therefore runs that section in f16 (`trunk_dtype_of`; the product's counterpart is
tgsr_amd.lp_pipeline.F16_TRUNK) and meets the >= 50 dB SURVEY.md 8c states for it.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import tgsr_oracle as O

Tensor = torch.Tensor


def rnd(x: Tensor, dtype) -> Tensor:
    """Round to `dtype` (nearest even), return as fp32."""
    return x.to(dtype).to(torch.float32)


def _fold(sd, p):
    """BatchNorm2d eval as (scale, shift) (tgsr_bn_fold)."""
    s = sd[p + "weight"] / torch.sqrt(sd[p + "running_var"] + O.BN_EPS)
    return s, sd[p + "bias"] - sd[p + "running_mean"] * s


def subpixel_upconv(x: Tensor, w: Tensor, dtype) -> Tensor:
    """conv3x3(nearest_x2(x)) as tgsr_lp_upconv_glu_fwd computes it: output phase (a, b) is a 2x2 convolution of the
    low-res x with the 3x3 taps that fall on the same low-res pixel pre-summed in fp32 and rounded ONCE to `dtype`
    (a = 0: rows {y-1: w0, y: w1+w2}; a = 1: rows {y: w0+w1, y+1: w2}; same for columns).  Mathematically equal to
    the direct form; the rounding of the summed taps is what differs, so the model restates it."""
    B, _, H, W = x.shape
    rows = [[w[:, :, 0], w[:, :, 1] + w[:, :, 2]], [w[:, :, 0] + w[:, :, 1], w[:, :, 2]]]      # [a][row slot] -> [Co,Ci,3]
    out = x.new_empty(B, w.shape[0], 2 * H, 2 * W)
    for a in (0, 1):
        for b in (0, 1):
            k = torch.stack([torch.stack([r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if b == 0 else
                                         [r[:, :, 0] + r[:, :, 1], r[:, :, 2]], dim=-1) for r in rows[a]], dim=-2)
            xp = F.pad(x, (1 - b, b, 1 - a, a))                  # a/b = 0 reach up/left, 1 reach down/right
            out[:, :, a::2, b::2] = F.conv2d(xp, rnd(k, dtype))
    return out


def conv_block(x: Tensor, w: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], dtype, glu: bool = False,
               upsample: bool = False, residual: Optional[Tensor] = None, round_w: bool = True,
               subpixel: bool = False) -> Tensor:
    """tgsr_lp_conv3x3_fwd (subpixel=True: tgsr_lp_upconv_glu_fwd): x, residual hold `dtype`-representable values
    (NCHW fp32 tensors)."""
    if upsample and subpixel:
        y = subpixel_upconv(x, w, dtype)
    else:
        if upsample:
            x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        y = F.conv2d(x, rnd(w, dtype) if round_w else w, None, 1, 1)
    if scale is not None:
        y = y * scale[None, :, None, None] + shift[None, :, None, None]
    if glu:
        y = O.glu(y)
    if residual is not None:
        y = y + residual
    return rnd(y, dtype)


def word_attention(h: Tensor, words: Tensor, w_ctx: Tensor, mask: Optional[Tensor], dtype, correct_mask=False):
    """tgsr_lp_word_attention_fwd: h holds `dtype` values; returns (c_code rounded to dtype, attn fp32)."""
    B, idf, ih, iw = h.shape
    T = words.shape[2]
    Q = ih * iw
    src = rnd(torch.einsum("ic,bct->bit", w_ctx.reshape(idf, -1), words), dtype)
    s = torch.einsum("biq,bit->bqt", h.reshape(B, idf, Q), src)
    if mask is not None:
        if correct_mask:
            m = mask[:, None, :].expand(B, Q, T)
        else:
            rows = torch.arange(B * Q) % B
            m = mask[rows].reshape(B, Q, T)
        s = s.masked_fill(m, float("-inf"))
    p = torch.softmax(s, dim=2)
    wc = torch.einsum("bit,bqt->biq", src, rnd(p, dtype))
    return rnd(wc, dtype).reshape(B, idf, ih, iw), p.transpose(1, 2).reshape(B, T, ih, iw)


def _res_block(x, sd, p, dtype):
    s0, t0 = _fold(sd, p + "block.1.")
    s1, t1 = _fold(sd, p + "block.4.")
    y = conv_block(x, sd[p + "block.0.weight"], s0, t0, dtype, glu=True)
    return conv_block(y, sd[p + "block.3.weight"], s1, t1, dtype, residual=x)


def _up_block(x, sd, p, dtype):
    s, t = _fold(sd, p + "2.")
    return conv_block(x, sd[p + "1.weight"], s, t, dtype, glu=True, upsample=True, subpixel=True)


def _stem(x, sd, p, dtype):
    s, t = _fold(sd, p + "1.")
    return conv_block(x, sd[p + "0.weight"], s, t, dtype, glu=True, round_w=False)


def _stage(sd, p, h, words, mask, dtype, correct_mask):
    c, att = word_attention(h, words, sd[p + "att.conv_context.weight"], mask, dtype, correct_mask)
    x = torch.cat((h, c), 1)
    r = 0
    while (p + "residual.%d.block.0.weight" % r) in sd:
        x = _res_block(x, sd, p + "residual.%d." % r, dtype)
        r += 1
    return _up_block(x, sd, p + "upsample.", dtype), att


def g_sr_net_low(sd, LR, sent_emb, words, mask, dtype, correct_mask=False):
    mu, logvar = O.ca_net(sd, sent_emb, "ca_net.")
    imgs, atts = [], []
    h = _stem(LR, sd, "h_net1.im2f.", dtype)
    for k in (1, 2, 3):
        h, a = _stage(sd, "h_net%d." % k, h, words, mask, dtype, correct_mask)
        imgs.append(F.conv2d(h, rnd(sd["img_net%d.img.0.weight" % k], dtype), None, 1, 1))
        atts.append(a)
    return imgs, atts, mu, logvar


def trunk_dtype_of(dtype):
    """The bf16 configuration runs NetG_highweight's 32x32 trunk (convin + the chained ResBlocks, model.py:258-262) with
    f16 operands and storage (tgsr_amd/lp_pipeline.py: F16_TRUNK); its output is rounded once to bf16 for the up-scales."""
    return torch.float16 if dtype == torch.bfloat16 else dtype


def _gh_trunk(sd, x, dtype, trunk_dtype):
    td = trunk_dtype_of(dtype) if trunk_dtype is None else trunk_dtype
    out = _stem(x, sd, "convin.", td)
    r = 0
    while ("residual.%d.block.0.weight" % r) in sd:
        out = _res_block(out, sd, "residual.%d." % r, td)
        r += 1
    return out if td == dtype else rnd(out, dtype)            # tgsr_lp_convert


def netg_highweight(sd, LR, SRb: Sequence[Tensor], LRb, dtype, low="lr", trunk_dtype=None):
    x = LRb if low == "lrblur" else (LR - LRb if low == "lr-lrblur" else LR)
    out = _gh_trunk(sd, x, dtype, trunk_dtype)
    w5 = rnd(sd["conv_output.0.weight"], dtype)

    def head(o, sr):
        return torch.tanh(F.conv2d(o, w5, None, 1, 2)) + 0.5 * sr

    def nosum(x, p):
        s0, t0 = _fold(sd, p + "1.")
        s1, t1 = _fold(sd, p + "4.")
        y = conv_block(x, sd[p + "0.weight"], s0, t0, dtype, glu=True)
        return conv_block(y, sd[p + "3.weight"], s1, t1, dtype)

    out = _up_block(out, sd, "upscale2x.", dtype)
    ims2 = head(out, SRb[0])
    out = _up_block(nosum(out, "residual24."), sd, "upscale4x.", dtype)
    ims4 = head(out, SRb[1])
    out = _up_block(nosum(out, "residual48."), sd, "upscale8x.", dtype)
    return [ims2, ims4, head(out, SRb[2])]


def g_sr_net_low16(sd, LR, sent_emb, words, mask, dtype, correct_mask=False):
    """oracle.tgsr_oracle.g_sr_net_low16 (models16.py:5-39) with the lp path's rounding points: stages 2-4 through the
    ONE `h_net2` module, four tanh heads through the ONE `img_net1` filter."""
    mu, logvar = O.ca_net(sd, sent_emb, "ca_net.")
    w_img = rnd(sd["img_net1.img.0.weight"], dtype)
    imgs, atts = [], []
    h = _stem(LR, sd, "h_net1.im2f.", dtype)
    for k in range(4):
        h, a = _stage(sd, "h_net1." if k == 0 else "h_net2.", h, words, mask, dtype, correct_mask)
        imgs.append(torch.tanh(F.conv2d(h, w_img, None, 1, 1)))
        atts.append(a)
    return imgs, atts, mu, logvar


def netg_highweight16(sd, LR, SRb: Sequence[Tensor], LRb, dtype, low="lr", trunk_dtype=None):
    """oracle.tgsr_oracle.netg_highweight16 (models16.py:97-179) with the lp path's rounding points."""
    x = LRb if low == "lrblur" else (LR - LRb if low == "lr-lrblur" else LR)
    out = _gh_trunk(sd, x, dtype, trunk_dtype)
    w5, a = rnd(sd["conv_output.0.weight"], dtype), sd["a"]

    def head(o, sr):
        return torch.tanh(F.conv2d(o, w5, None, 1, 2)) + a * sr

    def nosum(x, p):
        s0, t0 = _fold(sd, p + "1.")
        s1, t1 = _fold(sd, p + "4.")
        y = conv_block(x, sd[p + "0.weight"], s0, t0, dtype, glu=True)
        return conv_block(y, sd[p + "3.weight"], s1, t1, dtype)

    out = _up_block(out, sd, "upscale2x.", dtype)
    ims = [head(out, SRb[0])]
    out = _up_block(nosum(out, "residual24."), sd, "upscale4x.", dtype)
    ims.append(head(out, SRb[1]))
    for k in (2, 3):
        out = _up_block(nosum(out, "residual48."), sd, "upscale8x.", dtype)
        ims.append(head(out, SRb[k]))
    return ims


def sr_forward16(sd_E, sd_GL, sd_GH, captions, cap_lens, LR, LRb, dtype=None, low="lr", correct_mask=False,
                 trunk_dtype=None):
    """The x16 caller wiring (trainer_objective.py:74-87 with BRANCH_NUM != 4); dtype None = the fp32 oracle."""
    words, sent = O.rnn_encoder(sd_E, captions, cap_lens)
    mask = (captions == 0)[:, :words.shape[2]]
    if dtype is None:
        imgs, atts, mu, logvar = O.g_sr_net_low16(sd_GL, LR, sent, words, mask, correct_mask=correct_mask)
        fine = O.netg_highweight16(sd_GH, LR, imgs, LRb, low)[0]
    else:
        imgs, atts, mu, logvar = g_sr_net_low16(sd_GL, LR, sent, words, mask, dtype, correct_mask)
        fine = netg_highweight16(sd_GH, LR, imgs, LRb, dtype, low, trunk_dtype)
    return {"words_emb": words, "sent_emb": sent, "mask": mask, "fake": imgs, "att": atts, "mu": mu, "logvar": logvar,
            "fine": fine}


def sr_forward(sd_E, sd_GL, sd_GH, captions, cap_lens, LR, LRb, dtype, low="lr", correct_mask=False, trunk_dtype=None):
    """`oracle.tgsr_oracle.sr_forward` with the lp path's rounding points (trunk_dtype None: the shipped choice,
    trunk_dtype_of(dtype); pass `dtype` itself for the uniform-type model)."""
    words, sent = O.rnn_encoder(sd_E, captions, cap_lens)
    mask = (captions == 0)[:, :words.shape[2]]
    imgs, atts, mu, logvar = g_sr_net_low(sd_GL, LR, sent, words, mask, dtype, correct_mask)
    fine = netg_highweight(sd_GH, LR, imgs, LRb, dtype, low, trunk_dtype)
    return {"words_emb": words, "sent_emb": sent, "mask": mask, "fake": imgs, "att": atts, "mu": mu, "logvar": logvar,
            "fine": fine}


def psnr(a: Tensor, b: Tensor, peak: float = 2.0) -> float:
    """10 log10(peak^2 / MSE); peak 2.0 = the [-1, 1] range images are normalised to (datasets.py:286-288)."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * torch.log10(torch.tensor(peak * peak / mse)).item()
