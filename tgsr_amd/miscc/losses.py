"""Drop-in for the hot-path losses of the reference's miscc/losses.py.

`words_loss` / `sent_loss` (DAMSM, losses.py:21-136) keep the reference's signatures and return tuples; the
per-caption Python loop of words_loss is one launch of the batched DAMSM kernel (tgsr_damsm_words_fwd).  The class
mask, the x gamma3 scale and the two cross entropies act on a [B, B] matrix and stay in torch, like sent_loss's
[B,256]x[256,B] product (a plain library GEMM).  `KL_loss` and `MSE` (losses.py:779-810) are scalar reductions.
Both are differentiable: words_loss through the HIP backward of the DAMSM kernel (tgsr_damsm_words_bwd, autograd.DamsmWords),
sent_loss through torch autograd over its [B, B] matrix.
"""
import contextlib

import numpy as np

import torch
import torch.nn.functional as F

from .. import custom_ops as C
from .config import cfg


def _func_attention(query, context, gamma1):
    return C.func_attention(query, context, float(gamma1))


def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """losses.py:12-18: <x1, x2> / max(|x1| |x2|, eps) along `dim`."""
    dot = (x1 * x2).sum(dim)
    scale = (x1.norm(p=2, dim=dim) * x2.norm(p=2, dim=dim)).clamp_min(eps)
    return (dot / scale).squeeze()


def _class_masks(class_ids, batch_size, device):
    """losses.py:25-35 / 75-80: masks[i][j] = (class_ids[j] == class_ids[i]) and i != j."""
    if class_ids is None:
        return None
    ids = np.asarray(class_ids)
    m = ids[None, :] == ids[:, None]
    np.fill_diagonal(m, False)
    return torch.from_numpy(m).to(device)


def _ranking_ce(scores, labels):
    """The two cross entropies of a [B, B] matching matrix (rows = images, columns = captions) - losses.py:56-59."""
    if labels is None:
        return None, None
    return F.cross_entropy(scores, labels), F.cross_entropy(scores.t(), labels)


def sent_loss(cnn_code, rnn_code, labels, class_ids, batch_size, eps=1e-8):
    """losses.py:21-62: gamma3-scaled cosine matrix between the image codes and the sentence codes, same-class pairs
    masked out, cross entropy along both axes."""
    img = cnn_code.reshape(-1, cnn_code.shape[-1])                    # the reference's [1, B, D] form is the same matrix
    txt = rnn_code.reshape(-1, rnn_code.shape[-1])
    norms = img.norm(p=2, dim=1, keepdim=True) * txt.norm(p=2, dim=1, keepdim=True).t()
    scores = img @ txt.t() / norms.clamp_min(eps) * cfg.TRAIN.SMOOTH.GAMMA3
    same_class = _class_masks(class_ids, batch_size, scores.device)
    if same_class is not None:
        scores = scores.masked_fill(same_class, float("-inf"))
    return _ranking_ce(scores, labels)


def words_loss(img_features, words_emb, labels, cap_lens, class_ids, batch_size):
    """losses.py:65-136.  words_emb(query): batch x nef x seq_len; img_features(context): batch x nef x 17 x 17.
    Returns (loss0, loss1, att_maps) with att_maps[i] = [1, cap_len_i, 17, 17]."""
    lens = cap_lens.data.tolist() if torch.is_tensor(cap_lens) else list(cap_lens)
    if torch.is_grad_enabled() and (img_features.requires_grad or words_emb.requires_grad):
        from ..autograd import DamsmWords
        sim, att = DamsmWords.apply(img_features, words_emb, lens, cfg.TRAIN.SMOOTH.GAMMA1, cfg.TRAIN.SMOOTH.GAMMA2)
    else:
        sim, att = C.damsm_words(img_features, words_emb, [int(v) for v in lens], float(cfg.TRAIN.SMOOTH.GAMMA1),
                                 float(cfg.TRAIN.SMOOTH.GAMMA2))
    att_maps = [att[i:i + 1, :lens[i]].contiguous() for i in range(batch_size)]
    similarities = sim * cfg.TRAIN.SMOOTH.GAMMA3
    masks = _class_masks(class_ids, batch_size, sim.device)
    if masks is not None:
        similarities = similarities.masked_fill(masks, -float('inf'))
    loss0, loss1 = _ranking_ce(similarities, labels)
    return loss0, loss1, att_maps


def MSE(fake, label):
    """losses.py:779-784: sum over the image scales of the mean squared error."""
    return sum(F.mse_loss(f, t) for f, t in zip(fake, label))


def KL_loss(mu, logvar):
    """losses.py:806-810: -0.5 * mean(1 + logvar - mu^2 - exp(logvar))."""
    return -0.5 * torch.mean(1.0 + logvar - mu * mu - logvar.exp())


def _bce(logits, target):
    return F.binary_cross_entropy_with_logits(logits, target)


_BCE_WEIGHTS = {}


def _bce_weights(device, *runs):
    """The per-logit weights of a sum of mean-reduced BCE terms, `runs` = (count, weight of every element) in order: a device
    tensor built from fill kernels once per distinct argument list (no host copy: also legal inside a hipGraph capture)."""
    key = (str(device),) + runs
    w = _BCE_WEIGHTS.get(key)
    if w is None:
        w = _BCE_WEIGHTS[key] = torch.cat([torch.full((int(n),), float(v), dtype=torch.float32, device=device) for n, v in runs])
    return w


def _fused_bce_ok(*ts):
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in ts)


def discriminator_loss(netD, real_imgs, fake_imgs, conditions, real_labels, fake_labels):
    """losses.py:290-316.  The reference ships no discriminator class; any module exposing `COND_DNET` /
    `UNCOND_DNET` like AttnGAN's D_NET* works: real / fake / wrong-caption terms, the wrong pair being the batch
    shifted by one (:302).
    Under data parallelism (decided here): the shift stays INSIDE the shard - rank r pairs its image i with its
    caption i + 1, so world x (B - 1) wrong pairs are scored instead of world x B - 1 (the pair that would straddle two shards
    is dropped) and each weighs 1 / (world (B - 1)) after the gradient all-reduce's average.  The discriminators normalise with
    per-shard BatchNorm statistics anyway (parallel.py), so this loss is a per-shard quantity by construction; no halo exchange
    of conditions is made for one pair in B."""
    n = real_imgs.size(0)
    if getattr(netD, "supports_groups", False) and n > 1:
        # the build's D_NET*: the real and the fake pass as ONE batch whose two halves keep their own BatchNorm batch
        # statistics (and update the running statistics in the reference's order: real, then fake) - one convolution,
        # data-gradient and weight-gradient launch per layer instead of two, and every parameter receives one gradient;
        # the three conditional heads (real, fake, mismatched) likewise.  Numerically the passes run one by one.
        feats = netD(torch.cat((real_imgs, fake_imgs.detach())), groups=(n, n))
        feat_real, feat_fake = feats[:n], feats[n:]
        lc = netD.COND_DNET(torch.cat((feats, feat_real[:n - 1])), torch.cat((conditions, conditions, conditions[1:n])),
                            groups=(n, n, n - 1))
        lu = netD.UNCOND_DNET(feats) if netD.UNCOND_DNET is not None else None
        if _fused_bce_ok(lc, lu, real_labels, fake_labels):
            # the five mean-reduced BCE terms and their /2, /3 combination as ONE launch (tgsr::weighted_bce; its backward another):
            # ~40 small aten launches per discriminator and step otherwise (BCE, mean, slice backward fills, adds)
            tc = (real_labels, fake_labels, fake_labels[1:n])
            if lu is None:
                return C.weighted_bce(lc, None, torch.cat(tc), _bce_weights(lc.device, (n, 1. / n), (n, .5 / n), (n - 1, .5 / (n - 1))))
            return C.weighted_bce(lc, lu, torch.cat(tc + (real_labels, fake_labels)),
                                  _bce_weights(lc.device, (n, .5 / n), (n, 1. / (3 * n)), (n - 1, 1. / (3 * (n - 1))), (n, .5 / n),
                                               (n, 1. / (3 * n))))
        cond = {"real": _bce(lc[:n], real_labels), "fake": _bce(lc[n:2 * n], fake_labels),
                "wrong": _bce(lc[2 * n:], fake_labels[1:n])}
        if lu is None:
            return cond["real"] + (cond["fake"] + cond["wrong"]) / 2.
        return (_bce(lu[:n], real_labels) + cond["real"]) / 2. + (_bce(lu[n:], fake_labels) + cond["fake"] + cond["wrong"]) / 3.
    feat_real, feat_fake = netD(real_imgs), netD(fake_imgs.detach())
    cond = {"real": _bce(netD.COND_DNET(feat_real, conditions), real_labels),
            "fake": _bce(netD.COND_DNET(feat_fake, conditions), fake_labels),
            "wrong": _bce(netD.COND_DNET(feat_real[:n - 1], conditions[1:n]), fake_labels[1:n])}
    if netD.UNCOND_DNET is None:
        return cond["real"] + (cond["fake"] + cond["wrong"]) / 2.
    unc_real = _bce(netD.UNCOND_DNET(feat_real), real_labels)
    unc_fake = _bce(netD.UNCOND_DNET(feat_fake), fake_labels)
    return (unc_real + cond["real"]) / 2. + (unc_fake + cond["fake"] + cond["wrong"]) / 3.


class _LazyLog:
    """The log string of generator_loss, formatted when somebody looks at it: the reference builds it with `.item()` on
    every term (losses.py:366-390) - three device synchronisations inside the generator step that nothing needs unless
    the string is printed.  Behaves like the str it stands for under str(), format(), +, +=, len, in, slicing and ==; it
    is NOT a str instance (a str subclass cannot fill its character buffer later, so C-level consumers - file.write,
    ''.join - would silently see an empty string).  Only handed out on request: generator_loss(lazy_log=True), which
    train.SRTrainer passes; the default return is the reference's plain str.  The loss tensors are released once the
    text has been formatted."""

    def __init__(self, parts):
        self._parts = parts                      # [(format, tensor, ...), ...]; dropped once formatted
        self._text = None

    def __str__(self):
        if self._text is None:
            self._text = "".join(f % tuple(float(t) for t in ts) for f, *ts in self._parts)
            self._parts = None
        return self._text

    __repr__ = __str__

    def __format__(self, spec):
        return format(str(self), spec)

    def __add__(self, other):
        return str(self) + str(other)

    def __radd__(self, other):
        return str(other) + str(self)

    def __len__(self):
        return len(str(self))

    def __contains__(self, item):
        return item in str(self)

    def __getitem__(self, k):
        return str(self)[k]

    def __eq__(self, other):
        return str(self) == other

    def __hash__(self):
        return hash(str(self))


def damsm_terms(regions, code, words_embs, sent_emb, cap_lens, class_ids, gather=False):
    """(w_loss0, w_loss1, s_loss0, s_loss1, scale, att_maps) of one shard's images and captions.  gather=False: the shard's own
    B x B matching matrices (what the single-process reference computes), scale 1.  gather=True under torch.distributed: the
    matrices of the GLOBAL batch (parallel.gather_damsm_batch; the same four numbers on every rank = the single-process
    losses of the concatenated batch) and scale = world - multiply the term by it before backward, the gradient bucket
    averages over ranks what a replicated loss needs summed."""
    scale = 1
    if gather:
        from .. import parallel
        regions, code, words_embs, sent_emb, cap_lens, class_ids, B, scale = parallel.gather_damsm_batch(
            regions, code, words_embs, sent_emb, cap_lens, class_ids, cfg.TEXT.WORDS_NUM)
    else:
        B = sent_emb.shape[0]
    labels = torch.arange(B, device=sent_emb.device)
    w0, w1, att = words_loss(regions, words_embs, labels, cap_lens, class_ids, B)
    s0, s1 = sent_loss(code, sent_emb, labels, class_ids, B)
    return w0, w1, s0, s1, scale, att


def generator_loss(netsD, image_encoder, fake_imgs, real_labels, words_embs, sent_emb, match_labels, cap_lens,
                   class_ids, w=1, s=1, g=1, streams=None, lazy_log=False, gather_negatives=False, enc_out=None):
    """losses.py:351-391: per-scale adversarial terms + the DAMSM words / sentence ranking loss on the last scale
    (x TRAIN.SMOOTH.LAMBDA).  Returns (total, log) like the reference: `log` is the same str; with `lazy_log=True` (not a
    reference argument) a _LazyLog stands in for it - the same text, formatted (and the device synchronised) only when it
    is looked at.  `image_encoder=None` (the reference
    always has one; its Inception-v3 weights are third-party and not shipped) leaves the ranking term out.  `enc_out` (not a
    reference argument): `image_encoder(fake_imgs[-1])` computed by the caller already - the encoder reads the fake image only, not
    the discriminators, so train.SRTrainer runs it beside the discriminator updates that precede this call."""
    B = real_labels.size(0)
    total, parts = 0, []
    advs = []
    main = torch.cuda.current_stream(real_labels.device) if streams else None
    for k, (netD, img) in enumerate(zip(netsD, fake_imgs)):
        # `streams` (not a reference argument): one stream per discriminator - their forward passes, and through
        # autograd's stream rule their backward passes, run side by side; the terms are added up on the calling stream
        st = streams[k] if streams else None
        if st is not None:
            st.wait_stream(main)
        with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
            if st is not None:
                for t in (img, sent_emb, real_labels):
                    t.record_stream(st)                    # allocated on the calling stream, read on this one
            feat = netD(img)
            lc = netD.COND_DNET(feat, sent_emb)
            lu = netD.UNCOND_DNET(feat) if netD.UNCOND_DNET is not None else None
            if _fused_bce_ok(lc, lu, real_labels) and lc.dim() == 1:
                # g * (BCE(cond logits, real) + BCE(uncond logits, real)), one launch
                adv = C.weighted_bce(lc, lu, real_labels if lu is None else torch.cat((real_labels, real_labels)),
                                     _bce_weights(lc.device, (lc.numel() + (0 if lu is None else lu.numel()), float(g) / B)))
            else:
                adv = _bce(lc, real_labels)
                if lu is not None:
                    adv = adv + _bce(lu, real_labels)
                adv = g * adv
        advs.append(adv)
    # the ranking term's image encoder (the frozen trunk: ~190 launches forward) is issued BEFORE the calling stream joins the
    # discriminators' streams: it reads the last fake image only, so it runs beside their forward passes (and, through autograd's
    # stream rule, its backward beside theirs)
    if enc_out is None and image_encoder is not None and len(netsD) > 0:
        enc_out = image_encoder(fake_imgs[len(netsD) - 1])
    for k, (netD, img) in enumerate(zip(netsD, fake_imgs)):
        adv = advs[k]
        if streams:
            main.wait_stream(streams[k])
            adv.record_stream(main)
        total = total + adv
        parts.append(("g_loss%d: %%.5f " % k, adv.detach()))
        if k == len(netsD) - 1 and image_encoder is not None:
            regions, code = enc_out
            if gather_negatives:        # (not a reference argument) data parallel: the ranking term of the global batch
                w0, w1, s0, s1, scale, _ = damsm_terms(regions, code, words_embs, sent_emb, cap_lens, class_ids, gather=True)
            else:
                w0, w1, _ = words_loss(regions, words_embs, match_labels, cap_lens, class_ids, B)
                s0, s1 = sent_loss(code, sent_emb, match_labels, class_ids, B)
                scale = 1
            w_term = w * (w0 + w1) * cfg.TRAIN.SMOOTH.LAMBDA
            s_term = s * (s0 + s1) * cfg.TRAIN.SMOOTH.LAMBDA
            total = total + (w_term + s_term) * scale
            parts.append(("w_loss: %.5f s_loss: %.5f ", w_term.detach(), s_term.detach()))
    log = _LazyLog(parts)
    return total, (log if lazy_log else str(log))
