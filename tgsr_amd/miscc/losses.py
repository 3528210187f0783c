"""Drop-in for the hot-path losses of the reference's miscc/losses.py.

`words_loss` / `sent_loss` (DAMSM, losses.py:21-136) keep the reference's signatures and return tuples; the
per-caption Python loop of words_loss is one launch of the batched DAMSM kernel (tgsr_damsm_words_fwd).  The class
mask, the x gamma3 scale and the two cross entropies act on a [B, B] matrix and stay in torch, like sent_loss's
[B,256]x[256,B] product (a plain library GEMM).  `KL_loss` and `MSE` (losses.py:779-810) are scalar reductions.
Both are differentiable: words_loss through the HIP backward of the DAMSM kernel (tgsr_damsm_words_bwd, autograd.DamsmWords),
sent_loss through torch autograd over its [B, B] matrix.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .config import cfg


def _func_attention(query, context, gamma1):
    return ops.func_attention(query, context, gamma1)


def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """losses.py:12-18."""
    w12 = torch.sum(x1 * x2, dim)
    w1 = torch.norm(x1, 2, dim)
    w2 = torch.norm(x2, 2, dim)
    return (w12 / (w1 * w2).clamp(min=eps)).squeeze()


def _class_masks(class_ids, batch_size, device):
    """losses.py:25-35 / 75-80: masks[i][j] = (class_ids[j] == class_ids[i]) and i != j."""
    if class_ids is None:
        return None
    ids = np.asarray(class_ids)
    m = ids[None, :] == ids[:, None]
    np.fill_diagonal(m, False)
    return torch.from_numpy(m).to(device)


def sent_loss(cnn_code, rnn_code, labels, class_ids, batch_size, eps=1e-8):
    """losses.py:21-62."""
    masks = _class_masks(class_ids, batch_size, cnn_code.device)
    if cnn_code.dim() == 2:
        cnn_code = cnn_code.unsqueeze(0)
        rnn_code = rnn_code.unsqueeze(0)
    cnn_code_norm = torch.norm(cnn_code, 2, dim=2, keepdim=True)
    rnn_code_norm = torch.norm(rnn_code, 2, dim=2, keepdim=True)
    scores0 = torch.bmm(cnn_code, rnn_code.transpose(1, 2))
    norm0 = torch.bmm(cnn_code_norm, rnn_code_norm.transpose(1, 2))
    scores0 = scores0 / norm0.clamp(min=eps) * cfg.TRAIN.SMOOTH.GAMMA3
    scores0 = scores0.squeeze(dim=0)
    if masks is not None:
        scores0 = scores0.masked_fill(masks, -float('inf'))
    scores1 = scores0.transpose(0, 1)
    if labels is not None:
        loss0 = nn.CrossEntropyLoss()(scores0, labels)
        loss1 = nn.CrossEntropyLoss()(scores1, labels)
    else:
        loss0, loss1 = None, None
    return loss0, loss1


def words_loss(img_features, words_emb, labels, cap_lens, class_ids, batch_size):
    """losses.py:65-136.  words_emb(query): batch x nef x seq_len; img_features(context): batch x nef x 17 x 17.
    Returns (loss0, loss1, att_maps) with att_maps[i] = [1, cap_len_i, 17, 17]."""
    lens = cap_lens.data.tolist() if torch.is_tensor(cap_lens) else list(cap_lens)
    if torch.is_grad_enabled() and (img_features.requires_grad or words_emb.requires_grad):
        from ..autograd import DamsmWords
        sim, att = DamsmWords.apply(img_features, words_emb, lens, cfg.TRAIN.SMOOTH.GAMMA1, cfg.TRAIN.SMOOTH.GAMMA2)
    else:
        sim, att = ops.damsm_words_similarity(img_features, words_emb, lens, cfg.TRAIN.SMOOTH.GAMMA1,
                                              cfg.TRAIN.SMOOTH.GAMMA2, need_att=True)
    att_maps = [att[i:i + 1, :lens[i]].contiguous() for i in range(batch_size)]
    similarities = sim * cfg.TRAIN.SMOOTH.GAMMA3
    masks = _class_masks(class_ids, batch_size, sim.device)
    if masks is not None:
        similarities = similarities.masked_fill(masks, -float('inf'))
    similarities1 = similarities.transpose(0, 1)
    if labels is not None:
        loss0 = nn.CrossEntropyLoss()(similarities, labels)
        loss1 = nn.CrossEntropyLoss()(similarities1, labels)
    else:
        loss0, loss1 = None, None
    return loss0, loss1, att_maps


def MSE(fake, label):
    """losses.py:779-784."""
    mseloss = 0
    for i in range(len(fake)):
        mseloss += nn.MSELoss()(fake[i], label[i])
    return mseloss


def KL_loss(mu, logvar):
    """losses.py:806-810."""
    KLD_element = mu.pow(2).add(logvar.exp()).mul(-1).add(1).add(logvar)
    return torch.mean(KLD_element).mul(-0.5)


def discriminator_loss(netD, real_imgs, fake_imgs, conditions, real_labels, fake_labels):
    """losses.py:290-316.  The reference ships no discriminator class; any module exposing `COND_DNET` /
    `UNCOND_DNET` like AttnGAN's D_NET* works (real / fake / wrong-caption terms, the wrong pair is the batch shifted
    by one, :302)."""
    real_features = netD(real_imgs)
    fake_features = netD(fake_imgs.detach())
    bce = nn.BCEWithLogitsLoss()
    cond_real_errD = bce(netD.COND_DNET(real_features, conditions), real_labels)
    cond_fake_errD = bce(netD.COND_DNET(fake_features, conditions), fake_labels)
    batch_size = real_features.size(0)
    cond_wrong_logits = netD.COND_DNET(real_features[:(batch_size - 1)], conditions[1:batch_size])
    cond_wrong_errD = bce(cond_wrong_logits, fake_labels[1:batch_size])
    if netD.UNCOND_DNET is not None:
        real_errD = bce(netD.UNCOND_DNET(real_features), real_labels)
        fake_errD = bce(netD.UNCOND_DNET(fake_features), fake_labels)
        return (real_errD + cond_real_errD) / 2. + (fake_errD + cond_fake_errD + cond_wrong_errD) / 3.
    return cond_real_errD + (cond_fake_errD + cond_wrong_errD) / 2.


def generator_loss(netsD, image_encoder, fake_imgs, real_labels, words_embs, sent_emb, match_labels, cap_lens,
                   class_ids, w=1, s=1, g=1):
    """losses.py:351-391: per-scale adversarial terms + the DAMSM words/sentence ranking loss on the last scale."""
    numDs = len(netsD)
    batch_size = real_labels.size(0)
    logs = ''
    errG_total = 0
    bce = nn.BCEWithLogitsLoss()
    for i in range(numDs):
        features = netsD[i](fake_imgs[i])
        cond_errG = bce(netsD[i].COND_DNET(features, sent_emb), real_labels)
        if netsD[i].UNCOND_DNET is not None:
            g_loss = bce(netsD[i].UNCOND_DNET(features), real_labels) + cond_errG
        else:
            g_loss = cond_errG
        g_loss = g * g_loss
        errG_total += g_loss
        logs += 'g_loss%d: %.5f ' % (i, g_loss.item())
        if i == (numDs - 1):
            region_features, cnn_code = image_encoder(fake_imgs[i])
            w_loss0, w_loss1, _ = words_loss(region_features, words_embs, match_labels, cap_lens, class_ids, batch_size)
            w_loss = w * (w_loss0 + w_loss1) * cfg.TRAIN.SMOOTH.LAMBDA
            s_loss0, s_loss1 = sent_loss(cnn_code, sent_emb, match_labels, class_ids, batch_size)
            s_loss = s * (s_loss0 + s_loss1) * cfg.TRAIN.SMOOTH.LAMBDA
            errG_total += w_loss + s_loss
            logs += 'w_loss: %.5f s_loss: %.5f ' % (w_loss.item(), s_loss.item())
    return errG_total, logs
