"""SR generator training step (the loop the reference never shipped - SURVEY section 3.3).

What the reference pins down and this harness uses: the loss functions and their conventions (`MSE` losses.py:779,
`KL_loss` :806, `generator_loss` / `discriminator_loss` :290-391 when a discriminator is supplied), labels
(`prepare_labels`, trainer_objective.py:43-53), Adam(lr 2e-4, betas (0.5, 0.999)) (config.py:37-38,
pretrain_DAMSM.py:270), the EMA helpers `copy_G_params` / `load_params` (miscc/utils.py:467-474), BatchNorm in
training mode.  What it does NOT pin down (no caller exists): loss weights, update order, the discriminator
architecture (no class anywhere in the reference) and the Inception image encoder (third-party weights).  This
harness therefore trains the two generators on the pixel + KL terms,
    errG = MSE(fake_imgL, HR pyramid) + MSE(fine_im, HR pyramid) + KL(mu, logvar),
and takes the adversarial / DAMSM terms only when the caller supplies `netsD` / `image_encoder`.
Every forward and backward kernel of the generators is HIP (tgsr_amd.autograd); the text encoder is frozen (eval).
Data parallel: gradients live in one flat bucket, one all-reduce per step (tgsr_amd.parallel.FlatGradBucket).
"""
from copy import deepcopy

import torch

from .miscc import losses
from .miscc.config import cfg
from .model import G_SR_NET_low, NetG_highweight, RNN_ENCODER
from .parallel import FlatGradBucket
from .trainer import caption_mask


def copy_G_params(model):
    """miscc/utils.py:472-474."""
    return deepcopy(list(p.data for p in model.parameters()))


def load_params(model, new_param):
    """miscc/utils.py:467-469."""
    for p, new_p in zip(model.parameters(), new_param):
        p.data.copy_(new_p)


def prepare_labels(batch_size, device):
    """trainer_objective.py:43-53."""
    return (torch.ones(batch_size, device=device), torch.zeros(batch_size, device=device),
            torch.arange(batch_size, device=device))


class SRTrainer:
    def __init__(self, n_words, device="cuda", low="lr", lr=None, ema_decay=0.999):
        self.device = torch.device(device)
        self.text_encoder = RNN_ENCODER(n_words, nhidden=cfg.TEXT.EMBEDDING_DIM).to(self.device).eval()
        for p in self.text_encoder.parameters():
            p.requires_grad = False
        self.netGL = G_SR_NET_low().to(self.device).train()
        self.netGH = NetG_highweight(weightmap=False, low=low).to(self.device).train()
        self.params = list(self.netGL.parameters()) + list(self.netGH.parameters())
        self.bucket = FlatGradBucket(self.params).attach()
        self.opt = torch.optim.Adam(self.params, lr=lr or cfg.TRAIN.GENERATOR_LR, betas=(0.5, 0.999))
        self.ema_decay = ema_decay
        self.avg_param_G = copy_G_params(self.netGL) + copy_G_params(self.netGH)

    def loss(self, captions, cap_lens, LR, LRb, hr_pyramid):
        """hr_pyramid: the 3 target scales [B,3,2s,2s], [B,3,4s,4s], [B,3,8s,8s]."""
        with torch.no_grad():
            words_embs, sent_emb = self.text_encoder(captions, cap_lens, self.text_encoder.init_hidden(captions.shape[0]))
        mask = caption_mask(captions, words_embs.size(2))
        fake_imgL, _att, mu, logvar = self.netGL(LR, sent_emb, words_embs, mask)
        fine_im, _a, _one = self.netGH(LR, fake_imgL, LRb)
        errG = losses.MSE(fake_imgL, hr_pyramid) + losses.MSE(fine_im, hr_pyramid) + losses.KL_loss(mu, logvar)
        return errG, fake_imgL, fine_im

    def step(self, captions, cap_lens, LR, LRb, hr_pyramid):
        """forward + backward + gradient all-reduce (if distributed) + Adam + EMA.  Returns the loss tensor."""
        self.bucket.flat.zero_()                      # p.grad are views of the flat bucket (attach()): one memset
        for p, v in zip(self.bucket.params, self.bucket.views):
            p.grad = v
        errG, _, _ = self.loss(captions, cap_lens, LR, LRb, hr_pyramid)
        errG.backward()
        self.bucket.all_reduce_mean()
        self.opt.step()
        with torch.no_grad():
            torch._foreach_mul_(self.avg_param_G, self.ema_decay)
            torch._foreach_add_(self.avg_param_G, [p.data for p in self.params], alpha=1.0 - self.ema_decay)
        return errG.detach()
