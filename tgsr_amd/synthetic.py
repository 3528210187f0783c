"""Synthetic workload of the headline benchmark (SURVEY.md 8d / BASELINE.md 4): seeded inputs of the CelebA x8
shape and, when no checkpoint is at hand, seeded random parameters with the reference's key names and shapes."""
import math

import torch


def synthetic_batch(B, n_words=41, seed=100, lr=32, width=18, fixed_len=None):
    """LR, LRb ~ U(-1,1) [B,3,lr,lr]; captions int64 [B,18], lengths from {4..14} sorted descending (seed 100 =
    test1.py:170)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(4, 15, (B,), generator=g) if fixed_len is None else torch.full((B,), fixed_len)
    lens = torch.sort(lens, descending=True)[0]
    cap = torch.zeros(B, width, dtype=torch.int64)
    for i in range(B):
        cap[i, :int(lens[i])] = torch.randint(1, n_words, (int(lens[i]),), generator=g)
    LR = torch.rand(B, 3, lr, lr, generator=g) * 2 - 1
    LRb = torch.rand(B, 3, lr, lr, generator=g) * 2 - 1
    return cap, lens, LR, LRb


def random_init_(module, seed=0):
    """Variance-preserving seeded init for benches without a checkpoint (conv ~ N(0, 1/fan_in), BN identity-ish)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.Conv2d):
                fan = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) / math.sqrt(fan))
            elif isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1 + 0.02 * torch.randn(m.weight.shape, generator=g))
                m.bias.zero_()
                m.running_mean.zero_()
                m.running_var.fill_(1.0)
    return module
