#!/usr/bin/env python3
"""How many parameter gradients of one SRTrainer step were written straight into the flat bucket (adopted by autograd
without an accumulation kernel) vs copied in afterwards.  python tools/check_grad_slots.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.train import SRTrainer
tr = SRTrainer(41, device="cuda")
B = 4
cap, lens, LR, LRb = synthetic_batch(B)
g = torch.Generator().manual_seed(7)
hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).cuda() for s in (64, 128, 256)]
tr._zero(tr.bucket)
loss, _, _ = tr.loss(cap.cuda(), lens.tolist(), LR.cuda(), LRb.cuda(), hr)
loss.backward()
adopted = sum(1 for p, v in zip(tr.bucket.params, tr.bucket.views) if p.grad is not None and p.grad.data_ptr() == v.data_ptr())
none = sum(1 for p in tr.bucket.params if p.grad is None)
print("params %d adopted-in-place %d  elsewhere %d  no-grad %d" % (len(tr.bucket.params), adopted, len(tr.bucket.params) - adopted - none, none))
tr.bucket.end_step()
