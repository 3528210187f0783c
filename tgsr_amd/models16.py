"""Drop-in for the reference's models16.py (x16, TREE.BRANCH_NUM = 5): same kernels as model.py, different wiring.

Reference quirks kept: ONE NEXT_STAGE_G object serves stages 2-4 and ONE GET_IMAGE_G serves the four
heads (`h_net4 = h_net3 = h_net2`, `img_net4 = ... = img_net1`, models16.py:13-14), so the state_dict lists the tied
tensors under every alias, exactly like the reference.  In NetG_highweight the 16x stage re-uses `residual48` /
`upscale8x` (models16.py:172-173; `residual816` / `upscale16x` hold parameters but are never called).
One deviation, because the shipped line cannot execute: models16.py:178 adds the 8x image `SRb8` to the 16x tensor
(a shape error in the reference - pinned by tests/golden/nets16_small.npz `gh16_runs == 0`); here the 16x head adds
`SRb16`.
"""
import torch
import torch.nn as nn

from . import custom_ops as C
from . import ops
from .miscc.config import cfg
from .model import *  # noqa: F401,F403  (the reference does `from model import *`)
from .util import (CA_NET, GET_IMAGE_G, INIT_STAGE_GImgup, NEXT_STAGE_G, ResBlock, _ConvBnGlu, _ResidualNoSum,
                   conv5x5, upBlock)


class G_SR_NET_low(nn.Module):
    """models16.py:5-39."""

    def __init__(self):
        super(G_SR_NET_low, self).__init__()
        ngf = cfg.GAN.GF_DIM
        nef = cfg.TEXT.EMBEDDING_DIM
        ncf = cfg.GAN.CONDITION_DIM
        self.ca_net = CA_NET()
        self.h_net1 = INIT_STAGE_GImgup(ngf, ncf, nef)
        self.h_net4 = self.h_net3 = self.h_net2 = NEXT_STAGE_G(ngf, nef, ncf)
        self.img_net4 = self.img_net3 = self.img_net2 = self.img_net1 = GET_IMAGE_G(ngf)

    def attention_modules(self):
        """The distinct GlobalAttentionGeneral modules in stage order (h_net1's and the tied stages')."""
        return [self.h_net1.att, self.h_net2.att]

    def forward(self, LR, sent_emb, word_embs, mask, ca=None, proj=None):
        """ca / proj: optional precomputed `self.ca_net(sent_emb)` and conv_context projections of `attention_modules()`
        (SRPipeline computes them, and the mask, in one launch - like the x8 model)."""
        fake_imgs, att_maps = [], []
        c_code, mu, logvar = self.ca_net(sent_emb) if ca is None else ca
        src1 = src2 = None
        if proj is not None:
            src1, src2 = proj
        elif not self.training:
            # two distinct conv_context projections (h_net1's and the tied stages'): one launch for both
            src1, src2 = C.word_project(word_embs, [a.conv_context.weight.detach() for a in self.attention_modules()])
        h_code, att = self.h_net1(None, LR, word_embs, mask, wide_out=True, src=src1)
        fake_imgs.append(self.img_net1(h_code))
        att_maps.append(att)
        for k, (stage, head) in enumerate(((self.h_net2, self.img_net2), (self.h_net3, self.img_net3),
                                           (self.h_net4, self.img_net4))):
            h_code, att = stage(h_code, None, word_embs, mask, wide_out=(k < 2), src=src2)
            fake_imgs.append(head(h_code))
            att_maps.append(att)
        return fake_imgs, att_maps, mu, logvar


class NetG_highweight(nn.Module):
    """models16.py:97-179.  weightmap=False: `a` IS a registered parameter here (no `.cuda()` on it, models16.py:126), initial
    value 0.5; `one` is the constant 1.  weightmap=True (models16.py:119-125): four trainable maps `a1..a4` of 32 / 64 / 128 /
    256 pixels (for 16 x 16 inputs), initial value 1, no `a`; forward returns (ims, a4, one4)."""

    MAP_SIZES = (32, 64, 128, 256)

    def __init__(self, weightmap=False, low='lr-lrblur'):
        super(NetG_highweight, self).__init__()
        ngf = cfg.GAN.GF_DIM
        self.low = low
        self.residual = nn.Sequential(*[ResBlock(channel_num=32) for _ in range(6)])
        self.upscale4x = upBlock(ngf, ngf)
        self.upscale2x = upBlock(ngf, ngf)
        self.upscale8x = upBlock(ngf, ngf)
        self.upscale16x = upBlock(ngf, ngf)          # parameters only: never called (models16.py:173)
        self.conv_output = nn.Sequential(conv5x5(ngf, 3), nn.Tanh())
        self.convin = _ConvBnGlu(3, ngf)
        self.residual24 = _ResidualNoSum(ngf)
        self.residual48 = _ResidualNoSum(ngf)
        self.residual816 = _ResidualNoSum(ngf)       # parameters only: never called (models16.py:172)
        self.weightmap = bool(weightmap)
        if self.weightmap:
            for k, n in enumerate(self.MAP_SIZES):
                setattr(self, "a%d" % (k + 1), nn.Parameter(torch.ones([n, n], dtype=torch.float32)))
        else:
            self.a = nn.Parameter(torch.FloatTensor([0.5]))
        self._one = {}
        self._a_host = (None, 0.5)       # (version key, host copy of `a`): one D2H sync per weight version, not per head

    def maps(self):
        return [getattr(self, "a%d" % (k + 1)) for k in range(len(self.MAP_SIZES))] if self.weightmap else None

    def _head(self, out, SRb, k=0):
        if self.weightmap:                                    # one_k * conv_output(out) + a_k * SRb  (models16.py:150, 159, 167, 175)
            amap = self.maps()[k]
            if tuple(amap.shape) != tuple(out.shape[2:]):
                raise ValueError("models16.NetG_highweight(weightmap=True): a%d is %s but scale %d of this input is %s (the maps "
                                 "are sized for 16 x 16 inputs, models16.py:120-123)" % (k + 1, tuple(amap.shape), k, tuple(out.shape[2:])))
            return C.axpy_map(C.conv_to3(out, self.conv_output[0].weight, True, None, 0.0), SRb, amap)
        if self.training:
            from .autograd import ConvTo3
            return ConvTo3.apply(out, self.conv_output[0].weight, SRb, True, self.a)   # d/da = sum(dy * SRb)
        return C.conv_to3(out, self.conv_output[0].weight.detach(), True, SRb, self.alpha())

    def trunk(self, LR, LRb):
        """Everything that does not need the low-frequency images (see model.NetG_highweight.trunk): the four feature
        maps the heads read."""
        if self.low == 'lrblur':
            x = LRb
        elif self.low == 'lr-lrblur':
            x = LR - LRb
        else:
            x = LR
        out2 = self.upscale2x(self.residual(self.convin(x)))
        out4 = self.upscale4x(self.residual24(out2))
        out8 = self.upscale8x(self.residual48(out4))
        out16 = self.upscale8x(self.residual48(out8))   # models16.py:172-173: the 16x stage re-uses the 8x modules
        return out2, out4, out8, out16

    def heads(self, feats, SRb):
        """ims_k = one * tanh(conv5x5(out_k)) + a * SRb_k; the 16x head adds SRb16 (models16.py:178 says SRb8: a shape
        error as shipped)."""
        return [self._head(f, sr, k) for k, (f, sr) in enumerate(zip(feats, SRb[:4]))]

    def tanh_heads(self, feats):
        """tanh(conv5x5(out_k)) of every scale - needs no low-frequency image (see model.NetG_highweight.tanh_heads)."""
        w = self.conv_output[0].weight.detach()
        return [C.conv_to3(f, w, True, None, 0.0 if self.weightmap else self.alpha()) for f in feats]

    def finish_heads(self, ts, SRb):
        if self.weightmap:
            return [C.axpy_map(t, s.contiguous(), a.detach()) for t, s, a in zip(ts, SRb, self.maps())]
        return list(C.axpy_images(list(ts), [s.contiguous() for s in SRb[:len(ts)]], self.alpha()))

    def forward(self, LR, SRb, LRb):
        ims = self.heads(self.trunk(LR, LRb), SRb)
        one = self._one.get(LR.device)
        if one is None:
            one = self._one[LR.device] = LR.new_ones(1)
        return ims, (self.a4 if self.weightmap else self.a), one

    def alpha(self):
        """Host value of `a` for the inference kernels: one D2H copy per weight version, none per step (and none inside
        a hipGraph capture once a warm-up step has run)."""
        key = (self.a.data_ptr(), self.a._version)
        if self._a_host[0] != key:
            self._a_host = (key, float(self.a.item()))
        return self._a_host[1]
