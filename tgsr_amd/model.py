"""Drop-in for the reference's model.py (x8 generators): `G_SR_NET_low` and `NetG_highweight` on HIP kernels.

`from model import RNN_ENCODER, G_SR_NET_low, NetG_highweight` (trainer_objective.py:8, 75-88) keeps working:
same constructors (hyper-parameters read from the global `cfg`, model.py:37-39), forward signatures, return
tuples and state_dict keys (GL 104 tensors, GH 121 tensors - tests/golden/ckpt_manifest.json).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Variable      # re-exported: the reference's model.py does `from util import *` (model.py:3,
                                         # util.py:1-11) and its callers import `Variable, torch, cfg` FROM `model`
                                         # (trainer_objective.py:8)

from . import custom_ops as C
from . import ops
from .miscc.config import cfg
from .util import (CA_NET, CNN_ENCODER, D_GET_LOGITS, GET_IMAGE_G, GET_IMAGE_G_noAct, GLU, INIT_STAGE_GImgup, NEXT_STAGE_G,
                   RNN_ENCODER, ResBlock, Block3x3_leakRelu, Block3x3_relu, _ConvBnGlu, _ResidualNoSum, conv1x1, conv3x3,
                   conv5x5, downBlock, encode_image_by_16times, upBlock)

__all__ = ["G_SR_NET_low", "G_SR_NET_low_stage1", "NetG_highweight", "RNN_ENCODER", "CNN_ENCODER", "CA_NET",
           "INIT_STAGE_GImgup", "NEXT_STAGE_G", "GET_IMAGE_G", "GET_IMAGE_G_noAct", "ResBlock", "GLU", "upBlock",
           "Block3x3_relu", "conv1x1", "conv3x3", "conv5x5", "Variable", "cfg", "torch", "nn", "F",
           "D_NET64", "D_NET128", "D_NET256", "D_GET_LOGITS", "downBlock", "Block3x3_leakRelu"]


class G_SR_NET_low_stage1(nn.Module):
    """model.py:81-130.  Importable because trainer_objective.py:8 imports the name; never constructed on the shipped
    path (`stage1 = False` is hard-coded, trainer_objective.py:56) and not built here: constructing it raises."""

    def __init__(self, *args, **kwargs):
        super(G_SR_NET_low_stage1, self).__init__()
        raise NotImplementedError("G_SR_NET_low_stage1 is dead code in the reference (stage1=False, "
                                  "trainer_objective.py:56); tgsr_amd builds G_SR_NET_low")


class G_SR_NET_low(nn.Module):
    """model.py:34-78: low-frequency SR generator, three x2 stages with word attention + three image heads."""

    def __init__(self):
        super(G_SR_NET_low, self).__init__()
        ngf = cfg.GAN.GF_DIM
        nef = cfg.TEXT.EMBEDDING_DIM
        ncf = cfg.GAN.CONDITION_DIM
        self.ca_net = CA_NET()
        self.h_net1 = INIT_STAGE_GImgup(ngf, ncf, nef)
        self.h_net2 = NEXT_STAGE_G(ngf, nef, ncf)
        self.h_net3 = NEXT_STAGE_G(ngf, nef, ncf)
        self.img_net1 = GET_IMAGE_G_noAct(ngf)
        self.img_net2 = GET_IMAGE_G_noAct(ngf)
        self.img_net3 = GET_IMAGE_G_noAct(ngf)

    def attention_modules(self):
        """The distinct GlobalAttentionGeneral modules in stage order: `proj` of forward() holds their word projections."""
        return [self.h_net1.att, self.h_net2.att, self.h_net3.att]

    def forward(self, LR, sent_emb, word_embs, mask, outmiddle=False, ca=None, proj=None):
        """ca: optional precomputed `self.ca_net(sent_emb)` (nothing downstream reads c_code, model.py:51-52, only mu /
        logvar are returned); proj: optional precomputed conv_context projections of `attention_modules()` (SRPipeline
        computes both, and the mask, in one launch: ops.text_tail)."""
        fake_imgs, att_maps = [], []
        c_code, mu, logvar = self.ca_net(sent_emb) if ca is None else ca   # c_code unused downstream (model.py:51-52)
        srcs = [None, None, None]
        if proj is not None:
            srcs = proj
        elif not self.training:
            # the three stages attend to the same words: their conv_context projections go out as one launch
            srcs = C.word_project(word_embs, [a.conv_context.weight.detach() for a in self.attention_modules()])
        h_code1, att0 = self.h_net1(None, LR, word_embs, mask, wide_out=True, src=srcs[0])
        fake_imgs.append(self.img_net1(h_code1))
        att_maps.append(att0)
        h_code2, att1 = self.h_net2(h_code1, None, word_embs, mask, wide_out=True, src=srcs[1])
        fake_imgs.append(self.img_net2(h_code2))
        att_maps.append(att1)
        h_code3, att2 = self.h_net3(h_code2, None, word_embs, mask, src=srcs[2])
        fake_imgs.append(self.img_net3(h_code3))
        att_maps.append(att2)
        if outmiddle:
            return fake_imgs, att_maps, mu, logvar, [h_code1, h_code2, h_code3]
        return fake_imgs, att_maps, mu, logvar


class NetG_highweight(nn.Module):
    """model.py:212-298: SRResNet-style high-frequency generator whose three heads are `one_k * conv_output(out_k) + a_k * SRb_k`
    (conv_output = conv5x5 [+ Tanh when useAct]).
    weightmap=False: `a` = 0.5 and `one` = 1 are constants, not parameters - in the reference `nn.Parameter(...).cuda()` leaves a
    plain tensor that is neither trained nor saved (model.py:246-248; netGH_epoch_7.pth has no key `a`).
    weightmap=True (model.py:235-245): `a1`, `a2`, `a3` are trainable [64,64] / [128,128] / [256,256] maps (initial value 1;
    here `nn.Parameter(tensor.cuda())` IS a registered parameter: state_dict keys `a1..a3`), broadcast over batch and channels:
    tgsr::axpy_map and its backward; forward returns (ims, a3, one3) like the reference (:293-295)."""

    MAP_SIZES = (64, 128, 256)

    def __init__(self, weightmap=False, low='lr-lrblur', useAct=True):
        super(NetG_highweight, self).__init__()
        ngf = cfg.GAN.GF_DIM
        self.low = low
        self.useAct = bool(useAct)
        self.residual = nn.Sequential(*[ResBlock(channel_num=32) for _ in range(6)])   # model.py:258-262
        self.upscale4x = upBlock(ngf, ngf)
        self.upscale2x = upBlock(ngf, ngf)
        self.upscale8x = upBlock(ngf, ngf)
        self.conv_output = nn.Sequential(conv5x5(ngf, 3), nn.Tanh()) if useAct else nn.Sequential(conv5x5(ngf, 3))
        self.convin = _ConvBnGlu(3, ngf)
        self.residual24 = _ResidualNoSum(ngf)
        self.residual48 = _ResidualNoSum(ngf)
        self.weightmap = bool(weightmap)
        if self.weightmap:
            for k, n in enumerate(self.MAP_SIZES):
                setattr(self, "a%d" % (k + 1), nn.Parameter(torch.ones([n, n], dtype=torch.float32)))
        self._a = 0.5
        self._consts = {}

    def _const(self, ref):
        """(a, one) on ref's device, created once per device (no per-step H2D copy)."""
        c = self._consts.get(ref.device)
        if c is None:
            c = self._consts[ref.device] = (ref.new_tensor([self._a]), ref.new_ones(1))
        return c

    def maps(self):
        return [getattr(self, "a%d" % (k + 1)) for k in range(len(self.MAP_SIZES))] if self.weightmap else None

    def _head(self, out, SRb, k=0):
        w = self.conv_output[0].weight
        if self.weightmap:                                    # one_k * conv_output(out) + a_k * SRb  (model.py:277, 286, 294)
            amap = self.maps()[k]
            if tuple(amap.shape) != tuple(out.shape[2:]):
                raise ValueError("NetG_highweight(weightmap=True): a%d is %s but scale %d of this input is %s (the maps are sized "
                                 "for 32 x 32 inputs, model.py:236-239)" % (k + 1, tuple(amap.shape), k, tuple(out.shape[2:])))
            return C.axpy_map(C.conv_to3(out, w, self.useAct, None, 0.0), SRb, amap)
        if self.useAct:
            return C.conv_to3(out, w, True, SRb, self._a)     # torch.ops.tgsr.conv_to3 (+ autograd): tanh(conv) + a * SRb, one launch
        from .autograd import AxpyImage
        return AxpyImage.apply(C.conv_to3(out, w, False, None, 0.0), SRb, self._a)      # conv + a * SRb (useAct=False, model.py:226)

    def trunk(self, LR, LRb):
        """Everything of forward() that does not need the low-frequency images: convin -> 6 ResBlocks -> the three
        up-scales with residual24/48 between them.  Returns the three feature maps the heads read.  (In the
        reference the heads' outputs never feed back into this chain, model.py:264-298, so the whole trunk is
        independent of G_SR_NET_low and can run concurrently with it.)"""
        if self.low == 'lrblur':
            x = LRb
        elif self.low == 'lr-lrblur':
            x = LR - LRb
        else:
            x = LR
        out2 = self.upscale2x(self.residual(self.convin(x)))
        out4 = self.upscale4x(self.residual24(out2))
        out8 = self.upscale8x(self.residual48(out4))
        return out2, out4, out8

    def heads(self, feats, SRb):
        """ims_k = one * conv_output(out_k) + a * SRb_k   (model.py:280, 288, 297)."""
        return [self._head(f, sr, k) for k, (f, sr) in enumerate(zip(feats, SRb))]

    def tanh_heads(self, feats):
        """The part of heads() that needs no low-frequency image: conv_output(out_k).  SRPipeline runs it on the
        high-frequency branch's stream, beside G_SR_NET_low; `finish_heads` adds a * SRb_k (one launch for all scales) -
        the same fma the fused epilogue evaluates, bit-identical images (tests/test_hip_parity.py)."""
        w = self.conv_output[0].weight.detach()
        return [C.conv_to3(f, w, self.useAct, None, self._a) for f in feats]

    def finish_heads(self, ts, SRb):
        if self.weightmap:
            return [C.axpy_map(t, s.contiguous(), a.detach()) for t, s, a in zip(ts, SRb, self.maps())]
        return list(C.axpy_images(list(ts), [s.contiguous() for s in SRb[:len(ts)]], self._a))

    def forward(self, LR, SRb, LRb):
        ims = self.heads(self.trunk(LR, LRb), SRb[:3])
        a, one = self._const(LR)
        if self.weightmap:
            return ims, self.a3, one
        return ims, a, one


# ----------------------------------------------------------------------------------------------- discriminators
class _D_NET(nn.Module):
    """The reference CALLS discriminators (`netD(img)`, `netD.COND_DNET(features, sent_emb)`, `netD.UNCOND_DNET(features)`,
    miscc/losses.py:290-316, 351-366) but ships no class for them (`grep D_NET` only hits losses.py).  These are the
    build's declaration of that interface in the AttnGAN topology TGSR was forked from (README.md:4): an image encoder
    down to [B, 8 ndf, 4, 4] built from downBlock (util.py:92-98), plus the two logit heads.  The ARCHITECTURE is
    therefore parity-unpinned; its kernels are pinned against a torch restatement in the oracle."""

    def __init__(self, extra_down, b_jcu=True):
        super().__init__()
        ndf, nef = cfg.GAN.DF_DIM, cfg.TEXT.EMBEDDING_DIM
        self.img_code_s16 = encode_image_by_16times(ndf)
        ch = ndf * 8
        self.extra = nn.ModuleList()
        for _ in range(extra_down):                                # one more downBlock per extra factor of 2 ...
            self.extra.append(downBlock(ch, ch * 2))
            ch *= 2
        self.reduce = nn.ModuleList()
        while ch > ndf * 8:                                        # ... then conv3x3 blocks back down to 8 ndf channels
            self.reduce.append(Block3x3_leakRelu(ch, ch // 2))
            ch //= 2
        self.UNCOND_DNET = D_GET_LOGITS(ndf, nef, bcondition=False) if b_jcu else None
        self.COND_DNET = D_GET_LOGITS(ndf, nef, bcondition=True)

    supports_groups = True      # forward(x, groups=...): several passes as one batch with per-slice BatchNorm statistics

    def forward(self, x_var, groups=None):
        x = self.img_code_s16(x_var, groups)
        for m in self.extra:
            x = m(x, groups)
        for m in self.reduce:
            x = m(x, groups)
        return x                                                   # [B, 8 ndf, 4, 4]


class D_NET64(_D_NET):
    """Discriminator for 64 x 64 images (the first output scale of the x8 generators)."""

    def __init__(self, b_jcu=True):
        super().__init__(0, b_jcu)


class D_NET128(_D_NET):
    def __init__(self, b_jcu=True):
        super().__init__(1, b_jcu)


class D_NET256(_D_NET):
    def __init__(self, b_jcu=True):
        super().__init__(2, b_jcu)
