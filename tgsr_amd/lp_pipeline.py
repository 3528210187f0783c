"""Reduced-precision executor of the x8 inference path (BASELINE.json configs[4]: bf16 storage, MFMA bf16 attention +
fused conv, hipGraph-capturable): G_SR_NET_low.forward (model.py:48-78) and NetG_highweight.forward (model.py:264-298)
of an `SRPipeline`'s modules, run on lp images (tgsr_amd.lp) instead of fp32 NCHW tensors.

Same parameters (read from the fp32 drop-in modules, packed once per weight version), same call order and argument
wiring, same return values (fp32 images and attention maps) - only the storage / MFMA operand type of the activations
between the first and the last convolution differs.  Rounding points: oracle/tgsr_oracle_lp.py (the CPU model the GPU
tests compare against).  Every activation buffer is allocated (zeroed) once per (batch, LR size): the step touches no
allocator and captures into a hipGraph as is.
"""
import os

import torch

from . import lp, ops
from .util import _ver

# upBlocks by sub-pixel decomposition (tgsr_lp_upconv_glu_fwd: 2.25x fewer MFMAs); TGSR_LP_SUBPIXEL=0 keeps the direct
# 9-tap form on the up-sampled grid (tgsr_lp_conv3x3_fwd(upsample=1)) for A/B runs
SUBPIXEL = os.environ.get("TGSR_LP_SUBPIXEL", "1") != "0"


class _Conv:
    """Packed weight + folded BatchNorm affine of one conv(+bn) pair of the fp32 modules."""

    def __init__(self, conv, bn, dtype):
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.wpack = lp.pack_conv3x3_weight(conv.weight, dtype)
        self.scale, self.shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def __call__(self, x, glu=False, upsample=False, residual=None, out=None, out_coff=0):
        return lp.conv3x3(x, self.wpack, self.cin, self.cout, self.scale, self.shift, glu=glu, upsample=upsample,
                          residual=residual, out=out, out_coff=out_coff)


class _UpConv:
    """upBlock: sub-pixel pack where the kernel takes the shape, else the direct form."""

    def __init__(self, conv, bn, dtype):
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.sub = SUBPIXEL and self.cout == 64 and self.cin in (32, 64)
        self.wsub = lp.pack_upconv_weight(conv.weight, dtype) if self.sub else None
        self.wpack = lp.pack_conv3x3_weight(conv.weight, dtype)
        self.scale, self.shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def __call__(self, x, out):
        if self.sub and lp.upconv_supported(self.cin, self.cout, x.shape[1] - 2, x.shape[2] - 2):
            return lp.upconv_glu(x, self.wsub, self.cin, self.cout, self.scale, self.shift, out=out)
        return lp.conv3x3(x, self.wpack, self.cin, self.cout, self.scale, self.shift, glu=True, upsample=True, out=out)


class _Stem:
    def __init__(self, seq):                      # _ConvBnGlu: [conv3x3(3, 2C), BatchNorm2d, GLU]
        conv, bn = seq[0], seq[1]
        self.w = conv.weight.detach().contiguous()
        self.scale, self.shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def __call__(self, x, out, out_coff=0):
        return lp.stem(x, self.w, self.scale, self.shift, out=out, out_coff=out_coff)


class LpExecutor:
    def __init__(self, netGL, netGH, dtype):
        self.netGL, self.netGH = netGL, netGH
        self.dtype = lp.torch_dtype(dtype)
        self.key = None
        self.bufs = {}
        self.force_bufs = None       # set by a hipGraph capture: the buffer set the captured step is bound to

    # ------------------------------------------------------------------ weights
    def _params_key(self):
        return _ver(*self.netGL.parameters(), *self.netGL.buffers(), *self.netGH.parameters(), *self.netGH.buffers())

    def refresh(self, force=False):
        """(Re)build the packed weights when any parameter / running statistic of the two generators changed."""
        key = self._params_key()
        if key == self.key and not force:
            return
        dt, GL, GH = self.dtype, self.netGL, self.netGH
        if GL.training or GH.training:
            raise RuntimeError("the reduced-precision path is inference only: call .eval() on the generators")

        def res(rb):
            return (_Conv(rb.block[0], rb.block[1], dt), _Conv(rb.block[3], rb.block[4], dt))

        self.gl_stem = _Stem(GL.h_net1.im2f)
        self.gl_stage = []
        for st, img in ((GL.h_net1, GL.img_net1), (GL.h_net2, GL.img_net2), (GL.h_net3, GL.img_net3)):
            self.gl_stage.append({"res": [res(rb) for rb in st.residual], "up": _UpConv(st.upsample[1], st.upsample[2], dt),
                                  "head": lp.pack_to3_weight(img.img[0].weight, dt), "att": st.att})
        self.gh_stem = _Stem(GH.convin)
        self.gh_res = [res(rb) for rb in GH.residual]
        self.gh_up = [_UpConv(u[1], u[2], dt) for u in (GH.upscale2x, GH.upscale4x, GH.upscale8x)]
        self.gh_mid = [(_Conv(m[0], m[1], dt), _Conv(m[3], m[4], dt)) for m in (GH.residual24, GH.residual48)]
        self.gh_head = lp.pack_to3_weight(GH.conv_output[0].weight, dt)
        self.key = key

    # ------------------------------------------------------------------ activation buffers
    def alloc(self, B, H, W, dev):
        """One set of zero-bordered activation images for a batch of B LR images of H x W."""
        def im(s, c):
            return lp.new_image(B, H * s, W * s, c, self.dtype, dev)
        return {"gl": [{"wide": im(s, 64), "tmp": im(s, 64), "a": im(s, 64), "b": im(s, 64)} for s in (1, 2, 4)],
                "h3": im(8, 32),
                "gh": {"x": im(1, 32), "y": im(1, 32), "t": im(1, 32)},
                "u": [im(2, 32), im(4, 32), im(8, 32)],
                "m": [{"t": im(2, 32), "v": im(2, 32)}, {"t": im(4, 32), "v": im(4, 32)}]}

    def _buffers(self, B, H, W, dev):
        """The buffer set of the calling stream (concurrent lanes must not share activations); allocated on first use."""
        if self.force_bufs is not None:
            return self.force_bufs
        k = (B, H, W, str(dev), torch.cuda.current_stream(dev).cuda_stream)
        b = self.bufs.get(k)
        if b is None:
            b = self.bufs[k] = self.alloc(B, H, W, dev)
        return b

    # ------------------------------------------------------------------ the two generators
    def low(self, bufs, LR, sent_emb, word_embs, mask, ca=None):
        """G_SR_NET_low.forward (model.py:48-78) -> (fake_imgs, att_maps, mu, logvar, [h image of each stage])."""
        GL = self.netGL
        c_code, mu, logvar = GL.ca_net(sent_emb) if ca is None else ca
        T = word_embs.size(2)
        srcs = ops.word_project(word_embs, [st["att"].conv_context.weight for st in self.gl_stage])
        fake, atts = [], []
        wide = bufs["gl"][0]["wide"]
        self.gl_stem(LR, out=wide)                                             # im2f -> channels [0, 32)
        for k, st in enumerate(self.gl_stage):
            bb = bufs["gl"][k]
            atts.append(lp.word_attention(bb["wide"], srcs[k], mask, T, correct_mask=st["att"].correct_mask))
            x = bb["wide"]
            for (c0, c1), o in zip(st["res"], (bb["a"], bb["b"])):             # R_NUM = 2 ResBlocks (util.py:110-130)
                c0(x, glu=True, out=bb["tmp"])
                c1(bb["tmp"], residual=x, out=o)
                x = o
            nxt = bufs["gl"][k + 1]["wide"] if k < 2 else bufs["h3"]
            st["up"](x, out=nxt)                                               # upBlock -> channels [0, 32) of the next stage
            fake.append(lp.conv_to3(nxt, st["head"], 3))
        return fake, atts, mu, logvar

    def high_trunk(self, bufs, LR, LRb):
        """NetG_highweight.trunk: the three feature images the heads read."""
        GH = self.netGH
        x = LRb if GH.low == 'lrblur' else (LR - LRb if GH.low == 'lr-lrblur' else LR)
        g = bufs["gh"]
        self.gh_stem(x, out=g["x"])
        cur, other = g["x"], g["y"]
        for c0, c1 in self.gh_res:
            c0(cur, glu=True, out=g["t"])
            c1(g["t"], residual=cur, out=other)
            cur, other = other, cur
        feats = []
        for k in range(3):
            if k > 0:
                m, (c0, c1) = bufs["m"][k - 1], self.gh_mid[k - 1]
                c0(cur, glu=True, out=m["t"])
                c1(m["t"], out=m["v"])
                cur = m["v"]
            self.gh_up[k](cur, out=bufs["u"][k])
            cur = bufs["u"][k]
            feats.append(cur)
        return feats

    def high_heads(self, feats, SRb):
        return [lp.conv_to3(f, self.gh_head, 5, tanh_axpy=True, addend=sr, alpha=self.netGH._a)
                for f, sr in zip(feats, SRb)]
