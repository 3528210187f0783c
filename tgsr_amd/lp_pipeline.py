"""Reduced-precision executor of the inference path (BASELINE.json configs[4]: bf16 storage, MFMA bf16 attention +
fused conv, hipGraph-capturable): G_SR_NET_low.forward (model.py:48-78) and NetG_highweight.forward (model.py:264-298)
of an `SRPipeline`'s modules - or their x16 counterparts (models16.py:5-39, 97-179: weight-tied stages 2-4, tanh heads,
a fourth stage through `residual48` / `upscale8x` again) - run on lp images (tgsr_amd.lp) instead of fp32 NCHW tensors.

Same parameters (read from the fp32 drop-in modules, packed once per weight version), same call order and argument
wiring, same return values (fp32 images and attention maps) - only the storage / MFMA operand type of the activations
between the first and the last convolution differs.  Rounding points: oracle/tgsr_oracle_lp.py (the CPU model the GPU
tests compare against).  Every activation buffer is allocated (zeroed) once per (batch, LR size): the step touches no
allocator and captures into a hipGraph as is.
"""
import os

import torch

from . import custom_ops as C
from . import lp, ops
from .util import _ver

# upBlocks by sub-pixel decomposition (tgsr_lp_upconv_glu_fwd: 2.25x fewer MFMAs); TGSR_LP_SUBPIXEL=0 keeps the direct
# 9-tap form on the up-sampled grid (tgsr_lp_conv3x3_fwd(upsample=1)) for A/B runs
SUBPIXEL = os.environ.get("TGSR_LP_SUBPIXEL", "1") != "0"
# image heads computed inside the upBlock that produces their input (tgsr_lp_upconv_glu_head_fwd: per-tile partial sums)
# and finished for all scales of both generators by ONE tgsr_lp_head_combine launch; the last stage's 256^2 feature
# images then never go to HBM.  TGSR_LP_FUSE_HEADS=0 keeps the six stand-alone head launches (tgsr_lp_conv_to3_fwd).
FUSE_HEADS = os.environ.get("TGSR_LP_FUSE_HEADS", "1") != "0"
# the bf16 configuration runs NetG_highweight's 32^2 trunk (convin + the six ResBlocks of model.py:258-262, 1.2 % of the
# step's MACs) with f16 operands and storage: six chained residual additions rounded to 8-bit mantissas are where bf16 loses
# the finest image 7 dB (46.9 -> 53.7 dB against fp32 on the CPU model of the shipped checkpoint with ONLY this section in
# f16, oracle/tgsr_oracle_lp.py; everything else - all of G_SR_NET_low, every 64^2..256^2 layer, the heads - stays bf16).
# One tgsr_lp_convert launch (a 1.2 MB image at batch 16) hands the trunk's output to the bf16 up-scales.
# TGSR_LP_BF16_TRUNK=bf16 keeps the trunk in bf16 (A/B).
F16_TRUNK = os.environ.get("TGSR_LP_BF16_TRUNK", "f16") != "bf16"
# the word attention of every stage computed in the epilogue of the kernel that produces its h (lp_stem_kernel for stage 1,
# the previous stage's lp_upconv_glu_kernel for the others; same device function as the stand-alone kernel: same bits):
# three launches - and the re-read of h - leave G_SR_NET_low's dependent chain.  TGSR_LP_FUSE_ATT=0: stand-alone launches.
FUSE_ATT = os.environ.get("TGSR_LP_FUSE_ATT", "1") != "0"
# the two ResBlocks of a G_SR_NET_low stage (four dependent convolutions) as ONE launch, tiles synchronised through device flags
# (tgsr_lp_resblocks_fwd; same per-tile code, bit-identical).  Built in round 5 for the 32^2 / 64^2 stages, whose layers are shorter
# than the ~7 us a dependent kernel costs in a replayed graph - and measured SLOWER: on MI355X an in-kernel hand-off between
# workgroups needs an agent-scope release (write-back of the XCD's L2: ~6.5 us with a freshly written 16 KB tile) and an acquire
# (L1 invalidate, ~1.7 us) per layer, more than the kernel boundary it replaces (bf16 batch 16 one lane: 28.1 k images/s with it,
# 32.0 k without; profiles/HISTORY.md 3.17).  OFF by default; TGSR_LP_CHAIN=1 switches it on (the tests do).
CHAIN = os.environ.get("TGSR_LP_CHAIN", "0") == "1"
CHAIN_MAX_PIXELS = 64 * 64


def trunk_dtype_of(dtype):
    """Storage / operand type of NetG_highweight's 32^2 trunk for a pipeline of `dtype`."""
    return torch.float16 if (dtype == torch.bfloat16 and F16_TRUNK) else dtype


class _Conv:
    """Packed weight + folded BatchNorm affine of one conv(+bn) pair of the fp32 modules."""

    def __init__(self, conv, bn, dtype):
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.wpack = lp.pack_conv3x3_weight(conv.weight, dtype)
        self.scale, self.shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def __call__(self, x, glu=False, upsample=False, residual=None, out=None, out_coff=0):
        C.lp_conv3x3(x, self.wpack, self.cin, self.cout, self.scale, self.shift, glu, upsample, residual, 0, out, out_coff)
        return out


class _UpConv:
    """upBlock: sub-pixel pack where the kernel takes the shape, else the direct form."""

    def __init__(self, conv, bn, dtype):
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.sub = SUBPIXEL and self.cout == 64 and self.cin in (32, 64)
        self.wsub = lp.pack_upconv_weight(conv.weight, dtype) if self.sub else None
        self.wpack = lp.pack_conv3x3_weight(conv.weight, dtype)
        self.scale, self.shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def __call__(self, x, out, att=None):
        if att is not None:
            C.lp_upconv_glu_att(x, self.wsub, self.cin, self.cout, self.scale, self.shift, out, 0, *att)
        elif self.sub and lp.upconv_supported(self.cin, self.cout, x.shape[1] - 2, x.shape[2] - 2):
            C.lp_upconv_glu(x, self.wsub, self.cin, self.cout, self.scale, self.shift, out, 0)
        else:
            C.lp_conv3x3(x, self.wpack, self.cin, self.cout, self.scale, self.shift, True, True, None, 0, out, 0)
        return out

    def att_fusable(self, x):
        """The next stage's attention can ride this upBlock's epilogue: the sub-pixel kernel at 64 input channels."""
        return self.sub and self.cin == 64 and lp.upconv_supported(self.cin, self.cout, x.shape[1] - 2, x.shape[2] - 2)

    def fusable(self, x):
        return FUSE_HEADS and self.sub and lp.head_fusable(self.cin, self.cout, x.shape[1] - 2, x.shape[2] - 2)

    def with_head(self, x, head_wpack, K, partial, out, att=None):
        """upBlock + its image head's partial sums in one launch; out None: the feature image is not written."""
        if att is not None:
            C.lp_upconv_glu_head_att(x, self.wsub, self.cin, self.cout, self.scale, self.shift, head_wpack, K, partial, out, 0,
                                     *att)
        else:
            C.lp_upconv_glu_head(x, self.wsub, self.cin, self.cout, self.scale, self.shift, head_wpack, K, partial, out, 0)
        return out, partial


class _Stem:
    def __init__(self, seq):                      # _ConvBnGlu: [conv3x3(3, 2C), BatchNorm2d, GLU]
        conv, bn = seq[0], seq[1]
        self.w = conv.weight.detach().contiguous()
        self.scale, self.shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def __call__(self, x, out, out_coff=0, att=None):
        """att = (att_pack, nsets, index, T, use_mask, correct_mask, c_coff, attn): + the first stage's word attention."""
        if att is not None:
            C.lp_stem_att(x, self.w, self.scale, self.shift, out, out_coff, *att)
        else:
            C.lp_stem(x, self.w, self.scale, self.shift, out, out_coff)
        return out


class LpExecutor:
    def __init__(self, netGL, netGH, dtype):
        self.netGL, self.netGH = netGL, netGH
        self.dtype = lp.torch_dtype(dtype)
        self.key = None
        self.bufs = {}
        self.force_bufs = None       # set by a hipGraph capture: the buffer set the captured step is bound to
        self.fuse_attention = FUSE_ATT

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
        if getattr(GH, "weightmap", False) or not getattr(GH, "useAct", True):
            raise NotImplementedError("the reduced-precision path builds NetG_highweight's shipped heads (weightmap=False, "
                                      "useAct=True: trainer_objective.py:58, 88); the fp32 path runs the other two forms")

        def res(rb, dt=dt):
            return (_Conv(rb.block[0], rb.block[1], dt), _Conv(rb.block[3], rb.block[4], dt))
        self.trunk_dtype = trunk_dtype_of(dt)

        # x16 (models16): ONE NEXT_STAGE_G object serves stages 2-4 and ONE image head all four (models16.py:13-14); the
        # 16x stage of NetG_highweight re-uses residual48 / upscale8x (:172-173).  A module is packed once however many
        # stages use it.
        self.x16 = hasattr(GL, "h_net4")
        nst = 4 if self.x16 else 3
        memo = {}

        def once(mod, build):
            if id(mod) not in memo:
                memo[id(mod)] = build(mod)
            return memo[id(mod)]

        self.gl_stem = _Stem(GL.h_net1.im2f)
        self.gl_stage = []
        for k in range(1, nst + 1):
            st, img = getattr(GL, "h_net%d" % k), getattr(GL, "img_net%d" % k)
            self.gl_stage.append({"res": once(st.residual, lambda m: [res(rb) for rb in m]),
                                  "up": once(st.upsample, lambda m: _UpConv(m[1], m[2], dt)),
                                  "head": once(img, lambda m: lp.pack_to3_weight(m.img[0].weight, dt)), "att": st.att})
        self.gl_head_tanh = self.x16                       # GET_IMAGE_G (util.py:894-905) vs GET_IMAGE_G_noAct (:909-919)
        self.gh_stem = _Stem(GH.convin)
        self.gh_res = [res(rb, self.trunk_dtype) for rb in GH.residual]
        ups = [GH.upscale2x, GH.upscale4x, GH.upscale8x] + ([GH.upscale8x] if self.x16 else [])
        mids = [GH.residual24, GH.residual48] + ([GH.residual48] if self.x16 else [])
        self.gh_up = [once(u, lambda m: _UpConv(m[1], m[2], dt)) for u in ups]
        self.gh_mid = [once(m_, lambda m: (_Conv(m[0], m[1], dt), _Conv(m[3], m[4], dt))) for m_ in mids]
        self.gh_head = lp.pack_to3_weight(GH.conv_output[0].weight, dt)
        self.key = key

    # ------------------------------------------------------------------ activation buffers
    def alloc(self, B, H, W, dev):
        """One set of zero-bordered activation images for a batch of B LR images of H x W."""
        def im(s, c, dt=None):
            return lp.new_image(B, H * s, W * s, c, dt or self.dtype, dev)
        n = len(self.gl_stage)
        td = self.trunk_dtype
        return {"gl": [{"wide": im(1 << k, 64), "tmp": im(1 << k, 64), "a": im(1 << k, 64), "b": im(1 << k, 64)}
                       for k in range(n)],
                "h3": im(1 << n, 32),                                   # the last stage's 32-channel output
                "gh": {"x": im(1, 32, td), "y": im(1, 32, td), "t": im(1, 32, td),
                       "c": im(1, 32) if td != self.dtype else None},      # the trunk's output in the pipeline's type
                "u": [im(2 << k, 32) for k in range(n)],
                "m": [{"t": im(2 << k, 32), "v": im(2 << k, 32)} for k in range(n - 1)],
                # per-tile partial sums of the fused image heads (fp32; allocated on first use)
                "pl": [None] * n, "ph": [None] * n}

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
    def low(self, bufs, LR, sent_emb, word_embs, mask, ca=None, proj=None, defer_heads=False):
        """G_SR_NET_low.forward (model.py:48-78) -> (fake_imgs, att_maps, mu, logvar).

        Image heads computed inside their upBlocks leave per-tile partial sums.  By default they are combined before this
        returns: the images are finished.  `defer_heads=True` (SRPipeline: NetG_highweight's heads follow and one combine
        launch then finishes both generators' images) returns a fifth value, the list of pending partial-sum tensors (None
        where an image is already complete), and the corresponding `fake_imgs` entries are NOT WRITTEN YET: pass that list
        to `high_heads(feats, fake_imgs, pend)`, which fills them."""
        GL = self.netGL
        c_code, mu, logvar = GL.ca_net(sent_emb) if ca is None else ca
        T = word_embs.size(2)
        atts_m = GL.attention_modules()                                        # distinct attention modules (x16: two)
        if proj is None:
            proj = C.word_project(word_embs, [m.conv_context.weight.detach() for m in atts_m])
        set_of = [[i for i, m in enumerate(atts_m) if m is st["att"]][0] for st in self.gl_stage]
        srcs = [proj[i] for i in set_of]
        last = len(self.gl_stage) - 1
        fake, atts, pend = [], [], []
        wide = bufs["gl"][0]["wide"]
        B, H0, W0 = LR.shape[0], LR.shape[2], LR.shape[3]
        # The attention of stage k rides the kernel that produces its h when the step brought the attention pack
        # (SRPipeline._text_tail -> proj.att_pack) and the shapes allow: att_args(k) = the producer's extra arguments
        pack = getattr(proj, "att_pack", None) if self.fuse_attention else None

        def att_args(k):
            H, W = H0 << k, W0 << k
            atts.append(torch.empty(B, T, H, W, dtype=torch.float32, device=LR.device))
            return (pack, len(atts_m), set_of[k], T, mask is not None, bool(self.gl_stage[k]["att"].correct_mask), 32, atts[-1])

        fused_next = pack is not None and W0 % 32 == 0                         # stage 0: inside the stem
        self.gl_stem(LR, out=wide, att=att_args(0) if fused_next else None)    # im2f -> channels [0, 32) (+ c_code -> [32, 64))
        for k, st in enumerate(self.gl_stage):
            bb = bufs["gl"][k]
            if not fused_next:
                atts.append(C.lp_word_attention(bb["wide"], srcs[k], mask, T, st["att"].correct_mask, 32))
            x = bb["wide"]
            Hk, Wk = x.shape[1] - 2, x.shape[2] - 2
            if (CHAIN and len(st["res"]) == 2 and Hk * Wk <= CHAIN_MAX_PIXELS and lp.resblocks_supported(st["res"][0][0].cin, Hk, Wk)
                    and all(c.cin == 64 for pair in st["res"] for c in pair)):
                if bb.get("flags") is None:
                    bb["flags"] = lp.resblocks_flags(B, Hk, Wk, x.device)
                cs = [c for pair in st["res"] for c in pair]
                C.lp_resblocks(x, [c.wpack for c in cs], [c.scale for c in cs], [c.shift for c in cs], bb["tmp"], bb["a"], bb["b"],
                               bb["flags"])
                x = bb["b"]
            else:
                for (c0, c1), o in zip(st["res"], (bb["a"], bb["b"])):         # R_NUM = 2 ResBlocks (util.py:110-130)
                    c0(x, glu=True, out=bb["tmp"])
                    c1(bb["tmp"], residual=x, out=o)
                    x = o
            nxt = bufs["gl"][k + 1]["wide"] if k < last else bufs["h3"]
            # the NEXT stage's attention inside this upBlock (its output is that stage's h)
            fused_next = pack is not None and k < last and st["up"].att_fusable(x)
            if st["up"].fusable(x):
                # upBlock + the partial sums of its 3x3 head; the last stage's feature image is read by nothing else
                Ho, Wo = 2 * (x.shape[1] - 2), 2 * (x.shape[2] - 2)
                if bufs["pl"][k] is None:
                    bufs["pl"][k] = torch.empty(lp.head_partial_elems(B, Ho, Wo, 3), dtype=torch.float32, device=x.device)
                st["up"].with_head(x, st["head"], 3, bufs["pl"][k], nxt if k < last else None,
                                   att=att_args(k + 1) if fused_next else None)
                fake.append(torch.empty(B, 3, Ho, Wo, dtype=torch.float32, device=x.device))
                pend.append(bufs["pl"][k])
            else:
                st["up"](x, out=nxt, att=att_args(k + 1) if fused_next else None)   # upBlock -> channels [0, 32) of the next stage
                fake.append(C.lp_conv_to3(nxt, st["head"], 3, self.gl_head_tanh, None, 0.0))
                pend.append(None)
        if defer_heads:
            return fake, atts, mu, logvar, pend
        todo = [k for k, p in enumerate(pend) if p is not None]
        if todo:
            _combine([tuple(fake[k].shape[2:]) for k in todo], [pend[k] for k in todo], [None] * len(todo),
                     [fake[k] for k in todo], [None] * len(todo), self.gl_head_tanh, 0.0)
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
        if g["c"] is not None:
            C.lp_convert(cur, g["c"])              # f16 trunk -> the bf16 up-scales (one rounding per element)
            cur = g["c"]
        feats = []
        for k in range(len(self.gh_up)):
            if k > 0:
                m, (c0, c1) = bufs["m"][k - 1], self.gh_mid[k - 1]
                c0(cur, glu=True, out=m["t"])
                c1(m["t"], out=m["v"])
                cur = m["v"]
            nlast = len(self.gh_up) - 1
            if self.gh_up[k].fusable(cur):
                B, Ho, Wo = cur.shape[0], 2 * (cur.shape[1] - 2), 2 * (cur.shape[2] - 2)
                if bufs["ph"][k] is None:
                    bufs["ph"][k] = torch.empty(lp.head_partial_elems(B, Ho, Wo, 5), dtype=torch.float32, device=cur.device)
                self.gh_up[k].with_head(cur, self.gh_head, 5, bufs["ph"][k], bufs["u"][k] if k < nlast else None)
                feats.append(_Partial(bufs["ph"][k], B, Ho, Wo))
            else:
                self.gh_up[k](cur, out=bufs["u"][k])
                feats.append(bufs["u"][k])
            cur = bufs["u"][k]
        return feats

    def high_heads(self, feats, SRb, pend=None):
        """NetG_highweight's heads: `tanh(conv5x5(out_k)) + a * SRb_k`.  Scales whose heads were computed inside their
        upBlocks (partial sums) are finished here by one combine launch - together with the low-frequency images `SRb`
        whose partial sums `low(..., defer_heads=True)` handed back as `pend` (pend[k] is None / pend is None: SRb[k] is a
        finished image)."""
        alpha = self.netGH.alpha() if self.x16 else self.netGH._a      # x16: `a` is a parameter (models16.py:126)
        pend = [None] * len(SRb) if pend is None else list(pend)
        if len(pend) != len(SRb):
            raise ValueError("high_heads: %d pending entries for %d low-frequency images" % (len(pend), len(SRb)))
        fine, sizes, pl, ph, lo, hi = [], [], [], [], [], []
        for k, (f, sr) in enumerate(zip(feats, SRb)):
            fused_h = isinstance(f, _Partial)
            pk = pend[k]
            if not fused_h:
                if pk is not None:          # the low image must exist before an unfused 5x5 head can add it: combine it alone
                    _combine([tuple(sr.shape[2:])], [pk], [None], [sr], [None], self.gl_head_tanh, alpha)
                    pk = None
                fine.append(C.lp_conv_to3(f, self.gh_head, 5, True, sr, float(alpha)))
            else:
                fine.append(torch.empty(f.B, 3, f.H, f.W, dtype=torch.float32, device=sr.device))
            if fused_h or pk is not None:
                sizes.append(tuple(sr.shape[2:]))
                pl.append(pk)
                ph.append(f.t if fused_h else None)
                lo.append(sr)
                hi.append(fine[-1] if fused_h else None)
        if sizes:
            _combine(sizes, pl, ph, lo, hi, self.gl_head_tanh, alpha)
        return fine


def _combine(sizes, pl, ph, lo, hi, low_tanh, alpha):
    """torch.ops.tgsr.lp_head_combine (tensor lists cannot hold None: an empty tensor stands for an absent entry)."""
    e = lo[0].new_empty(0)
    C.lp_head_combine([s[0] for s in sizes], [s[1] for s in sizes], [e if t is None else t for t in pl],
                      [e if t is None else t for t in ph], list(lo), [e if t is None else t for t in hi], bool(low_tanh),
                      float(alpha))


class _Partial:
    """A NetG_highweight head whose per-tile partial sums an upBlock has written (LpExecutor.high_trunk)."""

    def __init__(self, t, B, H, W):
        self.t, self.B, self.H, self.W = t, B, H, W

    def record_stream(self, stream):        # SRPipeline tags the trunk's outputs for the allocator
        self.t.record_stream(stream)
