"""Reduced-precision (bf16 / f16 storage, fp32 accumulate) inference ops over the C ABI's `tgsr_lp_*` entry points.

BASELINE.json configs[4] ("bf16 ... MFMA bf16 attention + fused conv, hipGraph-captured step").  Activations are
"lp images": zero-bordered channels-last tensors [B, H+2, W+2, cpitch] of torch.bfloat16 / torch.float16 (layout and
border rule: include/tgsr_hip.h, tgsr_lp_common.h).  As in `tgsr_amd.ops`, nothing here computes on the CPU or through
eager torch arithmetic: torch allocates (zeroed) buffers and supplies the stream.
"""
import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import TgsrError, check
from .ops import _need_hip, _p, _stream

DT = {torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16, "bf16": _lib.DT_BF16, "f16": _lib.DT_F16}
TORCH_DT = {"bf16": torch.bfloat16, "f16": torch.float16, torch.bfloat16: torch.bfloat16, torch.float16: torch.float16}


def torch_dtype(dtype):
    try:
        return TORCH_DT[dtype]
    except KeyError:
        raise TgsrError("reduced-precision dtype must be 'bf16' or 'f16', got %r" % (dtype,))


def new_image(B: int, H: int, W: int, cpitch: int, dtype, device) -> torch.Tensor:
    """A zeroed lp image [B, H+2, W+2, cpitch]; kernels only ever write its interior, so the border stays zero."""
    return torch.zeros(B, H + 2, W + 2, cpitch, dtype=torch_dtype(dtype), device=device)


def _img(t: torch.Tensor, name: str):
    if t.dim() != 4 or t.dtype not in (torch.bfloat16, torch.float16) or not t.is_contiguous():
        raise TgsrError("%s must be a contiguous lp image [B,H+2,W+2,C] of bf16/f16, got %s %s" %
                        (name, tuple(t.shape), t.dtype))
    return t.shape[0], t.shape[1] - 2, t.shape[2] - 2, t.shape[3]


def from_nchw(x: torch.Tensor, dtype=None, cpitch: Optional[int] = None, coff: int = 0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 NCHW -> channels [coff, coff+C) of an lp image (a new zeroed one unless `out` is given)."""
    _need_hip(x, out)
    x = x.contiguous()
    if x.dtype != torch.float32:
        raise TgsrError("from_nchw: x must be float32")
    B, C, H, W = x.shape
    if out is None:
        out = new_image(B, H, W, cpitch or C, dtype, x.device)
    ob, oh, ow, ocp = _img(out, "out")
    if (ob, oh, ow) != (B, H, W):
        raise TgsrError("from_nchw: out %s does not match x %s" % (tuple(out.shape), tuple(x.shape)))
    check(_lib.lib().tgsr_lp_from_nchw(DT[out.dtype], _p(x), _p(out), B, C, H, W, ocp, coff, _stream()),
          "tgsr_lp_from_nchw")
    return out


def to_nchw(img: torch.Tensor, C: Optional[int] = None, coff: int = 0) -> torch.Tensor:
    """Channels [coff, coff+C) of an lp image -> fp32 NCHW."""
    _need_hip(img)
    B, H, W, cp = _img(img, "img")
    C = cp - coff if C is None else C
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=img.device)
    check(_lib.lib().tgsr_lp_to_nchw(DT[img.dtype], _p(img), _p(out), B, C, H, W, cp, coff, _stream()),
          "tgsr_lp_to_nchw")
    return out


def convert(src: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """An lp image in the other 2-byte type (f16 -> bf16: one rounding; bf16 -> f16: exact in range): the whole buffer,
    zero border included (tgsr_lp_convert)."""
    _need_hip(src, out)
    _img(src, "src")
    _img(out, "out")
    if src.shape != out.shape or src.dtype == out.dtype or src.numel() % 8:
        raise TgsrError("lp.convert: %s %s -> %s %s" % (tuple(src.shape), src.dtype, tuple(out.shape), out.dtype))
    check(_lib.lib().tgsr_lp_convert(DT[src.dtype], _p(src), DT[out.dtype], _p(out), src.numel(), _stream()),
          "tgsr_lp_convert")
    return out


def pack_conv3x3_weight(w: torch.Tensor, dtype) -> torch.Tensor:
    """[Cout,Cin,3,3] fp32 -> MFMA fragment order, rounded to `dtype` (tgsr_lp_pack_conv3x3_weight)."""
    _need_hip(w)
    w = w.detach().contiguous()
    if w.dtype != torch.float32 or w.dim() != 4 or w.shape[2:] != (3, 3):
        raise TgsrError("pack_conv3x3_weight: weight %s %s" % (tuple(w.shape), w.dtype))
    Cout, Cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    out = torch.empty(L.tgsr_lp_packed_conv3x3_elems(Cout, Cin), dtype=torch_dtype(dtype), device=w.device)
    check(L.tgsr_lp_pack_conv3x3_weight(DT[out.dtype], _p(w), _p(out), Cout, Cin, _stream()),
          "tgsr_lp_pack_conv3x3_weight")
    return out


def conv3x3(x: torch.Tensor, wpack: torch.Tensor, cin: int, cout: int, scale, shift, glu: bool = False,
            upsample: bool = False, residual: Optional[torch.Tensor] = None, res_coff: int = 0,
            out: Optional[torch.Tensor] = None, out_coff: int = 0, out_cpitch: Optional[int] = None) -> torch.Tensor:
    """conv3x3 [+ nearest x2 in front] + affine + (GLU | + residual) on lp images in one launch.  Reads channels
    [0, cin) of `x`; writes channels [out_coff, out_coff + cout') of `out` (a new zeroed image unless given)."""
    _need_hip(x, wpack, scale, shift, residual, out)
    B, Hi, Wi, xcp = _img(x, "x")
    H, W = (2 * Hi, 2 * Wi) if upsample else (Hi, Wi)
    co = cout // 2 if glu else cout
    if out is None:
        out = new_image(B, H, W, out_cpitch or (out_coff + co), x.dtype, x.device)
    ob, oh, ow, ocp = _img(out, "out")
    if (ob, oh, ow) != (B, H, W) or out.dtype != x.dtype or wpack.dtype != x.dtype:
        raise TgsrError("lp.conv3x3: out %s / dtypes do not match" % (tuple(out.shape),))
    if cin > xcp or wpack.numel() != _lib.lib().tgsr_lp_packed_conv3x3_elems(cout, cin):
        raise TgsrError("lp.conv3x3: a [%d, %d, 3, 3] filter does not fit the input (%d channels) / the pack (%d values)"
                        % (cout, cin, xcp, wpack.numel()))
    rcp = 0
    if residual is not None:
        rb, rh, rw, rcp = _img(residual, "residual")
        if (rb, rh, rw) != (B, H, W) or residual.dtype != x.dtype:
            raise TgsrError("lp.conv3x3: residual %s" % (tuple(residual.shape),))
    from . import ops
    e0 = ops._ev() if ops.profile is not None else None
    rc = _lib.lib().tgsr_lp_conv3x3_fwd(DT[x.dtype], _p(x), xcp, B, cin, H, W, _p(wpack), cout, _p(scale), _p(shift),
                                        _p(residual), rcp, res_coff, _p(out), ocp, out_coff,
                                        _lib.EPI_AFFINE_GLU if glu else _lib.EPI_AFFINE, 1 if upsample else 0, _stream())
    check(rc, "tgsr_lp_conv3x3_fwd")
    if ops.profile is not None:
        nbytes = 2 * (B * cin * Hi * Wi + B * co * H * W * (2 if residual is not None else 1) + cout * cin * 9)
        ops.profile.append(("lp_conv3x3_kernel", 2.0 * B * H * W * cout * cin * 9, nbytes, e0, ops._ev()))
    return out


def resblocks_flags(B: int, H: int, W: int, device) -> torch.Tensor:
    """The flag buffer of `resblocks` for images of this size: zeroed here once, then owned by the kernel across launches (one
    per set of activation images)."""
    n = _lib.lib().tgsr_lp_resblocks_flag_elems(B, H, W)
    if n <= 0:
        raise TgsrError("lp.resblocks: unsupported image size %d x %d" % (H, W))
    return torch.zeros(n, dtype=torch.int32, device=device)


def resblocks_supported(cin: int, H: int, W: int) -> bool:
    return cin == 64 and W % 32 == 0 and H % 4 == 0


def resblocks(x: torch.Tensor, wpacks, scales, shifts, tmp: torch.Tensor, a: torch.Tensor, b: torch.Tensor,
              flags: torch.Tensor) -> torch.Tensor:
    """Two ResBlocks (four dependent conv3x3 on 64 channels: GLU, + residual, GLU, + residual) in one launch
    (tgsr_lp_resblocks_fwd): x -> tmp -> a -> tmp -> b, bit-identical to four `conv3x3` launches.  Returns b."""
    import ctypes
    _need_hip(x, tmp, a, b, flags, *wpacks, *scales, *shifts)
    B, H, W, xcp = _img(x, "x")
    dims = [_img(t, n) for t, n in ((tmp, "tmp"), (a, "a"), (b, "b"))]
    if any(d[:3] != (B, H, W) for d in dims) or any(t.dtype != x.dtype for t in (tmp, a, b)) or len(wpacks) != 4:
        raise TgsrError("lp.resblocks: images %s / dtypes do not match x %s" % ([tuple(t.shape) for t in (tmp, a, b)], tuple(x.shape)))
    L = _lib.lib()
    for i, wp in enumerate(wpacks):
        if wp.dtype != x.dtype or wp.numel() != L.tgsr_lp_packed_conv3x3_elems(128 if i % 2 == 0 else 64, 64):
            raise TgsrError("lp.resblocks: pack %d is not a [%d, 64, 3, 3] filter of the images' type" % (i, 128 if i % 2 == 0 else 64))
    if flags.dtype != torch.int32 or flags.numel() != L.tgsr_lp_resblocks_flag_elems(B, H, W):
        raise TgsrError("lp.resblocks: flags must come from lp.resblocks_flags(%d, %d, %d)" % (B, H, W))
    wp = (ctypes.c_void_p * 4)(*[w.data_ptr() for w in wpacks])
    sc = (ctypes.c_void_p * 4)(*[None if s_ is None else s_.data_ptr() for s_ in scales])
    sh = (ctypes.c_void_p * 4)(*[None if s_ is None else s_.data_ptr() for s_ in shifts])
    from . import ops
    e0 = ops._ev() if ops.profile is not None else None
    rc = L.tgsr_lp_resblocks_fwd(DT[x.dtype], _p(x), xcp, B, H, W, wp, sc, sh, _p(tmp), dims[0][3], _p(a), dims[1][3], _p(b),
                                 dims[2][3], _p(flags), _stream())
    check(rc, "tgsr_lp_resblocks_fwd")
    if ops.profile is not None:
        px = B * H * W
        ops.profile.append(("lp_conv3x3_kernel", 2.0 * px * 9 * 64 * (128 + 64 + 128 + 64), 2 * (px * 64 * 10 + 9 * 64 * 384), e0, ops._ev()))
    return b


class AttFuse:
    """Arguments of a word attention fused into the kernel that produces h (tgsr_lp_stem_att_fwd,
    tgsr_lp_upconv_glu_att_fwd): pack / nsets = ops.text_tail(..., lp_dtype)'s att_pack of this batch, `index` = which
    projection the stage attends through, T words, use_mask / correct_mask as lp.word_attention, c_coff = first channel of
    c_code in the producer's output image, attn = fp32 [B, T, H, W] attention maps to fill (or None)."""

    def __init__(self, pack, nsets, index, T, use_mask, correct_mask, c_coff, attn):
        self.pack, self.nsets, self.index, self.T = pack, int(nsets), int(index), int(T)
        self.use_mask, self.correct_mask, self.c_coff, self.attn = bool(use_mask), bool(correct_mask), int(c_coff), attn

    def check(self, B, H, W, dtype):
        if self.pack.dtype != torch.uint8 or self.pack.numel() != _lib.lib().tgsr_lp_att_pack_bytes(self.nsets, B):
            raise TgsrError("fused attention: att_pack of %d bytes does not fit %d projections of a batch of %d"
                            % (self.pack.numel(), self.nsets, B))
        if self.attn is not None and (tuple(self.attn.shape) != (B, self.T, H, W) or self.attn.dtype != torch.float32 or
                                      not self.attn.is_contiguous()):
            raise TgsrError("fused attention: attn %s, expected %s" % (tuple(self.attn.shape), (B, self.T, H, W)))

    def args(self):
        return (_p(self.pack), self.nsets, self.index, 1 if self.use_mask else 0, 1 if self.correct_mask else 0, self.T,
                self.c_coff, _p(self.attn))


def _att_profile(att, B, H, W, e0):
    """The fused attention is accounted under its own name (zero-duration marker: its time is inside the producer's)."""
    from . import ops
    if ops.profile is not None and att is not None:
        nbytes = B * H * W * (2 * 32 + (4 * att.T if att.attn is not None else 0))       # h comes from LDS: c_code + attn only
        ops.profile.append(("lp_attention_fused", 4.0 * B * H * W * 32 * att.T, nbytes, e0, e0))


def stem(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, dtype=None,
         out: Optional[torch.Tensor] = None, out_coff: int = 0, out_cpitch: Optional[int] = None,
         att: Optional[AttFuse] = None) -> torch.Tensor:
    """conv3x3 3 -> 2C + affine + GLU from the fp32 NCHW image into C channels of an lp image (tgsr_lp_stem_fwd);
    att: + the first stage's word attention on those channels in the same launch (tgsr_lp_stem_att_fwd)."""
    _need_hip(x, w, scale, shift, out, *(() if att is None else (att.pack, att.attn)))
    x = x.contiguous()
    w = w.detach().contiguous()
    if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3 or tuple(w.shape[1:]) != (3, 3, 3):
        raise TgsrError("lp.stem: x %s / w %s" % (tuple(x.shape), tuple(w.shape)))
    B, _, H, W = x.shape
    C = w.shape[0] // 2
    if out is None:
        out = new_image(B, H, W, out_cpitch or (out_coff + C), dtype, x.device)
    ob, oh, ow, ocp = _img(out, "out")
    if (ob, oh, ow) != (B, H, W):
        raise TgsrError("lp.stem: out %s" % (tuple(out.shape),))
    if att is not None:
        att.check(B, H, W, out.dtype)
        from . import ops
        e0 = ops._ev() if ops.profile is not None else None
        check(_lib.lib().tgsr_lp_stem_att_fwd(DT[out.dtype], _p(x), B, H, W, _p(w), C, _p(scale), _p(shift), _p(out), ocp,
                                              out_coff, *att.args(), _stream()), "tgsr_lp_stem_att_fwd")
        _att_profile(att, B, H, W, e0)
        return out
    check(_lib.lib().tgsr_lp_stem_fwd(DT[out.dtype], _p(x), B, H, W, _p(w), C, _p(scale), _p(shift), _p(out), ocp,
                                      out_coff, _stream()), "tgsr_lp_stem_fwd")
    return out


def pack_to3_weight(w: torch.Tensor, dtype) -> torch.Tensor:
    """[3,32,K,K] fp32 -> the 16-row-padded MFMA fragments of tgsr_lp_conv_to3_fwd."""
    _need_hip(w)
    w = w.detach().contiguous()
    if w.dtype != torch.float32 or w.dim() != 4 or w.shape[0] != 3 or w.shape[2] != w.shape[3]:
        raise TgsrError("pack_to3_weight: weight %s" % (tuple(w.shape),))
    K = int(w.shape[2])
    out = torch.empty(K * 512, dtype=torch_dtype(dtype), device=w.device)
    check(_lib.lib().tgsr_lp_pack_to3_weight(DT[out.dtype], _p(w), _p(out), int(w.shape[1]), K, _stream()),
          "tgsr_lp_pack_to3_weight")
    return out


def conv_to3(x: torch.Tensor, wpack: torch.Tensor, K: int, tanh_axpy: bool = False,
             addend: Optional[torch.Tensor] = None, alpha: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """KxK conv of channels [0,32) of an lp image to a 3-channel fp32 NCHW image [+ tanh + alpha * addend]."""
    _need_hip(x, wpack, addend, out)
    B, H, W, xcp = _img(x, "x")
    if addend is not None:
        addend = addend.contiguous()
        if addend.dtype != torch.float32 or tuple(addend.shape) != (B, 3, H, W):
            raise TgsrError("lp.conv_to3: addend %s" % (tuple(addend.shape),))
    if out is None:
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
    from . import ops
    e0 = ops._ev() if ops.profile is not None else None
    rc = _lib.lib().tgsr_lp_conv_to3_fwd(DT[x.dtype], _p(x), xcp, B, 32, H, W, _p(wpack), K,
                                         _lib.ACT_TANH_AXPY if tanh_axpy else _lib.ACT_NONE, _p(addend), float(alpha),
                                         _p(out), _stream())
    check(rc, "tgsr_lp_conv_to3_fwd")
    if ops.profile is not None:
        nbytes = B * H * W * (2 * 32 + 4 * 3 * (2 if addend is not None else 1))
        ops.profile.append(("lp_to3_kernel", 2.0 * B * H * W * 3 * 32 * K * K, nbytes, e0, ops._ev()))
    return out


def word_attention(h_img: torch.Tensor, src: torch.Tensor, mask: Optional[torch.Tensor], T: int,
                   correct_mask: bool = False, c_coff: int = 32, attn: Optional[torch.Tensor] = None,
                   need_attn: bool = True):
    """GlobalAttentionGeneral.forward on an lp image: reads h = channels [0,32), writes c_code to channels
    [c_coff, c_coff+32) of the SAME image (the reference's cat), returns the fp32 attention maps [B,T,H,W] (or None).
    src = this stage's fp32 word projection [B,32,32] (ops.word_project)."""
    _need_hip(h_img, src, mask, attn)
    B, H, W, cp = _img(h_img, "h_img")
    if tuple(src.shape) != (B, 32, 32) or src.dtype != torch.float32 or not src.is_contiguous():
        raise TgsrError("lp.word_attention: src %s" % (tuple(src.shape),))
    from . import ops
    m8 = None
    if mask is not None:
        if tuple(mask.shape) != (B, T):
            raise TgsrError("lp.word_attention: mask shape %s, expected %s" % (tuple(mask.shape), (B, T)))
        m8 = ops._mask_u8(mask)
    if attn is None and need_attn:
        attn = torch.empty(B, T, H, W, dtype=torch.float32, device=h_img.device)
    e0 = ops._ev() if ops.profile is not None else None
    rc = _lib.lib().tgsr_lp_word_attention_fwd(DT[h_img.dtype], _p(h_img), cp, _p(src), _p(m8), 1 if correct_mask else 0,
                                               B, 32, T, H, W, _p(h_img), cp, c_coff, _p(attn), _stream())
    check(rc, "tgsr_lp_word_attention_fwd")
    if ops.profile is not None:
        nbytes = B * H * W * (2 * 2 * 32 + (4 * T if attn is not None else 0))
        ops.profile.append(("lp_word_attention_kernel", 4.0 * B * H * W * 32 * T, nbytes, e0, ops._ev()))
    return attn


def pack_upconv_weight(w: torch.Tensor, dtype) -> torch.Tensor:
    """[64,Cin,3,3] fp32 -> the pre-summed sub-pixel taps of tgsr_lp_upconv_glu_fwd (fp32 sums, rounded once)."""
    _need_hip(w)
    w = w.detach().contiguous()
    if w.dtype != torch.float32 or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise TgsrError("pack_upconv_weight: weight %s %s" % (tuple(w.shape), w.dtype))
    Cout, Cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    out = torch.empty(L.tgsr_lp_packed_upconv_elems(Cout, Cin), dtype=torch_dtype(dtype), device=w.device)
    check(L.tgsr_lp_pack_upconv_weight(DT[out.dtype], _p(w), _p(out), Cout, Cin, _stream()),
          "tgsr_lp_pack_upconv_weight")
    return out


def upconv_supported(cin: int, cout: int, Hi: int, Wi: int) -> bool:
    return cout == 64 and cin in (32, 64) and Wi % 32 == 0 and Hi % 4 == 0


def upconv_glu(x: torch.Tensor, wpack: torch.Tensor, cin: int, cout: int, scale, shift,
               out: Optional[torch.Tensor] = None, out_coff: int = 0, out_cpitch: Optional[int] = None,
               att: Optional[AttFuse] = None) -> torch.Tensor:
    """upBlock (Upsample x2 -> conv3x3 -> affine -> GLU) on lp images by sub-pixel decomposition, one launch;
    att: + the NEXT stage's word attention on the output tile (tgsr_lp_upconv_glu_att_fwd)."""
    _need_hip(x, wpack, scale, shift, out, *(() if att is None else (att.pack, att.attn)))
    B, Hi, Wi, xcp = _img(x, "x")
    co = cout // 2
    if out is None:
        out = new_image(B, 2 * Hi, 2 * Wi, out_cpitch or (out_coff + co), x.dtype, x.device)
    ob, oh, ow, ocp = _img(out, "out")
    if (ob, oh, ow) != (B, 2 * Hi, 2 * Wi) or out.dtype != x.dtype or wpack.dtype != x.dtype:
        raise TgsrError("lp.upconv_glu: out %s / dtypes do not match" % (tuple(out.shape),))
    if cin > xcp or wpack.numel() != _lib.lib().tgsr_lp_packed_upconv_elems(cout, cin):
        raise TgsrError("lp.upconv_glu: a [%d, %d, 3, 3] filter does not fit the input (%d channels) / the pack (%d values)"
                        % (cout, cin, xcp, wpack.numel()))
    from . import ops
    e0 = ops._ev() if ops.profile is not None else None
    if att is not None:
        att.check(B, 2 * Hi, 2 * Wi, x.dtype)
        rc = _lib.lib().tgsr_lp_upconv_glu_att_fwd(DT[x.dtype], _p(x), xcp, B, cin, Hi, Wi, _p(wpack), cout, _p(scale),
                                                   _p(shift), _p(out), ocp, out_coff, None, 0, None, *att.args(), _stream())
        check(rc, "tgsr_lp_upconv_glu_att_fwd")
    else:
        rc = _lib.lib().tgsr_lp_upconv_glu_fwd(DT[x.dtype], _p(x), xcp, B, cin, Hi, Wi, _p(wpack), cout, _p(scale), _p(shift),
                                               _p(out), ocp, out_coff, _stream())
        check(rc, "tgsr_lp_upconv_glu_fwd")
    if ops.profile is not None:
        nbytes = 2 * (B * cin * Hi * Wi + B * co * 4 * Hi * Wi + cout * cin * 16)
        ops.profile.append(("lp_upconv_glu_kernel", 2.0 * B * 4 * Hi * Wi * cout * cin * 9, nbytes, e0, ops._ev()))
        _att_profile(att, B, 2 * Hi, 2 * Wi, e0)
    return out


def head_partial_elems(B: int, H: int, W: int, K: int) -> int:
    return int(_lib.lib().tgsr_lp_head_partial_elems(B, H, W, K))


def head_fusable(cin: int, cout: int, Hi: int, Wi: int) -> bool:
    """Shapes tgsr_lp_upconv_glu_head_fwd + tgsr_lp_head_combine take: the sub-pixel upBlock kernel's, and an output
    image whose size is a multiple of the 8 x 64 workgroup tile (always true for Hi % 4 == 0, Wi % 32 == 0)."""
    return upconv_supported(cin, cout, Hi, Wi)


def upconv_glu_head(x: torch.Tensor, wpack: torch.Tensor, cin: int, cout: int, scale, shift, head_wpack: torch.Tensor,
                    K: int, partial: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                    out_coff: int = 0, write_out: bool = True, att: Optional[AttFuse] = None):
    """upBlock + the image head reading it, one launch: returns (out lp image or None, partial head sums fp32).  With
    write_out=False the 32-channel feature image is not written at all (its only consumer is the head).
    att (K == 3, the image is written): + the NEXT stage's word attention on the output tile, same launch."""
    _need_hip(x, wpack, scale, shift, head_wpack, partial, out, *(() if att is None else (att.pack, att.attn)))
    B, Hi, Wi, xcp = _img(x, "x")
    L = _lib.lib()
    n = L.tgsr_lp_head_partial_elems(B, 2 * Hi, 2 * Wi, K)
    if partial is None:
        partial = torch.empty(n, dtype=torch.float32, device=x.device)
    if partial.numel() != n or partial.dtype != torch.float32 or head_wpack.dtype != x.dtype or wpack.dtype != x.dtype:
        raise TgsrError("lp.upconv_glu_head: partial buffer / dtypes do not match")
    ocp = 0
    if write_out:
        if out is None:
            out = new_image(B, 2 * Hi, 2 * Wi, out_coff + cout // 2, x.dtype, x.device)
        ob, oh, ow, ocp = _img(out, "out")
        if (ob, oh, ow) != (B, 2 * Hi, 2 * Wi) or out.dtype != x.dtype:
            raise TgsrError("lp.upconv_glu_head: out %s" % (tuple(out.shape),))
    else:
        out = None
    if cin > xcp or wpack.numel() != L.tgsr_lp_packed_upconv_elems(cout, cin) or head_wpack.numel() != K * 512:
        raise TgsrError("lp.upconv_glu_head: filter packs do not fit (%d input channels, %d / %d values)"
                        % (xcp, wpack.numel(), head_wpack.numel()))
    from . import ops
    e0 = ops._ev() if ops.profile is not None else None
    if att is not None:
        if not write_out or K != 3:
            raise TgsrError("lp.upconv_glu_head: a fused attention needs the feature image written and the 3x3 head")
        att.check(B, 2 * Hi, 2 * Wi, x.dtype)
        rc = L.tgsr_lp_upconv_glu_att_fwd(DT[x.dtype], _p(x), xcp, B, cin, Hi, Wi, _p(wpack), cout, _p(scale), _p(shift),
                                          _p(out), ocp, out_coff, _p(head_wpack), K, _p(partial), *att.args(), _stream())
        check(rc, "tgsr_lp_upconv_glu_att_fwd")
    else:
        rc = L.tgsr_lp_upconv_glu_head_fwd(DT[x.dtype], _p(x), xcp, B, cin, Hi, Wi, _p(wpack), cout, _p(scale), _p(shift),
                                           _p(out), ocp, out_coff, _p(head_wpack), K, _p(partial), _stream())
        check(rc, "tgsr_lp_upconv_glu_head_fwd")
    if ops.profile is not None:
        co = cout // 2
        nbytes = 2 * (B * cin * Hi * Wi + (B * co * 4 * Hi * Wi if write_out else 0) + cout * cin * 16) + 4 * n
        flops = 2.0 * B * 4 * Hi * Wi * (cout * cin * 9 + 3 * co * K * K)
        ops.profile.append(("lp_upconv_glu_kernel", flops, nbytes, e0, ops._ev()))
        _att_profile(att, B, 2 * Hi, 2 * Wi, e0)
    return out, partial


def head_combine(B: int, sizes, partial_low, partial_high, low, high, low_tanh: bool, alpha: float):
    """tgsr_lp_head_combine over len(sizes) <= 4 scales: sizes[s] = (H, W); partial_low / partial_high / low / high lists
    (entries may be None as the header describes).  One launch."""
    n = len(sizes)
    ts = [t for lst in (partial_low, partial_high, low, high) for t in lst if t is not None]
    _need_hip(*ts)
    L = _lib.lib()
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TgsrError("lp.head_combine: fp32 contiguous tensors only")
    for k, (H, W) in enumerate(sizes):
        for t in (low[k], high[k]):
            if t is not None and tuple(t.shape) != (B, 3, H, W):
                raise TgsrError("lp.head_combine: image %s, expected %s" % (tuple(t.shape), (B, 3, H, W)))
        if partial_low[k] is not None and partial_low[k].numel() != L.tgsr_lp_head_partial_elems(B, H, W, 3):
            raise TgsrError("lp.head_combine: partial_low[%d] has %d values" % (k, partial_low[k].numel()))
        if partial_high[k] is not None and partial_high[k].numel() != L.tgsr_lp_head_partial_elems(B, H, W, 5):
            raise TgsrError("lp.head_combine: partial_high[%d] has %d values" % (k, partial_high[k].numel()))
    arr_i = ctypes.c_int * n
    arr_p = ctypes.c_void_p * n
    ptr = lambda lst: arr_p(*[None if t is None else t.data_ptr() for t in lst])      # noqa: E731
    from . import ops
    e0 = ops._ev() if ops.profile is not None else None
    rc = L.tgsr_lp_head_combine(n, B, arr_i(*[s[0] for s in sizes]), arr_i(*[s[1] for s in sizes]), ptr(partial_low),
                                ptr(partial_high), ptr(low), ptr(high), 1 if low_tanh else 0, float(alpha), _stream())
    check(rc, "tgsr_lp_head_combine")
    if ops.profile is not None:
        nbytes = 4 * sum(t.numel() for t in ts)
        ops.profile.append(("lp_head_combine_kernel", 0.0, nbytes, e0, ops._ev()))
    return low, high
