"""`torch.ops.tgsr.*`: the HIP entry points as PyTorch-ROCm custom operators (BASELINE.json north_star: "driven from
Python through PyTorch-ROCm custom ops over a thin C-ABI").

Each op is a schema in the `tgsr` namespace whose CUDA(=HIP) kernel is the ctypes call into libtgsr_hip.so
(`tgsr_amd.ops`), plus a fake (meta) kernel for shape propagation and - for the two differentiable ones below - an
autograd formula registered with `torch.library.register_autograd` whose backward is itself a `tgsr::` op.  The drop-in
modules (`tgsr_amd.util`) call these operators, not the ctypes wrappers directly.

Registered through `torch.library.Library.define / impl` rather than the `@custom_op` decorator: measured on this
torch build the decorator's Python wrapper costs ~11 us per call against ~2 us for a Library-registered op, and one SR
forward issues ~60 of them (0.7 ms of host time per 1.7 ms step otherwise).

Functional ops return fresh tensors; the `*_out` variants write into a caller-provided channel-slice view (how the
reference's torch.cat((h_code, c_code), 1), util.py:771/817, disappears) and are declared as mutating that argument.
"""
from typing import Optional

import torch

from . import ops

_lib = torch.library.Library("tgsr", "DEF")
_T = torch.Tensor


def _define(schema, fn, fake=None):
    name = schema.split("(")[0]
    _lib.define(schema)
    _lib.impl(name, fn, "CUDA")
    if fake is not None:
        torch.library.register_fake("tgsr::" + name, fake, lib=_lib)
    return getattr(torch.ops.tgsr, name).default


def _co(cout, glu):
    return cout // 2 if glu else cout


# ------------------------------------------------------------------------------------------------ fused conv3x3 (direct)
def _conv3x3_fused(x, wpack, cout, scale, shift, glu, upsample, residual):
    return ops.conv3x3_fused(x, wpack, cout, scale, shift, glu=glu, upsample=upsample, residual=residual)


def _conv3x3_fused_fake(x, wpack, cout, scale, shift, glu, upsample, residual):
    B, _, H, W = x.shape
    m = 2 if upsample else 1
    return x.new_empty(B, _co(cout, glu), H * m, W * m)


def _conv3x3_fused_out(x, wpack, cout, scale, shift, glu, upsample, residual, out):
    ops.conv3x3_fused(x, wpack, cout, scale, shift, glu=glu, upsample=upsample, residual=residual, out=out)


conv3x3_fused = _define("conv3x3_fused(Tensor x, Tensor wpack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                        "bool upsample, Tensor? residual) -> Tensor", _conv3x3_fused, _conv3x3_fused_fake)
conv3x3_fused_out = _define("conv3x3_fused_out(Tensor x, Tensor wpack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                            "bool upsample, Tensor? residual, Tensor(a!) out) -> ()", _conv3x3_fused_out,
                            lambda *a: None)


# ------------------------------------------------------------------------------------------------ Winograd conv3x3
def _conv3x3_wino(x, upack, cout, scale, shift, glu, residual):
    return ops.conv3x3_wino(x, upack, cout, scale, shift, glu=glu, residual=residual)


def _conv3x3_wino_out(x, upack, cout, scale, shift, glu, residual, out):
    ops.conv3x3_wino(x, upack, cout, scale, shift, glu=glu, residual=residual, out=out)


conv3x3_wino = _define("conv3x3_wino(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                       "Tensor? residual) -> Tensor", _conv3x3_wino,
                       lambda x, upack, cout, scale, shift, glu, residual:
                       x.new_empty(x.shape[0], _co(cout, glu), x.shape[2], x.shape[3]))
conv3x3_wino_out = _define("conv3x3_wino_out(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                           "Tensor? residual, Tensor(a!) out) -> ()", _conv3x3_wino_out, lambda *a: None)


# ------------------------------------------------------------------------------------------------ upBlock forms
def _up_fake(x, pack, cout, scale, shift):
    return x.new_empty(x.shape[0], cout // 2, 2 * x.shape[2], 2 * x.shape[3])


upwino_glu = _define("upwino_glu(Tensor x, Tensor upack, int cout, Tensor scale, Tensor shift) -> Tensor",
                     lambda x, p, cout, s, t: ops.upwino_glu(x, p, cout, s, t), _up_fake)
upwino_glu_out = _define("upwino_glu_out(Tensor x, Tensor upack, int cout, Tensor scale, Tensor shift, Tensor(a!) out) -> ()",
                         lambda x, p, cout, s, t, out: (ops.upwino_glu(x, p, cout, s, t, out=out), None)[1], lambda *a: None)
upconv3x3_glu = _define("upconv3x3_glu(Tensor x, Tensor wpack, int cout, Tensor scale, Tensor shift) -> Tensor",
                        lambda x, p, cout, s, t: ops.upconv3x3_glu(x, p, cout, s, t), _up_fake)
upconv3x3_glu_out = _define("upconv3x3_glu_out(Tensor x, Tensor wpack, int cout, Tensor scale, Tensor shift, "
                            "Tensor(a!) out) -> ()",
                            lambda x, p, cout, s, t, out: (ops.upconv3x3_glu(x, p, cout, s, t, out=out), None)[1],
                            lambda *a: None)


# ------------------------------------------------------------------------------------------------ word attention
def _word_attention(h, words, w_ctx, mask, correct_mask, src):
    return ops.word_attention(h, words, w_ctx, mask, correct_mask, src=src)


def _word_attention_fake(h, words, w_ctx, mask, correct_mask, src):
    return torch.empty_like(h), h.new_empty(h.shape[0], words.shape[2], h.shape[2], h.shape[3])


def _word_attention_out(h, words, w_ctx, mask, correct_mask, src, out):
    return ops.word_attention(h, words, w_ctx, mask, correct_mask, out=out, src=src)[1]


word_attention = _define("word_attention(Tensor h, Tensor words, Tensor w_ctx, Tensor? mask, bool correct_mask, "
                         "Tensor? src) -> (Tensor, Tensor)", _word_attention, _word_attention_fake)
word_attention_out = _define("word_attention_out(Tensor h, Tensor words, Tensor w_ctx, Tensor? mask, bool correct_mask, "
                             "Tensor? src, Tensor(a!) out) -> Tensor", _word_attention_out,
                             lambda h, words, w_ctx, mask, cm, src, out:
                             h.new_empty(h.shape[0], words.shape[2], h.shape[2], h.shape[3]))


# ------------------------------------------------------------------------------------------------ image heads (differentiable)
def _conv_to3(x, w, tanh_axpy, addend, alpha):
    return ops.conv_to3(x, w, tanh_axpy=tanh_axpy, addend=addend, alpha=alpha)


def _conv_to3_bwd(dy, out, addend, alpha, x, w, tanh_axpy, need_dx, need_dw):
    """(dx, dw) of conv_to3; a gradient that is not needed comes back as an empty tensor."""
    from . import _lib as L
    from ._lib import check
    from .ops import _p, _stream
    lib = L.lib()
    dy, x, w = dy.contiguous(), x.contiguous(), w.contiguous()
    addend = None if addend is None else addend.contiguous()     # the kernel reads it as dense NCHW
    B, Cin, H, W = x.shape
    K = w.shape[2]
    dx = torch.empty_like(x) if need_dx else x.new_empty(0)
    dw = torch.empty_like(w) if need_dw else x.new_empty(0)
    ws = x.new_empty(lib.tgsr_conv_to3_bwd_ws_elems(B, Cin, H, W, K)) if need_dw else None
    rc = lib.tgsr_conv_to3_bwd(_p(dy), _p(out), _p(addend), float(alpha), _p(x), Cin * H * W, _p(w), B, Cin, H, W, K,
                               L.ACT_TANH_AXPY if tanh_axpy else L.ACT_NONE, _p(dx) if need_dx else None, _p(ws),
                               _p(dw) if need_dw else None, _stream())
    check(rc, "tgsr_conv_to3_bwd")
    return dx, dw


conv_to3 = _define("conv_to3(Tensor x, Tensor w, bool tanh_axpy, Tensor? addend, float alpha) -> Tensor", _conv_to3,
                   lambda x, w, tanh_axpy, addend, alpha: x.new_empty(x.shape[0], 3, x.shape[2], x.shape[3]))
conv_to3_bwd = _define("conv_to3_bwd(Tensor dy, Tensor? out, Tensor? addend, float alpha, Tensor x, Tensor w, bool tanh_axpy, "
                       "bool need_dx, bool need_dw) -> (Tensor, Tensor)", _conv_to3_bwd,
                       lambda dy, out, addend, alpha, x, w, t, ndx, ndw:
                       (torch.empty_like(x) if ndx else x.new_empty(0), torch.empty_like(w) if ndw else x.new_empty(0)))


def _conv_to3_setup(ctx, inputs, output):
    x, w, tanh_axpy, addend, alpha = inputs
    ctx.save_for_backward(x, w, output if tanh_axpy else None, addend)
    ctx.tanh_axpy, ctx.alpha = tanh_axpy, alpha


def _conv_to3_backward(ctx, dy):
    x, w, out, addend = ctx.saved_tensors
    ndx, ndw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    dx = dw = None
    if ndx or ndw:
        dx, dw = conv_to3_bwd(dy, out, addend, ctx.alpha, x, w, ctx.tanh_axpy, ndx, ndw)
    dadd = dy * ctx.alpha if (addend is not None and ctx.needs_input_grad[3]) else None
    return (dx if ndx else None), (dw if ndw else None), None, dadd, None


torch.library.register_autograd("tgsr::conv_to3", _conv_to3_backward, setup_context=_conv_to3_setup, lib=_lib)


# ------------------------------------------------------------------------------------------------ downBlock conv (differentiable)
conv4x4s2 = _define("conv4x4s2(Tensor x, Tensor w, bool leaky) -> Tensor",
                    lambda x, w, leaky: ops.conv4x4s2(x, w, leaky=leaky),
                    lambda x, w, leaky: x.new_empty(x.shape[0], w.shape[0], x.shape[2] // 2, x.shape[3] // 2))
conv4x4s2_dgrad = _define("conv4x4s2_dgrad(Tensor dy, Tensor w, int H, int W) -> Tensor",
                          lambda dy, w, H, W: ops.conv4x4s2_dgrad(dy, w, H, W),
                          lambda dy, w, H, W: dy.new_empty(dy.shape[0], w.shape[1], H, W))
conv4x4s2_wgrad = _define("conv4x4s2_wgrad(Tensor dy, Tensor x) -> Tensor",
                          lambda dy, x: ops.conv4x4s2_wgrad(dy, x),
                          lambda dy, x: dy.new_empty(dy.shape[1], x.shape[1], 4, 4))
leaky_relu_bwd = _define("leaky_relu_bwd(Tensor dy, Tensor y) -> Tensor", lambda dy, y: ops.leaky_relu_bwd(dy, y),
                         lambda dy, y: torch.empty_like(dy))


def _conv4x4s2_setup(ctx, inputs, output):
    x, w, leaky = inputs
    ctx.save_for_backward(x, w, output if leaky else None)


def _conv4x4s2_backward(ctx, dy):
    x, w, out = ctx.saved_tensors
    g = dy.contiguous() if out is None else leaky_relu_bwd(dy.contiguous(), out)
    dx = conv4x4s2_dgrad(g, w, x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
    dw = conv4x4s2_wgrad(g, x.contiguous()) if ctx.needs_input_grad[1] else None
    return dx, dw, None


torch.library.register_autograd("tgsr::conv4x4s2", _conv4x4s2_backward, setup_context=_conv4x4s2_setup, lib=_lib)


# ------------------------------------------------------------------------------------------------ stand-alone GLU (differentiable)
glu = _define("glu(Tensor x) -> Tensor", lambda x: ops.glu(x),
              lambda x: x.new_empty((x.shape[0], x.shape[1] // 2) + tuple(x.shape[2:])))
glu_bwd = _define("glu_bwd(Tensor dy, Tensor x) -> Tensor", lambda dy, x: ops.glu_bwd(dy, x), lambda dy, x: torch.empty_like(x))
torch.library.register_autograd("tgsr::glu", lambda ctx, dy: glu_bwd(dy, ctx.saved_tensors[0]),
                                setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]), lib=_lib)
