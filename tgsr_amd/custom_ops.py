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


def _refuse_cpu(name):
    def kernel(*args, **kwargs):
        raise ops.TgsrError("torch.ops.tgsr.%s got CPU tensors: tgsr_amd ops run only on HIP tensors; there is no CPU fallback" % name)
    return kernel


def _define(schema, fn, fake=None):
    name = schema.split("(")[0]
    _lib.define(schema)
    _lib.impl(name, fn, "CUDA")
    _lib.impl(name, _refuse_cpu(name), "CPU")        # a loud refusal (the same TgsrError the ctypes wrappers raise), not a fallback
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


def _wino_stats_fake(x, upack, cout):
    n = ops.wino_stats_nslots(x.shape[0], x.shape[2], x.shape[3], cout)
    return x.new_empty(x.shape[0], cout, x.shape[2], x.shape[3]), x.new_empty(cout, n, 2)


conv3x3_wino_stats = _define("conv3x3_wino_stats(Tensor x, Tensor upack, int cout) -> (Tensor, Tensor)",
                             lambda x, upack, cout: ops.conv3x3_wino_stats(x, upack, cout), _wino_stats_fake)
conv3x3_wino = _define("conv3x3_wino(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                       "Tensor? residual) -> Tensor", _conv3x3_wino,
                       lambda x, upack, cout, scale, shift, glu, residual:
                       x.new_empty(x.shape[0], _co(cout, glu), x.shape[2], x.shape[3]))
conv3x3_wino_out = _define("conv3x3_wino_out(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                           "Tensor? residual, Tensor(a!) out) -> ()", _conv3x3_wino_out, lambda *a: None)


def _wino4_stats_fake(x, upack, cout):
    n = ops.wino4_stats_nslots(x.shape[0], x.shape[2], x.shape[3], cout)
    return x.new_empty(x.shape[0], cout, x.shape[2], x.shape[3]), x.new_empty(cout, n, 2)


conv3x3_wino4_stats = _define("conv3x3_wino4_stats(Tensor x, Tensor upack, int cout) -> (Tensor, Tensor)",
                              lambda x, upack, cout: ops.conv3x3_wino4_stats(x, upack, cout), _wino4_stats_fake)
conv3x3_wino4 = _define("conv3x3_wino4(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                        "Tensor? residual) -> Tensor",
                        lambda x, upack, cout, scale, shift, glu, residual:
                        ops.conv3x3_wino4(x, upack, cout, scale, shift, glu=glu, residual=residual),
                        lambda x, upack, cout, scale, shift, glu, residual:
                        x.new_empty(x.shape[0], _co(cout, glu), x.shape[2], x.shape[3]))
conv3x3_wino4_out = _define("conv3x3_wino4_out(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                            "Tensor? residual, Tensor(a!) out) -> ()",
                            lambda x, upack, cout, scale, shift, glu, residual, out:
                            (ops.conv3x3_wino4(x, upack, cout, scale, shift, glu=glu, residual=residual, out=out), None)[1],
                            lambda *a: None)


pack_wino4w_weight = _define("pack_wino4w_weight(Tensor w, bool glu, bool dgrad) -> Tensor",
                             lambda w, g, d: ops.pack_wino4w_weight(w, glu=g, dgrad=d),
                             lambda w, g, d: w.new_empty(((w.shape[0 if d else 1] + 3) // 4) * 144 * w.shape[1 if d else 0]))


def _wino4w_stats_fake(x, upack, cout):
    n = ops.wino4_stats_nslots(x.shape[0], x.shape[2], x.shape[3], cout, wide=True)
    return x.new_empty(x.shape[0], cout, x.shape[2], x.shape[3]), x.new_empty(cout, n, 2)


conv3x3_wino4w_stats = _define("conv3x3_wino4w_stats(Tensor x, Tensor upack, int cout) -> (Tensor, Tensor)",
                               lambda x, upack, cout: ops.conv3x3_wino4_stats(x, upack, cout, wide=True), _wino4w_stats_fake)
conv3x3_wino4w = _define("conv3x3_wino4w(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                         "Tensor? residual) -> Tensor",
                         lambda x, upack, cout, scale, shift, glu, residual:
                         ops.conv3x3_wino4(x, upack, cout, scale, shift, glu=glu, residual=residual, wide=True),
                         lambda x, upack, cout, scale, shift, glu, residual:
                         x.new_empty(x.shape[0], _co(cout, glu), x.shape[2], x.shape[3]))
conv3x3_wino4w_out = _define("conv3x3_wino4w_out(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu, "
                             "Tensor? residual, Tensor(a!) out) -> ()",
                             lambda x, upack, cout, scale, shift, glu, residual, out:
                             (ops.conv3x3_wino4(x, upack, cout, scale, shift, glu=glu, residual=residual, out=out, wide=True), None)[1],
                             lambda *a: None)


# ------------------------------------------------------------------------------------------------ upBlock forms
def _up_fake(x, pack, cout, scale, shift):
    return x.new_empty(x.shape[0], cout // 2, 2 * x.shape[2], 2 * x.shape[3])


upwino4_glu = _define("upwino4_glu(Tensor x, Tensor upack, int cout, Tensor scale, Tensor shift) -> Tensor",
                      lambda x, p, cout, s, t: ops.upwino4_glu(x, p, cout, s, t), _up_fake)
upwino4_glu_out = _define("upwino4_glu_out(Tensor x, Tensor upack, int cout, Tensor scale, Tensor shift, Tensor(a!) out) -> ()",
                          lambda x, p, cout, s, t, out: (ops.upwino4_glu(x, p, cout, s, t, out=out), None)[1], lambda *a: None)
pack_upwino4_weight = _define("pack_upwino4_weight(Tensor w, bool glu) -> Tensor", lambda w, g: ops.pack_upwino4_weight(w, glu=g),
                              lambda w, g: w.new_empty(((w.shape[1] + 3) // 4) * 112 * w.shape[0]))
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


word_project = _define("word_project(Tensor words, Tensor[] w_ctxs) -> Tensor[]", lambda words, ws: ops.word_project(words, list(ws)),
                       lambda words, ws: list(words.new_empty(len(ws), words.shape[0], ws[0].shape[0], 32).unbind(0)))


# ------------------------------------------------------------------------------------------------ image heads (differentiable)
def _conv_to3(x, w, tanh_axpy, addend, alpha):
    return ops.conv_to3(x, w, tanh_axpy=tanh_axpy, addend=addend, alpha=alpha)


def _conv_to3_bwd(dy, out, addend, alpha, x, w, tanh_axpy, need_dx, need_dw):
    """(dx, dw) of conv_to3; a gradient that is not needed comes back as an empty tensor."""
    dx, dw = ops.conv_to3_bwd(dy, out, addend, alpha, x, w, tanh_axpy, need_dx, need_dw)
    return (dx if dx is not None else x.new_empty(0)), (dw if dw is not None else x.new_empty(0))


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


# ================================================================================================ training primitives
# The blocks of tgsr_amd.autograd (ConvBnAct, ResBlockFn, ConvBnLeaky) are formulas over these operators; gradient
# tensors that must land in a given place (a slot of the flat gradient bucket, parallel.grad_slot) are mutable arguments.
def _bn_train_fwd(raw, gamma, beta, eps, momentum, running_mean, running_var, act, residual, nbt):
    return ops.bn_train_fwd(raw, gamma, beta, eps, momentum, running_mean, running_var, act, residual, nbt)


bn_train_fwd = _define("bn_train_fwd(Tensor raw, Tensor gamma, Tensor beta, float eps, float momentum, Tensor(a!)? running_mean, "
                       "Tensor(b!)? running_var, int act, Tensor? residual, Tensor(c!)? num_batches_tracked) -> (Tensor, Tensor)",
                       _bn_train_fwd,
                       lambda raw, g, b, eps, mom, rm, rv, act, res, nbt:
                       (raw.new_empty(raw.shape[0], raw.shape[1] // 2 if act == 1 else raw.shape[1], raw.shape[2], raw.shape[3]),
                        raw.new_empty(4, raw.shape[1])))
bn_train_fwd_from_stats = _define(
    "bn_train_fwd_from_stats(Tensor raw, Tensor gamma, Tensor beta, float eps, float momentum, Tensor(a!)? running_mean, "
    "Tensor(b!)? running_var, int act, Tensor? residual, Tensor(c!)? num_batches_tracked, Tensor stat_partial) -> (Tensor, Tensor)",
    lambda raw, g, b, eps, mom, rm, rv, act, res, nbt, sp: ops.bn_train_fwd(raw, g, b, eps, mom, rm, rv, act, res, nbt, stat_partial=sp),
    lambda raw, g, b, eps, mom, rm, rv, act, res, nbt, sp:
    (raw.new_empty(raw.shape[0], raw.shape[1] // 2 if act == 1 else raw.shape[1], raw.shape[2], raw.shape[3]),
     raw.new_empty(4, raw.shape[1])))
bn_train_fwd_out = _define("bn_train_fwd_out(Tensor raw, Tensor gamma, Tensor beta, float eps, float momentum, "
                           "Tensor(a!)? running_mean, Tensor(b!)? running_var, int act, Tensor(c!)? num_batches_tracked, "
                           "Tensor(d!) out, Tensor(e!) stats) -> ()",
                           lambda raw, g, b, eps, mom, rm, rv, act, nbt, out, stats:
                           (ops.bn_train_fwd(raw, g, b, eps, mom, rm, rv, act, None, nbt, out=out, stats=stats), None)[1],
                           lambda *a: None)
bn_train_bwd = _define("bn_train_bwd(Tensor dout, Tensor raw, Tensor stats, int act, Tensor(a!) dgamma, Tensor(b!) dbeta, "
                       "Tensor(c!)? draw) -> Tensor",
                       lambda dout, raw, stats, act, dg, db, draw: ops.bn_train_bwd(dout, raw, stats, act, dg, db, draw)[0]
                       if draw is None else (ops.bn_train_bwd(dout, raw, stats, act, dg, db, draw), raw.new_empty(0))[1],
                       lambda dout, raw, stats, act, dg, db, draw: torch.empty_like(raw) if draw is None else raw.new_empty(0))
conv3x3_wgrad = _define("conv3x3_wgrad(Tensor draw, Tensor x, bool upsample, bool winograd, Tensor(a!) dw) -> ()",
                        lambda draw, x, up, wino, dw: (ops.conv3x3_wgrad(draw, x, up, wino, out=dw), None)[1], lambda *a: None)
sumpool2x2 = _define("sumpool2x2(Tensor x) -> Tensor", lambda x: ops.sumpool2x2(x),
                     lambda x: x.new_empty(x.shape[0], x.shape[1], x.shape[2] // 2, x.shape[3] // 2))
conv3x3_gemm = _define("conv3x3_gemm(Tensor x, Tensor w) -> Tensor", lambda x, w: ops.conv3x3_gemm(x, w),
                       lambda x, w: x.new_empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3]))
conv3x3_gemm_dgrad = _define("conv3x3_gemm_dgrad(Tensor dy, Tensor w) -> Tensor", lambda dy, w: ops.conv3x3_gemm_dgrad(dy, w),
                             lambda dy, w: dy.new_empty(dy.shape[0], w.shape[1], dy.shape[2], dy.shape[3]))
conv3x3_gemm_wgrad_out = _define("conv3x3_gemm_wgrad_out(Tensor dy, Tensor x, Tensor(a!) dw) -> ()",
                                 lambda dy, x, dw: (ops.conv3x3_gemm_wgrad(dy, x, out=dw), None)[1], lambda *a: None)
conv4x4s2_wgrad_out = _define("conv4x4s2_wgrad_out(Tensor dy, Tensor x, Tensor(a!) dw) -> ()",
                              lambda dy, x, dw: (ops.conv4x4s2_wgrad(dy, x, out=dw), None)[1], lambda *a: None)
pack_conv3x3_weight = _define("pack_conv3x3_weight(Tensor w, bool dgrad) -> Tensor", lambda w, d: ops.pack_conv3x3_weight(w, dgrad=d),
                              lambda w, d: w.new_empty(((w.shape[0 if d else 1] + 3) // 4) * 36 * w.shape[1 if d else 0]))
pack_wino_weight = _define("pack_wino_weight(Tensor w, bool glu, bool dgrad) -> Tensor",
                           lambda w, g, d: ops.pack_wino_weight(w, glu=g, dgrad=d),
                           lambda w, g, d: w.new_empty(((w.shape[0 if d else 1] + 3) // 4) * 64 * w.shape[1 if d else 0]))
pack_wino4_weight = _define("pack_wino4_weight(Tensor w, bool glu, bool dgrad) -> Tensor",
                            lambda w, g, d: ops.pack_wino4_weight(w, glu=g, dgrad=d),
                            lambda w, g, d: w.new_empty(((w.shape[0 if d else 1] + 3) // 4) * 144 * w.shape[1 if d else 0]))
pack_upwino_weight = _define("pack_upwino_weight(Tensor w, bool glu) -> Tensor", lambda w, g: ops.pack_upwino_weight(w, glu=g),
                             lambda w, g: w.new_empty(((w.shape[1] + 3) // 4) * (w.shape[0] // 64) * 2304))
upwino = _define("upwino(Tensor x, Tensor upack, int cout, Tensor? scale, Tensor? shift, bool glu) -> Tensor",
                 lambda x, p, cout, s, t, glu: ops.upwino_glu(x, p, cout, s, t, glu=glu),
                 lambda x, p, cout, s, t, glu: x.new_empty(x.shape[0], cout // 2 if glu else cout, 2 * x.shape[2], 2 * x.shape[3]))


# ------------------------------------------------------------------------------------------------ word attention backward
word_attention_bwd = _define("word_attention_bwd(Tensor h, Tensor src, Tensor? mask, bool correct_mask, int T, Tensor dc) -> "
                             "(Tensor, Tensor)",
                             lambda h, src, mask, cm, T, dc: ops.word_attention_bwd(h, src, mask, cm, T, dc),
                             lambda h, src, mask, cm, T, dc: (torch.empty_like(h), h.new_empty(h.shape[0], h.shape[1], T)))


# ------------------------------------------------------------------------------------------------ logit heads (row dot)
rowdot = _define("rowdot(Tensor x, Tensor w, Tensor? bias) -> Tensor", lambda x, w, b: ops.rowdot(x, w, b),
                 lambda x, w, b: x.new_empty(x.shape[0]))


def _rowdot_bwd(dy, x, w, need_dx, need_dw):
    dx, dw = ops.rowdot_bwd(dy, x, w, need_dx, need_dw)
    return (dx if dx is not None else x.new_empty(0)), (dw if dw is not None else x.new_empty(0))


rowdot_bwd = _define("rowdot_bwd(Tensor dy, Tensor x, Tensor w, bool need_dx, bool need_dw) -> (Tensor, Tensor)", _rowdot_bwd,
                     lambda dy, x, w, ndx, ndw: (torch.empty_like(x) if ndx else x.new_empty(0),
                                                 x.new_empty(x.shape[1]) if ndw else x.new_empty(0)))


def _rowdot_backward(ctx, dy):
    x, w = ctx.saved_tensors
    ndx, ndw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    dx = dw = None
    if ndx or ndw:
        dx, dw = rowdot_bwd(dy, x, w, ndx, ndw)
    return (dx if ndx else None), (dw.reshape(w.shape) if ndw else None), (dy.sum().reshape(1) if ctx.needs_input_grad[2] else None)


torch.library.register_autograd("tgsr::rowdot", _rowdot_backward,
                                setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0], inputs[1]), lib=_lib)


# ------------------------------------------------------------------------------------------------ GEMM heads (differentiable)
linear = _define("linear(Tensor x, Tensor w, Tensor? bias) -> Tensor", lambda x, w, b: ops.linear(x, w, b),
                 lambda x, w, b: x.new_empty(x.shape[0], w.shape[0]))
conv1x1 = _define("conv1x1(Tensor x, Tensor w) -> Tensor", lambda x, w: ops.conv1x1(x, w),
                  lambda x, w: x.new_empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3]))


def _gemm_nt(a, b):
    """a [M,K] @ b[N,K]^T on the HIP GEMM kernel."""
    return linear(a.contiguous(), b.contiguous(), None)


def _linear_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = _gemm_nt(dy, w.detach().t()) if ctx.needs_input_grad[0] else None          # [B,N] @ W [N,K]
    dw = _gemm_nt(dy.t(), x.detach().t()) if ctx.needs_input_grad[1] else None      # dy^T x
    return dx, dw, (dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None)


def _linear_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.has_bias = inputs[2] is not None


torch.library.register_autograd("tgsr::linear", _linear_backward, setup_context=_linear_setup, lib=_lib)


def _conv1x1_backward(ctx, dy):
    x, w = ctx.saved_tensors
    Cout, Cin = w.shape[0], w.shape[1]
    dy = dy.contiguous()
    dx = dw = None
    if ctx.needs_input_grad[0]:                                    # (the frozen trunk's features need none)
        dx = conv1x1(dy, w.detach().reshape(Cout, Cin).t().contiguous())
    if ctx.needs_input_grad[1]:
        dw = ops.conv1x1_wgrad(dy, x.detach())                     # the implicit-GEMM kernel where the shape qualifies
        if dw is None:
            dy2 = dy.permute(1, 0, 2, 3).reshape(Cout, -1)         # [Cout, B*S]
            x2 = x.detach().permute(1, 0, 2, 3).reshape(Cin, -1)   # [Cin,  B*S]
            dw = _gemm_nt(dy2, x2)
        dw = dw.reshape(w.shape)
    return dx, dw


torch.library.register_autograd("tgsr::conv1x1", _conv1x1_backward,
                                setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0], inputs[1]), lib=_lib)


# ------------------------------------------------------------------------------------------------ text encoder
bilstm_table = _define("bilstm_table(Tensor captions, int[] cap_lens, Tensor table, Tensor w_hh) -> (Tensor, Tensor)",
                       lambda c, lens, table, w_hh: ops.bilstm_table(c, list(lens), table, w_hh),
                       lambda c, lens, table, w_hh: (table.new_empty(c.shape[0], 2 * w_hh.shape[2], max(lens)),
                                                     table.new_empty(c.shape[0], 2 * w_hh.shape[2])))
# lengths read from a device int32 tensor: no argument of the launch depends on their values (hipGraph replay on new batches)
bilstm_table_static = _define("bilstm_table_static(Tensor captions, Tensor cap_lens, Tensor table, Tensor w_hh) -> (Tensor, Tensor)",
                              lambda c, lens, table, w_hh: ops.bilstm_table(c, lens, table, w_hh),
                              lambda c, lens, table, w_hh: (table.new_empty(c.shape[0], 2 * w_hh.shape[2], c.shape[1]),
                                                            table.new_empty(c.shape[0], 2 * w_hh.shape[2])))
# the GRU branch of RNN_ENCODER (util.py:207-211), eval mode
gru_gate_table = _define("gru_gate_table(Tensor emb, Tensor w_ih, Tensor b_ih, Tensor b_hh) -> (Tensor, Tensor)",
                         lambda e, w, bi, bh: ops.gru_gate_table(e, w, bi, bh),
                         lambda e, w, bi, bh: (e.new_empty(e.shape[0], 2, w.shape[1]), e.new_empty(2, w.shape[1] // 3)))
bigru_table = _define("bigru_table(Tensor captions, int[] cap_lens, Tensor table, Tensor w_hh, Tensor b_hn) -> (Tensor, Tensor)",
                      lambda c, lens, table, w_hh, b_hn: ops.bigru_table(c, list(lens), table, w_hh, b_hn),
                      lambda c, lens, table, w_hh, b_hn: (table.new_empty(c.shape[0], 2 * w_hh.shape[2], max(lens)),
                                                          table.new_empty(c.shape[0], 2 * w_hh.shape[2])))
bigru_table_static = _define("bigru_table_static(Tensor captions, Tensor cap_lens, Tensor table, Tensor w_hh, Tensor b_hn) -> "
                             "(Tensor, Tensor)",
                             lambda c, lens, table, w_hh, b_hn: ops.bigru_table(c, lens, table, w_hh, b_hn),
                             lambda c, lens, table, w_hh, b_hn: (table.new_empty(c.shape[0], 2 * w_hh.shape[2], c.shape[1]),
                                                                 table.new_empty(c.shape[0], 2 * w_hh.shape[2])))
lstm_gate_table = _define("lstm_gate_table(Tensor emb, Tensor w_ih, Tensor b_ih, Tensor b_hh) -> Tensor",
                          lambda e, w, bi, bh: ops.lstm_gate_table(e, w, bi, bh),
                          lambda e, w, bi, bh: e.new_empty(e.shape[0], 2, w.shape[1]))
bilstm_train = _define("bilstm_train(Tensor x, Tensor w_ih, Tensor w_hh, Tensor b_ih, Tensor b_hh, int[] cap_lens) -> "
                       "(Tensor, Tensor, Tensor)",
                       lambda x, wi, wh, bi, bh, lens: ops.bilstm_train_fwd(x, list(lens), wi, wh, bi, bh),
                       lambda x, wi, wh, bi, bh, lens: (x.new_empty(x.shape[0], 2 * wh.shape[2], x.shape[1]),
                                                        x.new_empty(x.shape[0], 2 * wh.shape[2]),
                                                        x.new_empty(x.shape[0], x.shape[1], 2, 5, wh.shape[2])))
bilstm_bwd = _define("bilstm_bwd(int[] cap_lens, Tensor w_hh, Tensor acts, Tensor words, Tensor d_words, Tensor? d_sent) -> "
                     "(Tensor, Tensor, Tensor)",
                     lambda lens, wh, acts, words, dw, ds: ops.bilstm_bwd(list(lens), wh, acts, words, dw, ds),
                     lambda lens, wh, acts, words, dw, ds: (acts.new_empty(acts.shape[0], acts.shape[1], 2, 4 * acts.shape[4]),
                                                            acts.new_empty(acts.shape[0], acts.shape[1], 2, acts.shape[4]),
                                                            acts.new_empty(2, 4 * acts.shape[4])))
bigru_train = _define("bigru_train(Tensor x, Tensor w_ih, Tensor w_hh, Tensor b_ih, Tensor b_hh, int[] cap_lens) -> "
                      "(Tensor, Tensor, Tensor)",
                      lambda x, wi, wh, bi, bh, lens: ops.bigru_train_fwd(x, list(lens), wi, wh, bi, bh),
                      lambda x, wi, wh, bi, bh, lens: (x.new_empty(x.shape[0], 2 * wh.shape[2], x.shape[1]),
                                                       x.new_empty(x.shape[0], 2 * wh.shape[2]),
                                                       x.new_empty(x.shape[0], x.shape[1], 2, 4, wh.shape[2])))
bigru_bwd = _define("bigru_bwd(int[] cap_lens, Tensor w_hh, Tensor acts, Tensor words, Tensor d_words, Tensor? d_sent) -> "
                    "(Tensor, Tensor, Tensor, Tensor)",
                    lambda lens, wh, acts, words, dw, ds: ops.bigru_bwd(list(lens), wh, acts, words, dw, ds),
                    lambda lens, wh, acts, words, dw, ds: (acts.new_empty(acts.shape[0], acts.shape[1], 2, 3 * acts.shape[4]),
                                                           acts.new_empty(acts.shape[0], acts.shape[1], 2, 3 * acts.shape[4]),
                                                           acts.new_empty(acts.shape[0], acts.shape[1], 2, acts.shape[4]),
                                                           acts.new_empty(2, 2, 3 * acts.shape[4])))


# ------------------------------------------------------------------------------------------------ DAMSM
damsm_words = _define("damsm_words(Tensor img_features, Tensor words_emb, int[] cap_lens, float gamma1, float gamma2) -> "
                      "(Tensor, Tensor)",
                      lambda f, w, lens, g1, g2: ops.damsm_words_similarity(f, w, list(lens), g1, g2, need_att=True),
                      lambda f, w, lens, g1, g2: (f.new_empty(f.shape[0], f.shape[0]),
                                                  f.new_empty(f.shape[0], w.shape[2], f.shape[2], f.shape[3])))
damsm_words_bwd = _define("damsm_words_bwd(Tensor img_features, Tensor words_emb, int[] cap_lens, float gamma1, float gamma2, "
                          "Tensor grad_sim) -> (Tensor, Tensor)",
                          lambda f, w, lens, g1, g2, gs: ops.damsm_words_bwd(f, w, list(lens), g1, g2, gs),
                          lambda f, w, lens, g1, g2, gs: (torch.empty_like(f), torch.empty_like(w)))
func_attention = _define("func_attention(Tensor query, Tensor context, float gamma1) -> (Tensor, Tensor)",
                         lambda q, c, g1: ops.func_attention(q, c, g1),
                         lambda q, c, g1: (torch.empty_like(q), q.new_empty(q.shape[0], q.shape[2], c.shape[2], c.shape[3])))


# ------------------------------------------------------------------------------------------------ CA_NET (inference)
def _ca_net(sent_emb, w, b, ncf, eps):
    c, mu, logvar = ops.ca_net(sent_emb, w, b, ncf, eps)
    return (c if c is not None else mu.new_empty(0)), mu, logvar


ca_net = _define("ca_net(Tensor sent_emb, Tensor w, Tensor b, int ncf, Tensor? eps) -> (Tensor, Tensor, Tensor)", _ca_net,
                 lambda s_, w, b, ncf, eps: (s_.new_empty(s_.shape[0], ncf) if eps is not None else s_.new_empty(0),
                                             s_.new_empty(s_.shape[0], ncf), s_.new_empty(s_.shape[0], ncf)))
text_tail = _define("text_tail(Tensor words, Tensor[] w_ctxs, Tensor sent_emb, Tensor ca_w, Tensor ca_b, int ncf, Tensor captions) "
                    "-> (Tensor, Tensor, Tensor, Tensor)",
                    lambda words, ws, sent, cw, cb, ncf, cap: ops.text_tail(words, list(ws), sent, cw, cb, ncf, cap),
                    lambda words, ws, sent, cw, cb, ncf, cap:
                    (words.new_empty(len(ws), words.shape[0], ws[0].shape[0], 32), words.new_empty(words.shape[0], ncf),
                     words.new_empty(words.shape[0], ncf),
                     words.new_empty(words.shape[0], words.shape[2], dtype=torch.uint8)))
# ... and, for the reduced-precision generators, the `att_pack` their attention-fused kernels read (bool bf16: else f16)
text_tail_lp = _define("text_tail_lp(Tensor words, Tensor[] w_ctxs, Tensor sent_emb, Tensor ca_w, Tensor ca_b, int ncf, "
                       "Tensor captions, bool bf16) -> (Tensor, Tensor, Tensor, Tensor, Tensor)",
                       lambda words, ws, sent, cw, cb, ncf, cap, bf16:
                       ops.text_tail(words, list(ws), sent, cw, cb, ncf, cap,
                                     lp_dtype=torch.bfloat16 if bf16 else torch.float16),
                       lambda words, ws, sent, cw, cb, ncf, cap, bf16:
                       (words.new_empty(len(ws), words.shape[0], ws[0].shape[0], 32), words.new_empty(words.shape[0], ncf),
                        words.new_empty(words.shape[0], ncf),
                        words.new_empty(words.shape[0], words.shape[2], dtype=torch.uint8),
                        words.new_empty(len(ws) * words.shape[0] * 4096 + 4 * words.shape[0], dtype=torch.uint8)))
axpy_images = _define("axpy_images(Tensor[] ts, Tensor[] ss, float alpha) -> Tensor[]",
                      lambda ts, ss, alpha: ops.axpy_images(list(ts), list(ss), alpha), lambda ts, ss, alpha: [torch.empty_like(t) for t in ts])
# ------------------------------------------------------------------------------------------------ CNN_ENCODER's frozen trunk
gconv_pack = _define("gconv_pack(Tensor w, Tensor? scale, bool dgrad) -> Tensor", lambda w, sc, dg: ops.gconv_pack(w, sc, dg),
                     lambda w, sc, dg: w.new_empty((w.shape[1], w.shape[0] * w.shape[2] * w.shape[3]) if dg else
                                                   (w.shape[0], w.shape[1] * w.shape[2] * w.shape[3])))
gconv = _define("gconv(bool dgrad, Tensor A, Tensor S, int s_coff, int s_ch, Tensor(a!) out, int o_coff, int kh, int kw, int stride, "
                "int padh, int padw, Tensor? bias, bool relu, bool accumulate, Tensor(b!)? ws, Tensor? mask) -> ()",
                lambda dg, A, S, sc, sch, out, oc, kh, kw, st, ph, pw, bias, relu, acc, ws, mask:
                ops.gconv(dg, A, S, sc, sch, out, oc, kh, kw, st, ph, pw, bias, relu, acc, ws, mask), lambda *a: None)
maxpool3s2 = _define("maxpool3s2(Tensor x, Tensor(a!) out, int o_coff) -> ()", lambda x, out, oc: ops.maxpool3s2(x, out, oc),
                     lambda *a: None)
maxpool3s2_bwd = _define("maxpool3s2_bwd(Tensor x, Tensor dy, int dy_coff, Tensor(a!) dx, bool accumulate, Tensor? mask) -> ()",
                         lambda x, dy, dc, dx, acc, mask: ops.maxpool3s2_bwd(x, dy, dc, dx, acc, mask), lambda *a: None)
avgpool3 = _define("avgpool3(Tensor x, Tensor(a!) out, bool accumulate, Tensor? mask) -> ()",
                   lambda x, out, acc, mask: ops.avgpool3(x, out, acc, mask), lambda *a: None)
plane_mean = _define("plane_mean(Tensor x) -> Tensor", lambda x: ops.plane_mean(x), lambda x: x.new_empty(x.shape[0], x.shape[1]))
plane_mean_bwd = _define("plane_mean_bwd(Tensor dy, int H, int W) -> Tensor", lambda dy, H, W: ops.plane_mean_bwd(dy, H, W),
                         lambda dy, H, W: dy.new_empty(dy.shape[0], dy.shape[1], H, W))
relu_mask_ = _define("relu_mask_(Tensor(a!) dy, Tensor y, int coff, int ch) -> ()", lambda dy, y, c, ch: ops.relu_mask_(dy, y, c, ch),
                     lambda *a: None)
bilinear = _define("bilinear(Tensor x, int OH, int OW) -> Tensor", lambda x, OH, OW: ops.bilinear(x, OH, OW),
                   lambda x, OH, OW: x.new_empty(x.shape[0], x.shape[1], OH, OW))
sum_stack = _define("sum_stack(Tensor stack, int n, Tensor(a!) out) -> ()", lambda st, n, out: ops.sum_stack(st, n, out), lambda *a: None)
interleave2x2_ = _define("interleave2x2_(Tensor t00, Tensor t01, Tensor t10, Tensor t11, Tensor(a!) dx, bool accumulate, Tensor? mask) -> ()",
                         lambda a, b, c, d, dx, acc, m: ops.interleave2x2((a, b, c, d), dx, acc, m), lambda *a: None)
bilinear_bwd = _define("bilinear_bwd(Tensor dy, int H, int W) -> Tensor", lambda dy, H, W: ops.bilinear_bwd(dy, H, W),
                       lambda dy, H, W: dy.new_empty(dy.shape[0], dy.shape[1], H, W))
adam_flat_ = _define("adam_flat_(Tensor(a!) param, Tensor grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor(d!) state, float lr, "
                     "float beta1, float beta2, float eps, float weight_decay, bool advance) -> ()",
                     lambda p, g, m, v, st, lr, b1, b2, eps, wd, adv: ops.adam_flat(p, g, m, v, st, lr, b1, b2, eps, wd, adv),
                     lambda *a: None)
weighted_bce = _define("weighted_bce(Tensor a, Tensor? b, Tensor target, Tensor weight) -> Tensor",
                       lambda a, b, t, w: ops.weighted_bce(a, b, t, w), lambda a, b, t, w: a.new_empty(()))


def _weighted_bce_bwd(dy, a, b, t, w):
    da, db = ops.weighted_bce_bwd(dy, a, b, t, w)
    return da, (db if db is not None else a.new_empty(0))


weighted_bce_bwd = _define("weighted_bce_bwd(Tensor dy, Tensor a, Tensor? b, Tensor target, Tensor weight) -> (Tensor, Tensor)",
                           _weighted_bce_bwd, lambda dy, a, b, t, w: (torch.empty_like(a), torch.empty_like(b) if b is not None else a.new_empty(0)))


def _weighted_bce_backward(ctx, dy):
    a, b, t, w = ctx.saved_tensors if ctx.has_b else (ctx.saved_tensors[0], None, ctx.saved_tensors[1], ctx.saved_tensors[2])
    da, db = weighted_bce_bwd(dy.contiguous(), a, b, t, w)
    return (da if ctx.needs_input_grad[0] else None), (db if ctx.has_b and ctx.needs_input_grad[1] else None), None, None


def _weighted_bce_setup(ctx, inputs, output):
    a, b, t, w = inputs
    ctx.has_b = b is not None
    if b is not None:
        ctx.save_for_backward(a, b, t, w)
    else:
        ctx.save_for_backward(a, t, w)


torch.library.register_autograd("tgsr::weighted_bce", _weighted_bce_backward, setup_context=_weighted_bce_setup, lib=_lib)
axpy_map = _define("axpy_map(Tensor t, Tensor s, Tensor amap) -> Tensor", lambda t, s, a: ops.axpy_map(t, s, a),
                   lambda t, s, a: torch.empty_like(t))


def _axpy_map_bwd(dy, s, amap, need_ds, need_da):
    ds, da = ops.axpy_map_bwd(dy, s, amap, need_ds, need_da)
    return (ds if ds is not None else dy.new_empty(0)), (da if da is not None else dy.new_empty(0))


axpy_map_bwd = _define("axpy_map_bwd(Tensor dy, Tensor s, Tensor amap, bool need_ds, bool need_da) -> (Tensor, Tensor)", _axpy_map_bwd,
                       lambda dy, s, a, nds, nda: (torch.empty_like(dy) if nds else dy.new_empty(0),
                                                   torch.empty_like(a) if nda else dy.new_empty(0)))


def _axpy_map_backward(ctx, dy):
    s, amap = ctx.saved_tensors
    nds, nda = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
    ds = da = None
    if nds or nda:
        ds, da = axpy_map_bwd(dy.contiguous(), s, amap, nds, nda)
    return (dy if ctx.needs_input_grad[0] else None), (ds if nds else None), (da if nda else None)


torch.library.register_autograd("tgsr::axpy_map", _axpy_map_backward,
                                setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[1], inputs[2]), lib=_lib)
affine_act = _define("affine_act(Tensor raw, Tensor scale, Tensor shift, int act) -> Tensor",
                     lambda raw, sc, sh, act: ops.affine_act(raw, sc, sh, act), lambda raw, sc, sh, act: torch.empty_like(raw))
affine_act_bwd = _define("affine_act_bwd(Tensor dy, Tensor? out, Tensor scale, int act) -> Tensor",
                         lambda dy, out, sc, act: ops.affine_act_bwd(dy, out, sc, act), lambda dy, out, sc, act: torch.empty_like(dy))
multi_copy = _define("multi_copy(Tensor(a!)[] dsts, Tensor[] srcs) -> ()",
                     lambda dsts, srcs: ops.multi_copy(list(dsts), list(srcs)), lambda dsts, srcs: None)
to_uint8 = _define("to_uint8(Tensor img) -> Tensor", lambda x: ops.to_uint8(x), lambda x: torch.empty_like(x, dtype=torch.uint8))


# ================================================================================================ reduced-precision path
# (lp images are mutable arguments: every kernel writes a channel slice of a caller-owned zero-bordered image)
def _lp():
    from . import lp
    return lp


lp_conv3x3 = _define("lp_conv3x3(Tensor x, Tensor wpack, int cin, int cout, Tensor? scale, Tensor? shift, bool glu, bool upsample, "
                     "Tensor? residual, int res_coff, Tensor(a!) out, int out_coff) -> ()",
                     lambda x, wp, cin, cout, s, t, glu, up, res, rco, out, oco:
                     (_lp().conv3x3(x, wp, cin, cout, s, t, glu=glu, upsample=up, residual=res, res_coff=rco, out=out, out_coff=oco),
                      None)[1], lambda *a: None)
lp_resblocks = _define("lp_resblocks(Tensor x, Tensor[] wpacks, Tensor[] scales, Tensor[] shifts, Tensor(a!) tmp, Tensor(b!) a, "
                       "Tensor(c!) b, Tensor(d!) flags) -> ()",
                       lambda x, wp, sc, sh, tmp, a, b, flags: (_lp().resblocks(x, list(wp), list(sc), list(sh), tmp, a, b, flags), None)[1],
                       lambda *a: None)
lp_upconv_glu = _define("lp_upconv_glu(Tensor x, Tensor wpack, int cin, int cout, Tensor? scale, Tensor? shift, Tensor(a!) out, "
                        "int out_coff) -> ()",
                        lambda x, wp, cin, cout, s, t, out, oco: (_lp().upconv_glu(x, wp, cin, cout, s, t, out=out, out_coff=oco), None)[1],
                        lambda *a: None)
lp_upconv_glu_head = _define("lp_upconv_glu_head(Tensor x, Tensor wpack, int cin, int cout, Tensor? scale, Tensor? shift, "
                             "Tensor head_wpack, int K, Tensor(a!) partial, Tensor(b!)? out, int out_coff) -> ()",
                             lambda x, wp, cin, cout, s, t, hw, K, part, out, oco:
                             (_lp().upconv_glu_head(x, wp, cin, cout, s, t, hw, K, partial=part, out=out, out_coff=oco,
                                                    write_out=out is not None), None)[1], lambda *a: None)
lp_stem = _define("lp_stem(Tensor x, Tensor w, Tensor scale, Tensor shift, Tensor(a!) out, int out_coff) -> ()",
                  lambda x, w, s, t, out, oco: (_lp().stem(x, w, s, t, out=out, out_coff=oco), None)[1], lambda *a: None)



def _att(pack, nsets, index, T, use_mask, correct_mask, c_coff, attn):
    return _lp().AttFuse(pack, nsets, index, T, use_mask, correct_mask, c_coff, attn)


# the producers of h that also attend to the words for the pixels they have just computed (GlobalAttention.py:87-130 at
# util.py:768-771 / 814-817): att_pack = text_tail_lp's fifth output, index = which projection the stage attends through
_ATT_SCHEMA = "Tensor att_pack, int nsets, int index, int T, bool use_mask, bool correct_mask, int c_coff"
lp_stem_att = _define("lp_stem_att(Tensor x, Tensor w, Tensor scale, Tensor shift, Tensor(a!) out, int out_coff, " + _ATT_SCHEMA +
                      ", Tensor(b!)? attn) -> ()",
                      lambda x, w, s, t, out, oco, pack, ns, ix, T, um, cm, cco, attn:
                      (_lp().stem(x, w, s, t, out=out, out_coff=oco, att=_att(pack, ns, ix, T, um, cm, cco, attn)), None)[1],
                      lambda *a: None)
lp_upconv_glu_att = _define("lp_upconv_glu_att(Tensor x, Tensor wpack, int cin, int cout, Tensor? scale, Tensor? shift, "
                            "Tensor(a!) out, int out_coff, " + _ATT_SCHEMA + ", Tensor(b!)? attn) -> ()",
                            lambda x, wp, cin, cout, s, t, out, oco, pack, ns, ix, T, um, cm, cco, attn:
                            (_lp().upconv_glu(x, wp, cin, cout, s, t, out=out, out_coff=oco,
                                              att=_att(pack, ns, ix, T, um, cm, cco, attn)), None)[1], lambda *a: None)
lp_upconv_glu_head_att = _define("lp_upconv_glu_head_att(Tensor x, Tensor wpack, int cin, int cout, Tensor? scale, Tensor? shift, "
                                 "Tensor head_wpack, int K, Tensor(a!) partial, Tensor(b!) out, int out_coff, " + _ATT_SCHEMA +
                                 ", Tensor(c!)? attn) -> ()",
                                 lambda x, wp, cin, cout, s, t, hw, K, part, out, oco, pack, ns, ix, T, um, cm, cco, attn:
                                 (_lp().upconv_glu_head(x, wp, cin, cout, s, t, hw, K, partial=part, out=out, out_coff=oco,
                                                        att=_att(pack, ns, ix, T, um, cm, cco, attn)), None)[1],
                                 lambda *a: None)
lp_convert = _define("lp_convert(Tensor src, Tensor(a!) out) -> ()", lambda src, out: (_lp().convert(src, out), None)[1],
                     lambda *a: None)
lp_conv_to3 = _define("lp_conv_to3(Tensor x, Tensor wpack, int K, bool tanh_axpy, Tensor? addend, float alpha) -> Tensor",
                      lambda x, wp, K, act, add, alpha: _lp().conv_to3(x, wp, K, tanh_axpy=act, addend=add, alpha=alpha),
                      lambda x, wp, K, act, add, alpha: x.new_empty(x.shape[0], 3, x.shape[1] - 2, x.shape[2] - 2, dtype=torch.float32))
lp_word_attention = _define("lp_word_attention(Tensor(a!) h_img, Tensor src, Tensor? mask, int T, bool correct_mask, int c_coff) -> Tensor",
                            lambda h, src, mask, T, cm, cco: _lp().word_attention(h, src, mask, T, correct_mask=cm, c_coff=cco),
                            lambda h, src, mask, T, cm, cco: h.new_empty(h.shape[0], T, h.shape[1] - 2, h.shape[2] - 2, dtype=torch.float32))


def _lp_head_combine(sizes_h, sizes_w, partial_low, partial_high, low, high, low_tanh, alpha):
    n = len(sizes_h)
    none = lambda t: None if (t is None or t.numel() == 0) else t          # noqa: E731  (an empty tensor stands for "absent")
    _lp().head_combine(low[0].shape[0], list(zip(sizes_h, sizes_w)), [none(t) for t in partial_low[:n]],
                       [none(t) for t in partial_high[:n]], list(low[:n]), [none(t) for t in high[:n]], low_tanh, alpha)


lp_head_combine = _define("lp_head_combine(int[] H, int[] W, Tensor[] partial_low, Tensor[] partial_high, Tensor(a!)[] low, "
                          "Tensor(b!)[] high, bool low_tanh, float alpha) -> ()", _lp_head_combine, lambda *a: None)
