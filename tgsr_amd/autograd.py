"""Training path: torch.autograd.Function wrappers whose forward AND backward are HIP kernels.

The reference trains through torch autograd over nn.Conv2d / nn.BatchNorm2d(train) / GLU / nn.Upsample
(util.py:74-80, 110-130).  Here one Function = one fused block:
    ConvBnAct: [nearest x2 ->] conv3x3 -> BatchNorm2d(batch statistics) -> GLU | (+ residual)
        forward : tgsr_wino_conv3x3_fwd | tgsr_conv3x3_fwd (raw) -> tgsr_bn_train_fwd (stats, running update,
                  normalise + GLU/residual)
        backward: tgsr_bn_train_bwd (GLU', BN') -> data gradient = the same conv kernels on flipped/transposed
                  weights (+ tgsr_sumpool2x2 through the up-sample) and tgsr_conv3x3_wgrad
"""
import torch

from . import custom_ops as C
from . import ops
from ._lib import TgsrError


def _dgrad_weight(w: torch.Tensor) -> torch.Tensor:
    """conv_transpose of a stride-1 pad-1 3x3 conv = the same conv with the kernel flipped and in/out swapped."""
    return w.flip(2, 3).transpose(0, 1).contiguous()


# Weight gradients are leaves of the backward graph (nothing downstream waits for them until the optimizer): while a side
# stream is registered here they run beside the data-gradient chain instead of inside it.  Only train.SRTrainer registers
# one, for the duration of a step's backward, and joins it before anything reads the gradients; modules driven directly
# (`loss.backward()` then `p.grad`) stay on one stream.
WGRAD_SIDE = {}          # device index -> torch.cuda.Stream
# (parameter, data_ptr of its bucket slot) of every weight gradient issued on a side stream: `check_adopted` verifies after
# the join that autograd really took those tensors as `.grad` (had it cloned or added instead, that kernel would have run
# on the main stream while the side stream was still writing the slot)
_ADOPTED = []


def check_adopted():
    """Called AFTER the side stream has been joined (SRTrainer._wgrad_side): every gradient computed there must BE the
    parameter's `.grad`.  It detects after the fact - a failure invalidates the step's gradients (the caller zeroes the bucket)."""
    bad = [p for p, ptr in _ADOPTED if p.grad is None or p.grad.data_ptr() != ptr]
    _ADOPTED.clear()
    if bad:
        raise TgsrError("%d weight gradients were computed on the side stream but autograd did not adopt them in place "
                        "(it cloned or accumulated on the main stream while they were in flight): set TGSR_WGRAD_SIDE=0"
                        % len(bad))


def wgrad_stream(dev):
    return WGRAD_SIDE.get(dev.index if dev.index is not None else torch.cuda.current_device())


def on_wgrad_stream(dev, inputs, fn, adopted):
    """Run `fn` (a weight-gradient launch reading `inputs`) on the registered side stream, ordered after everything
    already queued on the current stream; without a registered stream: just run it.  `adopted`: the gradient goes
    straight into the parameter's bucket slot and autograd adopts it without a kernel - only then may it still be in
    flight when backward() moves on.  A gradient autograd will ADD to an existing .grad (a parameter used twice: the tied
    stages of models16, the discriminators' real + fake passes) is computed on the current stream, after the side stream
    has drained, because that add runs on the current stream."""
    side = wgrad_stream(dev)
    if side is None:
        return fn()
    if not adopted:
        torch.cuda.current_stream(dev).wait_stream(side)
        return fn()
    side.wait_stream(torch.cuda.current_stream(dev))
    # Also inside a graph capture: a block with a recorded side-stream use is not handed out again until the capture has ended
    # (the allocator defers its end-of-life events), so the next main-stream allocation of the capture cannot alias a tensor
    # the side branch of the graph is still reading.
    for t in inputs:
        if t is not None:
            t.record_stream(side)
    with torch.cuda.stream(side):
        return fn()


def _grad_out(param, shape, dev):
    """Where a parameter gradient is written: the parameter's region of its flat gradient bucket when that is open
    (parallel.grad_slot: no accumulation kernel afterwards), else a fresh tensor."""
    from .parallel import grad_slot
    slot = grad_slot(param) if param is not None else None
    return slot if slot is not None else torch.empty(tuple(shape), dtype=torch.float32, device=dev)


class PackCache:
    """The packed forms (Winograd U, direct-kernel stream order, up-sample-aware Winograd taps; forward and data-gradient
    variants) of the generator's conv weights across training steps.  The parameters change once per step, in the
    optimizer: `repack()` - called right after it - rewrites every pack seen so far IN PLACE on a stream of its own, so the
    ~70 pack launches of a step (4-5 us each, each in front of the conv that needs it) leave the critical stream and run
    under the EMA update and the text encoder of the next step.  `get` serves a pack while the weight's version counter
    still equals the one it was made from (a load_state_dict or any other in-place write makes it re-pack on the spot)."""

    def __init__(self):
        self.entries = {}            # (data_ptr, shape, kind) -> [weight alias, kind, packed, version]
        self.stream = None
        self.event = None

    @staticmethod
    def _pack(weight, kind, out):
        if kind in ("wino4", "wino4_dgrad"):
            return ops.pack_wino4_weight(weight, False, kind == "wino4_dgrad", out=out)
        if kind in ("wino4w", "wino4w_dgrad"):
            return ops.pack_wino4w_weight(weight, False, kind == "wino4w_dgrad", out=out)
        if kind == "wino":
            return ops.pack_wino_weight(weight, False, False, out=out)
        if kind == "wino_dgrad":
            return ops.pack_wino_weight(weight, False, True, out=out)
        if kind == "upwino":
            return ops.pack_upwino_weight(weight, False, out=out)
        return ops.pack_conv3x3_weight(weight, kind == "direct_dgrad", out=out)

    def get(self, weight, kind):
        if self.event is not None:                       # first use after a repack: this stream waits for it once
            torch.cuda.current_stream(weight.device).wait_event(self.event)
            self.event = None
        key = (weight.data_ptr(), tuple(weight.shape), kind)
        e = self.entries.get(key)
        if e is not None and e[3] == weight._version:
            return e[2]
        out = self._pack(weight, kind, e[2] if e is not None else None)
        self.entries[key] = [weight.detach(), kind, out, weight._version]
        return out

    def get_captured(self, weight, kind):
        """Inside a hipGraph capture of a train step (train.SRTrainer): the cached pack, without a launch - the captured
        optimizer segment rewrites every entry in place at the end of each replay (`repack_captured`), so whatever step is
        replayed next finds the packs of the current weights at these addresses.  None when the (weight, kind) pair was never
        packed by an eager step: the caller then packs inside the graph."""
        e = self.entries.get((weight.data_ptr(), tuple(weight.shape), kind))
        return None if e is None else e[2]

    def repack_captured(self, side):
        """`repack(force=True)` as nodes of the graph being captured on the current stream: the pack launches fork onto `side`
        (a branch beside the EMA update) - the caller joins it (`current.wait_stream(side)`) before the capture ends."""
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for e in self.entries.values():
                self._pack(e[0], e[1], e[2])

    def mark_fresh(self):
        """After a replay whose graph re-packed every entry: the entries match the weights' current version counters."""
        for e in self.entries.values():
            e[3] = e[0]._version

    def settle(self, dev):
        """Make the current stream wait for a pending eager repack (before a replay that reads the packs)."""
        if self.event is not None:
            torch.cuda.current_stream(dev).wait_event(self.event)
            self.event = None

    def repack(self, force=False):
        """Re-derive every cached pack from the current weights on the pack stream (ordered behind everything queued on
        the current stream: the optimizer step, and the previous backward's reads of the old packs).  `force`: whatever the
        version counters say - the fused Adam kernel (`torch._fused_adam_`) updates the parameters WITHOUT bumping them, and a
        cache that trusted the counters served the previous step's weights (caught by test_pack_cache_changes_nothing_over_steps)."""
        if not self.entries:
            return
        dev = next(iter(self.entries.values()))[0].device
        with torch.cuda.device(dev):
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=dev)
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.stream), torch.no_grad():
                for e in self.entries.values():
                    if force or e[3] != e[0]._version:
                        self._pack(e[0], e[1], e[2])
                        e[3] = e[0]._version
            self.event = torch.cuda.Event()
            self.event.record(self.stream)


import os as _os
BN_STATS_IN_CONV = _os.environ.get("TGSR_BN_STATS_IN_CONV", "1") != "0"
_PACKS = None        # the PackCache of the trainer whose step is running (train.SRTrainer sets it), else packs are per call


def _packed(weight, kind):
    if _PACKS is not None and weight.is_cuda:
        if not torch.cuda.is_current_stream_capturing():
            return _PACKS.get(weight, kind)
        hit = _PACKS.get_captured(weight, kind)
        if hit is not None:
            return hit
    if kind in ("wino4", "wino4_dgrad"):
        return C.pack_wino4_weight(weight, False, kind == "wino4_dgrad")
    if kind in ("wino4w", "wino4w_dgrad"):
        return C.pack_wino4w_weight(weight, False, kind == "wino4w_dgrad")
    if kind in ("wino", "wino_dgrad"):
        return C.pack_wino_weight(weight, False, kind == "wino_dgrad")
    if kind == "upwino":
        return C.pack_upwino_weight(weight, False)
    return C.pack_conv3x3_weight(weight, kind == "direct_dgrad")


def _conv_raw(x: torch.Tensor, weight: torch.Tensor, upsample: bool = False, dgrad: bool = False,
              residual: torch.Tensor = None) -> torch.Tensor:
    """conv3x3 without affine / activation (what BatchNorm's batch statistics are taken of, and the data gradient):
    the Winograd kernel where it applies (no up-sampling, Cout % 64 == 0, Cin % 4 == 0), else the direct kernel.
    The weights change every step: they are packed per call (a few microseconds each), or - inside a trainer's step -
    served from its PackCache, which re-packs them all behind the optimizer on a stream of its own.  `dgrad`: `weight` is the forward
    layer's [Cin_of_x... = weight.shape[0]] filter and the conv applied is its transpose (pack kernels read it
    transposed and flipped: no flip / transpose / copy kernels)."""
    from . import util
    Cout = weight.shape[1] if dgrad else weight.shape[0]
    if util.WINOGRAD and not upsample and util._wino4_takes(x, Cout, None, residual):
        # the large layers: F(4x4, 3x3), forward and data gradient alike (tgsr_winograd4.hip); the register-fed form where the
        # convolution has an even number of 4-channel stages
        if x.shape[1] % 8 == 0:
            return C.conv3x3_wino4w(x, _packed(weight, "wino4w_dgrad" if dgrad else "wino4w"), Cout, None, None, False, residual)
        return C.conv3x3_wino4(x, _packed(weight, "wino4_dgrad" if dgrad else "wino4"), Cout, None, None, False, residual)
    if util.WINOGRAD and not upsample and util._wino_pays(x, Cout, None, None):
        return C.conv3x3_wino(x, _packed(weight, "wino_dgrad" if dgrad else "wino"), Cout, None, None, False, residual)
    if util.WINOGRAD and upsample and ops.upwino_supported(x, Cout):
        assert not dgrad and residual is None
        return C.upwino(x, _packed(weight, "upwino"), Cout, None, None, False)
    return C.conv3x3_fused(x, _packed(weight, "direct_dgrad" if dgrad else "direct"), Cout, None, None, False, upsample, residual)


def _cba_forward(x, weight, gamma, beta, running_mean, running_var, residual, glu, upsample, momentum, eps, nbt):
    """conv3x3 (raw) -> BatchNorm batch statistics -> normalise (+ GLU | + residual).  Returns (out, raw, stats)."""
    from . import util
    w = weight.detach()
    if BN_STATS_IN_CONV and util.WINOGRAD and not upsample and util._wino4_takes(x, w.shape[0], None, None):
        raw, part = (C.conv3x3_wino4w_stats(x, _packed(w, "wino4w"), w.shape[0]) if x.shape[1] % 8 == 0 else
                     C.conv3x3_wino4_stats(x, _packed(w, "wino4"), w.shape[0]))
        out, stats = C.bn_train_fwd_from_stats(raw, gamma.detach(), beta.detach(), float(eps), float(momentum), running_mean,
                                               running_var, 1 if glu else 0, residual, nbt, part)
        return out, raw, stats
    if BN_STATS_IN_CONV and util.WINOGRAD and not upsample and util._wino_pays(x, w.shape[0], None, None):
        # BatchNorm's statistics pass rides the convolution's epilogue: one (sum, sumsq) pair per channel and wave tile
        raw, part = C.conv3x3_wino_stats(x, _packed(w, "wino"), w.shape[0])
        out, stats = C.bn_train_fwd_from_stats(raw, gamma.detach(), beta.detach(), float(eps), float(momentum), running_mean,
                                               running_var, 1 if glu else 0, residual, nbt, part)
        return out, raw, stats
    raw = _conv_raw(x, w, upsample)
    out, stats = C.bn_train_fwd(raw, gamma.detach(), beta.detach(), float(eps), float(momentum), running_mean, running_var,
                                1 if glu else 0, residual, nbt)
    return out, raw, stats


def _cba_backward(x, weight, raw, stats, gamma, beta, glu, upsample, dout, need_dx, need_dw, dx_addend=None):
    """Backward of _cba_forward: (dx, dw, dgamma, dbeta).  `dx_addend` (the gradient arriving at x through a skip
    connection) is added in the data-gradient conv's epilogue instead of by a separate elementwise kernel."""
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    dev = x.device
    dgamma, dbeta = _grad_out(gamma, (Cout,), dev), _grad_out(beta, (Cout,), dev)
    draw = C.bn_train_bwd(dout.contiguous(), raw, stats, 1 if glu else 0, dgamma, dbeta, None)
    dx = dw = None
    if need_dx:
        cpad = (Cin + 31) // 32 * 32                                  # the kernel tiles 32 output channels
        if cpad != Cin:                                               # stem convs (Cin = 3): zero-padded rows
            wT = _dgrad_weight(weight.detach())                       # [Cin, Cout, 3, 3]
            wT = torch.cat((wT, wT.new_zeros(cpad - Cin, Cout, 3, 3)), 0)
            dxu = _conv_raw(draw, wT)[:, :Cin].contiguous()           # [B, Cin, Ho, Wo]
            if dx_addend is not None:
                dxu = dxu + dx_addend
        else:
            dxu = _conv_raw(draw, weight.detach(), dgrad=True, residual=None if upsample else dx_addend)
        if upsample:
            dx = C.sumpool2x2(dxu)
            if dx_addend is not None:
                dx = dx + dx_addend
        else:
            dx = dxu
    if need_dw:
        from . import util
        dw = _grad_out(weight, weight.shape, dev)
        # a view of the flat bucket (parallel.grad_slot) that autograd's AccumulateGrad will take as `.grad` without
        # launching anything: only when no gradient sits there yet and no graph of the backward is being recorded
        adopted = dw._base is not None and weight.grad is None and not torch.is_grad_enabled()
        if adopted and wgrad_stream(dev) is not None:
            _ADOPTED.append((weight, dw.data_ptr()))
        # upBlock: 9 Winograd positions on the low-resolution pixels (4x fewer multiplies than 9 taps on the up-sampled
        # grid); plain conv: 16 positions per 2x2 output tile (2.25x fewer); else the direct 9-tap form (ops.conv3x3_wgrad)
        on_wgrad_stream(dev, (draw, x), lambda: C.conv3x3_wgrad(draw, x, upsample, util.WINOGRAD, dw), adopted)
    return dx, dw, dgamma, dbeta


class ConvBnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, residual, glu, upsample, momentum, eps, nbt=None):
        x = x.contiguous()
        out, raw, stats = _cba_forward(x, weight, gamma, beta, running_mean, running_var, residual, glu, upsample, momentum,
                                       eps, nbt)
        ctx.save_for_backward(x, weight, raw, stats)
        ctx.cfg = (glu, upsample, residual is not None)
        ctx.bn_params = (gamma, beta)          # only to look up their gradient slots (parallel.grad_slot) in backward
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, raw, stats = ctx.saved_tensors
        glu, upsample, has_res = ctx.cfg
        gamma, beta = ctx.bn_params
        dx, dw, dgamma, dbeta = _cba_backward(x, weight, raw, stats, gamma, beta, glu, upsample, dout,
                                              ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        dres = dout if has_res else None
        return dx, dw, dgamma, dbeta, None, None, dres, None, None, None, None, None


class ResBlockFn(torch.autograd.Function):
    """ResBlock (util.py:110-130): x + BN(conv(GLU(BN(conv(x))))) as ONE autograd node, so that the skip connection's
    gradient is added to the first conv's data gradient in that conv kernel's epilogue (its `residual` input) instead of
    by an elementwise add per block (10 per generator step on tensors of up to 67 MB)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, nbt1, mom1, eps1, w2, g2, b2, rm2, rv2, nbt2, mom2, eps2):
        x = x.contiguous()
        y, raw1, st1 = _cba_forward(x, w1, g1, b1, rm1, rv1, None, True, False, mom1, eps1, nbt1)
        out, raw2, st2 = _cba_forward(y, w2, g2, b2, rm2, rv2, x, False, False, mom2, eps2, nbt2)
        ctx.save_for_backward(x, w1, raw1, st1, y, w2, raw2, st2)
        ctx.bn_params = (g1, b1, g2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, raw1, st1, y, w2, raw2, st2 = ctx.saved_tensors
        g1, b1, g2, b2 = ctx.bn_params
        dout = dout.contiguous()
        ni = ctx.needs_input_grad
        dy, dw2, dg2, db2 = _cba_backward(y, w2, raw2, st2, g2, b2, False, False, dout, True, ni[9])
        dx, dw1, dg1, db1 = _cba_backward(x, w1, raw1, st1, g1, b1, True, False, dy, ni[0], ni[1],
                                          dx_addend=dout if ni[0] else None)
        return dx, dw1, dg1, db1, None, None, None, None, None, dw2, dg2, db2, None, None, None, None, None


def _nbt(bn):
    """bn.num_batches_tracked when the kernel can bump it in place (int64 scalar on the device), else None."""
    t = bn.num_batches_tracked if bn.track_running_stats else None
    if t is None:
        return None
    if t.dtype != torch.int64:
        raise TgsrError("BatchNorm num_batches_tracked must be an int64 tensor")
    if not t.is_cuda:
        if bn.running_mean is not None and bn.running_mean.is_cuda:
            raise TgsrError("BatchNorm num_batches_tracked is on the CPU while the module is on the GPU")
        return None    # a CPU module is refused by the first HIP op of the block (ops._need_hip): no fallback
    return t


def conv_bn_act_train(x, conv, bn, glu=False, upsample=False, residual=None):
    """Training-mode fused block over the parameter-holder modules (conv: nn.Conv2d, bn: nn.BatchNorm2d)."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not used by the reference")
    out = ConvBnAct.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                          bn.running_var if bn.track_running_stats else None, residual, glu, upsample, bn.momentum,
                          bn.eps, _nbt(bn))
    if bn.track_running_stats:
        # tgsr_bn_train_fwd wrote the running statistics (and num_batches_tracked) through raw pointers: tell autograd's
        # version counters, which the eval-mode caches (util._FusedParams, lp_pipeline.LpExecutor) key their folded
        # affines on
        torch.autograd.graph.increment_version(bn.running_mean)
        torch.autograd.graph.increment_version(bn.running_var)
    return out


def res_block_train(x, conv1, bn1, conv2, bn2):
    """Training-mode ResBlock over its parameter-holder modules (one autograd node: ResBlockFn)."""
    for bn in (bn1, bn2):
        if bn.momentum is None:
            raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not used by the reference")
    t1, t2 = bn1.track_running_stats, bn2.track_running_stats
    out = ResBlockFn.apply(x, conv1.weight, bn1.weight, bn1.bias, bn1.running_mean if t1 else None,
                           bn1.running_var if t1 else None, _nbt(bn1), bn1.momentum, bn1.eps,
                           conv2.weight, bn2.weight, bn2.bias, bn2.running_mean if t2 else None,
                           bn2.running_var if t2 else None, _nbt(bn2), bn2.momentum, bn2.eps)
    for bn in (bn1, bn2):
        if bn.track_running_stats:
            torch.autograd.graph.increment_version(bn.running_mean)
            torch.autograd.graph.increment_version(bn.running_var)
    return out


class DownConv(torch.autograd.Function):
    """nn.Conv2d(Cin, Cout, 4, 2, 1, bias=False) [+ LeakyReLU(0.2)] - the first layer of the discriminators'
    image encoder (no BatchNorm) - forward, data and weight gradient on the gather-GEMM kernels of tgsr_down.hip."""

    @staticmethod
    def forward(ctx, x, weight, leaky):
        out = C.conv4x4s2(x, weight.detach(), leaky)
        ctx.save_for_backward(x, weight, out if leaky else None)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, out = ctx.saved_tensors
        g = dy.contiguous() if out is None else C.leaky_relu_bwd(dy.contiguous(), out)
        dx = C.conv4x4s2_dgrad(g, weight.detach(), x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = C.conv4x4s2_wgrad(g, x.contiguous()) if ctx.needs_input_grad[1] else None
        return dx, dw, None


def _dconv_raw(x, weight, kind):
    """The discriminator blocks' convolution without BatchNorm / activation: kind "down" = conv4x4 stride 2 (downBlock), "3x3"."""
    if kind == "down":
        return C.conv4x4s2(x, weight.detach(), False)
    if ops.conv3x3_gemm_pays(x.shape[1], weight.shape[0], x.shape[2], x.shape[3]):
        return C.conv3x3_gemm(x, weight.detach())          # many channels on 4x4 pixels: the implicit-GEMM form
    return C.conv3x3_fused(x, C.pack_conv3x3_weight(weight.detach(), False), weight.shape[0], None, None, False, False, None)


def _dconv_grads(x, weight, draw, kind, need_dx, need_dw):
    """(dx, dw) of `_dconv_raw` given d(raw); dw goes into the weight's gradient-bucket slot when one is open."""
    dx = dw = None
    Cin, H, W = x.shape[1], x.shape[2], x.shape[3]
    Cc, dev = draw.shape[1], x.device
    w = weight.detach()
    def side(fn):
        # the weight gradient is a leaf of the backward graph: beside the data-gradient chain when the trainer has registered a
        # side stream for this update (train.SRTrainer._d_wgrad_side) and the gradient lands in its bucket slot unaccumulated
        adopted = dw._base is not None and weight.grad is None and not torch.is_grad_enabled()
        if adopted and wgrad_stream(dev) is not None:
            _ADOPTED.append((weight, dw.data_ptr()))
        on_wgrad_stream(dev, (draw, x), fn, adopted)
    if kind == "down":
        if need_dx:
            dx = C.conv4x4s2_dgrad(draw, w, H, W)
        if need_dw:
            dw = _grad_out(weight, weight.shape, dev)
            side(lambda: C.conv4x4s2_wgrad_out(draw, x, dw))
    elif ops.conv3x3_gemm_pays(Cin, Cc, H, W):
        if need_dx:
            dx = C.conv3x3_gemm_dgrad(draw, w)
        if need_dw:
            dw = _grad_out(weight, weight.shape, dev)
            side(lambda: C.conv3x3_gemm_wgrad_out(draw, x, dw))
    else:
        if need_dx:
            wT = _dgrad_weight(w)
            cpad = (Cin + 31) // 32 * 32
            if cpad != Cin:
                wT = torch.cat((wT, wT.new_zeros(cpad - Cin, Cc, 3, 3)), 0)
            dx = C.conv3x3_fused(draw, C.pack_conv3x3_weight(wT, False), cpad, None, None, False, False, None)
            if cpad != Cin:
                dx = dx[:, :Cin].contiguous()
        if need_dw:
            dw = torch.empty_like(weight)
            C.conv3x3_wgrad(draw, x, False, False, dw)
    return dx, dw


class ConvBnLeakyEval(torch.autograd.Function):
    """downBlock / Block3x3_leakRelu under .eval() (util.py:92-98 with nn.BatchNorm2d normalising by its running statistics):
    the same convolution kernels, then tgsr::affine_act with the folded scale / shift.  Differentiable (a frozen, eval-mode
    discriminator still passes the generator's gradient through): d(raw) = tgsr::affine_act_bwd, data / weight gradient on the
    training path's kernels; the rarely wanted BatchNorm parameter gradients are two torch reductions."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, kind, eps):
        x = x.contiguous()
        raw = _dconv_raw(x, weight, kind)
        scale, shift = ops.bn_fold(gamma.detach(), beta.detach(), running_mean, running_var, eps)
        out = C.affine_act(raw, scale, shift, 2)
        ctx.save_for_backward(x, weight, raw, out, scale, running_mean, running_var)
        ctx.kind, ctx.eps = kind, float(eps)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, raw, out, scale, rm, rv = ctx.saved_tensors
        dout = dout.contiguous()
        draw = C.affine_act_bwd(dout, out, scale, 2)
        dx, dw = _dconv_grads(x, weight, draw, ctx.kind, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        dgamma = dbeta = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            g = C.affine_act_bwd(dout, out, torch.ones_like(scale), 2)                    # dy * act'
            dbeta = g.sum((0, 2, 3))
            dgamma = (g * (raw - rm.view(1, -1, 1, 1))).sum((0, 2, 3)) * torch.rsqrt(rv + ctx.eps)
        return dx, dw, dgamma, dbeta, None, None, None, None


class ConvBnLeaky(torch.autograd.Function):
    """conv -> BatchNorm2d(batch statistics) -> LeakyReLU(0.2) with `kind` = "down" (conv4x4 stride 2: downBlock,
    util.py:92-98) or "3x3" (the discriminators' Block3x3_leakRelu).  BatchNorm + activation forward / backward =
    tgsr::bn_train_fwd_out / tgsr::bn_train_bwd (act = 2); the 3x3 convolution and its gradients are the generator's
    fp32 kernels."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, kind, momentum, eps, nbt=None, groups=None):
        """`groups`: sizes of consecutive slices of the batch that are normalised with their OWN batch statistics (and
        update the running statistics one after the other, in order) - several forward passes of the module (real /
        fake / mismatched images of discriminator_loss) as one convolution launch, one data-gradient launch and one
        weight-gradient launch, numerically the passes run one by one."""
        x = x.contiguous()
        raw = _dconv_raw(x, weight, kind)
        B, Cc = raw.shape[0], raw.shape[1]
        groups = (B,) if groups is None else tuple(int(g) for g in groups)
        if sum(groups) != B or min(groups) < 1:
            raise TgsrError("ConvBnLeaky: groups %s do not partition a batch of %d" % (groups, B))
        stats = torch.empty(len(groups), 4, Cc, dtype=torch.float32, device=x.device)
        out = torch.empty_like(raw)
        o = 0
        for gi, n in enumerate(groups):
            C.bn_train_fwd_out(raw[o:o + n], gamma.detach(), beta.detach(), float(eps), float(momentum), running_mean, running_var,
                               2, nbt, out[o:o + n], stats[gi])
            o += n
        ctx.save_for_backward(x, weight, raw, stats)
        ctx.kind = kind
        ctx.groups = groups
        ctx.bn_params = (gamma, beta)          # only to look up their gradient slots (parallel.grad_slot) in backward
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, raw, stats = ctx.saved_tensors
        dout = dout.contiguous()
        B, Cc = raw.shape[0], raw.shape[1]
        dev = x.device
        draw = torch.empty_like(raw)
        gamma, beta = ctx.bn_params
        dgamma, dbeta = _grad_out(gamma, (Cc,), dev), _grad_out(beta, (Cc,), dev)
        o = 0
        for gi, n in enumerate(ctx.groups):
            dg, db = (dgamma, dbeta) if gi == 0 else (torch.empty(Cc, dtype=torch.float32, device=dev),
                                                      torch.empty(Cc, dtype=torch.float32, device=dev))
            C.bn_train_bwd(dout[o:o + n], raw[o:o + n], stats[gi], 2, dg, db, draw[o:o + n])
            if gi:                                           # the groups' contributions in order, like separate passes
                dgamma += dg
                dbeta += db
            o += n
        dx, dw = _dconv_grads(x, weight, draw, ctx.kind, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, None


def conv_bn_leaky_train(x, conv, bn, kind, groups=None):
    """Training-mode conv + BatchNorm + LeakyReLU(0.2) over the parameter-holder modules (`groups`: ConvBnLeaky)."""
    out = ConvBnLeaky.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                            bn.running_var if bn.track_running_stats else None, kind, bn.momentum, bn.eps, _nbt(bn),
                            groups)
    if bn.track_running_stats:
        torch.autograd.graph.increment_version(bn.running_mean)
        torch.autograd.graph.increment_version(bn.running_var)
    return out


class ConvTo3(torch.autograd.Function):
    """KxK conv to 3 channels [+ tanh + alpha * addend] (the image heads) with `alpha` possibly a trainable 1-element
    tensor (models16.NetG_highweight's `a`, models16.py:126): tgsr::conv_to3 / tgsr::conv_to3_bwd + d/da = sum(dy * addend)."""

    @staticmethod
    def forward(ctx, x, weight, addend, tanh_axpy, alpha):
        ctx.alpha_is_tensor = torch.is_tensor(alpha)
        alpha_f = float(alpha.detach().item()) if ctx.alpha_is_tensor else float(alpha)
        if addend is not None:
            addend = addend.contiguous()     # backward hands the saved tensor's raw pointer to the kernel as dense NCHW
        with torch.no_grad():
            out = C.conv_to3(x, weight.detach(), tanh_axpy, addend, alpha_f)
        ctx.save_for_backward(x, weight, out if tanh_axpy else None, addend)
        ctx.cfg = (tanh_axpy, alpha_f)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, out, addend = ctx.saved_tensors
        tanh_axpy, alpha = ctx.cfg
        dy = dy.contiguous()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        if need_dx or need_dw:
            dx, dw = C.conv_to3_bwd(dy, out, addend, alpha, x.contiguous(), weight.detach(), tanh_axpy, need_dx, need_dw)
        dadd = dy * alpha if (addend is not None and ctx.needs_input_grad[2]) else None
        dalpha = None
        if ctx.alpha_is_tensor and ctx.needs_input_grad[4] and addend is not None:
            dalpha = (dy * addend).sum().reshape(1)              # out = tanh(conv) + alpha * addend
        return (dx if need_dx else None), (dw if need_dw else None), dadd, None, dalpha


class AxpyImage(torch.autograd.Function):
    """t + alpha * s with a host constant alpha (tgsr::axpy_images): NetG_highweight(useAct=False)'s `conv_output(out) + a * SRb`
    (model.py:226, 280) - the tanh form has the addend inside the convolution's epilogue instead."""

    @staticmethod
    def forward(ctx, t, s, alpha):
        ctx.alpha = float(alpha)
        return C.axpy_images([t.contiguous()], [s.contiguous()], ctx.alpha)[0]

    @staticmethod
    def backward(ctx, dy):
        return (dy if ctx.needs_input_grad[0] else None), (dy * ctx.alpha if ctx.needs_input_grad[1] else None), None


class WordAttention(torch.autograd.Function):
    """GlobalAttentionGeneral.forward with a HIP backward for h and conv_context.weight (and words when needed):
    tgsr::word_attention / tgsr::word_attention_bwd."""

    @staticmethod
    def forward(ctx, h, words, w_ctx, mask, correct_mask):
        c_code, attn = C.word_attention(h, words, w_ctx.detach(), mask, correct_mask, None)
        ctx.save_for_backward(h, words, w_ctx, mask)
        ctx.correct_mask = correct_mask
        ctx.mark_non_differentiable(attn)
        return c_code, attn

    @staticmethod
    def backward(ctx, dc, _dattn):
        h, words, w_ctx, mask = ctx.saved_tensors
        B, idf = h.shape[0], h.shape[1]
        cdf, T = words.shape[1], words.shape[2]
        w2 = w_ctx.detach().reshape(idf, cdf)
        # the 1x1 word projection (GlobalAttention.py:100-102) and its two gradients on the library's own MFMA kernels
        # (tgsr::word_project: [B, idf, 32] zero padded past T; tgsr::linear = gemm_bias_kernel) - no rocBLAS / Tensile
        # kernel on the training path
        src = C.word_project(words.detach().contiguous(), [w_ctx.detach()])[0]
        dh, dsrc = C.word_attention_bwd(h.contiguous(), src, mask, ctx.correct_mask, T, dc.contiguous())
        dwords = dw = None
        if ctx.needs_input_grad[1]:      # dwords[b] = W^T dsrc[b]: rows (b, t) x W [idf, cdf]
            dwt = C.linear(dsrc.permute(0, 2, 1).reshape(B * T, idf).contiguous(), w2.t().contiguous(), None)
            dwords = dwt.reshape(B, T, cdf).permute(0, 2, 1).contiguous()
        if ctx.needs_input_grad[2]:      # dW[i][c] = sum_{b,t} dsrc[b][i][t] words[b][c][t]
            dw = C.linear(dsrc.permute(1, 0, 2).reshape(idf, B * T).contiguous(),
                          words.detach().permute(1, 0, 2).reshape(cdf, B * T).contiguous(), None).reshape(w_ctx.shape)
        return dh, dwords, dw, None, None


class DamsmWords(torch.autograd.Function):
    """sim[j][i] of words_loss (losses.py:73-113) with its HIP backward: tgsr::damsm_words / tgsr::damsm_words_bwd.
    Differentiable w.r.t. the image region features and the word embeddings; the diagonal attention maps are returned
    without gradient, as the reference only plots them."""

    @staticmethod
    def forward(ctx, img_features, words_emb, lens, gamma1, gamma2):
        sim, att = C.damsm_words(img_features, words_emb, list(lens), float(gamma1), float(gamma2))
        ctx.save_for_backward(img_features, words_emb)
        ctx.meta = (list(lens), float(gamma1), float(gamma2))
        ctx.mark_non_differentiable(att)
        return sim, att

    @staticmethod
    def backward(ctx, grad_sim, _grad_att):
        img_features, words_emb = ctx.saved_tensors
        lens, gamma1, gamma2 = ctx.meta
        g_img, g_words = C.damsm_words_bwd(img_features.detach(), words_emb.detach(), lens, gamma1, gamma2, grad_sim)
        return g_img, g_words, None, None, None


class BiLSTM(torch.autograd.Function):
    """RNN_ENCODER's bidirectional LSTM (util.py:233-260) in training mode: forward = tgsr::bilstm_train on the embedded
    (and dropped-out) inputs, backward = tgsr::bilstm_bwd (BPTT, one workgroup per sample and direction) plus four GEMMs
    (tgsr::linear) for dW_ih, dW_hh and dx.  w_* are the two directions stacked: [2,4H,ninput], [2,4H,H], [2,4H]."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, lens):
        words, sent, acts = C.bilstm_train(x, w_ih.detach(), w_hh.detach(), b_ih.detach(), b_hh.detach(), list(lens))
        ctx.save_for_backward(x, w_ih, w_hh, acts, words)
        ctx.lens = list(lens)
        return words, sent

    @staticmethod
    def backward(ctx, d_words, d_sent):
        x, w_ih, w_hh, acts, words = ctx.saved_tensors
        B, Tmax, K = x.shape
        H = w_hh.shape[2]
        if d_words is None:
            d_words = torch.zeros_like(words)
        dgates, hprev, dbias = C.bilstm_bwd(ctx.lens, w_hh.detach(), acts, words, d_words.contiguous(),
                                            None if d_sent is None else d_sent.contiguous())
        gemm = C._gemm_nt
        g2 = dgates.reshape(B * Tmax, 8 * H)
        x2 = x.detach().reshape(B * Tmax, K)
        dw_ih = gemm(g2.t(), x2.t()).reshape(2, 4 * H, K)                        # dgates^T x
        dx = gemm(g2, w_ih.detach().reshape(8 * H, K).t()).reshape(B, Tmax, K)   # dgates W_ih
        dw_hh = torch.stack([gemm(dgates[:, :, d].reshape(B * Tmax, 4 * H).t(), hprev[:, :, d].reshape(B * Tmax, H).t())
                             for d in range(2)])
        return dx, dw_ih, dw_hh, dbias, dbias.clone(), None


class BiGRU(torch.autograd.Function):
    """RNN_ENCODER's bidirectional GRU (cfg.RNN_TYPE == 'GRU', util.py:207-211, 233-260) in training mode: forward =
    tgsr::bigru_train on the embedded (and dropped-out) inputs, backward = tgsr::bigru_bwd (BPTT, one workgroup per sample and
    direction) plus the GEMMs for dW_ih, dW_hh and dx.  w_* are the two directions stacked: [2,3H,ninput], [2,3H,H], [2,3H]
    (gate order r, z, n)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, lens):
        words, sent, acts = C.bigru_train(x, w_ih.detach(), w_hh.detach(), b_ih.detach(), b_hh.detach(), list(lens))
        ctx.save_for_backward(x, w_ih, w_hh, acts, words)
        ctx.lens = list(lens)
        return words, sent

    @staticmethod
    def backward(ctx, d_words, d_sent):
        x, w_ih, w_hh, acts, words = ctx.saved_tensors
        B, Tmax, K = x.shape
        H = w_hh.shape[2]
        if d_words is None:
            d_words = torch.zeros_like(words)
        dgx, dgh, hprev, dbias = C.bigru_bwd(ctx.lens, w_hh.detach(), acts, words, d_words.contiguous(),
                                             None if d_sent is None else d_sent.contiguous())
        gemm = C._gemm_nt
        g2 = dgx.reshape(B * Tmax, 6 * H)
        x2 = x.detach().reshape(B * Tmax, K)
        dw_ih = gemm(g2.t(), x2.t()).reshape(2, 3 * H, K)                        # dgx^T x
        dx = gemm(g2, w_ih.detach().reshape(6 * H, K).t()).reshape(B, Tmax, K)   # dgx W_ih
        dw_hh = torch.stack([gemm(dgh[:, :, d].reshape(B * Tmax, 3 * H).t(), hprev[:, :, d].reshape(B * Tmax, H).t())
                             for d in range(2)])
        return dx, dw_ih, dw_hh, dbias[0].contiguous(), dbias[1].contiguous(), None


class LinearFn:
    """y = x W^T + b with all three GEMMs on the HIP kernel (CNN_ENCODER.emb_cnn_code, util.py:301,364): the differentiable
    operator tgsr::linear (autograd registered on the op itself)."""

    @staticmethod
    def apply(x, w, bias):
        return C.linear(x, w, bias)


class RowDot:
    """out[b] = <x[b, :], w> + bias: the discriminators' logit heads (a 4x4 / stride-4 conv of a 4x4 map to one channel):
    the differentiable operator tgsr::rowdot."""

    @staticmethod
    def apply(x, w, bias):
        return C.rowdot(x.contiguous(), w, bias)


class Conv1x1Fn:
    """1x1 convolution without bias (CNN_ENCODER.emb_features, util.py:300,367): the differentiable operator tgsr::conv1x1
    (dx = the same kernel on the transposed weight, dw = one GEMM over all positions)."""

    @staticmethod
    def apply(x, w):
        return C.conv1x1(x, w)
