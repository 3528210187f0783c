"""Training path: torch.autograd.Function wrappers whose forward AND backward are HIP kernels.

The reference trains through torch autograd over nn.Conv2d / nn.BatchNorm2d(train) / GLU / nn.Upsample
(util.py:74-80, 110-130).  Here one Function = one fused block:
    ConvBnAct: [nearest x2 ->] conv3x3 -> BatchNorm2d(batch statistics) -> GLU | (+ residual)
        forward : tgsr_wino_conv3x3_fwd | tgsr_conv3x3_fwd (raw) -> tgsr_bn_train_fwd (stats, running update,
                  normalise + GLU/residual)
        backward: tgsr_bn_train_bwd (GLU', BN') -> data gradient = the same conv kernels on flipped/transposed
                  weights (+ tgsr_sumpool2x2 through the up-sample) and tgsr_conv3x3_wgrad
"""
import contextlib

import torch

from . import _lib, ops
from ._lib import TgsrError, check
from .ops import _p, _stream


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
    """Called once the side stream has been joined: every gradient computed there must BE the parameter's `.grad`."""
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
    if not torch.cuda.is_current_stream_capturing():     # (inside a graph capture the pool keeps its tensors alive)
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


def _conv_raw(x: torch.Tensor, weight: torch.Tensor, upsample: bool = False, dgrad: bool = False,
              residual: torch.Tensor = None) -> torch.Tensor:
    """conv3x3 without affine / activation (what BatchNorm's batch statistics are taken of, and the data gradient):
    the Winograd kernel where it applies (no up-sampling, Cout % 64 == 0, Cin % 4 == 0), else the direct kernel.
    The weights change every step, so they are packed per call (a few microseconds).  `dgrad`: `weight` is the forward
    layer's [Cin_of_x... = weight.shape[0]] filter and the conv applied is its transpose (pack kernels read it
    transposed and flipped: no flip / transpose / copy kernels)."""
    from . import util
    Cout = weight.shape[1] if dgrad else weight.shape[0]
    if util.WINOGRAD and not upsample and util._wino_pays(x, Cout, None, None):
        return ops.conv3x3_wino(x, ops.pack_wino_weight(weight, glu=False, dgrad=dgrad), Cout, None, None, residual=residual)
    if util.WINOGRAD and upsample and ops.upwino_supported(x, Cout):
        assert not dgrad and residual is None
        return ops.upwino_glu(x, ops.pack_upwino_weight(weight, glu=False), Cout, None, None, glu=False)
    return ops.conv3x3_fused(x, ops.pack_conv3x3_weight(weight, dgrad=dgrad), Cout, None, None, glu=False, upsample=upsample,
                             residual=residual)


def _cba_forward(x, weight, gamma, beta, running_mean, running_var, residual, glu, upsample, momentum, eps, nbt):
    """conv3x3 (raw) -> BatchNorm batch statistics -> normalise (+ GLU | + residual).  Returns (out, raw, stats)."""
    L = _lib.lib()
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    raw = _conv_raw(x, weight.detach(), upsample)
    Ho, Wo = raw.shape[2], raw.shape[3]
    HW = Ho * Wo
    dev = x.device
    nsplit = L.tgsr_bn_train_nsplit(B, Cout, HW)
    ws = torch.empty(Cout * nsplit * 4, dtype=torch.float32, device=dev)
    stats = torch.empty(4, Cout, dtype=torch.float32, device=dev)      # mean, invstd, scale, shift
    co = Cout // 2 if glu else Cout
    out = torch.empty(B, co, Ho, Wo, dtype=torch.float32, device=dev)
    res = None if residual is None else residual.contiguous()
    rc = L.tgsr_bn_train_fwd(_p(raw), B, Cout, HW, _p(gamma.detach()), _p(beta.detach()), float(eps),
                             float(momentum), _p(running_mean), _p(running_var), 1 if glu else 0, _p(res),
                             0 if res is None else co * HW, _p(ws), _p(stats[0]), _p(stats[1]), _p(stats[2]),
                             _p(stats[3]), _p(out), co * HW, _p(nbt), _stream())
    check(rc, "tgsr_bn_train_fwd")
    return out, raw, stats


def _cba_backward(x, weight, raw, stats, gamma, beta, glu, upsample, dout, need_dx, need_dw, dx_addend=None):
    """Backward of _cba_forward: (dx, dw, dgamma, dbeta).  `dx_addend` (the gradient arriving at x through a skip
    connection) is added in the data-gradient conv's epilogue instead of by a separate elementwise kernel."""
    L = _lib.lib()
    dout = dout.contiguous()
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    Ho, Wo = raw.shape[2], raw.shape[3]
    HW = Ho * Wo
    dev = x.device
    co = Cout // 2 if glu else Cout
    nsplit = L.tgsr_bn_train_nsplit(B, co, HW)
    ws = torch.empty(co * nsplit * 4, dtype=torch.float32, device=dev)
    draw = torch.empty_like(raw)
    dgamma, dbeta = _grad_out(gamma, (Cout,), dev), _grad_out(beta, (Cout,), dev)
    rc = L.tgsr_bn_train_bwd(_p(dout), _p(raw), B, Cout, HW, _p(stats[2]), _p(stats[3]), _p(stats[0]), _p(stats[1]),
                             1 if glu else 0, _p(ws), _p(ws), _p(draw), _p(dgamma), _p(dbeta), _stream())
    check(rc, "tgsr_bn_train_bwd")
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
            dx = torch.empty(B, Cin, H, W, dtype=torch.float32, device=dev)
            check(L.tgsr_sumpool2x2(_p(dxu), B * Cin, H, W, _p(dx), _stream()), "tgsr_sumpool2x2")
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

        def wgrad():
            e0 = ops._ev() if ops.profile is not None else None
            if upsample and util.WINOGRAD and Cout % 64 == 0 and Cin % 32 == 0:
                # upBlock: 9 Winograd positions on the low-resolution pixels (4x fewer multiplies than 9 taps on the
                # up-sampled grid)
                kname = "upwino_wgrad_kernel"
                wws = torch.empty(L.tgsr_upwino_wgrad_ws_elems(B, Cin, Cout, H, W), dtype=torch.float32, device=dev)
                rc = L.tgsr_upwino_wgrad(_p(draw), _p(x), Cin * H * W, B, Cin, H, W, Cout, _p(wws), _p(dw), _stream())
                check(rc, "tgsr_upwino_wgrad")
            elif (not upsample) and util.WINOGRAD and Cout % 64 == 0 and Cin % 32 == 0:
                # plain conv: 16 Winograd positions per 2x2 output tile (2.25x fewer multiplies than 9 taps per pixel)
                kname = "wino_wgrad_kernel"
                wws = torch.empty(L.tgsr_wino_wgrad_ws_elems(B, Cin, Cout, H, W), dtype=torch.float32, device=dev)
                rc = L.tgsr_wino_wgrad(_p(draw), _p(x), Cin * H * W, B, Cin, H, W, Cout, _p(wws), _p(dw), _stream())
                check(rc, "tgsr_wino_wgrad")
            else:
                kname = "conv3x3_wgrad_kernel"
                n = L.tgsr_conv3x3_wgrad_ws_elems(B, Cin, Cout, H, W, 1 if upsample else 0)
                wws = torch.empty(n, dtype=torch.float32, device=dev)
                rc = L.tgsr_conv3x3_wgrad(_p(draw), _p(x), Cin * H * W, B, Cin, H, W, Cout, 1 if upsample else 0,
                                          _p(wws), _p(dw), _stream())
                check(rc, "tgsr_conv3x3_wgrad")
            if ops.profile is not None:      # direct-form FLOPs of the weight gradient: one MAC per (output pixel, tap, ci, co)
                ops.profile.append((kname, 2.0 * B * HW * Cout * Cin * 9, 4.0 * (B * Cin * H * W + B * Cout * HW + Cout * Cin * 9),
                                    e0, ops._ev()))
        on_wgrad_stream(dev, (draw, x), wgrad, adopted)
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
        out = ops.conv4x4s2(x, weight, leaky=leaky)
        ctx.save_for_backward(x, weight, out if leaky else None)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, out = ctx.saved_tensors
        g = dy.contiguous() if out is None else ops.leaky_relu_bwd(dy, out)
        dx = ops.conv4x4s2_dgrad(g, weight, x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = ops.conv4x4s2_wgrad(g, x) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class ConvBnLeaky(torch.autograd.Function):
    """conv -> BatchNorm2d(batch statistics) -> LeakyReLU(0.2) with `kind` = "down" (conv4x4 stride 2: downBlock,
    util.py:92-98) or "3x3" (the discriminators' Block3x3_leakRelu).  BatchNorm + activation forward / backward =
    tgsr_bn_train_fwd / _bwd(act = 2); the 3x3 convolution and its gradients are the generator's fp32 kernels."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, kind, momentum, eps, nbt=None, groups=None):
        """`groups`: sizes of consecutive slices of the batch that are normalised with their OWN batch statistics (and
        update the running statistics one after the other, in order) - several forward passes of the module (real /
        fake / mismatched images of discriminator_loss) as one convolution launch, one data-gradient launch and one
        weight-gradient launch, numerically the passes run one by one."""
        L = _lib.lib()
        x = x.contiguous()
        if kind == "down":
            raw = ops.conv4x4s2(x, weight)
        elif ops.conv3x3_gemm_pays(x.shape[1], weight.shape[0], x.shape[2], x.shape[3]):
            raw = ops.conv3x3_gemm(x, weight)                 # many channels on 4x4 pixels: the implicit-GEMM form
        else:
            raw = ops.conv3x3_fused(x, ops.pack_conv3x3_weight(weight), weight.shape[0], None, None)
        B, C, Ho, Wo = raw.shape
        HW, dev = Ho * Wo, x.device
        groups = (B,) if groups is None else tuple(int(g) for g in groups)
        if sum(groups) != B or min(groups) < 1:
            raise TgsrError("ConvBnLeaky: groups %s do not partition a batch of %d" % (groups, B))
        stats = torch.empty(len(groups), 4, C, dtype=torch.float32, device=dev)
        out = torch.empty_like(raw)
        o = 0
        for gi, n in enumerate(groups):
            nsplit = L.tgsr_bn_train_nsplit(n, C, HW)
            ws = torch.empty(C * nsplit * 4, dtype=torch.float32, device=dev)
            st = stats[gi]
            rc = L.tgsr_bn_train_fwd(_p(raw[o:o + n]), n, C, HW, _p(gamma.detach()), _p(beta.detach()), float(eps),
                                     float(momentum), _p(running_mean), _p(running_var), 2, None, 0, _p(ws), _p(st[0]),
                                     _p(st[1]), _p(st[2]), _p(st[3]), _p(out[o:o + n]), C * HW, _p(nbt), _stream())
            check(rc, "tgsr_bn_train_fwd")
            o += n
        ctx.save_for_backward(x, weight, raw, stats)
        ctx.kind = kind
        ctx.groups = groups
        ctx.bn_params = (gamma, beta)          # only to look up their gradient slots (parallel.grad_slot) in backward
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x, weight, raw, stats = ctx.saved_tensors
        dout = dout.contiguous()
        B, C, Ho, Wo = raw.shape
        HW, dev = Ho * Wo, x.device
        draw = torch.empty_like(raw)
        gamma, beta = ctx.bn_params
        dgamma, dbeta = _grad_out(gamma, (C,), dev), _grad_out(beta, (C,), dev)
        o = 0
        for gi, n in enumerate(ctx.groups):
            nsplit = L.tgsr_bn_train_nsplit(n, C, HW)
            ws = torch.empty(C * nsplit * 4, dtype=torch.float32, device=dev)
            st = stats[gi]
            dg, db = (dgamma, dbeta) if gi == 0 else (torch.empty(C, dtype=torch.float32, device=dev),
                                                      torch.empty(C, dtype=torch.float32, device=dev))
            rc = L.tgsr_bn_train_bwd(_p(dout[o:o + n]), _p(raw[o:o + n]), n, C, HW, _p(st[2]), _p(st[3]), _p(st[0]),
                                     _p(st[1]), 2, _p(ws), _p(ws), _p(draw[o:o + n]), _p(dg), _p(db), _stream())
            check(rc, "tgsr_bn_train_bwd")
            if gi:                                           # the groups' contributions in order, like separate passes
                dgamma += dg
                dbeta += db
            o += n
        dx = dw = None
        Cin, H, W = x.shape[1], x.shape[2], x.shape[3]
        if ctx.kind == "down":
            if ctx.needs_input_grad[0]:
                dx = ops.conv4x4s2_dgrad(draw, weight, H, W)
            if ctx.needs_input_grad[1]:
                dw = ops.conv4x4s2_wgrad(draw, x, out=_grad_out(weight, weight.shape, dev))
        elif ops.conv3x3_gemm_pays(Cin, C, H, W):
            if ctx.needs_input_grad[0]:
                dx = ops.conv3x3_gemm_dgrad(draw, weight)
            if ctx.needs_input_grad[1]:
                dw = ops.conv3x3_gemm_wgrad(draw, x, out=_grad_out(weight, weight.shape, dev))
        else:
            if ctx.needs_input_grad[0]:
                wT = _dgrad_weight(weight.detach())
                cpad = (Cin + 31) // 32 * 32
                if cpad != Cin:
                    wT = torch.cat((wT, wT.new_zeros(cpad - Cin, C, 3, 3)), 0)
                dx = ops.conv3x3_fused(draw, ops.pack_conv3x3_weight(wT), cpad, None, None)
                if cpad != Cin:
                    dx = dx[:, :Cin].contiguous()
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                wws = torch.empty(L.tgsr_conv3x3_wgrad_ws_elems(B, Cin, C, H, W, 0), dtype=torch.float32, device=dev)
                check(L.tgsr_conv3x3_wgrad(_p(draw), _p(x), Cin * H * W, B, Cin, H, W, C, 0, _p(wws), _p(dw), _stream()),
                      "tgsr_conv3x3_wgrad")
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
    """KxK conv to 3 channels [+ tanh + alpha * addend] (the image heads), forward and backward on HIP."""

    @staticmethod
    def forward(ctx, x, weight, addend, tanh_axpy, alpha):
        """alpha: python float, or a 1-element tensor (models16.NetG_highweight's trainable `a`, models16.py:126)."""
        ctx.alpha_is_tensor = torch.is_tensor(alpha)
        alpha_f = float(alpha.detach().item()) if ctx.alpha_is_tensor else float(alpha)
        if addend is not None:
            addend = addend.contiguous()     # backward hands the saved tensor's raw pointer to the kernel as dense NCHW
        out = ops.conv_to3(x, weight, tanh_axpy=tanh_axpy, addend=addend, alpha=alpha_f)
        ctx.save_for_backward(x, weight, out if tanh_axpy else None, addend)
        ctx.cfg = (tanh_axpy, alpha_f)
        return out

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, weight, out, addend = ctx.saved_tensors
        tanh_axpy, alpha = ctx.cfg
        dy = dy.contiguous()
        x = x.contiguous()
        B, Cin, H, W = x.shape
        K = weight.shape[2]
        dev = x.device
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty_like(x) if need_dx else None
        dw = torch.empty_like(weight) if need_dw else None
        ws = torch.empty(L.tgsr_conv_to3_bwd_ws_elems(B, Cin, H, W, K), dtype=torch.float32, device=dev) if need_dw else None
        w = weight.detach().contiguous()
        rc = L.tgsr_conv_to3_bwd(_p(dy), _p(out), _p(addend), alpha, _p(x), Cin * H * W, _p(w), B, Cin, H, W, K,
                                 _lib.ACT_TANH_AXPY if tanh_axpy else _lib.ACT_NONE, _p(dx), _p(ws), _p(dw), _stream())
        check(rc, "tgsr_conv_to3_bwd")
        dadd = dy * alpha if (addend is not None and ctx.needs_input_grad[2]) else None
        dalpha = None
        if ctx.alpha_is_tensor and ctx.needs_input_grad[4] and addend is not None:
            dalpha = (dy * addend).sum().reshape(1)              # out = tanh(conv) + alpha * addend
        return dx, dw, dadd, None, dalpha


class WordAttention(torch.autograd.Function):
    """GlobalAttentionGeneral.forward with a HIP backward for h and conv_context.weight (and words when needed)."""

    @staticmethod
    def forward(ctx, h, words, w_ctx, mask, correct_mask):
        c_code, attn = ops.word_attention(h, words, w_ctx, mask, correct_mask)
        ctx.save_for_backward(h, words, w_ctx, mask)
        ctx.correct_mask = correct_mask
        ctx.mark_non_differentiable(attn)
        return c_code, attn

    @staticmethod
    def backward(ctx, dc, _dattn):
        L = _lib.lib()
        h, words, w_ctx, mask = ctx.saved_tensors
        h = h.contiguous()
        dc = dc.contiguous()
        B, idf, ih, iw = h.shape
        cdf, T = words.shape[1], words.shape[2]
        Q = ih * iw
        dev = h.device
        w2 = w_ctx.detach().reshape(idf, cdf)
        src = torch.zeros(B, idf, 32, dtype=torch.float32, device=dev)
        src[:, :, :T] = torch.matmul(w2, words.detach())               # tiny [idf,cdf] x [B,cdf,T] (library GEMM)
        nch = L.tgsr_word_attention_bwd_chunks(Q)
        part = torch.empty(B, nch, idf, 32, dtype=torch.float32, device=dev)
        dh = torch.empty_like(h)
        m8 = None if mask is None else ops._mask_u8(mask)
        rc = L.tgsr_word_attention_bwd(_p(h), idf * Q, _p(src), _p(m8), 1 if ctx.correct_mask else 0, B, idf, T, Q,
                                       _p(dc), _p(dh), _p(part), _stream())
        check(rc, "tgsr_word_attention_bwd")
        dsrc = part.sum(1)[:, :, :T]                                      # [B, idf, T]
        dwords = torch.matmul(w2.t(), dsrc) if ctx.needs_input_grad[1] else None
        dw = torch.einsum("bit,bct->ic", dsrc, words.detach()).reshape(w_ctx.shape) if ctx.needs_input_grad[2] else None
        return dh, dwords, dw, None, None


class DamsmWords(torch.autograd.Function):
    """sim[j][i] of words_loss (losses.py:73-113) with its HIP backward (tgsr_damsm_words_bwd): differentiable w.r.t.
    the image region features and the word embeddings; the diagonal attention maps are returned without gradient,
    as the reference only plots them."""

    @staticmethod
    def forward(ctx, img_features, words_emb, lens, gamma1, gamma2):
        sim, att = ops.damsm_words_similarity(img_features, words_emb, lens, gamma1, gamma2, need_att=True)
        ctx.save_for_backward(img_features, words_emb)
        ctx.meta = (list(lens), float(gamma1), float(gamma2))
        ctx.mark_non_differentiable(att)
        return sim, att

    @staticmethod
    def backward(ctx, grad_sim, _grad_att):
        img_features, words_emb = ctx.saved_tensors
        lens, gamma1, gamma2 = ctx.meta
        g_img, g_words = ops.damsm_words_bwd(img_features, words_emb, lens, gamma1, gamma2, grad_sim)
        return g_img, g_words, None, None, None


def _gemm_nt(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [M,K] @ b[N,K]^T on the HIP GEMM kernel (tgsr_linear_fwd)."""
    return ops.linear(a.contiguous(), b.contiguous(), None)


class BiLSTM(torch.autograd.Function):
    """RNN_ENCODER's bidirectional LSTM (util.py:233-260) in training mode: forward = tgsr_bilstm_train_fwd on the
    embedded (and dropped-out) inputs, backward = tgsr_bilstm_bwd (BPTT, one workgroup per sample and direction) plus
    four GEMMs for dW_ih, dW_hh and dx.  w_* are the two directions stacked: [2,4H,ninput], [2,4H,H], [2,4H]."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, lens):
        words, sent, acts = ops.bilstm_train_fwd(x, lens, w_ih, w_hh, b_ih, b_hh)
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
        dgates, hprev, dbias = ops.bilstm_bwd(ctx.lens, w_hh, acts, words, d_words, d_sent)
        g2 = dgates.reshape(B * Tmax, 8 * H)
        x2 = x.detach().reshape(B * Tmax, K)
        dw_ih = _gemm_nt(g2.t(), x2.t()).reshape(2, 4 * H, K)                        # dgates^T x
        dx = _gemm_nt(g2, w_ih.detach().reshape(8 * H, K).t()).reshape(B, Tmax, K)   # dgates W_ih
        dw_hh = torch.stack([_gemm_nt(dgates[:, :, d].reshape(B * Tmax, 4 * H).t(),
                                      hprev[:, :, d].reshape(B * Tmax, H).t()) for d in range(2)])
        return dx, dw_ih, dw_hh, dbias, dbias.clone(), None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b with all three GEMMs on the HIP kernel (CNN_ENCODER.emb_cnn_code, util.py:301,364)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return ops.linear(x, w, bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _gemm_nt(dy, w.detach().t()) if ctx.needs_input_grad[0] else None          # [B,N] @ W [N,K]
        dw = _gemm_nt(dy.t(), x.detach().t()) if ctx.needs_input_grad[1] else None      # dy^T x
        return dx, dw, (dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None)


class RowDot(torch.autograd.Function):
    """out[b] = <x[b, :], w> + bias: the discriminators' logit heads (a 4x4 / stride-4 conv of a 4x4 map to one
    channel), tgsr_rowdot_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, w, bias):
        L = _lib.lib()
        x = x.contiguous()
        w = w.detach().contiguous().view(-1)
        B, K = x.shape
        out = torch.empty(B, dtype=torch.float32, device=x.device)
        check(L.tgsr_rowdot_fwd(_p(x), _p(w), _p(None if bias is None else bias.detach()), _p(out), B, K, _stream()),
              "tgsr_rowdot_fwd")
        ctx.save_for_backward(x, w)
        ctx.wshape = None
        return out

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, K = x.shape
        ndx, ndw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty_like(x) if ndx else None
        dw = torch.empty(K, dtype=torch.float32, device=x.device) if ndw else None
        if ndx or ndw:
            check(L.tgsr_rowdot_bwd(_p(dy), _p(x), _p(w), _p(dx), _p(dw), B, K, _stream()), "tgsr_rowdot_bwd")
        return dx, dw, (dy.sum().reshape(1) if ctx.needs_input_grad[2] else None)


class Conv1x1Fn(torch.autograd.Function):
    """1x1 convolution without bias (CNN_ENCODER.emb_features, util.py:300,367): forward tgsr_conv1x1_fwd, backward as
    the same kernel on the transposed weight (dx) and one GEMM over all positions (dw)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return ops.conv1x1(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        Cout, Cin = w.shape[0], w.shape[1]
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:                                    # (the frozen trunk's features need none)
            dx = ops.conv1x1(dy, w.detach().reshape(Cout, Cin).t().contiguous())
        if ctx.needs_input_grad[1]:
            dy2 = dy.permute(1, 0, 2, 3).reshape(Cout, -1)             # [Cout, B*S]
            x2 = x.detach().permute(1, 0, 2, 3).reshape(Cin, -1)       # [Cin,  B*S]
            dw = _gemm_nt(dy2, x2).reshape(w.shape)
        return dx, dw
