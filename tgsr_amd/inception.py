"""CNN_ENCODER's frozen Inception-v3 trunk on the library's own kernels (csrc/tgsr_igemm.hip).

The reference copies sixteen blocks out of `torchvision.models.inception_v3` (util.py:281-298), freezes every parameter
(util.py:274-275) and walks them in `CNN_ENCODER.forward` (util.py:308-362): bilinear resize to 299 x 299, five stem convolutions
with two max pools, Mixed_5b..6e (-> the 17 x 17 x 768 region features), Mixed_7a..7c, the 8 x 8 average pool.  `generator_loss`
(losses.py:375-389) runs it on the finest fake image and needs the gradient back to that image - never a weight gradient.

This module executes that walk for ANY object carrying torchvision's attribute names (conv + bn pairs `m.conv`, `m.bn`; block
branches `branch1x1`, `branch5x5_1`, ... - the names the reference's checkpoints are keyed by): the layers are read off the modules,
eval-mode BatchNorm is folded into the filter once per weight version (scale) and into the convolution's epilogue (shift + ReLU), every
branch writes straight into its channel slice of the block's concatenation, and the backward is the recorded tape run in reverse on
the same kernels (ReLU mask in place, data gradients accumulated into the block input in a fixed order).  The trunk's ARITHMETIC is
third-party (torchvision's weights are absent here): parity is pinned against the same modules run by torch, not against the
reference (unpinned).
"""
import contextlib
import os

import torch

from . import custom_ops as C
from ._lib import TgsrError

RESIZE = 299
# The branches of a Mixed_* block are independent chains of small GEMMs (35 us launches that fill a fraction of the CUs):
# each branch runs on a stream of its own, forward and backward, as plain fork / join diamonds off the caller's stream per block (no
# edge between two side streams: a fork nested inside a forked branch is what ROCm 7.2's stream capture does not survive, and the
# walk is captured into the generators' hipGraphs).  TGSR_TRUNK_STREAMS=1: everything on the caller's stream (the same kernels in
# the same order of accumulation: bit-identical).
TRUNK_STREAMS = max(1, min(4, int(os.environ.get("TGSR_TRUNK_STREAMS", "4"))))
# stride-2 data gradients as four stride-1 class GEMMs + an interleave (_Layer.refresh); TGSR_TRUNK_CLASS_DGRAD=0: the direct form
CLASS_DGRAD = os.environ.get("TGSR_TRUNK_CLASS_DGRAD", "1") != "0"
# the branches' contributions to a block's input gradient each into a slot of their own, on their branch's stream, summed in the tape's
# order behind the join (tgsr::sum_stack); TGSR_TRUNK_HEADS_PARALLEL=0: accumulated one after the other behind the join (same bits)
HEADS_PARALLEL = os.environ.get("TGSR_TRUNK_HEADS_PARALLEL", "1") != "0"


class _Layer:
    """conv (bias-free) + BatchNorm2d(eval) + ReLU of one `m.conv` / `m.bn` pair, packed for tgsr::gconv."""

    def __init__(self, mod):
        conv, bn = mod.conv, mod.bn
        if conv.bias is not None or conv.groups != 1 or tuple(conv.dilation) != (1, 1) or conv.stride[0] != conv.stride[1]:
            raise TgsrError("the trunk kernels take bias-free, dense, undilated convolutions with one stride")
        self.conv, self.bn = conv, bn
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.kh, self.kw = conv.kernel_size
        self.stride = conv.stride[0]
        self.ph, self.pw = conv.padding
        self.key = None
        self.refresh()

    def _key(self):
        bn = self.bn
        return tuple((t.data_ptr(), t._version) for t in (self.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var))

    def refresh(self):
        k = self._key()
        if k == self.key:
            return
        from . import ops
        bn = self.bn
        scale, self.shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        self.wf = C.gconv_pack(self.conv.weight.detach(), scale, False)
        self.wd = C.gconv_pack(self.conv.weight.detach(), scale, True)
        # A stride-2 layer's data gradient by parity classes of the input pixel: input row y = 2 Y + py meets only the taps
        # ky = k0 + 2 a (k0 = (py + pad) % 2) - dx[2 Y + py] = sum_a g[Y + c0 - a] w[k0 + 2 a], c0 = (py + pad - k0) / 2 - which is the
        # STRIDE-1 data gradient of the kernel w[k0::2] with padding c0, dense at half resolution: four small GEMMs over a quarter
        # of the direct form's products (three quarters of those multiply the zeros between the strided samples), then one
        # interleaving pass (tgsr::interleave2x2_).  (py, px, pack, kh', kw', pad_h', pad_w')
        self.cls = None
        if self.stride == 2 and self.kh >= 2 and self.kw >= 2 and self.ph == 0 and self.pw == 0 and self.cin > 4 and CLASS_DGRAD:
            self.cls = []
            w = self.conv.weight.detach()
            for py in (0, 1):
                for px in (0, 1):
                    ky0, kx0 = (py + self.ph) % 2, (px + self.pw) % 2
                    wc = w[:, :, ky0::2, kx0::2].contiguous()
                    self.cls.append((py, px, C.gconv_pack(wc, scale, True), wc.shape[2], wc.shape[3], (py + self.ph - ky0) // 2,
                                     (px + self.pw - kx0) // 2))
        self.key = k

    def class_dgrad_ok(self, H, W):
        """The class form covers unpadded layers on input sizes the forward uses completely ((H - k) even: no trailing row / column
        that no output reads; a class is then exactly the stride-1 gradient's size) - every Inception-v3 stride-2 layer: 299, 35, 17."""
        return self.cls is not None and (H + 2 * self.ph - self.kh) % 2 == 0 and (W + 2 * self.pw - self.kw) % 2 == 0 and H > 1 and W > 1

    def out_hw(self, H, W):
        return (H + 2 * self.ph - self.kh) // self.stride + 1, (W + 2 * self.pw - self.kw) // self.stride + 1


def _pool_hw(H, W):
    return (H - 3) // 2 + 1, (W - 3) // 2 + 1


class InceptionTrunk:
    """The walk of util.py:308-362 (up to, not including, the two heads) over `enc`'s sixteen blocks.  `forward(x)` returns
    (features [B,768,17,17], pooled [B,2048]) and keeps the tape; `backward(d_features, d_pooled)` returns d(loss)/dx."""

    STEM = ("Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3", "Conv2d_3b_1x1", "Conv2d_4a_3x3")
    MIXED = ("Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e", "Mixed_7a", "Mixed_7b",
             "Mixed_7c")

    def __init__(self, enc):
        self.enc = enc
        self.layers = {}
        for name in self.STEM + self.MIXED:
            blk = getattr(enc, name)
            if hasattr(blk, "conv") and hasattr(blk, "bn"):
                self.layers[name] = _Layer(blk)
            else:
                for bname, sub in blk.named_children():
                    if hasattr(sub, "conv") and hasattr(sub, "bn"):
                        self.layers[name + "." + bname] = _Layer(sub)
        self.tape = None
        self.nstreams = TRUNK_STREAMS
        self.bwd_stream = None                 # see TrunkFn.backward
        self._side = None                      # the side streams (created on first use, distinct from the caller's)
        self._streams = None                   # [caller's stream] + side streams of the walk in progress

    def refresh(self):
        for L in self.layers.values():
            L.refresh()

    # ------------------------------------------------------------------ forward primitives (each records its backward)
    def _new(self, B, Cc, H, W, dev):
        t = torch.empty(B, Cc, H, W, dtype=torch.float32, device=dev)
        self.tensors.append(t)
        return len(self.tensors) - 1

    def _ws(self, need, dev, s=0):
        """Split-reduction workspace of stream index s (allocated on the caller's stream).  A buffer that is outgrown stays
        referenced until the next walk opens: kernels queued on stream s may still read it, and the allocator - which knows only
        the caller's stream - must not hand its memory to another stream's workspace meanwhile."""
        if need > self.wss[s].numel():
            self._ws_old.append(self.wss[s])
            self.wss[s] = torch.empty(need, dtype=torch.float32, device=dev)
        return self.wss[s] if need else None

    # ------------------------------------------------------------------ streams: a dataflow order over tensor ids
    def _open_streams(self, dev):
        """[the caller's stream, side streams...] for one walk; every tensor is allocated on the caller's stream (the `torch.empty`
        calls sit outside the stream scopes) and every side stream is joined back before the walk returns."""
        main = torch.cuda.current_stream(dev)
        if self.nstreams > 1:
            if self._side is None or any(int(st.cuda_stream) == int(main.cuda_stream) for st in self._side):
                from .trainer import distinct_streams
                self._side = distinct_streams(self.nstreams - 1, dev, avoid=[main.cuda_stream])
            self._streams = [main] + list(self._side)
        else:
            self._streams = [main]
        self.wss = [torch.empty(0, dtype=torch.float32, device=dev) for _ in self._streams]
        self._ws_old = []

    def _join_streams(self):
        """The caller's stream waits for everything queued on the side streams (the end of a walk)."""
        main = self._streams[0]
        for st in self._streams[1:]:
            main.wait_stream(st)

    def _fork_streams(self):
        """Every side stream waits for what the caller's stream holds so far (weight packs, tensors written outside `_run`)."""
        main = self._streams[0]
        for st in self._streams[1:]:
            st.wait_stream(main)

    def _sidx(self, s):
        return s % len(self._streams)

    @contextlib.contextmanager
    def _run(self, s):
        """Scope of a launch on stream index s.  Inside a block (between `_fork_streams` and `_join_streams`) every branch keeps to
        ONE stream, reads what existed at the fork (the block input / the gradient of the block output) or what its own stream
        produced, and writes tensors - or channel slices - no other branch touches; the gradient of the block input, which every
        branch adds to, is written after the join, on the caller's stream (`backward`)."""
        if s == 0:
            yield
            return
        with torch.cuda.stream(self._streams[s]):
            yield

    def _conv(self, name, src, dst=None, coff=0, s=0):
        """ConvBnRelu `name` on tensor id `src`; into channels [coff, coff + cout) of tensor id `dst` (a fresh tensor when None);
        on stream index s."""
        L = self.layers[name]
        s = self._sidx(s)
        x = self.tensors[src]
        B, Cs, H, W = x.shape
        if Cs != L.cin:
            raise TgsrError("%s expects %d channels, got %d" % (name, L.cin, Cs))
        OH, OW = L.out_hw(H, W)
        if dst is None:
            dst = self._new(B, L.cout, OH, OW, x.device)
        y = self.tensors[dst]
        ws = self._ws(ops_ws(B, L.cout, OH, OW, L.cin * L.kh * L.kw), x.device, s)
        with self._run(s):
            C.gconv(False, L.wf, x, 0, L.cin, y, coff, L.kh, L.kw, L.stride, L.ph, L.pw, L.shift, True, False, ws, None)
        self.tape.append(("conv", L, src, dst, coff, s))
        return dst

    def _maxpool(self, src, dst=None, coff=0, s=0):
        x = self.tensors[src]
        B, Cc, H, W = x.shape
        OH, OW = _pool_hw(H, W)
        if dst is None:
            dst = self._new(B, Cc, OH, OW, x.device)
        s = self._sidx(s)
        with self._run(s):
            C.maxpool3s2(x, self.tensors[dst], coff)
        self.tape.append(("maxpool", None, src, dst, coff, s))
        return dst

    def _avgpool(self, src, s=0):
        x = self.tensors[src]
        dst = self._new(*x.shape, x.device)
        s = self._sidx(s)
        with self._run(s):
            C.avgpool3(x, self.tensors[dst], False, None)
        self.tape.append(("avgpool", None, src, dst, 0, s))
        return dst

    def _block(self, name, src):
        """One Mixed_* block: its branches between a fork and a join of the side streams; returns the id of the concatenation."""
        self._fork_streams()
        self.tape.append(("fork", None, src, None, 0, 0))
        out = self._branches(name, src)
        self.tape.append(("join", None, src, out, 0, 0))
        self._join_streams()
        return out

    def _branches(self, name, src):
        """The branches of one Mixed_* block by torchvision's names (every branch on one stream index)."""
        blk = getattr(self.enc, name)
        x = self.tensors[src]
        B, _, H, W = x.shape
        dev = x.device
        has = lambda b: hasattr(blk, b)                                                   # noqa: E731
        co = lambda b: self.layers[name + "." + b].cout                                   # noqa: E731
        n = lambda b: name + "." + b                                                      # noqa: E731
        if has("branch5x5_1"):                              # InceptionA: 1x1 | 1x1-5x5 | 1x1-3x3-3x3 | avg pool-1x1
            widths = [co("branch1x1"), co("branch5x5_2"), co("branch3x3dbl_3"), co("branch_pool")]
            out = self._new(B, sum(widths), H, W, dev)
            self._conv(n("branch1x1"), src, out, 0, s=3)
            self._conv(n("branch5x5_2"), self._conv(n("branch5x5_1"), src, s=1), out, widths[0], s=1)
            t = self._conv(n("branch3x3dbl_2"), self._conv(n("branch3x3dbl_1"), src, s=0), s=0)
            self._conv(n("branch3x3dbl_3"), t, out, widths[0] + widths[1], s=0)
            self._conv(n("branch_pool"), self._avgpool(src, s=2), out, widths[0] + widths[1] + widths[2], s=2)
            return out
        if has("branch7x7x3_1"):                            # InceptionD: 1x1-3x3 s2 | 1x1-1x7-7x1-3x3 s2 | max pool
            OH, OW = _pool_hw(H, W)
            widths = [co("branch3x3_2"), co("branch7x7x3_4"), x.shape[1]]
            out = self._new(B, sum(widths), OH, OW, dev)
            self._conv(n("branch3x3_2"), self._conv(n("branch3x3_1"), src, s=1), out, 0, s=1)
            t = self._conv(n("branch7x7x3_3"), self._conv(n("branch7x7x3_2"), self._conv(n("branch7x7x3_1"), src, s=0), s=0), s=0)
            self._conv(n("branch7x7x3_4"), t, out, widths[0], s=0)
            self._maxpool(src, out, widths[0] + widths[1], s=2)
            return out
        if has("branch7x7_1"):                              # InceptionC: 1x1 | 1x1-1x7-7x1 | 1x1-7x1-1x7-7x1-1x7 | avg pool-1x1
            widths = [co("branch1x1"), co("branch7x7_3"), co("branch7x7dbl_5"), co("branch_pool")]
            out = self._new(B, sum(widths), H, W, dev)
            self._conv(n("branch1x1"), src, out, 0, s=3)
            self._conv(n("branch7x7_3"), self._conv(n("branch7x7_2"), self._conv(n("branch7x7_1"), src, s=1), s=1), out, widths[0], s=1)
            t = self._conv(n("branch7x7dbl_1"), src, s=0)
            for b in ("branch7x7dbl_2", "branch7x7dbl_3", "branch7x7dbl_4"):
                t = self._conv(n(b), t, s=0)
            self._conv(n("branch7x7dbl_5"), t, out, widths[0] + widths[1], s=0)
            self._conv(n("branch_pool"), self._avgpool(src, s=2), out, widths[0] + widths[1] + widths[2], s=2)
            return out
        if has("branch3x3_2a"):                             # InceptionE: 1x1 | 1x1-(1x3, 3x1) | 1x1-3x3-(1x3, 3x1) | avg pool-1x1
            widths = [co("branch1x1"), co("branch3x3_2a"), co("branch3x3_2b"), co("branch3x3dbl_3a"), co("branch3x3dbl_3b"),
                      co("branch_pool")]
            out = self._new(B, sum(widths), H, W, dev)
            offs = [sum(widths[:i]) for i in range(6)]
            self._conv(n("branch1x1"), src, out, 0, s=3)
            a = self._conv(n("branch3x3_1"), src, s=1)
            self._conv(n("branch3x3_2a"), a, out, offs[1], s=1)
            self._conv(n("branch3x3_2b"), a, out, offs[2], s=1)
            b_ = self._conv(n("branch3x3dbl_2"), self._conv(n("branch3x3dbl_1"), src, s=0), s=0)
            self._conv(n("branch3x3dbl_3a"), b_, out, offs[3], s=0)
            self._conv(n("branch3x3dbl_3b"), b_, out, offs[4], s=0)
            self._conv(n("branch_pool"), self._avgpool(src, s=2), out, offs[5], s=2)
            return out
        if has("branch3x3dbl_3") and has("branch3x3"):      # InceptionB: 3x3 s2 | 1x1-3x3-3x3 s2 | max pool
            OH, OW = _pool_hw(H, W)
            widths = [co("branch3x3"), co("branch3x3dbl_3"), x.shape[1]]
            out = self._new(B, sum(widths), OH, OW, dev)
            self._conv(n("branch3x3"), src, out, 0, s=1)
            t = self._conv(n("branch3x3dbl_2"), self._conv(n("branch3x3dbl_1"), src, s=0), s=0)
            self._conv(n("branch3x3dbl_3"), t, out, widths[0], s=0)
            self._maxpool(src, out, widths[0] + widths[1], s=2)
            return out
        raise TgsrError("%s: not one of torchvision's Inception-v3 block layouts" % name)

    # ------------------------------------------------------------------ the walk
    @torch.no_grad()
    def forward(self, x):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3):
            raise TgsrError("InceptionTrunk: a HIP fp32 image batch [B, 3, H, W] expected, got %s %s" % (x.dtype, tuple(x.shape)))
        self.refresh()
        self.tensors, self.tape = [], []
        self._open_streams(x.device)
        self.in_hw = (x.shape[2], x.shape[3])
        self.tensors.append(C.bilinear(x.contiguous(), RESIZE, RESIZE))                       # nn.Upsample(size=(299, 299), 'bilinear')
        t = 0
        self.marks = {"resize": 0}                                                             # name -> tensor id (diagnostics)
        for name in self.STEM[:3]:
            t = self.marks[name] = self._conv(name, t)
        t = self.marks["pool1"] = self._maxpool(t)
        for name in self.STEM[3:]:
            t = self.marks[name] = self._conv(name, t)
        t = self.marks["pool2"] = self._maxpool(t)
        for name in self.MIXED[:8]:
            t = self.marks[name] = self._block(name, t)
        self.feat_id = t                                                                       # 17 x 17 x 768 (after Mixed_6e)
        for name in self.MIXED[8:]:
            t = self.marks[name] = self._block(name, t)
        self.last_id = t
        return self.tensors[self.feat_id], C.plane_mean(self.tensors[t])                      # F.avg_pool2d(x, 8) on the 8 x 8 map

    @torch.no_grad()
    def backward(self, d_features, d_pooled):
        """The tape in reverse.  d_features / d_pooled: gradients of the two outputs (None = zero).  Returns d(loss)/d(image)."""
        if self.tape is None:
            raise TgsrError("InceptionTrunk.backward without a forward")
        T = self.tensors
        grads = [None] * len(T)
        self.snaps = {}
        last = T[self.last_id]
        self._open_streams(last.device)
        # Every tensor but the resized image is the output of ReLU'd convolutions (or a pool of such): the factor (y > 0) of its
        # gradient distributes over the sum of its consumers' contributions, so each consumer applies it to its own contribution in
        # its epilogue (`mask`), and the two gradients that arrive from outside are masked once here.  (A max / average pool's output is
        # zero only where everything it read is zero - positions whose gradient the producer's mask drops anyway.)
        if d_pooled is not None:
            grads[self.last_id] = C.plane_mean_bwd(d_pooled.contiguous(), last.shape[2], last.shape[3])
            C.relu_mask_(grads[self.last_id], last, 0, last.shape[1])
        if d_features is not None:
            grads[self.feat_id] = d_features.contiguous().clone()
            C.relu_mask_(grads[self.feat_id], T[self.feat_id], 0, T[self.feat_id].shape[1])

        def target(i):
            """(gradient buffer of tensor i, whether it already holds a contribution, its ReLU mask)"""
            m = T[i] if i != 0 else None
            if grads[i] is None:
                grads[i] = torch.empty_like(T[i])
                return grads[i], False, m
            return grads[i], True, m
        def one(kind, L, src, dst, coff, s, into=None):
            """The backward of one tape entry on stream index s; `into`: write the contribution (masked, not accumulated) there
            instead of adding it to the gradient of tensor `src`.  False: nothing reached this entry."""
            g = grads[dst]
            if g is None:                       # nothing downstream of this tensor reached the loss
                return False
            if self.keep_grads and dst not in self.snaps:
                if block_src is not None:
                    self._join_streams()
                self.snaps[dst] = g.clone()
                if block_src is not None:
                    self._fork_streams()
            if into is None:
                dx, acc, m = target(src)        # (allocated on the caller's stream, outside the launch's stream scope)
            else:
                dx, acc, m = into, False, (T[src] if src != 0 else None)
            if kind == "conv" and L.class_dgrad_ok(dx.shape[2], dx.shape[3]):
                Bn, _, H, W = dx.shape
                parts = [torch.empty(Bn, L.cin, (H - py + 1) // 2, (W - px + 1) // 2, dtype=torch.float32, device=dx.device)
                         for py, px, *_ in L.cls]
                self._ws_old.extend(parts)          # (alive until the next walk opens: stream s still reads them - see _ws)
                ws = self._ws(max(ops_ws(Bn, L.cin, t.shape[2], t.shape[3], L.cout * c[3] * c[4]) for t, c in zip(parts, L.cls)),
                              dx.device, s)
                with self._run(s):
                    for t, (_py, _px, A, khc, kwc, phc, pwc) in zip(parts, L.cls):
                        C.gconv(True, A, g, coff, L.cout, t, 0, khc, kwc, 1, phc, pwc, None, False, False, ws, None)
                    C.interleave2x2_(parts[0], parts[1], parts[2], parts[3], dx, acc, m)
            elif kind == "conv":
                ws = self._ws(ops_ws(dx.shape[0], L.cin, dx.shape[2], dx.shape[3], L.cout * L.kh * L.kw), dx.device, s)
                with self._run(s):
                    C.gconv(True, L.wd, g, coff, L.cout, dx, 0, L.kh, L.kw, L.stride, L.ph, L.pw, None, False, acc, ws, m)
            elif kind == "maxpool":
                with self._run(s):
                    C.maxpool3s2_bwd(T[src], g, coff, dx, acc, m)
            else:                               # avgpool3: symmetric
                with self._run(s):
                    C.avgpool3(g, dx, acc, m)
            return True

        # A block's backward: fork; every branch's layers on its stream.  The layers that feed the gradient of the block INPUT (the
        # head of every branch) all add to one tensor: each writes its masked contribution into a slot of its own - on its
        # branch's stream, beside the longer branches - and behind the join one pass sums the slots in the tape's order, so the
        # sum has the bits of the one-stream walk's accumulation (which is what runs with one stream, or with
        # TGSR_TRUNK_HEADS_PARALLEL=0: the heads one after the other behind the join).
        slots = HEADS_PARALLEL and len(self._streams) > 1
        block_src, heads, stack, nslot = None, [], None, 0
        for kind, L, src, dst, coff, s in reversed(self.tape):
            if kind == "join":                  # (the end of a block in the forward walk = where its backward begins)
                if grads[dst] is not None:
                    block_src, heads, nslot = src, [], 0
                    if slots:
                        # (allocated and - behind the join - summed on the caller's stream, written in between by side streams the
                        # caller has joined by then: it can go back to the allocator as soon as the sum is queued)
                        stack = torch.empty((5,) + tuple(T[src].shape), dtype=torch.float32, device=T[src].device)
                        if grads[src] is not None:          # a gradient that arrived earlier goes first, as it would in place
                            stack[0].copy_(grads[src])
                            nslot = 1
                    self._fork_streams()
                continue
            if kind == "fork":
                if block_src is not None:
                    self._join_streams()
                    bs, block_src = block_src, None
                    if slots:
                        if nslot == 1 and grads[bs] is None:
                            grads[bs] = stack[0]
                        elif nslot >= 1:
                            if grads[bs] is None:
                                grads[bs] = torch.empty_like(T[bs])
                            C.sum_stack(stack, nslot, grads[bs])
                        stack = None
                    else:
                        for h in heads:
                            one(*h[:5], 0)
                continue
            if block_src is not None and src == block_src:
                if slots:
                    if nslot >= 5:
                        raise TgsrError("a Mixed block with more than four branches reading its input")
                    if one(kind, L, src, dst, coff, self._sidx(s), into=stack[nslot]):
                        nslot += 1
                else:
                    heads.append((kind, L, src, dst, coff))
                continue
            one(kind, L, src, dst, coff, self._sidx(s) if block_src is not None else 0)
        d299 = grads[0]
        if self.keep_grads:
            self.snaps[0] = d299
        if d299 is None:
            return None
        return C.bilinear_bwd(d299, self.in_hw[0], self.in_hw[1])

    keep_grads = False          # diagnostics (tools/debug_trunk.py): keep every tensor's gradient of the last backward

    def take_state(self):
        """Detach the tape of the last forward (tensors, ops, bookkeeping) so that another forward can run before its backward."""
        st = (self.tensors, self.tape, self.feat_id, self.last_id, self.in_hw)
        self.tensors = self.tape = None
        return st

    def put_state(self, st):
        self.tensors, self.tape, self.feat_id, self.last_id, self.in_hw = st


def ops_ws(B, M, PH, PW, K):
    from . import ops
    return ops.gconv_ws_elems(B, M, PH, PW, K)


class TrunkFn(torch.autograd.Function):
    """images -> (features, pooled) through InceptionTrunk with its tape-driven backward (gradient to the images only: the trunk's
    parameters are frozen, util.py:274-275)."""

    @staticmethod
    def forward(ctx, x, runner):
        feats, pooled = runner.forward(x)
        st = runner.take_state()
        ctx.runner = runner
        ctx.state = st if ctx.needs_input_grad[0] else None
        return feats, pooled

    @staticmethod
    def backward(ctx, d_feats, d_pooled):
        if ctx.state is None:
            return None, None
        ctx.runner.put_state(ctx.state)
        ctx.state = None
        # autograd runs this node on the stream its forward ran on.  When the forward was issued early on a stream of its own
        # (train.SRTrainer: beside the discriminator updates), that stream is a forked branch of the step's backward - and the
        # walk's own forks off it would be forks nested in a forked branch, which ROCm 7.2's stream capture does not survive: the
        # caller names the stream the backward walk belongs on (`runner.bwd_stream`: the step's main / capture stream); the node's
        # own stream only hands the gradients over and takes the result back.
        here = torch.cuda.current_stream(d_feats.device if d_feats is not None else d_pooled.device)
        to = ctx.runner.bwd_stream
        try:
            if to is None or int(to.cuda_stream) == int(here.cuda_stream):
                return ctx.runner.backward(d_feats, d_pooled), None
            to.wait_stream(here)
            with torch.cuda.stream(to):
                dx = ctx.runner.backward(d_feats, d_pooled)
            here.wait_stream(to)
            if dx is not None and not torch.cuda.is_current_stream_capturing():
                dx.record_stream(here)           # (allocated on `to`, handed on by `here`)
            return dx, None
        finally:
            ctx.runner.take_state()
