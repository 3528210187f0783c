"""Tensor-level wrappers over the C ABI (include/tgsr_hip.h): validate, allocate outputs with torch, launch on
torch's current HIP stream.  PyTorch is plumbing here (device memory + streams); the arithmetic is in
libtgsr_hip.so.  No function in this file computes on the CPU or through eager torch ops.
"""
import ctypes
import functools
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import TgsrError, check

BN_EPS = 1e-5

# Per-launch measurement hook (bench.py): when `profile` is a list, every conv / attention launch is bracketed
# with HIP events on the launch stream and appended as (kernel, algorithmic flops, algorithmic bytes, e0, e1).
profile = None


def _ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current HIP stream of the current device as a void*.  The raw-handle query costs < 1 us; going through
    torch.cuda.current_stream() costs ~5 us, i.e. 0.25 ms of host time per forward at ~50 launches."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _rows4(stats: torch.Tensor):
    """Pointers of the four rows of a [4, C] fp32 tensor (mean, invstd, scale, shift).  Address arithmetic where the tensor is
    dense - four `stats[i]` views cost ~3 us of host time per call, on ~150 BatchNorm calls of a host-bound train step."""
    if stats.dim() == 2 and stats.shape[0] == 4 and stats.is_contiguous() and stats.dtype == torch.float32:
        base, step = stats.data_ptr(), stats.shape[1] * 4
        return (ctypes.c_void_p(base), ctypes.c_void_p(base + step), ctypes.c_void_p(base + 2 * step), ctypes.c_void_p(base + 3 * step))
    return _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3])


def _need_hip(*ts):
    """Every tensor on ONE HIP device, and that device the current one: the kernels launch on the current device's
    stream (`_stream()`), so a tensor living elsewhere would be read through a foreign pointer on the wrong queue."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise TgsrError("tgsr_amd ops run only on HIP tensors (got a %s tensor); there is no CPU fallback"
                            % t.device.type)
        if dev is None:
            dev = t.device.index
        elif t.device.index != dev:
            raise TgsrError("tgsr_amd op got tensors on different devices (cuda:%d and cuda:%d)" % (dev, t.device.index))
    if dev is not None and dev != torch.cuda.current_device():
        raise TgsrError("tensors live on cuda:%d but the current device is cuda:%d: run the call under "
                        "`with torch.cuda.device(%d):` (SRPipeline / the trainers do)" % (dev, torch.cuda.current_device(), dev))


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TgsrError("%s must be float32, got %s" % (name, t.dtype))
    return t


def _nchw_bstride(t: torch.Tensor, name: str) -> Tuple[torch.Tensor, int]:
    """Accept [B,C,H,W] with a dense (C,H,W) block and ANY batch stride (channel-slice views); else copy."""
    B, C, H, W = t.shape
    if t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W:
        return t, t.stride(0) if B > 1 else C * H * W
    t = t.contiguous()
    return t, C * H * W


# ----------------------------------------------------------------------------------------- weight prep
@functools.lru_cache(maxsize=None)
def _pack_elems(kind: str, cout: int, cin: int) -> int:
    L = _lib.lib()
    return int({"conv": lambda: L.tgsr_packed_weight_elems(cout, cin, 3), "wino": lambda: L.tgsr_packed_wino_weight_elems(cout, cin),
                "wino4": lambda: L.tgsr_packed_wino4_weight_elems(cout, cin),
                "upconv": lambda: L.tgsr_packed_upconv_weight_elems(cout, cin),
                "upwino": lambda: L.tgsr_packed_upwino_weight_elems(cout, cin),
                "upwino4": lambda: L.tgsr_packed_upwino4_weight_elems(cout, cin)}[kind]())


def _check_pack(what: str, kind: str, pack: torch.Tensor, cout: int, cin: int):
    """The kernels walk the packed filter by (Cout, Cin of the INPUT): a pack made for another channel count would be
    read past its end.  torch raises a shape error for the same mistake (weight [Cout, Cin', 3, 3] on a Cin-channel
    input); so does this."""
    if pack.numel() != _pack_elems(kind, cout, cin):
        raise TgsrError("%s: the packed weight holds %d values, not the %d of a [%d, %d, 3, 3] filter - the input has %d "
                        "channels, the weight was packed for another count" % (what, pack.numel(), _pack_elems(kind, cout, cin),
                                                                              cout, cin, cin))


def pack_conv3x3_weight(w: torch.Tensor, dgrad: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,3,3] -> the [ceil(Cin/4)][9][4][Cout] stream order of tgsr_conv3x3_fwd.  `dgrad`: w is the forward
    conv's weight and the pack is of its data-gradient conv (in/out swapped, taps flipped) - no flip/transpose copies."""
    _need_hip(w)
    w = _f32(w.detach(), "weight").contiguous()
    Cout, Cin, K, K2 = w.shape
    assert K == 3 and K2 == 3
    if dgrad:
        Cout, Cin = Cin, Cout
    L = _lib.lib()
    out = _pack_out(out, L.tgsr_packed_weight_elems(Cout, Cin, 3), w.device)
    if dgrad:
        check(L.tgsr_pack_conv_weight_dgrad(_p(w), _p(out), Cout, Cin, 3, _stream()), "tgsr_pack_conv_weight_dgrad")
    else:
        check(L.tgsr_pack_conv_weight(_p(w), _p(out), Cout, Cin, 3, _stream()), "tgsr_pack_conv_weight")
    return out


def bn_fold(weight, bias, running_mean, running_var, eps: float = BN_EPS):
    """BatchNorm2d(eval) -> (scale, shift) per channel."""
    _need_hip(weight, bias, running_mean, running_var)
    C = weight.numel()
    ts = [_f32(t.detach(), "bn tensor").contiguous() for t in (weight, bias, running_mean, running_var)]
    scale = torch.empty(C, dtype=torch.float32, device=weight.device)
    shift = torch.empty_like(scale)
    check(_lib.lib().tgsr_bn_fold(_p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]), eps, _p(scale), _p(shift), C, _stream()),
          "tgsr_bn_fold")
    return scale, shift


# ----------------------------------------------------------------------------------------- convolutions
def conv3x3_fused(x: torch.Tensor, wpack: torch.Tensor, cout: int, scale: Optional[torch.Tensor],
                  shift: Optional[torch.Tensor], glu: bool = False, upsample: bool = False,
                  residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3 [+nearest x2 in front] + per-channel affine + (GLU | residual add) in one launch.
    `out` may be a channel-slice view of a wider buffer (dense C,H,W block, any batch stride)."""
    _need_hip(x, wpack, scale, shift, residual, out)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    B, Cin, H, W = x.shape
    _check_pack("conv3x3_fused", "conv", wpack, cout, Cin)
    Ho, Wo = (2 * H, 2 * W) if upsample else (H, W)
    co = cout // 2 if glu else cout
    if out is None:
        out = torch.empty(B, co, Ho, Wo, dtype=torch.float32, device=x.device)
    if tuple(out.shape) != (B, co, Ho, Wo) or out.stride(3) != 1 or out.stride(2) != Wo or out.stride(1) != Ho * Wo:
        raise TgsrError("conv3x3_fused: bad `out` shape/strides %s %s" % (tuple(out.shape), out.stride()))
    obs = out.stride(0) if B > 1 else co * Ho * Wo
    rbs = 0
    if residual is not None:
        if glu:
            raise TgsrError("conv3x3_fused: residual with GLU is not a reference pattern")
        residual, rbs = _nchw_bstride(_f32(residual, "residual"), "residual")
        if tuple(residual.shape) != (B, co, Ho, Wo):
            raise TgsrError("conv3x3_fused: residual shape %s" % (tuple(residual.shape),))
    e0 = _ev() if profile is not None else None
    rc = _lib.lib().tgsr_conv3x3_fwd(_p(x), xbs, B, Cin, H, W, _p(wpack), cout, _p(scale), _p(shift), _p(residual), rbs,
                                     _p(out), obs, _lib.EPI_AFFINE_GLU if glu else _lib.EPI_AFFINE,
                                     1 if upsample else 0, _stream())
    check(rc, "tgsr_conv3x3_fwd")
    if profile is not None:
        nbytes = 4 * (B * Cin * H * W + B * co * Ho * Wo * (2 if residual is not None else 1) + cout * Cin * 9)
        profile.append(("conv3x3_mfma_kernel", 2.0 * B * Ho * Wo * cout * Cin * 9, nbytes, e0, _ev()))
    return out


def _pack_out(out, n, dev):
    """A pack's destination: fresh, or the caller's persistent buffer (autograd.PackCache re-packs in place)."""
    if out is None:
        return torch.empty(n, dtype=torch.float32, device=dev)
    if out.numel() != n or out.dtype != torch.float32 or out.device != dev or not out.is_contiguous():
        raise TgsrError("pack: out %s %s does not hold %d floats" % (tuple(out.shape), out.dtype, n))
    return out


def pack_wino_weight(w: torch.Tensor, glu: bool = False, dgrad: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,3,3] -> Winograd F(2x2,3x3) transformed weights [ceil(Cin/8)][Cout/64][16 pos][8][64]; `glu` must
    match the epilogue the pack is used with (it groups value channels with their gate channels).  `dgrad`: w is the
    forward conv's weight, the pack is of its data-gradient conv (see pack_conv3x3_weight)."""
    _need_hip(w)
    w = _f32(w.detach(), "weight").contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    if dgrad:
        Cout, Cin = Cin, Cout
    L = _lib.lib()
    out = _pack_out(out, L.tgsr_packed_wino_weight_elems(Cout, Cin), w.device)
    if dgrad:
        assert not glu
        check(L.tgsr_pack_wino_weight_dgrad(_p(w), _p(out), Cout, Cin, _stream()), "tgsr_pack_wino_weight_dgrad")
    else:
        check(L.tgsr_pack_wino_weight(_p(w), _p(out), Cout, Cin, 1 if glu else 0, _stream()), "tgsr_pack_wino_weight")
    return out


def wino_supported(x: torch.Tensor, cout: int, out: Optional[torch.Tensor] = None,
                   residual: Optional[torch.Tensor] = None) -> bool:
    """Shapes the Winograd kernel takes: Cout % 32 == 0, Cin % 4 == 0, width % 4 == 0, x planes 16-byte aligned,
    out / residual 8-byte aligned with even batch strides."""
    if not (x.dim() == 4 and cout % 32 == 0 and x.shape[1] % 4 == 0 and x.shape[3] % 4 == 0 and
            x.data_ptr() % 16 == 0 and (x.shape[0] == 1 or x.stride(0) % 4 == 0)):
        return False
    for t in (out, residual):
        if t is not None and (t.data_ptr() % 8 != 0 or (t.shape[0] > 1 and t.stride(0) % 2 != 0)):
            return False
    return True


def conv3x3_wino(x: torch.Tensor, upack: torch.Tensor, cout: int, scale, shift, glu: bool = False,
                 residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3 + affine + (GLU | residual) by Winograd F(2x2,3x3) (same contract as conv3x3_fused, no upsample)."""
    _need_hip(x, upack, scale, shift, residual, out)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    B, Cin, H, W = x.shape
    _check_pack("conv3x3_wino", "wino", upack, cout, Cin)
    co = cout // 2 if glu else cout
    if out is None:
        out = torch.empty(B, co, H, W, dtype=torch.float32, device=x.device)
    if tuple(out.shape) != (B, co, H, W) or out.stride(3) != 1 or out.stride(2) != W or out.stride(1) != H * W:
        raise TgsrError("conv3x3_wino: bad `out` shape/strides %s %s" % (tuple(out.shape), out.stride()))
    obs = out.stride(0) if B > 1 else co * H * W
    rbs = 0
    if residual is not None:
        residual, rbs = _nchw_bstride(_f32(residual, "residual"), "residual")
    e0 = _ev() if profile is not None else None
    rc = _lib.lib().tgsr_wino_conv3x3_fwd(_p(x), xbs, B, Cin, H, W, _p(upack), cout, _p(scale), _p(shift), _p(residual),
                                          rbs, _p(out), obs, _lib.EPI_AFFINE_GLU if glu else _lib.EPI_AFFINE, _stream())
    check(rc, "tgsr_wino_conv3x3_fwd")
    if profile is not None:
        nbytes = 4 * (B * Cin * H * W + B * co * H * W * (2 if residual is not None else 1) + cout * Cin * 9)
        profile.append(("wino_conv3x3_kernel", 2.0 * B * H * W * cout * Cin * 9, nbytes, e0, _ev()))
    return out


WINO4_MIN_PIXELS = 64 * 64        # per image; profiles/HISTORY.md 3.1e: the error study that keeps F(4x4) off the 32 x 32 layers
WINO4_MIN_PIXELS_64 = 128 * 128   # pixels per image from which layers with 64-channel groups only may use F(4x4) too
# The up-sample-aware F(4x4) form must keep +-1 among its interpolation points (its 25-of-36 structure depends on them,
# tgsr_upwino4.hip), so it does not get the better-conditioned points of tgsr_winograd4.hip: its OWN error is 1.9-2.2x the CPU
# fp32 op's on the same input (tests/test_hip_parity_margin.py, profiles/HISTORY.md 3.1g).  It therefore serves only upBlocks whose output is
# the generator's finest feature map (>= 256 x 256: nothing but an image head reads it); an upBlock in mid-network feeds the
# 128^2 stages that amplify whatever error they are handed.
UPWINO4_MIN_OUT_PIXELS = 256 * 256
WINO4_MIN_WORKGROUPS = 256        # below a full round of its (large) workgroup tiles F(2x2)'s four times smaller ones win


class _Routing:
    """Which layers go to the F(4x4) kernels.  Read from the environment ONCE, at import (the eager host path asks per layer and
    per forward); tests and tools/ change the attributes of `ops.ROUTING` instead.
      TGSR_WINO4=0                every layer stays on F(2x2)
      TGSR_WINO4_MIN_WG=<n>       the work rule's threshold (n = 32 gives the batch-2 golden case the routing of batch 16)
      TGSR_WINO4_PIN_BATCH=<b>    decide the work rule as if the batch were b: the same kernels - bit-identical images per
                                  sample - whatever the batch size (reproducibility knob; default: the real batch)
    diagnostics (tools/diag_precision_classes.py): min_cin / min_pixels / upwino4_min_cin switch layer classes off."""

    def __init__(self, env=os.environ):
        self.wino4 = env.get("TGSR_WINO4", "1") != "0"
        self.min_workgroups = int(env.get("TGSR_WINO4_MIN_WG", WINO4_MIN_WORKGROUPS))
        self.pin_batch = int(env.get("TGSR_WINO4_PIN_BATCH", "0"))
        self.min_cin = int(env.get("TGSR_WINO4_MIN_CIN", "0"))
        self.min_pixels = int(env.get("TGSR_WINO4_MIN_PIXELS", "0"))
        self.upwino4_min_cin = int(env.get("TGSR_UPWINO4_MIN_CIN", "0"))

    def reset(self, env=os.environ):
        self.__init__(env)
        return self


ROUTING = _Routing()


def wino4_wanted(cin: int, cout: int, H: int, W: int, B: int = 16) -> bool:
    """Does a conv3x3 layer go to the F(4x4, 3x3) kernels?  Three rules.
    Shape: Cout % 64, Cin % 4, W % 64, H % 8 (whole workgroup tiles).
    Numerics (profiles/HISTORY.md 3.1e / 3.1g; tests/test_hip_parity.py::test_fp32_parity_margin_*): layers of >= 128 x 128
    pixels; at 64 x 64 .. 128 x 128 only the convolutions with 128-channel groups; nothing below 64 x 64 (the early, small layers
    are the ones whose error the rest of the network amplifies).
    Work: >= 256 workgroups of the form the layer takes - register-fed (Cin % 8 == 0): 4 x 64 pixels x 128 rows, or x 64 rows
    where Cout % 128 != 0; else the LDS-fed form, 8 x 64 x 64 (measured at batch 4 .. 32: each routed layer faster than on
    F(2x2), each unrouted one slower; the batch-2 golden case stays on F(2x2) except for the last upBlocks).
    `ROUTING` holds the switches (environment, read at import)."""
    R = ROUTING
    if not R.wino4:
        return False
    if not (cout % 64 == 0 and cin % 4 == 0 and W % 64 == 0 and H % 8 == 0 and H * W >= WINO4_MIN_PIXELS):
        return False
    if cin < R.min_cin or H * W < R.min_pixels:          # diagnostics
        return False
    if H * W < WINO4_MIN_PIXELS_64 and cout % 128 != 0:
        return False
    if R.pin_batch:
        B = R.pin_batch
    if cin % 8 == 0:
        nwg = B * (H // 4) * (W // 64) * (cout // (128 if cout % 128 == 0 else 64))
    else:
        nwg = B * (H // 8) * (W // 64) * (cout // 64)
    return nwg >= R.min_workgroups


def pack_wino4_weight(w: torch.Tensor, glu: bool = False, dgrad: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,3,3] -> Winograd F(4x4,3x3) transformed weights [Cin/4][Cout/64][9 quads][4 ci][64 rows][4] (U = G g G^T in
    double, rounded once); `glu` must match the epilogue the pack is used with.  `dgrad`: w is the forward conv's weight, the
    pack is of its data-gradient conv (see pack_conv3x3_weight)."""
    _need_hip(w)
    w = _f32(w.detach(), "weight").contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    if dgrad:
        Cout, Cin = Cin, Cout
    L = _lib.lib()
    out = _pack_out(out, L.tgsr_packed_wino4_weight_elems(Cout, Cin), w.device)
    if dgrad:
        assert not glu
        check(L.tgsr_pack_wino4_weight_dgrad(_p(w), _p(out), Cout, Cin, _stream()), "tgsr_pack_wino4_weight_dgrad")
    else:
        check(L.tgsr_pack_wino4_weight(_p(w), _p(out), Cout, Cin, 1 if glu else 0, _stream()), "tgsr_pack_wino4_weight")
    return out


def pack_wino4w_weight(w: torch.Tensor, glu: bool = False, dgrad: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The pack of the register-fed F(4x4,3x3) kernel: per-wave fragment order [Cin/4][groups][8 | 4 blocks][9 quads][64 lanes][4]
    (128-row groups where Cout % 128 == 0, else 64-row groups) - not interchangeable with pack_wino4_weight's.  `dgrad`: w is the
    forward conv's weight, the pack is of its data-gradient conv."""
    _need_hip(w)
    w = _f32(w.detach(), "weight").contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    if dgrad:
        Cout, Cin = Cin, Cout
    L = _lib.lib()
    out = _pack_out(out, L.tgsr_packed_wino4_weight_elems(Cout, Cin), w.device)
    if dgrad:
        assert not glu
        check(L.tgsr_pack_wino4_wide_weight_dgrad(_p(w), _p(out), Cout, Cin, _stream()), "tgsr_pack_wino4_wide_weight_dgrad")
    else:
        check(L.tgsr_pack_wino4_wide_weight(_p(w), _p(out), Cout, Cin, 1 if glu else 0, _stream()), "tgsr_pack_wino4_wide_weight")
    return out


def conv3x3_wino4(x: torch.Tensor, upack: torch.Tensor, cout: int, scale, shift, glu: bool = False,
                  residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, wide: bool = False) -> torch.Tensor:
    """conv3x3 + affine + (GLU | residual) by Winograd F(4x4,3x3) (same contract as conv3x3_wino; 16-byte aligned tensors).
    `wide`: the register-fed form (Cin % 8 == 0), upack from pack_wino4w_weight."""
    _need_hip(x, upack, scale, shift, residual, out)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    B, Cin, H, W = x.shape
    _check_pack("conv3x3_wino4", "wino4", upack, cout, Cin)
    co = cout // 2 if glu else cout
    if out is None:
        out = torch.empty(B, co, H, W, dtype=torch.float32, device=x.device)
    if tuple(out.shape) != (B, co, H, W) or out.stride(3) != 1 or out.stride(2) != W or out.stride(1) != H * W:
        raise TgsrError("conv3x3_wino4: bad `out` shape/strides %s %s" % (tuple(out.shape), out.stride()))
    obs = out.stride(0) if B > 1 else co * H * W
    rbs = 0
    if residual is not None:
        residual, rbs = _nchw_bstride(_f32(residual, "residual"), "residual")
        if tuple(residual.shape) != (B, co, H, W):
            raise TgsrError("conv3x3_wino4: residual shape %s" % (tuple(residual.shape),))
    e0 = _ev() if profile is not None else None
    fn = _lib.lib().tgsr_wino4_wide_conv3x3_fwd if wide else _lib.lib().tgsr_wino4_conv3x3_fwd
    rc = fn(_p(x), xbs, B, Cin, H, W, _p(upack), cout, _p(scale), _p(shift), _p(residual), rbs, _p(out), obs,
            _lib.EPI_AFFINE_GLU if glu else _lib.EPI_AFFINE, _stream())
    check(rc, "tgsr_wino4_wide_conv3x3_fwd" if wide else "tgsr_wino4_conv3x3_fwd")
    if profile is not None:
        nbytes = 4 * (B * Cin * H * W + B * co * H * W * (2 if residual is not None else 1) + cout * Cin * 9)
        profile.append(("wino4w_conv3x3_kernel" if wide else "wino4_conv3x3_kernel", 2.0 * B * H * W * cout * Cin * 9, nbytes, e0, _ev()))
    return out


def wino4_stats_nslots(B: int, H: int, W: int, cout: int, wide: bool = False) -> int:
    """Partial-sum pairs per channel conv3x3_wino4_stats writes for this shape (0: unsupported)."""
    L = _lib.lib()
    return int((L.tgsr_wino4_wide_stats_nslots if wide else L.tgsr_wino4_stats_nslots)(int(B), int(H), int(W), int(cout)))


def conv3x3_wino4_stats(x: torch.Tensor, upack: torch.Tensor, cout: int, wide: bool = False):
    """conv3x3_wino_stats on the F(4x4, 3x3) kernels: (out [B,cout,H,W], stat_partial [cout, nslots, 2]).  `wide`: the
    register-fed form (upack from pack_wino4w_weight)."""
    _need_hip(x, upack)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    B, Cin, H, W = x.shape
    _check_pack("conv3x3_wino4_stats", "wino4", upack, cout, Cin)
    L = _lib.lib()
    nslots = (L.tgsr_wino4_wide_stats_nslots if wide else L.tgsr_wino4_stats_nslots)(B, H, W, cout)
    if nslots < 1:
        raise TgsrError("conv3x3_wino4_stats: unsupported shape %s -> %d channels" % (tuple(x.shape), cout))
    out = torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
    part = torch.empty(cout, nslots, 2, dtype=torch.float32, device=x.device)
    e0 = _ev() if profile is not None else None
    fn = L.tgsr_wino4_wide_conv3x3_stats_fwd if wide else L.tgsr_wino4_conv3x3_stats_fwd
    check(fn(_p(x), xbs, B, Cin, H, W, _p(upack), cout, _p(out), cout * H * W, _p(part), _stream()),
          "tgsr_wino4_wide_conv3x3_stats_fwd" if wide else "tgsr_wino4_conv3x3_stats_fwd")
    if profile is not None:
        profile.append(("wino4w_conv3x3_kernel" if wide else "wino4_conv3x3_kernel", 2.0 * B * H * W * cout * Cin * 9,
                        4 * (B * Cin * H * W + B * cout * H * W + cout * Cin * 9), e0, _ev()))
    return out, part


def wino_stats_nslots(B: int, H: int, W: int, cout: int) -> int:
    """Partial-sum pairs per channel conv3x3_wino_stats writes for this shape (0: unsupported)."""
    return int(_lib.lib().tgsr_wino_stats_nslots(int(B), int(H), int(W), int(cout)))


def conv3x3_wino_stats(x: torch.Tensor, upack: torch.Tensor, cout: int):
    """The raw Winograd convolution (no affine, no residual) whose epilogue also leaves BatchNorm's batch statistics as
    partial sums: returns (out [B,cout,H,W], stat_partial [cout, nslots, 2]) for bn_train_fwd(..., stat_partial=...)."""
    _need_hip(x, upack)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    B, Cin, H, W = x.shape
    _check_pack("conv3x3_wino_stats", "wino", upack, cout, Cin)
    L = _lib.lib()
    nslots = L.tgsr_wino_stats_nslots(B, H, W, cout)
    if nslots < 1:
        raise TgsrError("conv3x3_wino_stats: unsupported shape %s -> %d channels" % (tuple(x.shape), cout))
    out = torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
    part = torch.empty(cout, nslots, 2, dtype=torch.float32, device=x.device)
    e0 = _ev() if profile is not None else None
    check(L.tgsr_wino_conv3x3_stats_fwd(_p(x), xbs, B, Cin, H, W, _p(upack), cout, _p(out), cout * H * W, _p(part), _stream()),
          "tgsr_wino_conv3x3_stats_fwd")
    if profile is not None:
        profile.append(("wino_conv3x3_kernel", 2.0 * B * H * W * cout * Cin * 9,
                        4 * (B * Cin * H * W + B * cout * H * W + cout * Cin * 9), e0, _ev()))
    return out, part


def pack_upconv_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [ceil(Cin/4)][4 phases][4 taps][4][Cout] with the sub-pixel tap sums (tgsr_upconv3x3_glu_fwd)."""
    _need_hip(w)
    w = _f32(w.detach(), "weight").contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    out = torch.empty(L.tgsr_packed_upconv_weight_elems(Cout, Cin), dtype=torch.float32, device=w.device)
    check(L.tgsr_pack_upconv_weight(_p(w), _p(out), Cout, Cin, _stream()), "tgsr_pack_upconv_weight")
    return out


def upconv3x3_glu(x: torch.Tensor, wpack_up: torch.Tensor, cout: int, scale, shift,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """upBlock in one launch by sub-pixel decomposition (4 x 2x2 convs on the pre-upsample tensor)."""
    _need_hip(x, wpack_up, scale, shift, out)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    B, Cin, H, W = x.shape
    _check_pack("upconv3x3_glu", "upconv", wpack_up, cout, Cin)
    co, Ho, Wo = cout // 2, 2 * H, 2 * W
    if out is None:
        out = torch.empty(B, co, Ho, Wo, dtype=torch.float32, device=x.device)
    if tuple(out.shape) != (B, co, Ho, Wo) or out.stride(3) != 1 or out.stride(2) != Wo or out.stride(1) != Ho * Wo:
        raise TgsrError("upconv3x3_glu: bad `out` shape/strides %s %s" % (tuple(out.shape), out.stride()))
    obs = out.stride(0) if B > 1 else co * Ho * Wo
    e0 = _ev() if profile is not None else None
    rc = _lib.lib().tgsr_upconv3x3_glu_fwd(_p(x), xbs, B, Cin, H, W, _p(wpack_up), cout, _p(scale), _p(shift), _p(out),
                                           obs, _stream())
    check(rc, "tgsr_upconv3x3_glu_fwd")
    if profile is not None:
        nbytes = 4 * (B * Cin * H * W + B * co * Ho * Wo + cout * Cin * 9)
        profile.append(("upconv_glu_mfma_kernel", 2.0 * B * Ho * Wo * cout * Cin * 9, nbytes, e0, _ev()))
    return out


def pack_upwino_weight(w: torch.Tensor, glu: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,3,3] -> the 9 tap-sum positions of the up-sample-aware Winograd form (tgsr_upwino_glu_fwd; glu=False:
    the layout of tgsr_upwino_fwd)."""
    _need_hip(w)
    w = _f32(w.detach(), "weight").contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    out = _pack_out(out, L.tgsr_packed_upwino_weight_elems(Cout, Cin), w.device)
    check(L.tgsr_pack_upwino_weight(_p(w), _p(out), Cout, Cin, 1 if glu else 0, _stream()), "tgsr_pack_upwino_weight")
    return out


def upwino_supported(x: torch.Tensor, cout: int, out: Optional[torch.Tensor] = None) -> bool:
    """Shapes tgsr_upwino_glu_fwd takes: Cout % 64 == 0, Cin % 4 == 0, W % 4 == 0, aligned planes."""
    if not (x.dim() == 4 and cout % 64 == 0 and x.shape[1] % 4 == 0 and x.shape[3] % 4 == 0 and
            x.data_ptr() % 16 == 0 and (x.shape[0] == 1 or x.stride(0) % 4 == 0)):
        return False
    return out is None or (out.data_ptr() % 8 == 0 and (out.shape[0] == 1 or out.stride(0) % 2 == 0))


def upwino_glu(x: torch.Tensor, upack: torch.Tensor, cout: int, scale, shift,
               out: Optional[torch.Tensor] = None, glu: bool = True) -> torch.Tensor:
    """upBlock in one launch by Winograd on the up-sampled grid (9 products per 2x2 outputs); contract of upconv3x3_glu.
    glu=False: Upsample -> conv3x3 -> affine without the gate (out [B,cout,2H,2W]; scale/shift may be None)."""
    _need_hip(x, upack, scale, shift, out)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    B, Cin, H, W = x.shape
    _check_pack("upwino_glu", "upwino", upack, cout, Cin)
    co, Ho, Wo = (cout // 2 if glu else cout), 2 * H, 2 * W
    if out is None:
        out = torch.empty(B, co, Ho, Wo, dtype=torch.float32, device=x.device)
    if tuple(out.shape) != (B, co, Ho, Wo) or out.stride(3) != 1 or out.stride(2) != Wo or out.stride(1) != Ho * Wo:
        raise TgsrError("upwino_glu: bad `out` shape/strides %s %s" % (tuple(out.shape), out.stride()))
    obs = out.stride(0) if B > 1 else co * Ho * Wo
    e0 = _ev() if profile is not None else None
    fn = _lib.lib().tgsr_upwino_glu_fwd if glu else _lib.lib().tgsr_upwino_fwd
    rc = fn(_p(x), xbs, B, Cin, H, W, _p(upack), cout, _p(scale), _p(shift), _p(out), obs, _stream())
    check(rc, "tgsr_upwino_glu_fwd" if glu else "tgsr_upwino_fwd")
    if profile is not None:
        nbytes = 4 * (B * Cin * H * W + B * co * Ho * Wo + cout * Cin * 9)
        profile.append(("upwino_glu_kernel", 2.0 * B * Ho * Wo * cout * Cin * 9, nbytes, e0, _ev()))
    return out


def upwino4_wanted(cin: int, cout: int, H: int, W: int, B: int = 16) -> bool:
    """Does an upBlock (low-resolution input H x W) go to the F(4x4) form of the up-sample-aware kernel?  Shape support
    (Cout % 64, Cin % 8: an even number of 4-channel stages, whole 4 x 64 OUTPUT tiles), a numerics floor on the OUTPUT
    (>= 256 x 256 pixels: the generator's last upBlock, read by an image head only - see UPWINO4_MIN_OUT_PIXELS) and at least 256
    of its 4-wave workgroups.  `ROUTING.wino4 = False` (TGSR_WINO4=0) keeps the F(2x2) form."""
    R = ROUTING
    if not R.wino4:
        return False
    Ho, Wo = 2 * H, 2 * W
    if cin < R.upwino4_min_cin:                            # diagnostics: which upBlocks cost what
        return False
    if not (cout % 64 == 0 and cin % 8 == 0 and Wo % 64 == 0 and Ho % 4 == 0 and Ho * Wo >= UPWINO4_MIN_OUT_PIXELS):
        return False
    if R.pin_batch:
        B = R.pin_batch
    return B * (Ho // 4) * (Wo // 64) * (cout // 64) >= R.min_workgroups


def pack_upwino4_weight(w: torch.Tensor, glu: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,3,3] -> the 25 live positions of the up-sample-aware F(4x4,3x3) form (tgsr_upwino4_fwd), the -1/3 factors of
    the fifth transformed row / column folded in; computed in double, rounded once."""
    _need_hip(w)
    w = _f32(w.detach(), "weight").contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    out = _pack_out(out, L.tgsr_packed_upwino4_weight_elems(Cout, Cin), w.device)
    check(L.tgsr_pack_upwino4_weight(_p(w), _p(out), Cout, Cin, 1 if glu else 0, _stream()), "tgsr_pack_upwino4_weight")
    return out


def upwino4_glu(x: torch.Tensor, upack: torch.Tensor, cout: int, scale, shift,
                out: Optional[torch.Tensor] = None, glu: bool = True) -> torch.Tensor:
    """upBlock in one launch by F(4x4,3x3) on the up-sampled grid (25 products per 4x4 outputs); contract of upwino_glu with
    16-byte aligned tensors."""
    _need_hip(x, upack, scale, shift, out)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    B, Cin, H, W = x.shape
    _check_pack("upwino4_glu", "upwino4", upack, cout, Cin)
    co, Ho, Wo = (cout // 2 if glu else cout), 2 * H, 2 * W
    if out is None:
        out = torch.empty(B, co, Ho, Wo, dtype=torch.float32, device=x.device)
    if tuple(out.shape) != (B, co, Ho, Wo) or out.stride(3) != 1 or out.stride(2) != Wo or out.stride(1) != Ho * Wo:
        raise TgsrError("upwino4_glu: bad `out` shape/strides %s %s" % (tuple(out.shape), out.stride()))
    obs = out.stride(0) if B > 1 else co * Ho * Wo
    e0 = _ev() if profile is not None else None
    rc = _lib.lib().tgsr_upwino4_fwd(_p(x), xbs, B, Cin, H, W, _p(upack), cout, _p(scale), _p(shift), _p(out), obs,
                                     1 if glu else 0, _stream())
    check(rc, "tgsr_upwino4_fwd")
    if profile is not None:
        nbytes = 4 * (B * Cin * H * W + B * co * Ho * Wo + cout * Cin * 9)
        profile.append(("upwino4_kernel", 2.0 * B * Ho * Wo * cout * Cin * 9, nbytes, e0, _ev()))
    return out


def conv_to3(x: torch.Tensor, w: torch.Tensor, tanh_axpy: bool = False, addend: Optional[torch.Tensor] = None,
             alpha: float = 0.0) -> torch.Tensor:
    """KxK (3|5) conv to 3 channels; tanh_axpy: tanh(conv) + alpha * addend."""
    _need_hip(x, w, addend)
    x, xbs = _nchw_bstride(_f32(x, "x"), "x")
    w = _f32(w.detach(), "w").contiguous()
    B, Cin, H, W = x.shape
    if w.shape[0] != 3 or w.shape[1] != Cin or w.shape[2] != w.shape[3]:
        raise TgsrError("conv_to3: weight shape %s" % (tuple(w.shape),))
    if addend is not None:
        addend = _f32(addend, "addend").contiguous()
        if tuple(addend.shape) != (B, 3, H, W):
            raise TgsrError("conv_to3: addend shape %s" % (tuple(addend.shape),))
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
    e0 = _ev() if profile is not None else None
    rc = _lib.lib().tgsr_conv_to3_fwd(_p(x), xbs, B, Cin, H, W, _p(w), int(w.shape[2]),
                                      _lib.ACT_TANH_AXPY if tanh_axpy else _lib.ACT_NONE, _p(addend), float(alpha),
                                      _p(out), _stream())
    check(rc, "tgsr_conv_to3_fwd")
    if profile is not None:
        K = int(w.shape[2])
        nbytes = 4 * (B * Cin * H * W + B * 3 * H * W * (2 if addend is not None else 1))
        profile.append(("conv_to3_kernel", 2.0 * B * H * W * 3 * Cin * K * K, nbytes, e0, _ev()))
    return out


# ----------------------------------------------------------------------------------------- attention
_MASK_CACHE = [None, None, None]   # (tensor id/version key, source (kept alive), uint8 copy)


def _mask_u8(mask: torch.Tensor) -> torch.Tensor:
    """bool -> uint8 once per mask tensor (the three generator stages share one mask)."""
    if mask.dtype == torch.uint8 and mask.is_contiguous():
        return mask
    if mask.dtype == torch.bool and mask.is_contiguous():
        return mask.view(torch.uint8)                    # torch.bool storage is one byte 0 / 1: no cast kernel
    key = (mask.data_ptr(), mask._version, tuple(mask.shape), mask.dtype)
    if _MASK_CACHE[0] != key or _MASK_CACHE[1] is not mask:
        _MASK_CACHE[0], _MASK_CACHE[1], _MASK_CACHE[2] = key, mask, mask.to(torch.uint8).contiguous()
    return _MASK_CACHE[2]


def word_project(words: torch.Tensor, w_ctxs) -> list:
    """The conv1x1 of GlobalAttentionGeneral (GlobalAttention.py:100-102) for up to 4 weight sets over the same words in
    one launch: words [B,cdf,T], w_ctxs = list of [idf,cdf(,1,1)] -> list of src [B,idf,32] (zero padded words),
    to be handed to word_attention(src=...)."""
    import ctypes
    _need_hip(words, *w_ctxs)
    words = _f32(words, "words").contiguous()
    B, cdf, T = words.shape
    n = len(w_ctxs)
    idf = w_ctxs[0].shape[0]
    ws = [_f32(w.detach(), "w_ctx").reshape(idf, cdf).contiguous() for w in w_ctxs]
    out = torch.empty(n, B, idf, 32, dtype=torch.float32, device=words.device)
    ptrs = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    check(_lib.lib().tgsr_word_project_fwd(_p(words), ptrs, n, B, idf, cdf, T, _p(out), _stream()),
          "tgsr_word_project_fwd")
    return [out[i] for i in range(n)]


def word_attention(h: torch.Tensor, words: torch.Tensor, w_ctx: torch.Tensor, mask: Optional[torch.Tensor],
                   correct_mask: bool = False, out: Optional[torch.Tensor] = None, need_attn: bool = True,
                   src: Optional[torch.Tensor] = None):
    """GlobalAttentionGeneral.forward: h [B,idf,ih,iw], words [B,cdf,T], w_ctx [idf,cdf(,1,1)], mask bool [B,T].
    Returns (c_code [B,idf,ih,iw], attn [B,T,ih,iw]).  `src` = this layer's word_project output (skips the projection)."""
    _need_hip(h, words, w_ctx, mask, out)
    h, hbs = _nchw_bstride(_f32(h, "h"), "h")
    B, idf, ih, iw = h.shape
    words = _f32(words, "words").contiguous()
    cdf, T = words.shape[1], words.shape[2]
    w2 = _f32(w_ctx.detach(), "w_ctx").reshape(idf, cdf).contiguous()
    m8 = None
    if mask is not None:
        if tuple(mask.shape) != (B, T):
            raise TgsrError("word_attention: mask shape %s, expected %s" % (tuple(mask.shape), (B, T)))
        m8 = _mask_u8(mask)
    Q = ih * iw
    if out is None:
        out = torch.empty(B, idf, ih, iw, dtype=torch.float32, device=h.device)
    if tuple(out.shape) != (B, idf, ih, iw) or out.stride(3) != 1 or out.stride(2) != iw or out.stride(1) != Q:
        raise TgsrError("word_attention: bad `out`")
    cbs = out.stride(0) if B > 1 else idf * Q
    attn = torch.empty(B, T, ih, iw, dtype=torch.float32, device=h.device) if need_attn else None
    if src is not None:
        if tuple(src.shape) != (B, idf, 32) or not src.is_contiguous():
            raise TgsrError("word_attention: bad `src` %s" % (tuple(src.shape),))
        ws, wp, w2p = src, None, None
    else:
        ws, wp, w2p = torch.empty(B * idf * 32, dtype=torch.float32, device=h.device), _p(words), _p(w2)
    e0 = _ev() if profile is not None else None
    rc = _lib.lib().tgsr_word_attention_fwd(_p(h), hbs, wp, w2p, _p(m8), 1 if correct_mask else 0, B, idf,
                                            cdf, T, Q, _p(ws), _p(out), cbs, _p(attn), _stream())
    check(rc, "tgsr_word_attention_fwd")
    if profile is not None:
        nbytes = 4 * B * Q * (2 * idf + (T if need_attn else 0))
        profile.append(("word_attention_kernel", 4.0 * B * Q * idf * T, nbytes, e0, _ev()))
    return out, attn


# ----------------------------------------------------------------------------------------- text encoder
_LENS_CACHE = {}


def _lens_on_device(lens: tuple, dev) -> torch.Tensor:
    """int32 caption lengths on the device; cached so a steady-state step issues no blocking H2D copy."""
    key = (lens, str(dev))
    t = _LENS_CACHE.pop(key, None)
    if t is None:
        if len(_LENS_CACHE) >= 256:
            _LENS_CACHE.pop(next(iter(_LENS_CACHE)))         # least recently used (dicts keep insertion order; hits re-insert)
        t = torch.tensor(lens, dtype=torch.int32).to(dev)
    _LENS_CACHE[key] = t
    if t.is_cuda:
        # the caller reads it on the CURRENT stream, which need not be the one it was allocated on (stream lanes share the
        # cache): an evicted entry's memory must not be handed out again while such a read is pending
        t.record_stream(torch.cuda.current_stream(t.device))
    return t


def bilstm(captions: torch.Tensor, cap_lens, emb: torch.Tensor, w_ih: torch.Tensor, w_hh: torch.Tensor,
           b_ih: torch.Tensor, b_hh: torch.Tensor):
    """captions int64 [B,W]; cap_lens list/tensor (host or device) ; emb [ntoken,ninput]; w_* stacked over the two
    directions [2,4H,*].  Returns (words_emb [B,2H,Tmax], sent_emb [B,2H])."""
    _need_hip(captions, emb, w_ih, w_hh, b_ih, b_hh)
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    B, width = captions.shape
    if len(lens) != B or min(lens) < 1 or max(lens) > width:
        raise TgsrError("bilstm: cap_lens %s invalid for captions %s" % (lens, tuple(captions.shape)))
    Tmax = max(lens)
    H = w_hh.shape[2]
    dev = emb.device
    captions = captions.to(torch.int64).contiguous()
    lens_d = _lens_on_device(tuple(lens), dev)
    gates = torch.empty(B * Tmax * 8 * H, dtype=torch.float32, device=dev)
    words = torch.empty(B, 2 * H, Tmax, dtype=torch.float32, device=dev)
    sent = torch.empty(B, 2 * H, dtype=torch.float32, device=dev)
    ts = [_f32(t.detach(), "lstm tensor").contiguous() for t in (emb, w_ih, w_hh, b_ih, b_hh)]
    rc = _lib.lib().tgsr_bilstm_fwd(_p(captions), width, _p(lens_d), B, Tmax, _p(ts[0]), emb.shape[0], emb.shape[1],
                                    _p(ts[1]), _p(ts[2]), _p(ts[3]), _p(ts[4]), H, _p(gates), _p(words), _p(sent),
                                    _stream())
    check(rc, "tgsr_bilstm_fwd")
    return words, sent


def bilstm_train_fwd(x: torch.Tensor, cap_lens, w_ih, w_hh, b_ih, b_hh):
    """Training forward on embedded inputs x [B,Tmax,ninput]: returns (words_emb, sent_emb, acts) where acts
    [B,Tmax,2,5,H] holds the gate activations / cell states tgsr_bilstm_bwd needs."""
    _need_hip(x, w_ih, w_hh, b_ih, b_hh)
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    B, Tmax, K = x.shape
    if len(lens) != B or min(lens) < 1 or max(lens) > Tmax:
        raise TgsrError("bilstm_train_fwd: cap_lens %s invalid for x %s" % (lens, tuple(x.shape)))
    H = w_hh.shape[2]
    dev = x.device
    ts = [_f32(t.detach(), "lstm tensor").contiguous() for t in (x, w_ih, w_hh, b_ih, b_hh)]
    gates = torch.empty(B * Tmax * 8 * H, dtype=torch.float32, device=dev)
    acts = torch.zeros(B, Tmax, 2, 5, H, dtype=torch.float32, device=dev)
    words = torch.empty(B, 2 * H, Tmax, dtype=torch.float32, device=dev)
    sent = torch.empty(B, 2 * H, dtype=torch.float32, device=dev)
    rc = _lib.lib().tgsr_bilstm_train_fwd(_p(ts[0]), _p(_lens_on_device(tuple(lens), dev)), B, Tmax, K, _p(ts[1]),
                                          _p(ts[2]), _p(ts[3]), _p(ts[4]), H, _p(gates), _p(acts), _p(words), _p(sent),
                                          _stream())
    check(rc, "tgsr_bilstm_train_fwd")
    return words, sent, acts


def bilstm_bwd(cap_lens, w_hh, acts, words, d_words, d_sent):
    """BPTT of both directions: returns (dgates [B,Tmax,2,4H], hprev [B,Tmax,2,H], dbias [2,4H])."""
    _need_hip(w_hh, acts, words, d_words, d_sent)
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    B, Tmax, _, _, H = acts.shape
    dev = acts.device
    dgates = torch.empty(B, Tmax, 2, 4 * H, dtype=torch.float32, device=dev)
    hprev = torch.empty(B, Tmax, 2, H, dtype=torch.float32, device=dev)
    dbias = torch.empty(2, 4 * H, dtype=torch.float32, device=dev)
    dw = _f32(d_words, "d_words").contiguous()
    ds = None if d_sent is None else _f32(d_sent, "d_sent").contiguous()
    rc = _lib.lib().tgsr_bilstm_bwd(_p(_lens_on_device(tuple(lens), dev)), B, Tmax, H,
                                    _p(_f32(w_hh.detach(), "w_hh").contiguous()), _p(acts), _p(words.contiguous()),
                                    _p(dw), _p(ds), _p(dgates), _p(hprev), _p(dbias), _stream())
    check(rc, "tgsr_bilstm_bwd")
    return dgates, hprev, dbias


def bigru_train_fwd(x: torch.Tensor, cap_lens, w_ih, w_hh, b_ih, b_hh):
    """Training forward of the bidirectional GRU on embedded inputs x [B,Tmax,ninput] (w_* stacked over the two directions:
    [2,3H,ninput], [2,3H,H], [2,3H]): returns (words_emb [B,2H,Tmax], sent_emb [B,2H], acts [B,Tmax,2,4,H])."""
    _need_hip(x, w_ih, w_hh, b_ih, b_hh)
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    B, Tmax, K = x.shape
    if len(lens) != B or min(lens) < 1 or max(lens) > Tmax:
        raise TgsrError("bigru_train_fwd: cap_lens %s invalid for x %s" % (lens, tuple(x.shape)))
    H = w_hh.shape[2]
    dev = x.device
    ts = [_f32(t.detach(), "gru tensor").contiguous() for t in (x, w_ih, w_hh, b_ih, b_hh)]
    b_rz = ts[4].clone()
    b_rz[:, 2 * H:] = 0
    b_hn = ts[4][:, 2 * H:].contiguous()
    gates = torch.empty(B * Tmax * 6 * H, dtype=torch.float32, device=dev)
    acts = torch.empty(B, Tmax, 2, 4, H, dtype=torch.float32, device=dev)
    words = torch.empty(B, 2 * H, Tmax, dtype=torch.float32, device=dev)
    sent = torch.empty(B, 2 * H, dtype=torch.float32, device=dev)
    rc = _lib.lib().tgsr_bigru_train_fwd(_p(ts[0]), _p(_lens_on_device(tuple(lens), dev)), B, Tmax, K, _p(ts[1]), _p(ts[2]),
                                         _p(ts[3]), _p(b_rz), _p(b_hn), H, _p(gates), _p(acts), _p(words), _p(sent), _stream())
    check(rc, "tgsr_bigru_train_fwd")
    return words, sent, acts


def bigru_bwd(cap_lens, w_hh, acts, words, d_words, d_sent):
    """BPTT of both directions: (dgx [B,Tmax,2,3H], dgh [B,Tmax,2,3H], hprev [B,Tmax,2,H], dbias [2,2,3H] = (d b_ih, d b_hh))."""
    _need_hip(w_hh, acts, words, d_words, d_sent)
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    B, Tmax, _, _, H = acts.shape
    dev = acts.device
    dgx = torch.empty(B, Tmax, 2, 3 * H, dtype=torch.float32, device=dev)
    dgh = torch.empty(B, Tmax, 2, 3 * H, dtype=torch.float32, device=dev)
    hprev = torch.empty(B, Tmax, 2, H, dtype=torch.float32, device=dev)
    dbias = torch.empty(2, 2, 3 * H, dtype=torch.float32, device=dev)
    dw = _f32(d_words, "d_words").contiguous()
    ds = None if d_sent is None else _f32(d_sent, "d_sent").contiguous()
    rc = _lib.lib().tgsr_bigru_bwd(_p(_lens_on_device(tuple(lens), dev)), B, Tmax, H, _p(_f32(w_hh.detach(), "w_hh").contiguous()),
                                   _p(acts), _p(words.contiguous()), _p(dw), _p(ds), _p(dgx), _p(dgh), _p(hprev), _p(dbias), _stream())
    check(rc, "tgsr_bigru_bwd")
    return dgx, dgh, hprev, dbias


def lstm_gate_table(emb, w_ih, b_ih, b_hh):
    """[ntoken, 2, 4H] gate pre-activations of every token (eval mode, frozen weights); see tgsr_lstm_gate_table."""
    _need_hip(emb, w_ih, b_ih, b_hh)
    ts = [_f32(t.detach(), "lstm tensor").contiguous() for t in (emb, w_ih, b_ih, b_hh)]
    H = w_ih.shape[1] // 4
    table = torch.empty(emb.shape[0], 2, 4 * H, dtype=torch.float32, device=emb.device)
    check(_lib.lib().tgsr_lstm_gate_table(_p(ts[0]), emb.shape[0], emb.shape[1], _p(ts[1]), _p(ts[2]), _p(ts[3]), H,
                                          _p(table), _stream()), "tgsr_lstm_gate_table")
    return table


def gru_gate_table(emb, w_ih, b_ih, b_hh):
    """[ntoken, 2, 3H] gate pre-activations of every token for the GRU encoder (tgsr_gru_gate_table): b_hh's r and z thirds
    fold into the table, its n third stays with the recurrence (returned as b_hn [2, H])."""
    _need_hip(emb, w_ih, b_ih, b_hh)
    ts = [_f32(t.detach(), "gru tensor").contiguous() for t in (emb, w_ih, b_ih, b_hh)]
    H = w_ih.shape[1] // 3
    b_rz = ts[3].clone()
    b_rz[:, 2 * H:] = 0
    b_hn = ts[3][:, 2 * H:].contiguous()
    table = torch.empty(emb.shape[0], 2, 3 * H, dtype=torch.float32, device=emb.device)
    check(_lib.lib().tgsr_gru_gate_table(_p(ts[0]), emb.shape[0], emb.shape[1], _p(ts[1]), _p(ts[2]), _p(b_rz), H, _p(table),
                                         _stream()), "tgsr_gru_gate_table")
    return table, b_hn


def bigru_table(captions, cap_lens, table, w_hh, b_hn):
    """The bidirectional GRU recurrence over a per-token gate table (one launch): (words_emb, sent_emb); cap_lens on the host
    (-> T_max columns) or as a device int32 tensor (-> full caption width, see bilstm_table)."""
    _need_hip(captions, table, w_hh, b_hn)
    B, width = captions.shape
    H, dev = w_hh.shape[2], table.device
    if torch.is_tensor(cap_lens) and cap_lens.is_cuda:
        if cap_lens.dtype != torch.int32 or cap_lens.numel() != B or not cap_lens.is_contiguous():
            raise TgsrError("bigru: device cap_lens must be a contiguous int32 [%d] tensor" % B)
        Tmax, lens_d = width, cap_lens
    else:
        lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
        if len(lens) != B or min(lens) < 1 or max(lens) > width:
            raise TgsrError("bigru: cap_lens %s invalid for captions %s" % (lens, tuple(captions.shape)))
        Tmax, lens_d = max(lens), _lens_on_device(tuple(lens), dev)
    captions = captions.to(torch.int64).contiguous()
    words = torch.empty(B, 2 * H, Tmax, dtype=torch.float32, device=dev)
    sent = torch.empty(B, 2 * H, dtype=torch.float32, device=dev)
    rc = _lib.lib().tgsr_bigru_table_fwd(_p(captions), width, _p(lens_d), B, Tmax, _p(table), table.shape[0],
                                         _p(_f32(w_hh.detach(), "w_hh").contiguous()), _p(_f32(b_hn, "b_hn").contiguous()), H,
                                         _p(words), _p(sent), _stream())
    check(rc, "tgsr_bigru_table_fwd")
    return words, sent


def bilstm_table(captions, cap_lens, table, w_hh):
    """The BiLSTM recurrence over a per-token gate table (one launch).  Returns (words_emb, sent_emb).

    cap_lens on the HOST (list / CPU tensor): words_emb is [B, 2H, max(cap_lens)] like the reference's
    pad_packed_sequence output (util.py:250-253).  cap_lens as a DEVICE int32 tensor [B]: the lengths are read by the
    kernel only - nothing of the launch (shapes, grid, arguments) depends on their values, so the step can be captured
    once and replayed on any batch; words_emb is then [B, 2H, captions.size(1)] with zeros behind each caption, and the
    caller crops to the batch's longest caption on its side (SRPipeline / GraphedStep do)."""
    _need_hip(captions, table, w_hh)
    B, width = captions.shape
    H, dev = w_hh.shape[2], table.device
    if torch.is_tensor(cap_lens) and cap_lens.is_cuda:
        if cap_lens.dtype != torch.int32 or cap_lens.numel() != B or not cap_lens.is_contiguous():
            raise TgsrError("bilstm: device cap_lens must be a contiguous int32 [%d] tensor, got %s %s"
                            % (B, cap_lens.dtype, tuple(cap_lens.shape)))
        _need_hip(cap_lens)
        Tmax, lens_d = width, cap_lens
    else:
        lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
        if len(lens) != B or min(lens) < 1 or max(lens) > width:
            raise TgsrError("bilstm: cap_lens %s invalid for captions %s" % (lens, tuple(captions.shape)))
        Tmax, lens_d = max(lens), _lens_on_device(tuple(lens), dev)
    captions = captions.to(torch.int64).contiguous()
    words = torch.empty(B, 2 * H, Tmax, dtype=torch.float32, device=dev)
    sent = torch.empty(B, 2 * H, dtype=torch.float32, device=dev)
    w = _f32(w_hh.detach(), "w_hh").contiguous()
    rc = _lib.lib().tgsr_bilstm_table_fwd(_p(captions), width, _p(lens_d), B, Tmax, _p(table),
                                          table.shape[0], _p(w), H, _p(words), _p(sent), _stream())
    check(rc, "tgsr_bilstm_table_fwd")
    return words, sent


# ----------------------------------------------------------------------------------------- DAMSM
def damsm_words_similarity(img_features: torch.Tensor, words_emb: torch.Tensor, cap_lens, gamma1: float,
                           gamma2: float, need_att: bool = True):
    """All (image j, caption i) word-level similarities in one launch: returns (sim [B,B] = log sum_w exp(gamma2 *
    cos), att_diag [B,Tw,ih,iw] or None).  img_features [B,ndf,ih,iw], words_emb [B,ndf,Tw]."""
    _need_hip(img_features, words_emb)
    B, ndf, ih, iw = img_features.shape
    Tw = words_emb.shape[2]
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    if len(lens) != B or min(lens) < 1 or max(lens) > Tw:
        raise TgsrError("damsm: cap_lens %s invalid for words %s" % (lens, tuple(words_emb.shape)))
    ctx = _f32(img_features, "img_features").contiguous()
    words = _f32(words_emb, "words_emb").contiguous()
    dev = ctx.device
    sim = torch.empty(B, B, dtype=torch.float32, device=dev)
    att = torch.empty(B, Tw, ih, iw, dtype=torch.float32, device=dev) if need_att else None
    rc = _lib.lib().tgsr_damsm_words_fwd(_p(words), _p(_lens_on_device(tuple(lens), dev)), _p(ctx), B, ndf, Tw,
                                         ih * iw, float(gamma1), float(gamma2), _p(sim), _p(att), _stream())
    check(rc, "tgsr_damsm_words_fwd")
    return sim, att


def damsm_words_bwd(img_features: torch.Tensor, words_emb: torch.Tensor, cap_lens, gamma1: float, gamma2: float,
                    grad_sim: torch.Tensor):
    """Backward of damsm_words_similarity: grad_sim [B,B] -> (grad_img_features [B,ndf,ih,iw], grad_words [B,ndf,Tw])."""
    _need_hip(img_features, words_emb, grad_sim)
    B, ndf, ih, iw = img_features.shape
    Tw, S = words_emb.shape[2], ih * iw
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    ctx = _f32(img_features.detach(), "img_features").contiguous()
    words = _f32(words_emb.detach(), "words_emb").contiguous()
    gs = _f32(grad_sim, "grad_sim").contiguous()
    dev = ctx.device
    L = _lib.lib()
    ws = torch.empty(L.tgsr_damsm_words_bwd_ws_elems(B, ndf, S), dtype=torch.float32, device=dev)
    gw32 = torch.empty(B, ndf, 32, dtype=torch.float32, device=dev)
    gctx = torch.empty(B, ndf, ih, iw, dtype=torch.float32, device=dev)
    rc = L.tgsr_damsm_words_bwd(_p(words), _p(_lens_on_device(tuple(lens), dev)), _p(ctx), _p(gs), B, ndf, Tw, S,
                                float(gamma1), float(gamma2), _p(ws), _p(gw32), _p(gctx), _stream())
    check(rc, "tgsr_damsm_words_bwd")
    return gctx, gw32[:, :, :Tw].contiguous()


def func_attention(query: torch.Tensor, context: torch.Tensor, gamma1: float):
    """GlobalAttention.func_attention: query [B,ndf,L], context [B,ndf,ih,iw] -> (weightedContext [B,ndf,L],
    attn [B,L,ih,iw])."""
    _need_hip(query, context)
    B, ndf, L = query.shape
    ih, iw = context.shape[2], context.shape[3]
    q = _f32(query, "query").contiguous()
    c = _f32(context, "context").contiguous()
    wc = torch.empty(B, ndf, L, dtype=torch.float32, device=q.device)
    attn = torch.empty(B, L, ih, iw, dtype=torch.float32, device=q.device)
    rc = _lib.lib().tgsr_func_attention_fwd(_p(q), _p(c), B, ndf, L, ih * iw, float(gamma1), _p(wc), _p(attn),
                                            _stream())
    check(rc, "tgsr_func_attention_fwd")
    return wc, attn


# ----------------------------------------------------------------------------------------- CNN_ENCODER heads
CONV1X1_GCONV = os.environ.get("TGSR_CONV1X1_GCONV", "1") != "0"


def conv1x1(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """1x1 convolution [B,Cin,H,W] -> [B,Cout,H,W] (emb_features, util.py:300,367) as an MFMA GEMM."""
    _need_hip(x, w, bias)
    x = _f32(x, "x").contiguous()
    B, Cin, H, W = x.shape
    w2 = _f32(w.detach(), "w").reshape(w.shape[0], -1).contiguous()
    if w2.shape[1] != Cin:
        raise TgsrError("conv1x1: weight %s vs input channels %d" % (tuple(w.shape), Cin))
    out = torch.empty(B, w2.shape[0], H, W, dtype=torch.float32, device=x.device)
    if bias is None and Cin % 16 == 0 and B * H * W >= 1024 and CONV1X1_GCONV:
        # the implicit-GEMM kernel of CNN_ENCODER's trunk (three-piece bf16 operands: tgsr_gconv) takes a 1x1 filter as it is - the
        # weight [Cout, Cin] IS its A operand: emb_features on the 17 x 17 x 768 region features, forward and (through the
        # transposed weight, custom_ops._conv1x1_backward) data gradient, 80 -> ~20 us each at batch 16
        need = gconv_ws_elems(B, w2.shape[0], H, W, Cin)
        ws = torch.empty(need, dtype=torch.float32, device=x.device) if need else None
        gconv(False, w2, x, 0, Cin, out, 0, 1, 1, 1, 0, 0, None, False, False, ws, None)
        return out
    b = None if bias is None else _f32(bias.detach(), "bias").contiguous()
    check(_lib.lib().tgsr_conv1x1_fwd(_p(x), B, Cin, H * W, _p(w2), _p(b), w2.shape[0], _p(out), _stream()),
          "tgsr_conv1x1_fwd")
    return out


def conv1x1_wgrad(dy: torch.Tensor, x: torch.Tensor) -> Optional[torch.Tensor]:
    """Weight gradient of a bias-free 1x1 convolution, dW[co][ci] = sum over (b, pixel) of dy[b][co][p] x[b][ci][p], on the implicit-GEMM
    kernel: with K = B H W as the "channels" and Cin as the "pixels" it IS a 1x1 convolution - A = dy as [Cout, K], the image x as
    [1, K, Cin, 1] - split over K into slabs summed in a fixed order (emb_features in DAMSM pre-training, pretrain_DAMSM.py:79-98: the
    plain GEMM kernel took ~1 ms for these 1.8 GFLOP, half of the step's kernel time).  None when the shape does not qualify
    (K % 16 != 0, few channels): the caller keeps its own form."""
    _need_hip(dy, x)
    B, Cout, H, W = dy.shape
    Cin, K = x.shape[1], B * H * W
    if not CONV1X1_GCONV or K % 16 != 0 or K < 1024 or Cin < 64 or tuple(x.shape) != (B, Cin, H, W):
        return None
    a = _f32(dy, "dy").permute(1, 0, 2, 3).reshape(Cout, K).contiguous()
    xt = _f32(x, "x").permute(0, 2, 3, 1).reshape(1, K, Cin, 1).contiguous()
    out = torch.empty(1, Cout, Cin, 1, dtype=torch.float32, device=dy.device)
    need = gconv_ws_elems(1, Cout, Cin, 1, K)
    ws = torch.empty(need, dtype=torch.float32, device=dy.device) if need else None
    gconv(False, a, xt, 0, K, out, 0, 1, 1, 1, 0, 0, None, False, False, ws, None)
    return out.reshape(Cout, Cin)


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B,K] @ w[Cout,K]^T + bias (emb_cnn_code, util.py:301,364) as an MFMA GEMM."""
    _need_hip(x, w, bias)
    x = _f32(x, "x").contiguous()
    w2 = _f32(w.detach(), "w").contiguous()
    B, K = x.shape
    if w2.shape[1] != K:
        raise TgsrError("linear: weight %s vs input %s" % (tuple(w.shape), tuple(x.shape)))
    b = None if bias is None else _f32(bias.detach(), "bias").contiguous()
    if CONV1X1_GCONV and K % 16 == 0 and K >= 256:
        # long reductions over few rows (emb_cnn_code: 16 x 2048 -> 256; the recurrent weights' gradients: K = B T): on the plain
        # GEMM kernel a handful of workgroups walk K alone (~100 us for a few MFLOP); the implicit-GEMM kernel splits K over the
        # chip - x^T as a [1, K, B, 1] image, the weight as its A operand, the bias in its epilogue
        xt = x.t().contiguous().view(1, K, B, 1)
        img = torch.empty(1, w2.shape[0], B, 1, dtype=torch.float32, device=x.device)
        need = gconv_ws_elems(1, w2.shape[0], B, 1, K)
        ws = torch.empty(need, dtype=torch.float32, device=x.device) if need else None
        gconv(False, w2, xt, 0, K, img, 0, 1, 1, 1, 0, 0, b, False, False, ws, None)
        return img.view(w2.shape[0], B).t().contiguous()
    out = torch.empty(B, w2.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.lib().tgsr_linear_fwd(_p(x), B, K, _p(w2), _p(b), w2.shape[0], _p(out), _stream()), "tgsr_linear_fwd")
    return out


# ----------------------------------------------------------------------------------------- training path primitives
def conv_to3_set_pipe(on: bool) -> bool:
    """The image heads' streaming kernel with its copies two stages ahead (default) or the plain double buffer; returns the previous setting."""
    return bool(_lib.lib().tgsr_conv_to3_set_pipe(1 if on else 0))


def bn_set_fuse_small(on: bool) -> bool:
    """Small BatchNorm layers (one workgroup per channel) as one launch per direction (default) or two; returns the previous setting."""
    return bool(_lib.lib().tgsr_bn_set_fuse_small(1 if on else 0))


def bn_train_fwd(raw: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, momentum: float,
                 running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor], act: int = 0,
                 residual: Optional[torch.Tensor] = None, nbt: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None,
                 stat_partial: Optional[torch.Tensor] = None):
    """nn.BatchNorm2d(train) on the raw conv output [B,C,H,W] + act (0 none [+ residual], 1 GLU, 2 LeakyReLU(0.2)):
    batch statistics, running statistics / num_batches_tracked updated in place through their pointers.  Returns
    (out, stats [4, C] = mean, invstd, scale, shift).  `out` / `stats` may be given (slices of larger tensors).
    `stat_partial` [C, nslots, 2]: the (sum, sumsq) pairs conv3x3_wino_stats left - the pass over `raw` that would compute
    them is then skipped."""
    _need_hip(raw, gamma, beta, running_mean, running_var, residual, nbt, out, stats, stat_partial)
    L = _lib.lib()
    B, C, Ho, Wo = raw.shape
    HW, dev = Ho * Wo, raw.device
    raw = _f32(raw, "raw")
    if not raw.is_contiguous():
        raise TgsrError("bn_train_fwd: raw must be contiguous")
    co = C // 2 if act == 1 else C
    if out is None:
        out = torch.empty(B, co, Ho, Wo, dtype=torch.float32, device=dev)
    if stats is None:
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)
    res = None if residual is None else residual.contiguous()
    if stat_partial is not None:
        if (stat_partial.dim() != 3 or stat_partial.shape[0] != C or stat_partial.shape[2] != 2 or
                stat_partial.dtype != torch.float32 or not stat_partial.is_contiguous()):
            raise TgsrError("bn_train_fwd: stat_partial %s for %d channels" % (tuple(stat_partial.shape), C))
        s0, s1, s2, s3 = _rows4(stats)
        rc = L.tgsr_bn_train_fwd_from_stats(_p(raw), B, C, HW, _p(gamma), _p(beta), float(eps),
                                            float(momentum), _p(running_mean), _p(running_var), int(act), _p(res),
                                            0 if res is None else co * HW, _p(stat_partial), stat_partial.shape[1],
                                            s0, s1, s2, s3, _p(out), co * HW, _p(nbt), _stream())
        check(rc, "tgsr_bn_train_fwd_from_stats")
        return out, stats
    ws = torch.empty(C * L.tgsr_bn_train_nsplit(B, C, HW) * 4, dtype=torch.float32, device=dev)
    s0, s1, s2, s3 = _rows4(stats)
    rc = L.tgsr_bn_train_fwd(_p(raw), B, C, HW, _p(gamma), _p(beta), float(eps), float(momentum),
                             _p(running_mean), _p(running_var), int(act), _p(res), 0 if res is None else co * HW, _p(ws),
                             s0, s1, s2, s3, _p(out), co * HW, _p(nbt), _stream())
    check(rc, "tgsr_bn_train_fwd")
    return out, stats


def bn_train_bwd(dout: torch.Tensor, raw: torch.Tensor, stats: torch.Tensor, act: int, dgamma: Optional[torch.Tensor] = None,
                 dbeta: Optional[torch.Tensor] = None, draw: Optional[torch.Tensor] = None):
    """Backward of bn_train_fwd: (draw like raw, dgamma [C], dbeta [C]); the three may be given (gradient slots of a flat
    bucket, slices of a batched tensor)."""
    _need_hip(dout, raw, stats, dgamma, dbeta, draw)
    L = _lib.lib()
    B, C, Ho, Wo = raw.shape
    HW, dev = Ho * Wo, raw.device
    dout = _f32(dout, "dout").contiguous()
    co = C // 2 if act == 1 else C
    if draw is None:
        draw = torch.empty_like(raw)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    if dbeta is None:
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
    ws = torch.empty(co * L.tgsr_bn_train_nsplit(B, co, HW) * 4, dtype=torch.float32, device=dev)
    s0, s1, s2, s3 = _rows4(stats)
    rc = L.tgsr_bn_train_bwd(_p(dout), _p(raw), B, C, HW, s2, s3, s0, s1, int(act),
                             _p(ws), _p(ws), _p(draw), _p(dgamma), _p(dbeta), _stream())
    check(rc, "tgsr_bn_train_bwd")
    return draw, dgamma, dbeta


def sumpool2x2(x: torch.Tensor) -> torch.Tensor:
    """Backward of nn.Upsample(x2, nearest): [B,C,2H,2W] -> [B,C,H,W], each output the sum of its 2x2 block."""
    _need_hip(x)
    x = _f32(x, "x").contiguous()
    B, C, H2, W2 = x.shape
    out = torch.empty(B, C, H2 // 2, W2 // 2, dtype=torch.float32, device=x.device)
    check(_lib.lib().tgsr_sumpool2x2(_p(x), B * C, H2 // 2, W2 // 2, _p(out), _stream()), "tgsr_sumpool2x2")
    return out


def conv3x3_wgrad_kind(Cin: int, Cout: int, upsample: bool, winograd: bool = True) -> str:
    """Which weight-gradient kernel serves a conv3x3 layer: "wino" (16 positions per 2x2 output tile) where Cout % 32 == 0 and
    Cin % 32 == 0; for an upBlock "upwino" (9 Winograd positions on the low-resolution pixels) where Cout % 64 == 0 and
    Cin % 32 == 0 - that kernel has 64-row co-blocks only (tgsr_upwino_wgrad.hip), so upBlock(32, 16) / GF_DIM = 16 (Cout = 32)
    stay on the direct kernel; else "direct"."""
    if winograd and Cin % 32 == 0:
        if upsample:
            return "upwino" if Cout % 64 == 0 else "direct"
        if Cout % 32 == 0:
            return "wino"
    return "direct"


def conv3x3_wgrad(draw: torch.Tensor, x: torch.Tensor, upsample: bool = False, winograd: bool = True,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight gradient [Cout,Cin,3,3] of conv3x3(upsample ? nearest_x2(x) : x): draw [B,Cout,Ho,Wo] (gradient at the raw
    conv output), x [B,Cin,H,W]; per-workgroup partial slabs summed in a fixed order (reproducible).  `out`: where to
    write it (a gradient slot)."""
    _need_hip(draw, x, out)
    L = _lib.lib()
    draw = _f32(draw, "draw").contiguous()
    x = _f32(x, "x").contiguous()
    B, Cin, H, W = x.shape
    Cout, HW, dev = draw.shape[1], draw.shape[2] * draw.shape[3], x.device
    dw = out if out is not None else torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=dev)
    kind = conv3x3_wgrad_kind(Cin, Cout, upsample, winograd)
    e0 = _ev() if profile is not None else None
    if kind == "upwino":
        ws = torch.empty(L.tgsr_upwino_wgrad_ws_elems(B, Cin, Cout, H, W), dtype=torch.float32, device=dev)
        check(L.tgsr_upwino_wgrad(_p(draw), _p(x), Cin * H * W, B, Cin, H, W, Cout, _p(ws), _p(dw), _stream()), "tgsr_upwino_wgrad")
    elif kind == "wino":
        ws = torch.empty(L.tgsr_wino_wgrad_ws_elems(B, Cin, Cout, H, W), dtype=torch.float32, device=dev)
        check(L.tgsr_wino_wgrad(_p(draw), _p(x), Cin * H * W, B, Cin, H, W, Cout, _p(ws), _p(dw), _stream()), "tgsr_wino_wgrad")
    else:
        ws = torch.empty(L.tgsr_conv3x3_wgrad_ws_elems(B, Cin, Cout, H, W, 1 if upsample else 0), dtype=torch.float32, device=dev)
        check(L.tgsr_conv3x3_wgrad(_p(draw), _p(x), Cin * H * W, B, Cin, H, W, Cout, 1 if upsample else 0, _p(ws), _p(dw),
                                   _stream()), "tgsr_conv3x3_wgrad")
    if profile is not None:      # direct-form FLOPs of the weight gradient: one MAC per (output pixel, tap, ci, co)
        profile.append(({"upwino": "upwino_wgrad_kernel", "wino": "wino_wgrad_kernel", "direct": "conv3x3_wgrad_kernel"}[kind],
                        2.0 * B * HW * Cout * Cin * 9, 4.0 * (B * Cin * H * W + B * Cout * HW + Cout * Cin * 9), e0, _ev()))
    return dw


def conv_to3_bwd(dy: torch.Tensor, out: Optional[torch.Tensor], addend: Optional[torch.Tensor], alpha: float, x: torch.Tensor,
                 w: torch.Tensor, tanh_axpy: bool, need_dx: bool = True, need_dw: bool = True):
    """(dx, dw) of conv_to3 (None for a gradient that is not needed); `out` = the forward output (tanh heads)."""
    _need_hip(dy, out, addend, x, w)
    L = _lib.lib()
    dy, x, w = _f32(dy, "dy").contiguous(), _f32(x, "x").contiguous(), _f32(w.detach(), "w").contiguous()
    addend = None if addend is None else addend.contiguous()     # the kernel reads it as dense NCHW
    B, Cin, H, W = x.shape
    K = w.shape[2]
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty_like(w) if need_dw else None
    ws = x.new_empty(L.tgsr_conv_to3_bwd_ws_elems(B, Cin, H, W, K)) if need_dw else None
    rc = L.tgsr_conv_to3_bwd(_p(dy), _p(out), _p(addend), float(alpha), _p(x), Cin * H * W, _p(w), B, Cin, H, W, K,
                             _lib.ACT_TANH_AXPY if tanh_axpy else _lib.ACT_NONE, _p(dx), _p(ws), _p(dw), _stream())
    check(rc, "tgsr_conv_to3_bwd")
    return dx, dw


def word_attention_bwd(h: torch.Tensor, src: torch.Tensor, mask: Optional[torch.Tensor], correct_mask: bool, T: int,
                       dc: torch.Tensor):
    """Backward of word_attention wrt h and the projected words: h [B,idf,ih,iw], src [B,idf,32] (zero padded past T),
    dc [B,idf,ih,iw] -> (dh like h, dsrc [B,idf,T]); P is recomputed, the per-chunk partial sums of d(src) are added in a
    fixed order."""
    _need_hip(h, src, mask, dc)
    L = _lib.lib()
    h, dc = _f32(h, "h").contiguous(), _f32(dc, "dc").contiguous()
    B, idf, ih, iw = h.shape
    Q = ih * iw
    part = torch.empty(B, L.tgsr_word_attention_bwd_chunks(Q), idf, 32, dtype=torch.float32, device=h.device)
    dh = torch.empty_like(h)
    m8 = None if mask is None else _mask_u8(mask)
    rc = L.tgsr_word_attention_bwd(_p(h), idf * Q, _p(src.contiguous()), _p(m8), 1 if correct_mask else 0, B, idf, T, Q,
                                   _p(dc), _p(dh), _p(part), _stream())
    check(rc, "tgsr_word_attention_bwd")
    return dh, part.sum(1)[:, :, :T].contiguous()


def rowdot(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """out[b] = <x[b, :], w> + bias (the discriminators' logit heads: a 4x4 / stride-4 conv of a 4x4 map to one channel)."""
    _need_hip(x, w, bias)
    x = _f32(x, "x").contiguous()
    w = _f32(w.detach(), "w").contiguous().view(-1)
    B, K = x.shape
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    check(_lib.lib().tgsr_rowdot_fwd(_p(x), _p(w), _p(None if bias is None else bias.detach()), _p(out), B, K, _stream()),
          "tgsr_rowdot_fwd")
    return out


def rowdot_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, need_dx: bool = True, need_dw: bool = True):
    """(dx [B,K], dw [K]) of rowdot (None where not needed)."""
    _need_hip(dy, x, w)
    dy, x = _f32(dy, "dy").contiguous(), _f32(x, "x").contiguous()
    w = _f32(w.detach(), "w").contiguous().view(-1)
    B, K = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty(K, dtype=torch.float32, device=x.device) if need_dw else None
    if need_dx or need_dw:
        check(_lib.lib().tgsr_rowdot_bwd(_p(dy), _p(x), _p(w), _p(dx), _p(dw), B, K, _stream()), "tgsr_rowdot_bwd")
    return dx, dw


def ca_net(sent_emb: torch.Tensor, w: torch.Tensor, b: torch.Tensor, ncf: int, eps: Optional[torch.Tensor]):
    """CA_NET.forward (util.py:372-400) in one launch: (c_code or None, mu, logvar); eps [B,ncf] = the standard normals of
    the re-parametrisation (None: c_code is not produced - the x8 / x16 generators discard it, model.py:51-52)."""
    _need_hip(sent_emb, w, b, eps)
    x = _f32(sent_emb, "sent_emb").contiguous()
    w, b = _f32(w.detach(), "w").contiguous(), _f32(b.detach(), "b").contiguous()
    B, tdim = x.shape
    mu = torch.empty(B, ncf, dtype=torch.float32, device=x.device)
    logvar = torch.empty_like(mu)
    c_code = torch.empty_like(mu) if eps is not None else None
    check(_lib.lib().tgsr_ca_net_fwd(_p(x), _p(w), _p(b), _p(eps), B, tdim, ncf, _p(c_code), _p(mu), _p(logvar), _stream()),
          "tgsr_ca_net_fwd")
    return c_code, mu, logvar


def text_tail(words: torch.Tensor, w_ctxs, sent_emb: torch.Tensor, ca_w: torch.Tensor, ca_b: torch.Tensor, ncf: int,
              captions: torch.Tensor, lp_dtype=None):
    """What an inference step needs between the text encoder and the generator, in one launch (tgsr_text_tail_fwd):
    word_project(words, w_ctxs), CA_NET's (mu, logvar) and mask = (captions[:, :T] == 0).
    Returns (src [nsets,B,idf,32], mu, logvar, mask uint8 [B,T] - `.view(torch.bool)` is the reference's mask).
    lp_dtype (torch.bfloat16 | torch.float16; idf == 32): a fifth value, the uint8 buffer `att_pack` of
    tgsr_text_tail_lp_fwd - the projections as MFMA A fragments of that type + the packed mask rows, which the
    reduced-precision kernels that attend in their epilogue read (lp.stem / lp.upconv_glu_head with att=...)."""
    import ctypes
    _need_hip(words, sent_emb, ca_w, ca_b, captions, *w_ctxs)
    words = _f32(words, "words").contiguous()
    B, cdf, T = words.shape
    n = len(w_ctxs)
    idf = w_ctxs[0].shape[0]
    ws = [_f32(w.detach(), "w_ctx").reshape(idf, cdf).contiguous() for w in w_ctxs]
    x = _f32(sent_emb, "sent_emb").contiguous()
    w, b = _f32(ca_w.detach(), "ca_w").contiguous(), _f32(ca_b.detach(), "ca_b").contiguous()
    if captions.dtype != torch.int64 or captions.dim() != 2 or captions.shape[0] != B or captions.shape[1] < T:
        raise TgsrError("text_tail: captions %s %s for words %s" % (tuple(captions.shape), captions.dtype, tuple(words.shape)))
    captions = captions.contiguous()
    if x.shape[0] != B or w.shape != (4 * ncf, x.shape[1]):
        raise TgsrError("text_tail: sent_emb %s / fc weight %s" % (tuple(x.shape), tuple(w.shape)))
    out = torch.empty(n, B, idf, 32, dtype=torch.float32, device=words.device)
    mu = torch.empty(B, ncf, dtype=torch.float32, device=words.device)
    logvar = torch.empty_like(mu)
    mask = torch.empty(B, T, dtype=torch.uint8, device=words.device)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ws])
    if lp_dtype is not None:
        L = _lib.lib()
        dt = {torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16}[lp_dtype]
        pack = torch.empty(L.tgsr_lp_att_pack_bytes(n, B), dtype=torch.uint8, device=words.device)
        check(L.tgsr_text_tail_lp_fwd(_p(words), ptrs, n, B, idf, cdf, T, _p(out), _p(x), _p(w), _p(b), x.shape[1], ncf,
                                      _p(mu), _p(logvar), _p(captions), captions.shape[1], _p(mask), dt, _p(pack), _stream()),
              "tgsr_text_tail_lp_fwd")
        return out, mu, logvar, mask, pack
    check(_lib.lib().tgsr_text_tail_fwd(_p(words), ptrs, n, B, idf, cdf, T, _p(out), _p(x), _p(w), _p(b), x.shape[1], ncf,
                                        _p(mu), _p(logvar), _p(captions), captions.shape[1], _p(mask), _stream()),
          "tgsr_text_tail_fwd")
    return out, mu, logvar, mask


def multi_copy(dsts, srcs) -> None:
    """dst[i].copy_(src[i]) for up to 16 dense same-shape, same-dtype device tensors in one launch (tgsr_multi_copy)."""
    import ctypes
    n = len(dsts)
    if n == 0:
        return
    if n != len(srcs) or n > 16:
        raise TgsrError("multi_copy: %d destinations, %d sources" % (n, len(srcs)))
    _need_hip(*dsts, *srcs)
    for d, s_ in zip(dsts, srcs):
        if d.shape != s_.shape or d.dtype != s_.dtype or not d.is_contiguous() or not s_.is_contiguous():
            raise TgsrError("multi_copy: %s %s <- %s %s" % (tuple(d.shape), d.dtype, tuple(s_.shape), s_.dtype))
    dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    sp = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_ in srcs])
    nb = (ctypes.c_int64 * n)(*[d.numel() * d.element_size() for d in dsts])
    check(_lib.lib().tgsr_multi_copy(n, dp, sp, nb, _stream()), "tgsr_multi_copy")


# ----------------------------------------------------------------------------------------- RCCL behind the C ABI
def comm_available() -> bool:
    """librccl could be opened by the library (tgsr_comm_available)."""
    return bool(_lib.lib().tgsr_comm_available())


def comm_unique_id() -> bytes:
    """The 128-byte id rank 0 creates and every rank passes to comm_init (ncclGetUniqueId)."""
    import ctypes
    buf = ctypes.create_string_buffer(128)
    check(_lib.lib().tgsr_comm_unique_id(buf), "tgsr_comm_unique_id")
    return bytes(buf.raw)


def comm_init(unique_id: bytes, rank: int, world: int) -> int:
    """An RCCL communicator on the current device (ncclCommInitRank; collective over the ranks); returns an opaque handle."""
    import ctypes
    if len(unique_id) != 128:
        raise TgsrError("comm_init: the unique id has %d bytes, not 128" % len(unique_id))
    comm = ctypes.c_void_p()
    check(_lib.lib().tgsr_comm_init(ctypes.byref(comm), ctypes.create_string_buffer(unique_id, 128), int(rank), int(world)),
          "tgsr_comm_init")
    return comm.value


def allreduce_flat(comm: int, flat: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """flat <- scale * sum over ranks of flat, in place, on torch's current stream (tgsr_allreduce_flat: ncclAllReduce + one
    scaling launch): the gradient bucket's collective without torch.distributed in the way."""
    _need_hip(flat)
    if flat.dtype != torch.float32 or not flat.is_contiguous():
        raise TgsrError("allreduce_flat: a dense fp32 buffer, got %s %s" % (flat.dtype, tuple(flat.shape)))
    check(_lib.lib().tgsr_allreduce_flat(comm, _p(flat), flat.numel(), float(scale), _stream()), "tgsr_allreduce_flat")
    return flat


def comm_count(comm: int) -> tuple:
    """(world, rank) as RCCL reports them for a communicator (ncclCommCount, ncclCommUserRank)."""
    import ctypes
    w, r = ctypes.c_int(-1), ctypes.c_int(-1)
    check(_lib.lib().tgsr_comm_count(comm, ctypes.byref(w), ctypes.byref(r)), "tgsr_comm_count")
    return int(w.value), int(r.value)


def comm_destroy(comm: int) -> None:
    check(_lib.lib().tgsr_comm_destroy(comm), "tgsr_comm_destroy")


def axpy_images(ts, ss, alpha: float, outs=None):
    """[t + alpha * s for t, s in zip(ts, ss)] for up to 4 dense fp32 images in one launch (tgsr_axpy_images): the closing
    `+ a * SRb` of NetG_highweight's heads when tanh(conv5x5(.)) was computed ahead of the low-frequency images."""
    import ctypes
    n = len(ts)
    if n == 0:
        return []
    if n != len(ss) or n > 4:
        raise TgsrError("axpy_images: %d / %d images (at most 4)" % (n, len(ss)))
    _need_hip(*ts, *ss)
    ts = [_f32(t, "t").contiguous() for t in ts]
    ss = [_f32(s_, "s").contiguous() for s_ in ss]
    for t, s_ in zip(ts, ss):
        if t.shape != s_.shape:
            raise TgsrError("axpy_images: %s + alpha * %s" % (tuple(t.shape), tuple(s_.shape)))
    if outs is None:
        outs = [torch.empty_like(t) for t in ts]
    tp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    sp = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_ in ss])
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    ne = (ctypes.c_int64 * n)(*[t.numel() for t in ts])
    check(_lib.lib().tgsr_axpy_images(n, op, tp, sp, ne, float(alpha), _stream()), "tgsr_axpy_images")
    return outs



def adam_flat(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, state: torch.Tensor,
              lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, advance: bool) -> None:
    """One Adam update over flat buffers, in place (tgsr_adam_flat): `state` = 3 device floats [step, 1 - b1^t, sqrt(1 - b2^t)],
    advanced on the device first when `advance`."""
    _need_hip(param, grad, exp_avg, exp_avg_sq, state)
    n = param.numel()
    for t in (param, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise TgsrError("adam_flat: dense fp32 buffers of one length expected")
    if state.dtype != torch.float32 or state.numel() != 3 or not state.is_contiguous():
        raise TgsrError("adam_flat: state = 3 fp32 values")
    check(_lib.lib().tgsr_adam_flat(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(state), n, float(lr), float(beta1),
                                    float(beta2), float(eps), float(weight_decay), 1 if advance else 0, _stream()), "tgsr_adam_flat")

def weighted_bce(a: torch.Tensor, b: Optional[torch.Tensor], target: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """sum_i weight[i] * BCEWithLogits([a; b][i], target[i]) as a 0-dim tensor (the adversarial terms of the GAN losses)."""
    _need_hip(a, b, target, weight)
    a = _f32(a, "a").contiguous().view(-1)
    b = None if b is None else _f32(b, "b").contiguous().view(-1)
    n = a.numel() + (0 if b is None else b.numel())
    if target.numel() != n or weight.numel() != n or target.dtype != torch.float32 or weight.dtype != torch.float32:
        raise TgsrError("weighted_bce: %d logits, %d targets, %d weights" % (n, target.numel(), weight.numel()))
    out = torch.empty((), dtype=torch.float32, device=a.device)
    check(_lib.lib().tgsr_weighted_bce_fwd(_p(a), a.numel(), _p(b), 0 if b is None else b.numel(), _p(target.contiguous()),
                                           _p(weight.contiguous()), _p(out), _stream()), "tgsr_weighted_bce_fwd")
    return out


def weighted_bce_bwd(dy: torch.Tensor, a: torch.Tensor, b: Optional[torch.Tensor], target: torch.Tensor, weight: torch.Tensor):
    """(da, db) of weighted_bce."""
    _need_hip(dy, a, b, target, weight)
    dy = _f32(dy, "dy").contiguous()
    af = a.contiguous().view(-1)
    bf = None if b is None else b.contiguous().view(-1)
    da = torch.empty_like(a, memory_format=torch.contiguous_format)
    db = None if b is None else torch.empty_like(b, memory_format=torch.contiguous_format)
    check(_lib.lib().tgsr_weighted_bce_bwd(_p(dy), _p(af), af.numel(), _p(bf), 0 if bf is None else bf.numel(),
                                           _p(target.contiguous()), _p(weight.contiguous()), _p(da), _p(db), _stream()),
          "tgsr_weighted_bce_bwd")
    return da, db


def axpy_map(t: torch.Tensor, s: torch.Tensor, amap: torch.Tensor) -> torch.Tensor:
    """t + amap * s with amap [H, W] broadcast over batch and channels (NetG_highweight(weightmap=True), model.py:276-297)."""
    _need_hip(t, s, amap)
    t, s, amap = _f32(t, "t").contiguous(), _f32(s, "s").contiguous(), _f32(amap.detach(), "amap").contiguous()
    if t.shape != s.shape or t.dim() != 4 or tuple(amap.shape) != tuple(t.shape[2:]):
        raise TgsrError("axpy_map: %s + %s * %s" % (tuple(t.shape), tuple(amap.shape), tuple(s.shape)))
    out = torch.empty_like(t)
    check(_lib.lib().tgsr_axpy_map_fwd(_p(t), _p(s), _p(amap), _p(out), t.shape[0] * t.shape[1], t.shape[2] * t.shape[3], _stream()),
          "tgsr_axpy_map_fwd")
    return out


def axpy_map_bwd(dy: torch.Tensor, s: torch.Tensor, amap: torch.Tensor, need_ds: bool = True, need_da: bool = True):
    """(ds, damap) of axpy_map: ds = amap * dy, damap = sum over batch and channels of dy * s."""
    _need_hip(dy, s, amap)
    dy, s, amap = _f32(dy, "dy").contiguous(), _f32(s, "s").contiguous(), _f32(amap.detach(), "amap").contiguous()
    ds = torch.empty_like(dy) if need_ds else None
    da = torch.empty_like(amap) if need_da else None
    if ds is None and da is None:
        return None, None
    check(_lib.lib().tgsr_axpy_map_bwd(_p(dy), _p(s), _p(amap), _p(ds), _p(da), dy.shape[0] * dy.shape[1], dy.shape[2] * dy.shape[3],
                                       _stream()), "tgsr_axpy_map_bwd")
    return ds, da


def affine_act(raw: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, act: int) -> torch.Tensor:
    """act(raw * scale[c] + shift[c]) on a dense NCHW tensor (eval-mode BatchNorm + LeakyReLU(0.2) for act = 2)."""
    _need_hip(raw, scale, shift)
    raw = _f32(raw, "raw").contiguous()
    B, Cc, H, W = raw.shape
    out = torch.empty_like(raw)
    check(_lib.lib().tgsr_affine_act_fwd(_p(raw), _p(scale.contiguous()), _p(shift.contiguous()), _p(out), B, Cc, H * W, int(act),
                                         _stream()), "tgsr_affine_act_fwd")
    return out


def affine_act_bwd(dy: torch.Tensor, out: Optional[torch.Tensor], scale: torch.Tensor, act: int) -> torch.Tensor:
    _need_hip(dy, out, scale)
    dy = _f32(dy, "dy").contiguous()
    B, Cc, H, W = dy.shape
    draw = torch.empty_like(dy)
    check(_lib.lib().tgsr_affine_act_bwd(_p(dy), _p(None if out is None else out.contiguous()), _p(scale.contiguous()), _p(draw),
                                         B, Cc, H * W, int(act), _stream()), "tgsr_affine_act_bwd")
    return draw


# ----------------------------------------------------------------------------------------- CNN_ENCODER's frozen trunk (tgsr_igemm.hip)
def _slice_ptr(t: torch.Tensor, coff: int):
    """Pointer to channel `coff` of a dense NCHW tensor and its batch stride in elements."""
    if t.dim() != 4 or not t.is_contiguous() or t.dtype != torch.float32:
        raise TgsrError("dense fp32 NCHW tensor expected, got %s %s" % (t.dtype, tuple(t.shape)))
    return t.data_ptr() + 4 * coff * t.shape[2] * t.shape[3], t.shape[1] * t.shape[2] * t.shape[3]


def gconv_pack(w: torch.Tensor, scale: Optional[torch.Tensor], dgrad: bool) -> torch.Tensor:
    """The filter [Cout, Cin, KH, KW] times the folded BatchNorm scale, as the A operand of gconv: forward [Cout, Cin KH KW] or data
    gradient [Cin, Cout KH KW]."""
    _need_hip(w, scale)
    w = _f32(w.detach(), "w").contiguous()
    Cout, Cin, KH, KW = w.shape
    out = torch.empty((Cin, Cout * KH * KW) if dgrad else (Cout, Cin * KH * KW), dtype=torch.float32, device=w.device)
    check(_lib.lib().tgsr_gconv_pack(_p(w), _p(scale), _p(out), Cout, Cin, KH * KW, 1 if dgrad else 0, _stream()), "tgsr_gconv_pack")
    return out


def gconv_set_form(split: bool) -> bool:
    """The arithmetic form of gconv: True (default) = three-piece bf16 operands on the bf16 matrix pipe where the shape qualifies,
    False = fp32 MFMA everywhere.  Returns the previous setting."""
    return bool(_lib.lib().tgsr_gconv_set_form(1 if split else 0))


def gconv_ws_elems(B: int, M: int, PH: int, PW: int, K: int) -> int:
    return int(_lib.lib().tgsr_gconv_ws_elems(B, M, PH, PW, K))


def gconv(dgrad: bool, A: torch.Tensor, S: torch.Tensor, s_coff: int, s_ch: int, out: torch.Tensor, o_coff: int, kh: int, kw: int,
          stride: int, padh: int, padw: int, bias: Optional[torch.Tensor], relu: bool, accumulate: bool, ws: Optional[torch.Tensor],
          mask: Optional[torch.Tensor] = None):
    """Convolution (dgrad False) or its data gradient (True) as one implicit GEMM: reads channels [s_coff, s_coff + s_ch) of S,
    writes channels [o_coff, o_coff + A.shape[0]) of out (+= when accumulate).  mask (shaped like out): the contribution is kept
    where mask > 0 (the ReLU of the tensor whose gradient `out` is)."""
    _need_hip(A, S, out, bias, ws, mask)
    if mask is not None and (mask.shape != out.shape or not mask.is_contiguous() or mask.dtype != torch.float32):
        raise TgsrError("gconv: the mask must be a dense fp32 tensor shaped like the output")
    M, K = A.shape
    B, Hs, Ws = S.shape[0], S.shape[2], S.shape[3]
    PH, PW = out.shape[2], out.shape[3]
    if K != s_ch * kh * kw or s_coff + s_ch > S.shape[1] or o_coff + M > out.shape[1] or out.shape[0] != B:
        raise TgsrError("gconv: A %s against %d channels of %s -> channels %d.. of %s" % (tuple(A.shape), s_ch, tuple(S.shape), o_coff,
                                                                                          tuple(out.shape)))
    if dgrad:
        ok = Hs == (PH + 2 * padh - kh) // stride + 1 and Ws == (PW + 2 * padw - kw) // stride + 1
    else:
        ok = PH == (Hs + 2 * padh - kh) // stride + 1 and PW == (Ws + 2 * padw - kw) // stride + 1
    if not ok:
        raise TgsrError("gconv: %dx%d / stride %d / pad (%d, %d) does not map %s to %s" % (kh, kw, stride, padh, padw, tuple(S.shape),
                                                                                          tuple(out.shape)))
    need = gconv_ws_elems(B, M, PH, PW, K)
    if need and (ws is None or ws.numel() < need):
        raise TgsrError("gconv: %d floats of workspace needed" % need)
    sp, sbs = _slice_ptr(S, s_coff)
    op, obs = _slice_ptr(out, o_coff)
    mp = None if mask is None else _slice_ptr(mask, o_coff)[0]
    check(_lib.lib().tgsr_gconv(1 if dgrad else 0, _p(A.contiguous()), sp, sbs, B, Hs, Ws, M, K, PH, PW, kh, kw, stride, padh, padw,
                                _p(bias), 1 if relu else 0, 1 if accumulate else 0, mp, op, obs, _p(ws), _stream()), "tgsr_gconv")




def sum_stack(stack: torch.Tensor, n: int, out: torch.Tensor):
    """out = ((stack[0] + stack[1]) + ...) + stack[n - 1]: the first n slices of a dense stack [N, ...] summed in order (tgsr_sum_stack)."""
    _need_hip(stack, out)
    if stack.dtype != torch.float32 or not stack.is_contiguous() or not out.is_contiguous() or out.dtype != torch.float32:
        raise TgsrError("sum_stack: dense fp32 tensors expected")
    if n < 1 or n > stack.shape[0] or tuple(stack.shape[1:]) != tuple(out.shape):
        raise TgsrError("sum_stack: %d slices of %s into %s" % (n, tuple(stack.shape), tuple(out.shape)))
    check(_lib.lib().tgsr_sum_stack(_p(stack), n, out.numel(), _p(out), _stream()), "tgsr_sum_stack")

def interleave2x2(parts, dx: torch.Tensor, accumulate: bool, mask: Optional[torch.Tensor] = None):
    """dx[:, :, py::2, px::2] (+)= parts[2 py + px] (dense [B, C, ceil((H - py) / 2), ceil((W - px) / 2)]), masked where mask <= 0:
    the four parity classes of a stride-2 convolution's data gradient woven into one tensor (tgsr_interleave2x2)."""
    _need_hip(dx, mask, *parts)
    B, Cc, H, W = dx.shape
    if not dx.is_contiguous() or dx.dtype != torch.float32 or len(parts) != 4:
        raise TgsrError("interleave2x2: a dense fp32 dx and four class tensors expected")
    for k, t in enumerate(parts):
        want = (B, Cc, (H - (k >> 1) + 1) // 2, (W - (k & 1) + 1) // 2)
        if tuple(t.shape) != want or not t.is_contiguous() or t.dtype != torch.float32:
            raise TgsrError("interleave2x2: class %d is %s, expected dense %s" % (k, tuple(t.shape), want))
    if mask is not None and (mask.shape != dx.shape or not mask.is_contiguous() or mask.dtype != torch.float32):
        raise TgsrError("interleave2x2: the mask must be a dense fp32 tensor shaped like dx")
    check(_lib.lib().tgsr_interleave2x2(_p(parts[0]), _p(parts[1]), _p(parts[2]), _p(parts[3]), _p(dx), B * Cc, H, W,
                                        1 if accumulate else 0, _p(mask), _stream()), "tgsr_interleave2x2")

def maxpool3s2(x: torch.Tensor, out: torch.Tensor, o_coff: int):
    _need_hip(x, out)
    B, Cc, H, W = x.shape
    xp, xbs = _slice_ptr(x, 0)
    op, obs = _slice_ptr(out, o_coff)
    if out.shape[2] != (H - 3) // 2 + 1 or out.shape[3] != (W - 3) // 2 + 1 or o_coff + Cc > out.shape[1]:
        raise TgsrError("maxpool3s2: %s -> %s" % (tuple(x.shape), tuple(out.shape)))
    check(_lib.lib().tgsr_maxpool3s2_fwd(xp, xbs, B, Cc, H, W, op, obs, _stream()), "tgsr_maxpool3s2_fwd")


def maxpool3s2_bwd(x: torch.Tensor, dy: torch.Tensor, dy_coff: int, dx: torch.Tensor, accumulate: bool, mask: Optional[torch.Tensor] = None):
    _need_hip(x, dy, dx, mask)
    if mask is not None and (mask.shape != dx.shape or not mask.is_contiguous()):
        raise TgsrError("maxpool3s2_bwd: the mask must be dense and shaped like dx")
    B, Cc, H, W = x.shape
    xp, xbs = _slice_ptr(x, 0)
    gp, gbs = _slice_ptr(dy, dy_coff)
    dp, dbs = _slice_ptr(dx, 0)
    if dx.shape != x.shape or dy_coff + Cc > dy.shape[1]:
        raise TgsrError("maxpool3s2_bwd: shapes")
    check(_lib.lib().tgsr_maxpool3s2_bwd(xp, xbs, gp, gbs, B, Cc, H, W, dp, dbs, 1 if accumulate else 0, _p(mask), _stream()),
          "tgsr_maxpool3s2_bwd")


def avgpool3(x: torch.Tensor, out: torch.Tensor, accumulate: bool, mask: Optional[torch.Tensor] = None):
    _need_hip(x, out, mask)
    if mask is not None and (mask.shape != out.shape or not mask.is_contiguous()):
        raise TgsrError("avgpool3: the mask must be dense and shaped like the output")
    if x.shape != out.shape:
        raise TgsrError("avgpool3: %s -> %s" % (tuple(x.shape), tuple(out.shape)))
    B, Cc, H, W = x.shape
    xp, xbs = _slice_ptr(x, 0)
    op, obs = _slice_ptr(out, 0)
    check(_lib.lib().tgsr_avgpool3(xp, xbs, B, Cc, H, W, op, obs, 1 if accumulate else 0, _p(mask), _stream()), "tgsr_avgpool3")


def plane_mean(x: torch.Tensor) -> torch.Tensor:
    _need_hip(x)
    x = _f32(x, "x").contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    check(_lib.lib().tgsr_plane_mean(_p(x), _p(out), B * Cc, H * W, _stream()), "tgsr_plane_mean")
    return out


def plane_mean_bwd(dy: torch.Tensor, H: int, W: int) -> torch.Tensor:
    _need_hip(dy)
    dy = _f32(dy, "dy").contiguous()
    B, Cc = dy.shape
    dx = torch.empty(B, Cc, H, W, dtype=torch.float32, device=dy.device)
    check(_lib.lib().tgsr_plane_mean_bwd(_p(dy), _p(dx), B * Cc, H * W, _stream()), "tgsr_plane_mean_bwd")
    return dx


def relu_mask_(dy: torch.Tensor, y: torch.Tensor, coff: int, ch: int):
    """dy[:, coff:coff+ch] *= (y[:, coff:coff+ch] > 0) in place (two equally shaped dense NCHW tensors)."""
    _need_hip(dy, y)
    if dy.shape != y.shape or coff + ch > dy.shape[1]:
        raise TgsrError("relu_mask: %s / %s" % (tuple(dy.shape), tuple(y.shape)))
    gp, gbs = _slice_ptr(dy, coff)
    yp, ybs = _slice_ptr(y, coff)
    check(_lib.lib().tgsr_relu_mask(gp, gbs, yp, ybs, gp, gbs, dy.shape[0], ch * dy.shape[2] * dy.shape[3], _stream()), "tgsr_relu_mask")


def bilinear(x: torch.Tensor, OH: int, OW: int) -> torch.Tensor:
    _need_hip(x)
    x = _f32(x, "x").contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty(B, Cc, OH, OW, dtype=torch.float32, device=x.device)
    check(_lib.lib().tgsr_bilinear_fwd(_p(x), B * Cc, H, W, OH, OW, _p(out), _stream()), "tgsr_bilinear_fwd")
    return out


def bilinear_bwd(dy: torch.Tensor, H: int, W: int) -> torch.Tensor:
    _need_hip(dy)
    dy = _f32(dy, "dy").contiguous()
    B, Cc, OH, OW = dy.shape
    dx = torch.empty(B, Cc, H, W, dtype=torch.float32, device=dy.device)
    check(_lib.lib().tgsr_bilinear_bwd(_p(dy), B * Cc, H, W, OH, OW, _p(dx), _stream()), "tgsr_bilinear_bwd")
    return dx


# ----------------------------------------------------------------------------------------- image pyramid (uint8)
def resize_bilinear_u8(x: torch.Tensor, out_h: int, out_w: int, htab, vtab) -> torch.Tensor:
    """Pillow's `resize(BILINEAR)` of planar uint8 images [..., H, W]; htab / vtab = (bounds, coefficients, ksize) device
    tables of the horizontal / vertical pass, or None when that size does not change."""
    _need_hip(x)
    if x.dtype != torch.uint8:
        raise TgsrError("resize: uint8 images expected, got %s" % x.dtype)
    x = x.contiguous()
    H, W = x.shape[-2], x.shape[-1]
    N = x.numel() // (H * W)
    out = torch.empty(x.shape[:-2] + (out_h, out_w), dtype=torch.uint8, device=x.device)
    hb, hk, hks = htab if htab is not None else (None, None, 0)
    vb, vk, vks = vtab if vtab is not None else (None, None, 0)
    tmp = torch.empty(N * H * out_w, dtype=torch.uint8, device=x.device) if (hb is not None and vb is not None) else None
    check(_lib.lib().tgsr_resize_bilinear_u8(_p(x), N, H, W, out_h, out_w, _p(hb), _p(hk), hks, _p(vb), _p(vk), vks, _p(tmp),
                                             _p(out), _stream()), "tgsr_resize_bilinear_u8")
    return out


def gaussian_blur_u8(x: torch.Tensor, r: int, ww: int, fw: int, passes: int = 3) -> torch.Tensor:
    """Pillow's GaussianBlur (extended box blur, `passes` per axis) of planar uint8 images [..., H, W]."""
    _need_hip(x)
    x = x.contiguous()
    H, W = x.shape[-2], x.shape[-1]
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    check(_lib.lib().tgsr_gaussian_blur_u8(_p(x), x.numel() // (H * W), H, W, r, ww, fw, passes, _p(tmp), _p(out), _stream()),
          "tgsr_gaussian_blur_u8")
    return out


def u8_normalize(x: torch.Tensor) -> torch.Tensor:
    """ToTensor + Normalize((0.5,)*3, (0.5,)*3) (datasets.py:286-288): uint8 -> float32 in [-1, 1]."""
    _need_hip(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_lib.lib().tgsr_u8_normalize(_p(x), _p(out), x.numel(), _stream()), "tgsr_u8_normalize")
    return out


# ----------------------------------------------------------------------------------------- stand-alone GLU
def glu(x: torch.Tensor) -> torch.Tensor:
    """GLU.forward (util.py:45-53): x[:, :C/2] * sigmoid(x[:, C/2:]) for x [B, C, ...] with C even."""
    _need_hip(x)
    x = _f32(x, "x").contiguous()
    if x.dim() < 2 or x.shape[1] % 2 != 0:
        raise TgsrError("glu: channels dont divide 2! (shape %s)" % (tuple(x.shape),))
    B, nc = x.shape[0], x.shape[1] // 2
    out = torch.empty((B, nc) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    if out.numel():
        check(_lib.lib().tgsr_glu(_p(x), None, _p(out), B, out.numel() // B, _stream()), "tgsr_glu")
    return out


def glu_bwd(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Backward of glu: dy [B, C/2, ...], x [B, C, ...] -> dx like x."""
    _need_hip(dy, x)
    x = _f32(x, "x").contiguous()
    dy = _f32(dy, "dy").contiguous()
    dx = torch.empty_like(x)
    if dx.numel():
        check(_lib.lib().tgsr_glu(_p(x), _p(dy), _p(dx), x.shape[0], dy.numel() // x.shape[0], _stream()), "tgsr_glu")
    return dx


# ----------------------------------------------------------------------------------------- image epilogue
def to_uint8(img: torch.Tensor) -> torch.Tensor:
    """trainer_objective.py:153-155 on the device: round(clip((x + 1) * 127.5, 0, 255)) -> uint8, same shape.
    Byte-identical to the reference's numpy expression (float32 add, multiply, round-half-even)."""
    _need_hip(img)
    x = _f32(img.detach(), "img").contiguous()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(_lib.lib().tgsr_to_uint8(_p(x), _p(out), x.numel(), _stream()), "tgsr_to_uint8")
    return out


# ----------------------------------------------------------------------------------------- discriminator convolutions
def _dconv(kind: int):
    L = _lib.lib()
    if kind == 4:
        return L.tgsr_conv4x4s2_ws_elems, L.tgsr_conv4x4s2_fwd, L.tgsr_conv4x4s2_dgrad, L.tgsr_conv4x4s2_wgrad, "tgsr_conv4x4s2"
    return (L.tgsr_conv3x3_gemm_ws_elems, L.tgsr_conv3x3_gemm_fwd, L.tgsr_conv3x3_gemm_dgrad, L.tgsr_conv3x3_gemm_wgrad,
            "tgsr_conv3x3_gemm")


def _dconv_fwd(kind, x, w, leaky):
    _need_hip(x, w)
    x = _f32(x, "x").contiguous()
    w = _f32(w.detach(), "w").contiguous()
    B, Cin, H, W = x.shape
    if tuple(w.shape[1:]) != (Cin, kind, kind):
        raise TgsrError("conv%dx%d: weight %s vs input %s" % (kind, kind, tuple(w.shape), tuple(x.shape)))
    ws_elems, fwd, _, _, name = _dconv(kind)
    Cout = w.shape[0]
    Ho, Wo = (H // 2, W // 2) if kind == 4 else (H, W)
    out = torch.empty(B, Cout, Ho, Wo, dtype=torch.float32, device=x.device)
    ws = torch.empty(ws_elems(0, B, Cin, H, W, Cout), dtype=torch.float32, device=x.device)
    e0 = _ev() if profile is not None else None
    if kind == 4:
        check(fwd(_p(x), B, Cin, H, W, _p(w), Cout, 1 if leaky else 0, _p(ws), _p(out), _stream()), name + "_fwd")
    else:
        assert not leaky
        check(fwd(_p(x), B, Cin, H, W, _p(w), Cout, _p(ws), _p(out), _stream()), name + "_fwd")
    _dconv_profile(e0, kind, B, Cin, H, W, Cout, Ho, Wo)
    return out


def _dconv_profile(e0, kind, B, Cin, H, W, Cout, Ho, Wo, op=0):
    """bench.py's per-launch hook for the discriminators' implicit-GEMM kernels (forward / data / weight gradient alike:
    2 B Ho Wo Cout Cin K^2 FLOPs, every one of them issued - no Winograd saving here; `dconv_igemm6_kernel` when the library
    takes the three-piece bf16 form for this shape: six bf16 MFMAs per fp32 MAC)."""
    if profile is not None:
        L = _lib.lib()
        split = (L.tgsr_conv4x4s2_split_form if kind == 4 else L.tgsr_conv3x3_gemm_split_form)(op, B, Cin, H, W, Cout)
        profile.append(("dconv_igemm6_kernel" if split else "dconv_igemm_kernel", 2.0 * B * Ho * Wo * Cout * Cin * kind * kind,
                        4.0 * (B * Cin * H * W + B * Cout * Ho * Wo + Cout * Cin * kind * kind), e0, _ev()))


def _dconv_dgrad(kind, dy, w, H, W):
    _need_hip(dy, w)
    dy = _f32(dy, "dy").contiguous()
    w = _f32(w.detach(), "w").contiguous()
    B, Cout, Cin = dy.shape[0], w.shape[0], w.shape[1]
    ws_elems, _, dgrad, _, name = _dconv(kind)
    dx = torch.empty(B, Cin, H, W, dtype=torch.float32, device=dy.device)
    ws = torch.empty(ws_elems(1, B, Cin, H, W, Cout), dtype=torch.float32, device=dy.device)
    e0 = _ev() if profile is not None else None
    check(dgrad(_p(dy), B, Cin, H, W, _p(w), Cout, _p(ws), _p(dx), _stream()), name + "_dgrad")
    _dconv_profile(e0, kind, B, Cin, H, W, Cout, dy.shape[2], dy.shape[3], 1)
    return dx


def _dconv_wgrad(kind, dy, x, out=None):
    _need_hip(dy, x)
    dy = _f32(dy, "dy").contiguous()
    x = _f32(x, "x").contiguous()
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    ws_elems, _, _, wgrad, name = _dconv(kind)
    ws = torch.empty(ws_elems(2, B, Cin, H, W, Cout), dtype=torch.float32, device=x.device)
    dw = out if out is not None else torch.empty(Cout, Cin, kind, kind, dtype=torch.float32, device=x.device)
    e0 = _ev() if profile is not None else None
    check(wgrad(_p(dy), _p(x), B, Cin, H, W, Cout, _p(ws), _p(dw), _stream()), name + "_wgrad")
    _dconv_profile(e0, kind, B, Cin, H, W, Cout, dy.shape[2], dy.shape[3], 2)
    return dw


def dconv_set_split(on: int) -> int:
    """The discriminators' 4x4 convolutions on the bf16 matrix pipe with three-piece fp32 operands (default on; profiles/HISTORY.md 3.18) or
    on the fp32 MFMA.  Returns the previous setting (tgsr_dconv_set_split)."""
    return _lib.lib().tgsr_dconv_set_split(int(on))


def conv4x4s2(x: torch.Tensor, w: torch.Tensor, leaky: bool = False) -> torch.Tensor:
    """nn.Conv2d(Cin, Cout, 4, 2, 1, bias=False) (downBlock, util.py:92-98) [+ LeakyReLU(0.2)]: x [B,Cin,H,W] ->
    [B,Cout,H/2,W/2]."""
    return _dconv_fwd(4, x, w, leaky)


def conv4x4s2_dgrad(dy: torch.Tensor, w: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """Data gradient of conv4x4s2: dy [B,Cout,H/2,W/2] -> dx [B,Cin,H,W]."""
    return _dconv_dgrad(4, dy, w, H, W)


def conv4x4s2_wgrad(dy: torch.Tensor, x: torch.Tensor, out=None) -> torch.Tensor:
    """Weight gradient of conv4x4s2: dy [B,Cout,H/2,W/2], x [B,Cin,H,W] -> dw [Cout,Cin,4,4]."""
    return _dconv_wgrad(4, dy, x, out)


def conv3x3_gemm(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """conv3x3 stride 1 pad 1 without bias as an implicit GEMM: the discriminators' many-channel, few-pixel layers
    (Block3x3_leakRelu at 4x4 pixels, 512 ... 2048 channels)."""
    return _dconv_fwd(3, x, w, False)


def conv3x3_gemm_dgrad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return _dconv_dgrad(3, dy, w, dy.shape[2], dy.shape[3])


def conv3x3_gemm_wgrad(dy: torch.Tensor, x: torch.Tensor, out=None) -> torch.Tensor:
    return _dconv_wgrad(3, dy, x, out)


def conv3x3_gemm_pays(Cin: int, Cout: int, H: int, W: int) -> bool:
    """Many channels on few pixels: the GEMM form; the generator's shapes stay on the tiled conv kernels."""
    return Cin >= 256 and Cout >= 256 and H * W <= 64 * 64


def leaky_relu_bwd(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dy * (y > 0 ? 1 : 0.2) from the forward OUTPUT y of a LeakyReLU(0.2)."""
    _need_hip(dy, y)
    dy = _f32(dy, "dy").contiguous()
    y = _f32(y, "y").contiguous()
    out = torch.empty_like(dy)
    check(_lib.lib().tgsr_leaky_relu(_p(dy), _p(y), _p(out), dy.numel(), _stream()), "tgsr_leaky_relu")
    return out
