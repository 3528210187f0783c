// ABI bookkeeping + the small layout/prep kernels (weight packing, BatchNorm-eval folding).
#include <stdio.h>

#include "tgsr_common.h"

namespace tgsr {

static char g_last_error[256] = "no error";

int note_launch(hipError_t e, const char* what) {
  if (e == hipSuccess) return TGSR_OK;
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
  return TGSR_ELAUNCH;
}

int note_error(const char* what, const char* detail, int code) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", what, detail ? detail : "error", code);
  return TGSR_ELAUNCH;
}

// wpack[chunk][tap][ci][Cout] <- w[Cout][Cin][K][K]; channels past Cin are zero.
// tr != 0: the source is the FORWARD conv's weight [Cin][Cout][K][K] and the pack is of the data-gradient conv
// w'[co][c][tap] = w[c][co][KK - 1 - tap] (in/out swapped, taps flipped) - no flip/transpose/copy kernels beforehand
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KK,
                                        int tr, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    int64_t t = i / Cout;
    const int ci = (int)(t % kConvCK);
    t /= kConvCK;
    const int tap = (int)(t % KK);
    const int chunk = (int)(t / KK);
    const int c = chunk * kConvCK + ci;
    wp[i] = c < Cin ? (tr ? w[((int64_t)c * Cout + co) * KK + (KK - 1 - tap)] : w[((int64_t)co * Cin + c) * KK + tap]) : 0.f;
  }
}

__global__ void bn_fold_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ m,
                               const float* __restrict__ v, float eps, float* __restrict__ scale,
                               float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float s = g[c] / sqrtf(v[c] + eps);
    scale[c] = s;
    shift[c] = b[c] - m[c] * s;
  }
}

// uint8 image epilogue of the reference's caller (trainer_objective.py:153-155):
//   round(clip((x + 1) * 127.5, 0, 255)) with numpy's float32 arithmetic (add, then multiply - no FMA - and
//   round-half-to-even), so the bytes are identical to the host-side formula.  4 pixels per thread when aligned.
__global__ void to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto cvt = [](float v) {
    const float t = __fmul_rn(__fadd_rn(v, 1.0f), 127.5f);
    return (uint8_t)(int)rintf(fminf(255.f, fmaxf(0.f, t)));
  };
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 3) == 0)) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += stride) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      uchar4 o;
      o.x = cvt(v.x); o.y = cvt(v.y); o.z = cvt(v.z); o.w = cvt(v.w);
      reinterpret_cast<uchar4*>(out)[i] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = cvt(x[i]);
  }
}

// GLU (util.py:45-53): out[o][i] = x[o][0][i] * sigmoid(x[o][1][i]) over x viewed as [outer][2][half] (channel halves of
// [B, 2C, ...]).  The fused conv kernels carry this in their epilogues; this is the module's stand-alone form (CA_NET
// calls it on [B, 400], util.py:381).  dy != NULL: the backward, dx[o][0][i] = dy s, dx[o][1][i] = dy v s (1 - s).
__global__ void glu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out,
                           int64_t outer, int64_t half) {
  const int64_t n = outer * half, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t o = i / half, r = i - o * half;
    const float v = x[(2 * o) * half + r], g = x[(2 * o + 1) * half + r];
    const float sg = 1.f / (1.f + expf(-g));
    if (dy == nullptr) {
      out[i] = v * sg;
    } else {
      const float d = dy[i];
      out[(2 * o) * half + r] = d * sg;
      out[(2 * o + 1) * half + r] = d * v * sg * (1.f - sg);
    }
  }
}

// Up to 16 dense device buffers copied in ONE launch (the static inputs of a captured step: 3 per lane).  blockIdx.y =
// segment; 16-byte words when every pointer and size of the segment allows, 4-byte words otherwise.
struct MultiCopyArgs {
  void* dst[16];
  const void* src[16];
  uint32_t bytes[16];
};

__global__ __launch_bounds__(256) void multi_copy_kernel(MultiCopyArgs a) {
  const int sgm = blockIdx.y;
  const uint32_t nb = a.bytes[sgm];
  const uintptr_t both = reinterpret_cast<uintptr_t>(a.dst[sgm]) | reinterpret_cast<uintptr_t>(a.src[sgm]) | nb;
  if ((both & 15) == 0) {
    const uint4* s = static_cast<const uint4*>(a.src[sgm]);
    uint4* d = static_cast<uint4*>(a.dst[sgm]);
    for (uint32_t o = blockIdx.x * 256 + threadIdx.x; o < (nb >> 4); o += gridDim.x * 256) d[o] = s[o];
  } else {
    const uint32_t* s = static_cast<const uint32_t*>(a.src[sgm]);
    uint32_t* d = static_cast<uint32_t*>(a.dst[sgm]);
    for (uint32_t o = blockIdx.x * 256 + threadIdx.x; o < (nb >> 2); o += gridDim.x * 256) d[o] = s[o];
  }
}

// out_i = t_i + alpha * s_i for up to 4 dense images in one launch: NetG_highweight's `one * tanh(conv5x5(out)) + a * SRb`
// (model.py:280, 288, 297) with the tanh(conv) part computed EARLIER, beside G_SR_NET_low, by tgsr_conv_to3_fwd without an
// addend - the same fma(alpha, s, t) that kernel's own epilogue evaluates when it is handed the addend, so the images are
// bit-identical.  blockIdx.y = image; 16-byte words (sizes are multiples of 4 floats: host-checked).
struct AxpyArgs {
  float* out[4];
  const float* t[4];
  const float* s[4];
  uint32_t n4[4];
  float alpha;
};

__global__ __launch_bounds__(256) void axpy_images_kernel(AxpyArgs a) {
  const int k = blockIdx.y;
  const float4* __restrict__ t = reinterpret_cast<const float4*>(a.t[k]);
  const float4* __restrict__ sr = reinterpret_cast<const float4*>(a.s[k]);
  float4* __restrict__ o = reinterpret_cast<float4*>(a.out[k]);
  const float al = a.alpha;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.n4[k]; i += gridDim.x * 256) {
    const float4 tv = t[i], sv = sr[i];
    float4 r;
    r.x = fmaf(al, sv.x, tv.x);
    r.y = fmaf(al, sv.y, tv.y);
    r.z = fmaf(al, sv.z, tv.z);
    r.w = fmaf(al, sv.w, tv.w);
    o[i] = r;
  }
}

// out[b][c][p] = t[b][c][p] + amap[p] * s[b][c][p]: NetG_highweight(weightmap=True)'s heads (model.py:235-245, 276-297: `a_k` is a
// trainable [H, W] map broadcast over batch and channels).  One float4 of pixels per thread; HW % 4 == 0.
__global__ __launch_bounds__(256) void axpy_map_kernel(const float* __restrict__ t, const float* __restrict__ s,
                                                       const float* __restrict__ amap, float* __restrict__ out, int64_t n4,
                                                       uint32_t hw4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 tv = reinterpret_cast<const float4*>(t)[i], sv = reinterpret_cast<const float4*>(s)[i];
    const float4 av = reinterpret_cast<const float4*>(amap)[i % hw4];
    float4 r;
    r.x = fmaf(av.x, sv.x, tv.x);
    r.y = fmaf(av.y, sv.y, tv.y);
    r.z = fmaf(av.z, sv.z, tv.z);
    r.w = fmaf(av.w, sv.w, tv.w);
    reinterpret_cast<float4*>(out)[i] = r;
  }
}

// Its backward: ds = amap * dy (when ds != nullptr) and damap[p] = sum over the BC planes of dy * s, planes added in index order
// (one thread owns a float4 of pixels: deterministic, coalesced across the plane).
__global__ __launch_bounds__(256) void axpy_map_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                           const float* __restrict__ amap, float* __restrict__ ds,
                                                           float* __restrict__ damap, int BC, uint32_t hw4) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw4) return;
  const float4 av = reinterpret_cast<const float4*>(amap)[p];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < BC; ++k) {
    const int64_t i = (int64_t)k * hw4 + p;
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    if (s) {
      const float4 sv = reinterpret_cast<const float4*>(s)[i];
      acc.x = fmaf(g.x, sv.x, acc.x); acc.y = fmaf(g.y, sv.y, acc.y);
      acc.z = fmaf(g.z, sv.z, acc.z); acc.w = fmaf(g.w, sv.w, acc.w);
    }
    if (ds) reinterpret_cast<float4*>(ds)[i] = make_float4(av.x * g.x, av.y * g.y, av.z * g.z, av.w * g.w);
  }
  if (damap) reinterpret_cast<float4*>(damap)[p] = acc;
}

// Eval-mode BatchNorm (running statistics folded to scale / shift by bn_fold_kernel) + activation on a raw convolution output:
// downBlock / Block3x3_leakRelu under .eval() (util.py:92-98).  act 0: none, 2: LeakyReLU(0.2).  grid (C, splits), HW % 4 == 0.
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ out, int B, int C,
                                                         int HW, int act) {
  const int c = blockIdx.x;
  const float sv = scale[c], tv = shift[c];
  const int64_t total = (int64_t)B * HW;
  for (int64_t e = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 4; e < total; e += (int64_t)gridDim.y * 1024) {
    const int b = (int)(e / HW);
    const int64_t o = ((int64_t)b * C + c) * HW + (e - (int64_t)b * HW);
    const float4 v = *reinterpret_cast<const float4*>(raw + o);
    float4 y = make_float4(fmaf(v.x, sv, tv), fmaf(v.y, sv, tv), fmaf(v.z, sv, tv), fmaf(v.w, sv, tv));
    if (act == 2) {
      y.x = y.x > 0.f ? y.x : 0.2f * y.x; y.y = y.y > 0.f ? y.y : 0.2f * y.y;
      y.z = y.z > 0.f ? y.z : 0.2f * y.z; y.w = y.w > 0.f ? y.w : 0.2f * y.w;
    }
    *reinterpret_cast<float4*>(out + o) = y;
  }
}

// d(raw) = dy * act'(out) * scale[c]  (LeakyReLU's slope is positive: the sign of `out` is the sign of its argument)
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                                                             const float* __restrict__ scale, float* __restrict__ draw, int B,
                                                             int C, int HW, int act) {
  const int c = blockIdx.x;
  const float sv = scale[c];
  const int64_t total = (int64_t)B * HW;
  for (int64_t e = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 4; e < total; e += (int64_t)gridDim.y * 1024) {
    const int b = (int)(e / HW);
    const int64_t o = ((int64_t)b * C + c) * HW + (e - (int64_t)b * HW);
    const float4 g = *reinterpret_cast<const float4*>(dy + o);
    float4 r = make_float4(g.x * sv, g.y * sv, g.z * sv, g.w * sv);
    if (act == 2) {
      const float4 y = *reinterpret_cast<const float4*>(out + o);
      r.x = y.x > 0.f ? r.x : 0.2f * r.x; r.y = y.y > 0.f ? r.y : 0.2f * r.y;
      r.z = y.z > 0.f ? r.z : 0.2f * r.z; r.w = y.w > 0.f ? r.w : 0.2f * r.w;
    }
    *reinterpret_cast<float4*>(draw + o) = r;
  }
}

// sum_i w[i] * BCEWithLogits(l[i], t[i]) over the concatenation l = [a (na values); b (nb values)]: the terms of discriminator_loss
// / generator_loss (losses.py:290-316, 358-371: up to five mean-reduced F.binary_cross_entropy_with_logits calls, their /2 and /3
// weights folded into w) in ONE launch, one workgroup, the lanes' partial sums combined in a fixed order.
// BCEWithLogits(l, t) = max(l, 0) - l t + log1p(exp(-|l|)).
__global__ __launch_bounds__(256) void weighted_bce_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                                                           const float* __restrict__ t, const float* __restrict__ w,
                                                           float* __restrict__ out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < na + nb; i += 256) {
    const float l = i < na ? a[i] : b[i - na];
    acc += w[i] * (fmaxf(l, 0.f) - l * t[i] + log1pf(expf(-fabsf(l))));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// d/dl[i] = dy * w[i] * (sigmoid(l[i]) - t[i])
__global__ __launch_bounds__(256) void weighted_bce_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a, int na,
                                                               const float* __restrict__ b, int nb, const float* __restrict__ t,
                                                               const float* __restrict__ w, float* __restrict__ da,
                                                               float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= na + nb) return;
  const float l = i < na ? a[i] : b[i - na];
  const float g = dy[0] * w[i] * (1.f / (1.f + expf(-l)) - t[i]);
  if (i < na) { if (da) da[i] = g; }
  else if (db) db[i - na] = g;
}


// Adam over flat buffers (torch.optim.Adam's arithmetic: no amsgrad, L2 weight decay folded into the gradient): parameters,
// gradients and both moments of a network each live in ONE dense fp32 buffer (tgsr_amd/optim.py re-homes the parameters as views of
// theirs; the gradients are parallel.FlatGradBucket's), so the update is one pass - 16 bytes read, 12 written per element, HBM-bound -
// instead of the ~10 multi-tensor launches per group of torch's fused form (1.0 ms per G/D step over the discriminators' 100 M
// parameters against 0.45 at the roof).  The step count lives on the device (hipGraph replays advance it): `adam_advance_kernel`
// (one thread) bumps it and leaves the two bias corrections, `adam_flat_kernel` reads them.
//   m = m + (g - m) (1 - b1);  v = v b2 + (1 - b2) g g;  p = p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void adam_advance_kernel(float* __restrict__ state, double b1, double b2) {   // state: [step, 1 - b1^t, sqrt(1 - b2^t)]
  const float t = state[0] + 1.f;
  state[0] = t;
  // in double, from the double betas, as torch's default path forms them on the host (1 - 0.999^t in fp32 keeps ~5 digits for small t)
  state[1] = (float)(1.0 - pow(b1, (double)t));
  state[2] = (float)sqrt(1.0 - pow(b2, (double)t));
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const float* __restrict__ state, int64_t n4, int64_t n,
                                                        float lr, float omb1, float b2, float omb2, float eps, float wd) {
  // (omb1 = 1 - beta1, omb2 = 1 - beta2 rounded from the DOUBLE differences, as torch forms them: 1.f - 0.999f is 4.7e-5 off 0.001f)
  const float step_size = lr / state[1], rs2 = 1.f / state[2];
  auto upd = [&](float& pv, float gv, float& mv, float& vv) {
    if (wd != 0.f) gv = fmaf(wd, pv, gv);
    mv = fmaf(gv - mv, omb1, mv);
    vv = fmaf(vv, b2, omb2 * gv * gv);
    pv = pv - step_size * (mv / (sqrtf(vv) * rs2 + eps));
  };
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    upd(pv.x, gv.x, mv.x, vv.x);
    upd(pv.y, gv.y, mv.y, vv.y);
    upd(pv.z, gv.z, mv.z, vv.z);
    upd(pv.w, gv.w, mv.w, vv.w);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) upd(p[i], g[i], m[i], v[i]);
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_abi_version(void) { return TGSR_ABI_VERSION; }
extern "C" const char* tgsr_last_error(void) { return g_last_error; }

extern "C" int64_t tgsr_packed_weight_elems(int Cout, int Cin, int K) {
  return (int64_t)((Cin + kConvCK - 1) / kConvCK) * K * K * kConvCK * Cout;
}

static int pack_conv_weight(const float* w, float* wpack, int Cout, int Cin, int K, int tr, void* stream) {
  if (!w || !wpack || Cout < 1 || Cin < 1 || K < 1) return TGSR_EINVAL;
  const int64_t total = tgsr_packed_weight_elems(Cout, Cin, K);
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, wpack, Cout, Cin,
                     K * K, tr, total);
  return note_launch(hipGetLastError(), "pack_conv_weight_kernel");
}

extern "C" int tgsr_pack_conv_weight(const float* w, float* wpack, int Cout, int Cin, int K, void* stream) {
  return pack_conv_weight(w, wpack, Cout, Cin, K, 0, stream);
}

extern "C" int tgsr_pack_conv_weight_dgrad(const float* w, float* wpack, int Cout, int Cin, int K, void* stream) {
  return pack_conv_weight(w, wpack, Cout, Cin, K, 1, stream);
}

extern "C" int tgsr_bn_fold(const float* weight, const float* bias, const float* running_mean,
                            const float* running_var, float eps, float* scale, float* shift, int C, void* stream) {
  if (!weight || !bias || !running_mean || !running_var || !scale || !shift || C < 1) return TGSR_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), weight, bias,
                     running_mean, running_var, eps, scale, shift, C);
  return note_launch(hipGetLastError(), "bn_fold_kernel");
}

extern "C" int tgsr_to_uint8(const float* x, uint8_t* out, int64_t n, void* stream) {
  if (!x || !out || n < 1) return TGSR_EINVAL;
  const int64_t work = (n + 3) / 4;
  const int blocks = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
  hipLaunchKernelGGL(to_uint8_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, out, n);
  return note_launch(hipGetLastError(), "to_uint8_kernel");
}

extern "C" int tgsr_glu(const float* x, const float* dy, float* out, int64_t outer, int64_t half, void* stream) {
  if (!x || !out || outer < 1 || half < 1) return TGSR_EINVAL;
  const int64_t n = outer * half;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(glu_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, dy, out, outer, half);
  return note_launch(hipGetLastError(), "glu_kernel");
}

extern "C" int tgsr_multi_copy(int n, void* const* dst, const void* const* src, const int64_t* nbytes, void* stream) {
  if (n < 1 || !dst || !src || !nbytes) return TGSR_EINVAL;
  if (n > 16) return TGSR_EUNSUPPORTED;
  MultiCopyArgs a;
  uint32_t most = 0;
  for (int i = 0; i < 16; ++i) {
    a.dst[i] = nullptr; a.src[i] = nullptr; a.bytes[i] = 0;
    if (i >= n) continue;
    if (!dst[i] || !src[i] || nbytes[i] < 0) return TGSR_EINVAL;
    if (nbytes[i] > 0x7fffffff || (nbytes[i] & 3) ||
        ((reinterpret_cast<uintptr_t>(dst[i]) | reinterpret_cast<uintptr_t>(src[i])) & 3))
      return TGSR_EUNSUPPORTED;
    a.dst[i] = dst[i]; a.src[i] = src[i]; a.bytes[i] = (uint32_t)nbytes[i];
    most = a.bytes[i] > most ? a.bytes[i] : most;
  }
  if (most == 0) return TGSR_OK;
  const uint32_t words = (most + 15) / 16;
  const int bx = (int)((words + 255) / 256 < 64 ? (words + 255) / 256 : 64);
  hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, n), dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "multi_copy_kernel");
}

extern "C" int tgsr_axpy_images(int n, float* const* out, const float* const* t, const float* const* s, const int64_t* numel,
                                float alpha, void* stream) {
  if (n < 1 || !out || !t || !s || !numel) return TGSR_EINVAL;
  if (n > 4) return TGSR_EUNSUPPORTED;
  AxpyArgs a;
  uint32_t most = 0;
  for (int i = 0; i < 4; ++i) {
    a.out[i] = nullptr; a.t[i] = nullptr; a.s[i] = nullptr; a.n4[i] = 0;
    if (i >= n) continue;
    if (!out[i] || !t[i] || !s[i] || numel[i] < 1) return TGSR_EINVAL;
    if ((numel[i] & 3) || numel[i] > 0x7fffffff ||
        ((reinterpret_cast<uintptr_t>(out[i]) | reinterpret_cast<uintptr_t>(t[i]) | reinterpret_cast<uintptr_t>(s[i])) & 15))
      return TGSR_EUNSUPPORTED;
    a.out[i] = out[i]; a.t[i] = t[i]; a.s[i] = s[i]; a.n4[i] = (uint32_t)(numel[i] >> 2);
    most = a.n4[i] > most ? a.n4[i] : most;
  }
  a.alpha = alpha;
  const int bx = (int)((most + 255) / 256 < 512 ? (most + 255) / 256 : 512);
  hipLaunchKernelGGL(axpy_images_kernel, dim3(bx, n), dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "axpy_images_kernel");
}

extern "C" int tgsr_axpy_map_fwd(const float* t, const float* s, const float* amap, float* out, int BC, int HW, void* stream) {
  if (!t || !s || !amap || !out || BC < 1 || HW < 1) return TGSR_EINVAL;
  if ((HW & 3) || ((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(amap) |
                    reinterpret_cast<uintptr_t>(out)) & 15))
    return TGSR_EUNSUPPORTED;
  const int64_t n4 = (int64_t)BC * (HW >> 2);
  const int bx = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(axpy_map_kernel, dim3(bx), dim3(256), 0, as_stream(stream), t, s, amap, out, n4, (uint32_t)(HW >> 2));
  return note_launch(hipGetLastError(), "axpy_map_kernel");
}

extern "C" int tgsr_axpy_map_bwd(const float* dy, const float* s, const float* amap, float* ds, float* damap, int BC, int HW,
                                 void* stream) {
  if (!dy || !amap || BC < 1 || HW < 1 || (!ds && !damap) || (damap && !s)) return TGSR_EINVAL;
  if ((HW & 3) || ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(amap) |
                    reinterpret_cast<uintptr_t>(ds) | reinterpret_cast<uintptr_t>(damap)) & 15))
    return TGSR_EUNSUPPORTED;
  const uint32_t hw4 = (uint32_t)(HW >> 2);
  hipLaunchKernelGGL(axpy_map_bwd_kernel, dim3((hw4 + 255) / 256), dim3(256), 0, as_stream(stream), dy, damap ? s : nullptr, amap,
                     ds, damap, BC, hw4);
  return note_launch(hipGetLastError(), "axpy_map_bwd_kernel");
}

static int affine_grid_y(int B, int C, int HW) {
  const int64_t per = ((int64_t)B * HW + 1023) / 1024;
  int64_t want = (2048 + C - 1) / C;
  if (want > per) want = per;
  return (int)(want < 1 ? 1 : want);
}

extern "C" int tgsr_affine_act_fwd(const float* raw, const float* scale, const float* shift, float* out, int B, int C, int HW,
                                   int act, void* stream) {
  if (!raw || !scale || !shift || !out || B < 1 || C < 1 || HW < 1 || (act != 0 && act != 2)) return TGSR_EINVAL;
  if ((HW & 3) || ((reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(out)) & 15)) return TGSR_EUNSUPPORTED;
  hipLaunchKernelGGL(affine_act_kernel, dim3(C, affine_grid_y(B, C, HW)), dim3(256), 0, as_stream(stream), raw, scale, shift, out,
                     B, C, HW, act);
  return note_launch(hipGetLastError(), "affine_act_kernel");
}

extern "C" int tgsr_affine_act_bwd(const float* dy, const float* out, const float* scale, float* draw, int B, int C, int HW,
                                   int act, void* stream) {
  if (!dy || !scale || !draw || B < 1 || C < 1 || HW < 1 || (act != 0 && act != 2) || (act == 2 && !out)) return TGSR_EINVAL;
  if ((HW & 3) || ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(draw)) & 15))
    return TGSR_EUNSUPPORTED;
  hipLaunchKernelGGL(affine_act_bwd_kernel, dim3(C, affine_grid_y(B, C, HW)), dim3(256), 0, as_stream(stream), dy, out, scale,
                     draw, B, C, HW, act);
  return note_launch(hipGetLastError(), "affine_act_bwd_kernel");
}

extern "C" int tgsr_weighted_bce_fwd(const float* a, int na, const float* b, int nb, const float* target, const float* weight,
                                     float* out, void* stream) {
  if (!a || na < 1 || nb < 0 || (nb > 0 && !b) || !target || !weight || !out) return TGSR_EINVAL;
  hipLaunchKernelGGL(weighted_bce_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, na, b, nb, target, weight, out);
  return note_launch(hipGetLastError(), "weighted_bce_kernel");
}

extern "C" int tgsr_weighted_bce_bwd(const float* dy, const float* a, int na, const float* b, int nb, const float* target,
                                     const float* weight, float* da, float* db, void* stream) {
  if (!dy || !a || na < 1 || nb < 0 || (nb > 0 && !b) || !target || !weight || (!da && !db)) return TGSR_EINVAL;
  hipLaunchKernelGGL(weighted_bce_bwd_kernel, dim3((na + nb + 255) / 256), dim3(256), 0, as_stream(stream), dy, a, na, b, nb, target,
                     weight, da, db);
  return note_launch(hipGetLastError(), "weighted_bce_bwd_kernel");
}

extern "C" int tgsr_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* state, int64_t n, double lr,
                              double beta1, double beta2, double eps, double weight_decay, int advance, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !state || n < 1) return TGSR_EINVAL;
  if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(lr >= 0.0)) return TGSR_EINVAL;
  if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
       reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
    return TGSR_EUNSUPPORTED;
  if (advance) {
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), state, beta1, beta2);
    const int rc = note_launch(hipGetLastError(), "adam_advance_kernel");
    if (rc != TGSR_OK) return rc;
  }
  const int64_t n4 = n >> 2;
  const int64_t want = (n4 + 255) / 256;
  const int bx = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(adam_flat_kernel, dim3(bx), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, state, n4, n,
                     (float)lr, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay);
  return note_launch(hipGetLastError(), "adam_flat_kernel");
}
