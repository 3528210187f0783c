// Training-mode BatchNorm2d (+ GLU / residual) around the conv kernel, forward and backward, for gfx950.
//
// The reference trains with nn.BatchNorm2d batch statistics (util.py:77,116,119).  A grid-wide reduction
// sits between the convolution and the GLU, so training runs each block as
//     conv3x3 (tgsr_conv3x3_fwd, identity affine) -> raw [B][C][HW]
//     bn_stats       : per-channel sum / sum of squares, one float4 stream over raw     (HBM-bound)
//     bn_fin_act_fwd : every workgroup combines its channel's partials (double): mean, biased var, invstd, scale/shift,
//                      running-stat update (momentum, unbiased); then y = GLU(raw*scale+shift) | raw*scale+shift (+ residual)
// and the backward as
//     bn_act_bwd_reduce   : dz = d(BN output) from dy (GLU'/identity), per-channel sum dz, sum dz*xhat
//     bn_fin_act_bwd_apply: combines those partials, draw = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); dgamma, dbeta
// followed by the conv data/weight gradients (tgsr_conv3x3_fwd on flipped weights, tgsr_conv3x3_wgrad).
#include "tgsr_common.h"

namespace tgsr {

constexpr int kBnThreads = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// A workgroup's walk over its slice [lo, hi) of a channel's B x HW values, one float4 per thread and step: (b, p) advance by
// addition - the 64-bit division `e / HW` per step that this replaces was a fifth of the GLU passes' VALU work (they are
// VALU-bound: exp + rcp per value) - and the passes issue two steps' loads before they compute (`TGSR_BN_STEP2`).
struct BnWalk {
  int64_t e, hi;
  int b, p, HW;
  __device__ __forceinline__ BnWalk(int64_t lo, int64_t hi_, int HW_) : e(lo + 4 * threadIdx.x), hi(hi_), HW(HW_) {
    b = (int)(e / HW_);
    p = (int)(e - (int64_t)b * HW_);
  }
  __device__ __forceinline__ bool ok() const { return e < hi; }
  __device__ __forceinline__ void next() {
    e += 4 * kBnThreads;
    p += 4 * kBnThreads;
    while (p >= HW) { p -= HW; ++b; }
  }
};

// a thread's share of (sum, sumsq) over the slice [lo, hi) of channel c's B x HW values
__device__ __forceinline__ void bn_stats_slice(const float* __restrict__ raw, int64_t bstride, int HW, int c, int64_t lo, int64_t hi,
                                               float& s, float& q) {
  const bool vec = (HW & 3) == 0;
  if (vec) {
    BnWalk w(lo, hi, HW);
    while (w.ok()) {
      const float4 v = *reinterpret_cast<const float4*>(raw + w.b * bstride + (int64_t)c * HW + w.p);
      w.next();
      const bool two = w.ok();
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
      if (two) u = *reinterpret_cast<const float4*>(raw + w.b * bstride + (int64_t)c * HW + w.p);
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      if (two) {
        s += (u.x + u.y) + (u.z + u.w);
        q += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
        w.next();
      }
    }
  } else {
    for (int64_t e = lo + threadIdx.x; e < hi; e += kBnThreads) {
      const int b = (int)(e / HW);
      const int p = (int)(e - (int64_t)b * HW);
      const float v = raw[b * bstride + (int64_t)c * HW + p];
      s += v;
      q += v * v;
    }
  }
}

// grid (C, nsplit): partial[c][s] = (sum, sumsq) of raw[b][c][:] over the b's / pixel ranges of split s
__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const float* __restrict__ raw, int64_t bstride, int B,
                                                              int HW, float* __restrict__ partial, int nsplit) {
  __shared__ float red[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int64_t total = (int64_t)B * HW;
  const int64_t per = ((total + nsplit - 1) / nsplit + 3) & ~3ll;
  const int64_t lo = sp * per, hi = lo + per < total ? lo + per : total;
  float s = 0.f, q = 0.f;
  bn_stats_slice(raw, bstride, HW, c, lo, hi, s, q);
  s = block_sum(s, red);
  q = block_sum(q, red);
  if (threadIdx.x == 0) {
    partial[((int64_t)c * nsplit + sp) * 2] = s;
    partial[((int64_t)c * nsplit + sp) * 2 + 1] = q;
  }
}

// v_exp_f32 + v_rcp_f32 (1 ulp each), as the inference epilogues: the IEEE division expands to ~10 instructions, and the GLU
// passes of BatchNorm's forward / backward are VALU-bound (the step lost 14 % without packed math, csrc/Makefile)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// Backward, pass 1: a thread's share, over the slice [lo, hi) of output channel c (GLU -> BN channels c (value) and c+Co (gate)), of
// (sum dz_v, sum dz_v*xhat_v, sum dz_g, sum dz_g*xhat_g)   (non-GLU: the last two stay 0)
template <bool GLU>
__device__ __forceinline__ void bn_bwd_reduce_slice(const float* __restrict__ dout, const float* __restrict__ raw, int C, int HW, int c,
                                                    int64_t lo, int64_t hi, int64_t per, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, int leaky, float& a0, float& a1, float& a2,
                                                    float& a3) {
  const int Co = GLU ? C / 2 : C;
  const float sv = scale[c], tv = shift[c], mv = mean[c], iv = invstd[c];
  float sg = 0.f, tg = 0.f, mg = 0.f, ig = 0.f;
  if (GLU) { sg = scale[c + Co]; tg = shift[c + Co]; mg = mean[c + Co]; ig = invstd[c + Co]; }
  auto acc1 = [&](float dy, float rv, float rg) {
    if (GLU) {
      const float av = rv * sv + tv, s = sigm(rg * sg + tg);
      const float dzv = dy * s, dzg = dy * av * s * (1.f - s);
      a0 += dzv; a1 += dzv * ((rv - mv) * iv);
      a2 += dzg; a3 += dzg * ((rg - mg) * ig);
    } else {
      if (leaky && rv * sv + tv <= 0.f) dy *= 0.2f;
      a0 += dy; a1 += dy * ((rv - mv) * iv);
    }
  };
  if ((HW & 3) == 0 && (per & 3) == 0) {                 // float4 streams (every slice starts on a multiple of 4)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    BnWalk w(lo, hi, HW);
    while (w.ok()) {
      const float4 dy = *reinterpret_cast<const float4*>(dout + ((int64_t)w.b * Co + c) * HW + w.p);
      const float4 rv = *reinterpret_cast<const float4*>(raw + ((int64_t)w.b * C + c) * HW + w.p);
      float4 rg = z4;
      if (GLU) rg = *reinterpret_cast<const float4*>(raw + ((int64_t)w.b * C + c + Co) * HW + w.p);
      w.next();
      const bool two = w.ok();
      float4 dy2 = z4, rv2 = z4, rg2 = z4;
      if (two) {
        dy2 = *reinterpret_cast<const float4*>(dout + ((int64_t)w.b * Co + c) * HW + w.p);
        rv2 = *reinterpret_cast<const float4*>(raw + ((int64_t)w.b * C + c) * HW + w.p);
        if (GLU) rg2 = *reinterpret_cast<const float4*>(raw + ((int64_t)w.b * C + c + Co) * HW + w.p);
        w.next();
      }
      acc1(dy.x, rv.x, rg.x); acc1(dy.y, rv.y, rg.y); acc1(dy.z, rv.z, rg.z); acc1(dy.w, rv.w, rg.w);
      if (two) { acc1(dy2.x, rv2.x, rg2.x); acc1(dy2.y, rv2.y, rg2.y); acc1(dy2.z, rv2.z, rg2.z); acc1(dy2.w, rv2.w, rg2.w); }
    }
  } else {
    for (int64_t e = lo + threadIdx.x; e < hi; e += kBnThreads) {
      const int b = (int)(e / HW);
      const int p = (int)(e - (int64_t)b * HW);
      acc1(dout[((int64_t)b * Co + c) * HW + p], raw[((int64_t)b * C + c) * HW + p],
           GLU ? raw[((int64_t)b * C + c + Co) * HW + p] : 0.f);
    }
  }
}

// grid (Co, nsplit): partial[c][s] = the four sums of slice s
template <bool GLU>
__global__ __launch_bounds__(kBnThreads) void bn_act_bwd_reduce_kernel(
    const float* __restrict__ dout, const float* __restrict__ raw, int B, int C, int HW,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ partial, int nsplit, int leaky) {
  __shared__ float red[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int64_t total = (int64_t)B * HW;
  const int64_t per = ((total + nsplit - 1) / nsplit + 3) & ~3ll;
  const int64_t lo = sp * per, hi = lo + per < total ? lo + per : total;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  bn_bwd_reduce_slice<GLU>(dout, raw, C, HW, c, lo, hi, per, scale, shift, mean, invstd, leaky, a0, a1, a2, a3);
  a0 = block_sum(a0, red); a1 = block_sum(a1, red);
  if (GLU) { a2 = block_sum(a2, red); a3 = block_sum(a3, red); }
  if (threadIdx.x == 0) {
    float* o = partial + ((int64_t)c * nsplit + sp) * 4;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The finalize steps folded into their consumers (72 four-microsecond launches per generator step sat on the dependent
// chain between the statistics pass and the pass that uses them): every workgroup of the normalise / backward-apply pass
// owns ONE output channel and a slice of its (b, pixel) range - grid (Co, nsplit) like the statistics pass - and first
// combines that channel's partial sums itself (<= 64 values, lanes of wave 0, pairwise tree in double: every workgroup
// of a channel computes bit-identical statistics); the workgroup of slice 0 also writes what the old finalize kernels
// wrote (mean / invstd / scale / shift, running statistics, dgamma / dbeta).
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ONE thread: channel ch's (scale, shift) into bc[0..1] from its (sum, sum of squares); `publish` also writes the statistics.
__device__ __forceinline__ void bn_affine_thread0(double s, double q, int ch, double count, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                                  float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                                  bool publish, float* bc) {
  const double m = s / count;
  double var = q / count - m * m;
  var = var < 0.0 ? 0.0 : var;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[ch] * is, sh = beta[ch] - (float)m * sc;
  bc[0] = sc;
  bc[1] = sh;
  if (publish) {
    mean[ch] = (float)m;
    invstd[ch] = is;
    scale[ch] = sc;
    shift[ch] = sh;
    if (running_mean) {
      const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
    }
  }
}

// channel ch: (scale, shift) from the statistics partials; slice-0 workgroups publish the statistics.
// Up to 64 partials (the statistics kernel's splits): lanes of wave 0, as before.  More (a convolution's epilogue wrote one
// pair per wave tile, tgsr_wino_conv3x3_stats_fwd: 4096 per channel at 128^2): every thread of the workgroup sums its
// strided share with the loads in flight together, the 256 thread sums meet in LDS - a fixed order either way, so every
// workgroup of the channel gets the same bits.
__device__ __forceinline__ void bn_channel_affine(const float* __restrict__ partial, int nsplit, int ch, double count,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                  float momentum, float* running_mean, float* running_var, float* mean,
                                                  float* invstd, float* scale, float* shift, bool publish, float* bc) {
  __shared__ double wide_s[kBnThreads], wide_q[kBnThreads];
  const bool wide = nsplit > 64;
  if (wide) {
    double s = 0.0, q = 0.0;
    const float2* pp = reinterpret_cast<const float2*>(partial + (int64_t)ch * nsplit * 2);
    int k = threadIdx.x;
    for (; k + 3 * kBnThreads < nsplit; k += 4 * kBnThreads) {
      const float2 v0 = pp[k], v1 = pp[k + kBnThreads], v2 = pp[k + 2 * kBnThreads], v3 = pp[k + 3 * kBnThreads];
      s += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      q += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; k < nsplit; k += kBnThreads) {
      const float2 v = pp[k];
      s += (double)v.x;
      q += (double)v.y;
    }
    __syncthreads();                       // (a previous call's readers of wide_* are done)
    wide_s[threadIdx.x] = s;
    wide_q[threadIdx.x] = q;
    __syncthreads();
  }
  if (threadIdx.x < 64) {
    double s = 0.0, q = 0.0;
    if (wide) {
#pragma unroll
      for (int j = 0; j < kBnThreads / 64; ++j) {
        s += wide_s[threadIdx.x + 64 * j];
        q += wide_q[threadIdx.x + 64 * j];
      }
    } else if ((int)threadIdx.x < nsplit) {
      s = partial[((int64_t)ch * nsplit + threadIdx.x) * 2];
      q = partial[((int64_t)ch * nsplit + threadIdx.x) * 2 + 1];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (threadIdx.x == 0)
      bn_affine_thread0(s, q, ch, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, publish, bc);
  }
}

// FUSED (small layers - the discriminators' 16^2 .. 4^2 maps - where ONE workgroup per channel covers the whole batch, grid (C, 1),
// no GLU): the statistics pass runs inside this kernel - the same per-thread sums in the same order as bn_stats_kernel with one
// split, the same finalize: bit-identical to the two launches it replaces (the second read of the channel's <= 32 KB comes from L2).
template <bool GLU, bool FUSED = false>
__global__ __launch_bounds__(kBnThreads) void bn_fin_act_fwd_kernel(
    const float* __restrict__ raw, int B, int C, int HW, const float* __restrict__ partial, int nsplit_stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, float* running_mean,
    float* running_var, float* mean, float* invstd, float* scale, float* shift, long long* num_batches_tracked,
    const float* __restrict__ res, int64_t rbs, float* __restrict__ out, int64_t obs, int leaky, int nsplit) {
  static_assert(!(GLU && FUSED), "the fused form serves the plain / LeakyReLU layers");
  __shared__ float bc[4];
  const int Co = GLU ? C / 2 : C;
  const int c = blockIdx.x, sp = blockIdx.y;
  const double count = (double)B * HW;
  if (FUSED) {
    __shared__ float red[4];
    float s = 0.f, q = 0.f;
    bn_stats_slice(raw, (int64_t)C * HW, HW, c, 0, (int64_t)B * HW, s, q);
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0)
      bn_affine_thread0((double)s, (double)q, c, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale,
                        shift, true, bc);
  } else {
    bn_channel_affine(partial, nsplit_stats, c, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                      scale, shift, sp == 0, bc);
  }
  if (GLU) {
    __syncthreads();      // (bc[0..1] written by thread 0; the gate channel goes to bc[2..3])
    bn_channel_affine(partial, nsplit_stats, c + Co, count, gamma, beta, eps, momentum, running_mean, running_var, mean,
                      invstd, scale, shift, sp == 0, bc + 2);
  }
  if (c == 0 && sp == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
  __syncthreads();
  const float sv = bc[0], tv = bc[1], sg = GLU ? bc[2] : 0.f, tg = GLU ? bc[3] : 0.f;
  const int64_t total = (int64_t)B * HW;
  const int64_t per = ((total + nsplit - 1) / nsplit + 3) & ~3ll;
  const int64_t lo = sp * per, hi = lo + per < total ? lo + per : total;
  auto act4 = [&](const float4 v, const float4 g, const float4 r) {
    float4 y = make_float4(v.x * sv + tv, v.y * sv + tv, v.z * sv + tv, v.w * sv + tv);
    if (GLU) {
      y.x *= sigm(g.x * sg + tg);
      y.y *= sigm(g.y * sg + tg);
      y.z *= sigm(g.z * sg + tg);
      y.w *= sigm(g.w * sg + tg);
    } else if (res) {
      y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
    } else if (leaky) {                                    // LeakyReLU(0.2): downBlock / Block3x3_leakRelu (util.py:92-98)
      y.x = y.x > 0.f ? y.x : 0.2f * y.x; y.y = y.y > 0.f ? y.y : 0.2f * y.y;
      y.z = y.z > 0.f ? y.z : 0.2f * y.z; y.w = y.w > 0.f ? y.w : 0.2f * y.w;
    }
    return y;
  };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  BnWalk w(lo, hi, HW);
  while (w.ok()) {
    const int b0 = w.b, p0 = w.p;
    const float4 v = *reinterpret_cast<const float4*>(raw + ((int64_t)b0 * C + c) * HW + p0);
    float4 g = z4, r = z4;
    if (GLU) g = *reinterpret_cast<const float4*>(raw + ((int64_t)b0 * C + c + Co) * HW + p0);
    else if (res) r = *reinterpret_cast<const float4*>(res + b0 * rbs + (int64_t)c * HW + p0);
    w.next();
    const bool two = w.ok();
    const int b1 = w.b, p1 = w.p;
    float4 v2 = z4, g2 = z4, r2 = z4;
    if (two) {
      v2 = *reinterpret_cast<const float4*>(raw + ((int64_t)b1 * C + c) * HW + p1);
      if (GLU) g2 = *reinterpret_cast<const float4*>(raw + ((int64_t)b1 * C + c + Co) * HW + p1);
      else if (res) r2 = *reinterpret_cast<const float4*>(res + b1 * rbs + (int64_t)c * HW + p1);
      w.next();
    }
    *reinterpret_cast<float4*>(out + b0 * obs + (int64_t)c * HW + p0) = act4(v, g, r);
    if (two) *reinterpret_cast<float4*>(out + b1 * obs + (int64_t)c * HW + p1) = act4(v2, g2, r2);
  }
}

// Backward, pass 2 with the finalize folded in: grid (Co, nsplit) - the partials of bn_act_bwd_reduce_kernel on the same grid
// FUSED (as the forward's: one workgroup per channel, no GLU): pass 1 runs inside this kernel, the same sums in the same order.
template <bool GLU, bool FUSED = false>
__global__ __launch_bounds__(kBnThreads) void bn_fin_act_bwd_apply_kernel(
    const float* __restrict__ dout, const float* __restrict__ raw, int B, int C, int HW,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ partial, int nsplit, float invN,
    float* __restrict__ draw, float* __restrict__ dgamma, float* __restrict__ dbeta, int leaky) {
  static_assert(!(GLU && FUSED), "the fused form serves the plain / LeakyReLU layers");
  __shared__ float bs[4];
  const int Co = GLU ? C / 2 : C;
  const int c = blockIdx.x, sp = blockIdx.y;
  if (FUSED) {
    __shared__ float red[4];
    const int64_t tot = (int64_t)B * HW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    bn_bwd_reduce_slice<false>(dout, raw, C, HW, c, 0, tot, (tot + 3) & ~3ll, scale, shift, mean, invstd, leaky, a0, a1, a2, a3);
    a0 = block_sum(a0, red);
    a1 = block_sum(a1, red);
    if (threadIdx.x == 0) {
      bs[0] = a0; bs[1] = a1; bs[2] = 0.f; bs[3] = 0.f;
      dbeta[c] = a0;
      dgamma[c] = a1;
    }
  } else if (threadIdx.x < 64) {
    double a[4] = {0, 0, 0, 0};
    if ((int)threadIdx.x < nsplit)
      for (int j = 0; j < 4; ++j) a[j] = partial[((int64_t)c * nsplit + threadIdx.x) * 4 + j];
    for (int j = 0; j < 4; ++j) a[j] = wave_sum_d(a[j]);
    if (threadIdx.x == 0) {
      for (int j = 0; j < 4; ++j) bs[j] = (float)a[j];
      if (sp == 0) {
        dbeta[c] = (float)a[0];
        dgamma[c] = (float)a[1];
        if (GLU) {
          dbeta[c + Co] = (float)a[2];
          dgamma[c + Co] = (float)a[3];
        }
      }
    }
  }
  __syncthreads();
  const float s0 = bs[0] * invN, s1 = bs[1] * invN, s2 = bs[2] * invN, s3 = bs[3] * invN;
  const float sv = scale[c], tv = shift[c], mv = mean[c], isv = invstd[c];   // scale = gamma * invstd
  float sg = 0.f, tg = 0.f, mg = 0.f, isg = 0.f;
  if (GLU) { sg = scale[c + Co]; tg = shift[c + Co]; mg = mean[c + Co]; isg = invstd[c + Co]; }
  const int64_t total = (int64_t)B * HW;
  const int64_t per = ((total + nsplit - 1) / nsplit + 3) & ~3ll;
  const int64_t lo = sp * per, hi = lo + per < total ? lo + per : total;
  auto one = [&](float dy, float rv, float rg, float& ov, float& og) {
    const float xv = (rv - mv) * isv;
    if (GLU) {
      const float xg = (rg - mg) * isg;
      const float av = rv * sv + tv, s = sigm(rg * sg + tg);
      const float dzv = dy * s, dzg = dy * av * s * (1.f - s);
      ov = sv * (dzv - s0 - xv * s1);
      og = sg * (dzg - s2 - xg * s3);
    } else {
      if (leaky && rv * sv + tv <= 0.f) dy *= 0.2f;
      ov = sv * (dy - s0 - xv * s1);
    }
  };
  if ((HW & 3) == 0) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    BnWalk w(lo, hi, HW);
    while (w.ok()) {
      const int64_t iv_ = ((int64_t)w.b * C + c) * HW + w.p, ig_ = iv_ + (int64_t)Co * HW;
      const float4 dy = *reinterpret_cast<const float4*>(dout + ((int64_t)w.b * Co + c) * HW + w.p);
      const float4 rv = *reinterpret_cast<const float4*>(raw + iv_);
      float4 rg = z4, ov, og = z4;
      if (GLU) rg = *reinterpret_cast<const float4*>(raw + ig_);
      w.next();
      const bool two = w.ok();
      const int64_t iv2 = ((int64_t)w.b * C + c) * HW + w.p, ig2 = iv2 + (int64_t)Co * HW;
      float4 dy2 = z4, rv2 = z4, rg2 = z4, ov2, og2 = z4;
      if (two) {
        dy2 = *reinterpret_cast<const float4*>(dout + ((int64_t)w.b * Co + c) * HW + w.p);
        rv2 = *reinterpret_cast<const float4*>(raw + iv2);
        if (GLU) rg2 = *reinterpret_cast<const float4*>(raw + ig2);
        w.next();
      }
      one(dy.x, rv.x, rg.x, ov.x, og.x); one(dy.y, rv.y, rg.y, ov.y, og.y);
      one(dy.z, rv.z, rg.z, ov.z, og.z); one(dy.w, rv.w, rg.w, ov.w, og.w);
      *reinterpret_cast<float4*>(draw + iv_) = ov;
      if (GLU) *reinterpret_cast<float4*>(draw + ig_) = og;
      if (two) {
        one(dy2.x, rv2.x, rg2.x, ov2.x, og2.x); one(dy2.y, rv2.y, rg2.y, ov2.y, og2.y);
        one(dy2.z, rv2.z, rg2.z, ov2.z, og2.z); one(dy2.w, rv2.w, rg2.w, ov2.w, og2.w);
        *reinterpret_cast<float4*>(draw + iv2) = ov2;
        if (GLU) *reinterpret_cast<float4*>(draw + ig2) = og2;
      }
    }
  } else {
    for (int64_t e = lo + threadIdx.x; e < hi; e += kBnThreads) {
      const int b = (int)(e / HW);
      const int p = (int)(e - (int64_t)b * HW);
      const int64_t iv_ = ((int64_t)b * C + c) * HW + p, ig_ = iv_ + (int64_t)Co * HW;
      float ov, og = 0.f;
      one(dout[((int64_t)b * Co + c) * HW + p], raw[iv_], GLU ? raw[ig_] : 0.f, ov, og);
      draw[iv_] = ov;
      if (GLU) draw[ig_] = og;
    }
  }
}

// out[bc][y][x] = sum of the 2x2 block of in[bc][2y..2y+1][2x..2x+1]   (backward of the nearest x2 up-sample)
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const float* __restrict__ in, int64_t BC, int H, int W,
                                                         float* __restrict__ out) {
  const int64_t total = BC * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int64_t t = i / W;
    const int y = (int)(t % H);
    const int64_t bc = t / H;
    const float* r0 = in + (bc * 2 * H + 2 * y) * (int64_t)(2 * W) + 2 * x;
    const float2 u = *reinterpret_cast<const float2*>(r0);
    const float2 v = *reinterpret_cast<const float2*>(r0 + 2 * W);
    out[i] = (u.x + u.y) + (v.x + v.y);
  }
}

static int g_bn_fuse_small = [] {
  const char* e = getenv("TGSR_BN_FUSE_SMALL");
  return (e && e[0] == '0') ? 0 : 1;
}();
static inline bool bn_fuse_small() { return g_bn_fuse_small != 0; }

static inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_bn_set_fuse_small(int on) {
  const int was = g_bn_fuse_small;
  g_bn_fuse_small = on ? 1 : 0;
  return was;
}

extern "C" int tgsr_bn_train_nsplit(int B, int C, int HW) {
  // enough workgroups to fill the chip (>= 1024) without splitting below ~4k elements per workgroup
  const int64_t total = (int64_t)B * HW;
  int n = (int)((1024 + C - 1) / C);
  const int64_t cap = total / 4096 > 0 ? total / 4096 : 1;
  if (n > cap) n = (int)cap;
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}

static int bn_train_fwd(const float* raw, int B, int C, int HW, const float* gamma, const float* beta, float eps,
                        float momentum, float* running_mean, float* running_var, int glu, const float* residual,
                        int64_t res_bstride, float* partial_ws, int given_nsplit, float* mean, float* invstd, float* scale,
                        float* shift, float* out, int64_t out_bstride, int64_t* num_batches_tracked, void* stream) {
  if (!raw || !gamma || !beta || !partial_ws || !mean || !invstd || !scale || !shift || !out) return TGSR_EINVAL;
  if (glu < 0 || glu > 2) return TGSR_EINVAL;             // `glu` is the activation selector: 0 none, 1 GLU, 2 LeakyReLU(0.2)
  const int leaky = glu == 2 ? 1 : 0;
  glu = glu == 1 ? 1 : 0;
  if (B < 1 || C < 1 || HW < 1 || (glu && (C & 1)) || ((glu || leaky) && residual)) return TGSR_EINVAL;
  if ((HW & 3) != 0) return TGSR_EUNSUPPORTED;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TGSR_EINVAL;
  hipStream_t s = as_stream(stream);
  // given_nsplit > 0: partial_ws already holds that many (sum, sumsq) pairs per channel (a convolution's epilogue wrote
  // them): the statistics pass over the raw tensor is not launched
  const int nsplit = given_nsplit > 0 ? given_nsplit : tgsr_bn_train_nsplit(B, C, HW);
  // one workgroup per channel covers the batch: statistics and normalisation in ONE launch (bit-identical; TGSR_BN_FUSE_SMALL=0: two)
  const bool fused = given_nsplit <= 0 && nsplit == 1 && !glu && bn_fuse_small();
  if (given_nsplit <= 0 && !fused)
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, nsplit), dim3(kBnThreads), 0, s, raw, (int64_t)C * HW, B, HW,
                       partial_ws, nsplit);
  const int Co = glu ? C / 2 : C;
  const int ns2 = tgsr_bn_train_nsplit(B, Co, HW);
  long long* nbt = reinterpret_cast<long long*>(num_batches_tracked);
  if (fused) {
    hipLaunchKernelGGL((bn_fin_act_fwd_kernel<false, true>), dim3(C, 1), dim3(kBnThreads), 0, s, raw, B, C, HW, partial_ws, 1, gamma,
                       beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, nbt, residual, res_bstride, out,
                       out_bstride, leaky, 1);
    return note_launch(hipGetLastError(), "bn_train_fwd");
  }
  if (glu)
    hipLaunchKernelGGL(bn_fin_act_fwd_kernel<true>, dim3(Co, ns2), dim3(kBnThreads), 0, s, raw, B, C, HW, partial_ws, nsplit,
                       gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, nbt, nullptr,
                       (int64_t)0, out, out_bstride, 0, ns2);
  else
    hipLaunchKernelGGL(bn_fin_act_fwd_kernel<false>, dim3(Co, ns2), dim3(kBnThreads), 0, s, raw, B, C, HW, partial_ws, nsplit,
                       gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, nbt, residual,
                       res_bstride, out, out_bstride, leaky, ns2);
  return note_launch(hipGetLastError(), "bn_train_fwd");
}

extern "C" int tgsr_bn_train_fwd(const float* raw, int B, int C, int HW, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var, int glu,
                                 const float* residual, int64_t res_bstride, float* partial_ws, float* mean,
                                 float* invstd, float* scale, float* shift, float* out, int64_t out_bstride,
                                 int64_t* num_batches_tracked, void* stream) {
  return bn_train_fwd(raw, B, C, HW, gamma, beta, eps, momentum, running_mean, running_var, glu, residual, res_bstride,
                      partial_ws, 0, mean, invstd, scale, shift, out, out_bstride, num_batches_tracked, stream);
}

extern "C" int tgsr_bn_train_fwd_from_stats(const float* raw, int B, int C, int HW, const float* gamma, const float* beta,
                                            float eps, float momentum, float* running_mean, float* running_var, int glu,
                                            const float* residual, int64_t res_bstride, const float* stat_partial,
                                            int nslots, float* mean, float* invstd, float* scale, float* shift, float* out,
                                            int64_t out_bstride, int64_t* num_batches_tracked, void* stream) {
  if (nslots < 1) return TGSR_EINVAL;
  return bn_train_fwd(raw, B, C, HW, gamma, beta, eps, momentum, running_mean, running_var, glu, residual, res_bstride,
                      const_cast<float*>(stat_partial), nslots, mean, invstd, scale, shift, out, out_bstride,
                      num_batches_tracked, stream);
}

extern "C" int tgsr_bn_train_bwd(const float* dout, const float* raw, int B, int C, int HW, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, int glu,
                                 float* partial_ws, float* sums_ws, float* draw, float* dgamma, float* dbeta,
                                 void* stream) {
  if (!dout || !raw || !scale || !shift || !mean || !invstd || !partial_ws || !sums_ws || !draw || !dgamma || !dbeta)
    return TGSR_EINVAL;
  if (glu < 0 || glu > 2) return TGSR_EINVAL;
  const int leaky = glu == 2 ? 1 : 0;
  glu = glu == 1 ? 1 : 0;
  if (B < 1 || C < 1 || HW < 1 || (glu && (C & 1))) return TGSR_EINVAL;
  hipStream_t s = as_stream(stream);
  const int Co = glu ? C / 2 : C;
  const int nsplit = tgsr_bn_train_nsplit(B, Co, HW);
  const float invN = (float)(1.0 / ((double)B * HW));
  (void)sums_ws;                         // (was the finalize kernel's output; the apply pass combines the partials itself)
  if (nsplit == 1 && !glu && bn_fuse_small()) {            // both passes in one launch (see bn_fin_act_bwd_apply_kernel)
    hipLaunchKernelGGL((bn_fin_act_bwd_apply_kernel<false, true>), dim3(Co, 1), dim3(kBnThreads), 0, s, dout, raw, B, C, HW, scale,
                       shift, mean, invstd, partial_ws, 1, invN, draw, dgamma, dbeta, leaky);
    return note_launch(hipGetLastError(), "bn_train_bwd");
  }
  if (glu) {
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<true>, dim3(Co, nsplit), dim3(kBnThreads), 0, s, dout, raw, B, C, HW,
                       scale, shift, mean, invstd, partial_ws, nsplit, 0);
    hipLaunchKernelGGL(bn_fin_act_bwd_apply_kernel<true>, dim3(Co, nsplit), dim3(kBnThreads), 0, s, dout, raw, B, C, HW,
                       scale, shift, mean, invstd, partial_ws, nsplit, invN, draw, dgamma, dbeta, 0);
  } else {
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<false>, dim3(Co, nsplit), dim3(kBnThreads), 0, s, dout, raw, B, C,
                       HW, scale, shift, mean, invstd, partial_ws, nsplit, leaky);
    hipLaunchKernelGGL(bn_fin_act_bwd_apply_kernel<false>, dim3(Co, nsplit), dim3(kBnThreads), 0, s, dout, raw, B, C, HW,
                       scale, shift, mean, invstd, partial_ws, nsplit, invN, draw, dgamma, dbeta, leaky);
  }
  return note_launch(hipGetLastError(), "bn_train_bwd");
}

extern "C" int tgsr_sumpool2x2(const float* x, int64_t BC, int H, int W, float* out, void* stream) {
  if (!x || !out || BC < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  hipLaunchKernelGGL(sumpool2x2_kernel, dim3(grid_for(BC * H * W)), dim3(256), 0, as_stream(stream), x, BC, H, W, out);
  return note_launch(hipGetLastError(), "sumpool2x2_kernel");
}
