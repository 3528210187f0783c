// The discriminators' convolutions (the build's AttnGAN-style D_NET64/128/256 behind losses.py:290-374): downBlock's
// Conv2d(Cin, Cout, 4, 2, 1, bias=False) (util.py:92-98) and the 3x3 stride-1 convolutions of Block3x3_leakRelu at
// 4x4 pixels with 512 ... 2048 channels - forward, data gradient and weight gradient - as ONE fp32 MFMA (32x32x2)
// implicit-GEMM kernel.  These layers are GEMM-shaped the generator's are not: K = 16 Cin up to 16384 against as few as
// 256 output pixels, so the kernel is a 128 x 128 x 16 LDS-tiled GEMM whose operands are gathered from the NCHW tensors
// by index arithmetic (no im2col buffer), with the reduction split over blockIdx.z into slabs when M x N alone cannot
// fill 256 CUs; slabs are summed in a fixed order (bitwise reproducible, no float atomics).
//
//   forward (4x4 s2 | 3x3 s1): out[b][co][p] = sum_j w[co][j] S(p, j),  j = (ci, ky, kx)       M = Cout, N = pixels, K = j
//   dgrad 4x4 s2: input pixels of parity class (p, q) = ((iy+1)&1, (ix+1)&1) only see taps ky in {p, p+2}, kx in
//                 {q, q+2}: four GEMMs M = Cin, K = 4 Cout, N = B (H/2)(W/2) over re-grouped weights (pack kernel)
//   dgrad 3x3 s1: the same conv over dy with the filter read transposed and flipped                M = Cin, K = 9 Cout
//   wgrad       : dw[co][j] = sum_pixels dy[co][p] S(p, j)                                         M = Cout, N = j, K = pixels
// BatchNorm (batch statistics) + LeakyReLU(0.2) is tgsr_bn_train_fwd / _bwd with act = 2 (tgsr_bn.hip); the first layer
// (no BatchNorm) takes the LeakyReLU in this kernel's epilogue.  SURVEY.md 8(f)1.
#include "tgsr_common.h"

namespace tgsr {

enum { kFwd4 = 0, kFwd3 = 1, kDgrad4 = 2, kDgrad3 = 3, kWgrad4 = 4, kWgrad3 = 5 };

struct IgArgs {
  const float* A;        // fwd: w [Cout][K]; dgrad4: packed [4][Cin][4 Cout]; dgrad3: w [Cout][Cin][3][3]; wgrad: dy
  const float* S;        // gathered tensor: fwd / wgrad x [B][C][Hs][Ws]; dgrad: dy
  float* out;            // output, or slab 0 when nsplit > 1
  int M, N, K;           // GEMM sizes (K = reduction)
  int C, Hs, Ws;         // channels and spatial size of S
  int PH, PW;            // the pixel grid the pixel index runs over (output pixels; dgrad4: one parity class)
  int OH, OW;            // spatial size of the NCHW output (dgrad4: the full input)
  int nsplit, chunks_per_split;
  int64_t slab_stride;
  int act;               // forward, nsplit == 1 only: 1 = LeakyReLU(0.2) epilogue
};

constexpr int kIgKC = 16, kIgP = 132;   // K-chunk; LDS pitch: (4 k + m) mod 64 is a bijection over a wave's writes

// decode a column index j = (channel, taps) of the gathered operand into (element offset, row shift, column shift)
template <int MODE>
__device__ __forceinline__ void ig_decode_j(const IgArgs& a, int j, int cls, int& off, int& dyk, int& dxk) {
  if (MODE == kFwd4 || MODE == kWgrad4) {
    const int c = j >> 4;
    dyk = ((j >> 2) & 3) - 1;
    dxk = (j & 3) - 1;
    off = c * a.Hs * a.Ws;
  } else if (MODE == kDgrad4) {
    const int c = j >> 2;
    dyk = 1 - (cls >> 1) - ((j >> 1) & 1);
    dxk = 1 - (cls & 1) - (j & 1);
    off = c * a.Hs * a.Ws;
  } else {
    const int c = j / 9, t = j - 9 * c, ky = t / 3;
    dyk = ky - 1;
    dxk = t - 3 * ky - 1;
    off = c * a.Hs * a.Ws;
  }
}

// decode a pixel index into (element offset of (b, channel 0, row 0, col 0), top row, left column) of its window
template <int MODE>
__device__ __forceinline__ void ig_decode_p(const IgArgs& a, int p, int& off, int& iy0, int& ix0) {
  const int hw = a.PH * a.PW;
  const int b = p / hw, r = p - b * hw, py = r / a.PW, px = r - py * a.PW;
  constexpr int S = (MODE == kFwd4 || MODE == kWgrad4) ? 2 : 1;
  iy0 = S * py;
  ix0 = S * px;
  off = b * a.C * a.Hs * a.Ws;
}

// WIDE (forward / data gradient with M <= 64: the 64-channel side of the first layers): a 64 (M) x 256 (N) tile, the four waves
// side by side along N - the 128 x 128 tile would multiply 64 rows of zeros (64 -> 128 data gradient at 128^2: 52 TFLOP/s).
template <int MODE, bool WIDE = false>
__global__ __launch_bounds__(256) void dconv_igemm_kernel(IgArgs a) {
  constexpr bool WG = MODE == kWgrad4 || MODE == kWgrad3;
  static_assert(!(WG && WIDE), "the wide tile serves the pixel-column modes");
  constexpr int MB = WIDE ? 64 : 128, NB = WIDE ? 256 : 128, PA = MB + 4, PB = NB + 4, NLA = MB / 16, NLB = WIDE ? 16 : 8;
  __shared__ float a_s[2][kIgKC * PA];
  __shared__ float b_s[2][kIgKC * PB];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5, wave = tid >> 6;
  const int wm = WIDE ? 0 : wave >> 1, wn = WIDE ? wave : wave & 1;
  const int ncls = MODE == kDgrad4 ? 4 : 1;
  const int cls = MODE == kDgrad4 ? (int)blockIdx.z % ncls : 0;
  const int z = (int)blockIdx.z / ncls;
  const int m0 = blockIdx.y * MB, n0 = blockIdx.x * NB;
  // forward-like B tile [16 k][NB n]: this thread's column and the k rows it loads (i = 0 .. NLB - 1)
  const int bn = WIDE ? tid : tid & 127;
  auto bk = [&](int i) { return WIDE ? i : (tid >> 7) + 2 * i; };
  const int kbeg = z * a.chunks_per_split * kIgKC;
  const int kend = min(a.K, kbeg + a.chunks_per_split * kIgKC);
  const int nchunks = (kend - kbeg + kIgKC - 1) / kIgKC;
  const float* Ab = a.A + (MODE == kDgrad4 ? (int64_t)cls * a.M * a.K : 0);

  // ---- loader state.  A tile [128 m][16 k]: k = tid & 15, m = (tid >> 4) + 16 i.
  const int ak = tid & 15, am = tid >> 4;
  // B tile [16 k][128 n]: forward-like (n = pixel): n = tid & 127, k = (tid >> 7) + 2 i;  weight-gradient-like
  // (k = pixel, n = j): k = tid & 15, n = (tid >> 4) + 16 i - consecutive lanes walk consecutive pixels either way
  int poff = 0, piy = 0, pix_ = 0;            // forward-like: this thread's pixel
  bool pok = false;
  int joff[8], jsh[8];                        // weight-gradient-like: this thread's 8 columns j (shifts packed)
  if (!WG) {
    const int n = n0 + bn;
    pok = n < a.N;
    if (pok) ig_decode_p<MODE>(a, n, poff, piy, pix_);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = n0 + (tid >> 4) + 16 * i;
      int off = 0, dyk = 0, dxk = 0;
      if (j < a.N) ig_decode_j<MODE>(a, j, 0, off, dyk, dxk);
      joff[i] = j < a.N ? off : -1;
      jsh[i] = (dyk + 4) | ((dxk + 4) << 4);
    }
  }
  // forward / data gradient of the 4x4 convolution: a K-chunk of 16 is one whole input channel (16 taps) / four whole output
  // channels (4 taps each) and kbeg is a multiple of 16, so WHICH taps this thread gathers - their offsets inside a channel
  // plane and whether they fall inside the image for this thread's pixel - does not change from chunk to chunk: decoded once
  // (per chunk and element that leaves one add and the load where there were ~20 integer instructions)
  constexpr bool HOIST = MODE == kFwd4 || MODE == kDgrad4;
  int bofs[NLB];
  unsigned bval = 0;
  if (!WG && HOIST) {
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      int off = 0, dyk = 0, dxk = 0;
      ig_decode_j<MODE>(a, bk(i), cls, off, dyk, dxk);
      const int iy = piy + dyk, ix = pix_ + dxk;
      const bool ok = pok && (unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws;
      bofs[i] = ok ? off + iy * a.Ws + ix : 0;
      bval |= (ok ? 1u : 0u) << i;
    }
  }
  float ra[NLA], rb[NLB];
  auto load = [&](int c) {
    const int k0 = kbeg + c * kIgKC;
    // A
    {
      const int k = k0 + ak;
      const bool kok = k < kend;
      int64_t base = 0, stride = 0;
      if (MODE == kFwd4 || MODE == kFwd3 || MODE == kDgrad4) {
        base = k;
        stride = a.K;
      } else if (MODE == kDgrad3) {
        const int co = k / 9, t = k - 9 * co;
        base = (int64_t)co * a.M * 9 + 8 - t;
        stride = 9;
      } else {
        const int hw = a.PH * a.PW, b = k / hw, r = k - b * hw;
        base = (int64_t)b * a.M * hw + r;
        stride = hw;
      }
#pragma unroll
      for (int i = 0; i < NLA; ++i) {
        const int m = m0 + am + 16 * i;
        ra[i] = (kok && m < a.M) ? Ab[base + (int64_t)m * stride] : 0.f;
      }
    }
    // B
    if (!WG && HOIST) {
      const int cbase = poff + (MODE == kFwd4 ? (k0 >> 4) : (k0 >> 2)) * a.Hs * a.Ws;
#pragma unroll
      for (int i = 0; i < NLB; ++i) {
        const int j = k0 + bk(i);
        rb[i] = (((bval >> i) & 1u) && j < kend) ? a.S[cbase + bofs[i]] : 0.f;
      }
    } else if (!WG) {
#pragma unroll
      for (int i = 0; i < NLB; ++i) {
        const int j = k0 + bk(i);
        float v = 0.f;
        if (pok && j < kend) {
          int off, dyk, dxk;
          ig_decode_j<MODE>(a, j, cls, off, dyk, dxk);
          const int iy = piy + dyk, ix = pix_ + dxk;
          if ((unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws) v = a.S[poff + off + iy * a.Ws + ix];
        }
        rb[i] = v;
      }
    } else {
      const int p = k0 + ak;
      int off = 0, iy0 = 0, ix0 = 0;
      const bool ok = p < kend;
      if (ok) ig_decode_p<MODE>(a, p, off, iy0, ix0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v = 0.f;
        if (ok && joff[i] >= 0) {
          const int iy = iy0 + (jsh[i] & 15) - 4, ix = ix0 + (jsh[i] >> 4) - 4;
          if ((unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws) v = a.S[off + joff[i] + iy * a.Ws + ix];
        }
        rb[i] = v;
      }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) a_s[buf][ak * PA + am + 16 * i] = ra[i];
    if (!WG) {
#pragma unroll
      for (int i = 0; i < NLB; ++i) b_s[buf][bk(i) * PB + bn] = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < NLB; ++i) b_s[buf][ak * PB + am + 16 * i] = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  if (nchunks > 0) load(0);
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    store(buf);
    __syncthreads();                   // also: everybody finished computing from the other buffer one iteration ago
    if (c + 1 < nchunks) load(c + 1);  // global loads in flight under the MFMAs below
    const float* as = a_s[buf] + wm * 64 + l31;
    const float* bs = b_s[buf] + wn * 64 + l31;
#pragma unroll
    for (int s = 0; s < kIgKC / 2; ++s) {
      const int kra = (2 * s + hh) * PA, krb = (2 * s + hh) * PB;
      const float a0 = as[kra], a1 = as[kra + 32], b0 = bs[krb], b1 = bs[krb + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // ---- epilogue: D[row = acc_row(i, hh)][col = lane & 31]; consecutive lanes = consecutive n (pixels | j): coalesced
  float* ob = a.out + (int64_t)z * a.slab_stride;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + wn * 64 + nb * 32 + l31;
    if (n >= a.N) continue;
    int64_t obase;
    int64_t mstride;
    if (WG) {
      obase = n;
      mstride = a.N;
    } else {
      const int hw = a.PH * a.PW, b = n / hw, r = n - b * hw;
      if (MODE == kDgrad4) {
        const int py = r / a.PW, px = r - py * a.PW;
        obase = (int64_t)b * a.M * a.OH * a.OW + (2 * py + 1 - (cls >> 1)) * a.OW + 2 * px + 1 - (cls & 1);
        mstride = (int64_t)a.OH * a.OW;
      } else {
        obase = (int64_t)b * a.M * hw + r;
        mstride = hw;
      }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm * 64 + mb * 32 + acc_row(i, hh);
        if (m >= a.M) continue;
        float v = acc[mb][nb][i];
        if (a.act) v = v > 0.f ? v : 0.2f * v;
        ob[obase + m * mstride] = v;
      }
  }
}

// Data gradient of the FIRST downBlock (Cin = 3: the image).  As a GEMM it is M = Cin = 3 rows of a 128-row tile - 2.5 TFLOP/s,
// 1.3 ms at 256^2 and batch 32 (tools/exp_dconv.py) for 1.6 GFLOP and 150 MB of traffic.  Here a thread owns one low-resolution
// position (y', x') = the 2 x 2 input pixels (2y' + r, 2x' + s) of every input channel and walks the output channels: input
// row 2y' + r sees kernel rows ky of the parity of r + 1 only - r = 0: (ky 1, oy y'), (ky 3, oy y' - 1); r = 1: (ky 0, oy y' + 1),
// (ky 2, oy y') - so the 3 x 3 patch dy[co][y' - 1 .. y' + 1][x' - 1 .. x' + 1] feeds 4 pixels x 4 taps x CIN channels = 48 FMAs
// per output channel; the filter sits in LDS (broadcast reads).  Fixed summation order (co ascending, taps in a fixed order).
template <int CIN>
__global__ __launch_bounds__(256) void dconv_dgrad4_image_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int Cout, int H2, int W2) {
  extern __shared__ __attribute__((aligned(16))) float wl[];      // [co][CIN][4 ky][4 kx]
  for (int i = threadIdx.x; i < Cout * CIN * 16; i += 256) wl[i] = w[i];
  __syncthreads();
  const int xq = blockIdx.x * 64 + (threadIdx.x & 63), yq = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (xq >= W2 || yq >= H2) return;
  const int64_t plane = (int64_t)H2 * W2;
  const float* dyb = dy + (int64_t)b * Cout * plane;
  int ro[3], cofs[3];
  bool rok[3], cok[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int yy = yq + d - 1, xx = xq + d - 1;
    rok[d] = (unsigned)yy < (unsigned)H2;
    cok[d] = (unsigned)xx < (unsigned)W2;
    ro[d] = rok[d] ? yy * W2 : 0;
    cofs[d] = cok[d] ? xx : 0;
  }
  float acc[CIN][2][2];
#pragma unroll
  for (int c = 0; c < CIN; ++c) acc[c][0][0] = acc[c][0][1] = acc[c][1][0] = acc[c][1][1] = 0.f;
  for (int co = 0; co < Cout; ++co) {
    const float* dp = dyb + (int64_t)co * plane;
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) g[i][j] = (rok[i] && cok[j]) ? dp[ro[i] + cofs[j]] : 0.f;
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float4* wq = reinterpret_cast<const float4*>(wl + (co * CIN + c) * 16);
      const float4 k0 = wq[0], k1 = wq[1], k2 = wq[2], k3 = wq[3];       // kernel rows ky = 0 .. 3: (.x .y .z .w) = kx 0 .. 3
      // r = 0: (ky 1, row y' = g[1]), (ky 3, row y' - 1 = g[0]);  r = 1: (ky 0, row y' + 1 = g[2]), (ky 2, row y' = g[1])
      // s = 0: (kx 1, col x' = [1]),  (kx 3, col x' - 1 = [0]);   s = 1: (kx 0, col x' + 1 = [2]),  (kx 2, col x' = [1])
      acc[c][0][0] = fmaf(g[1][1], k1.y, fmaf(g[1][0], k1.w, fmaf(g[0][1], k3.y, fmaf(g[0][0], k3.w, acc[c][0][0]))));
      acc[c][0][1] = fmaf(g[1][2], k1.x, fmaf(g[1][1], k1.z, fmaf(g[0][2], k3.x, fmaf(g[0][1], k3.z, acc[c][0][1]))));
      acc[c][1][0] = fmaf(g[2][1], k0.y, fmaf(g[2][0], k0.w, fmaf(g[1][1], k2.y, fmaf(g[1][0], k2.w, acc[c][1][0]))));
      acc[c][1][1] = fmaf(g[2][2], k0.x, fmaf(g[2][1], k0.z, fmaf(g[1][2], k2.x, fmaf(g[1][1], k2.z, acc[c][1][1]))));
    }
  }
  const int W = 2 * W2;
  float* ob = dx + (int64_t)b * CIN * 4 * plane + (int64_t)(2 * yq) * W + 2 * xq;
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    float* o = ob + (int64_t)c * 4 * plane;
    *reinterpret_cast<float2*>(o) = make_float2(acc[c][0][0], acc[c][0][1]);
    *reinterpret_cast<float2*>(o + W) = make_float2(acc[c][1][0], acc[c][1][1]);
  }
}

// packed[cls = 2p + q][ci][co * 4 + 2a + c] = w[co][ci][p + 2a][q + 2c].  One thread per (ci, co) filter: its 16 taps are one
// 64-byte read, and the four taps of a parity class are one 16-byte store that is contiguous with the neighbouring thread's
// (co runs fastest over the lanes) - the per-element form of rounds 3-4 gathered 4 bytes at a time and took 0.49 ms per G/D step.
__global__ __launch_bounds__(256) void conv4x4s2_pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout,
                                                                   int Cin, int64_t total) {
  const int64_t nf = (int64_t)Cin * Cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout), ci = (int)(i / Cout);
    const float4* src = reinterpret_cast<const float4*>(w + ((int64_t)co * Cin + ci) * 16);
    const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];            // kernel rows ky = 0 .. 3
    float4* dst = reinterpret_cast<float4*>(out + ((int64_t)ci * Cout + co) * 4);
    const int64_t cls4 = nf;                                                      // float4s per parity class
    dst[0 * cls4] = make_float4(r0.x, r0.z, r2.x, r2.z);                          // p = 0, q = 0: (ky, kx) = (0|2, 0|2)
    dst[1 * cls4] = make_float4(r0.y, r0.w, r2.y, r2.w);                          // p = 0, q = 1
    dst[2 * cls4] = make_float4(r1.x, r1.z, r3.x, r3.z);                          // p = 1, q = 0
    dst[3 * cls4] = make_float4(r1.y, r1.w, r3.y, r3.w);                          // p = 1, q = 1
  }
}

// out[i] = sum_z slab[z][i] in a fixed order (+ LeakyReLU)
__global__ void slab_sum_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n, int nsplit, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += ws[(int64_t)z * n + i];
    out[i] = act ? (s > 0.f ? s : 0.2f * s) : s;
  }
}

// y = x > 0 ? x : 0.2 x (forward)  |  dx = dy * (y > 0 ? 1 : 0.2) (backward, from the OUTPUT y: same sign as x)
__global__ void leaky_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = g[i];
    out[i] = y ? (y[i] > 0.f ? v : 0.2f * v) : (v > 0.f ? v : 0.2f * v);
  }
}

// reduction splits: enough workgroups for two per CU, at least 8 chunks (128 reduction elements) per split
static bool ig_wide(int kind, int op, int64_t M) { return kind == 4 && op != 2 && M <= 64; }   // fwd / dgrad of the 4x4 conv

static int ig_nsplit(int64_t M, int64_t N, int64_t K, int ncls, bool wide = false) {
  const int64_t tiles = wide ? ((M + 63) / 64) * ((N + 255) / 256) * ncls : ((M + 127) / 128) * ((N + 127) / 128) * ncls;
  int64_t s = (512 + tiles - 1) / tiles;
  const int64_t cap = (K + 8 * kIgKC - 1) / (8 * kIgKC);
  if (s > cap) s = cap;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

// Shapes of one convolution in GEMM terms.  kind: 4 = 4x4 stride 2 pad 1, 3 = 3x3 stride 1 pad 1; op: 0 forward,
// 1 data gradient, 2 weight gradient.  out_elems = elements of the tensor the GEMM produces.
struct IgShape { int64_t M, N, K, out_elems; int ncls, nsplit; };
static IgShape ig_shape(int kind, int op, int B, int Cin, int H, int W, int Cout) {
  const int T = kind == 4 ? 16 : 9, Ho = kind == 4 ? H / 2 : H, Wo = kind == 4 ? W / 2 : W;
  IgShape s;
  s.ncls = 1;
  if (op == 0) { s.M = Cout; s.N = (int64_t)B * Ho * Wo; s.K = (int64_t)Cin * T; s.out_elems = (int64_t)B * Cout * Ho * Wo; }
  else if (op == 1 && kind == 4) { s.M = Cin; s.N = (int64_t)B * Ho * Wo; s.K = 4ll * Cout; s.ncls = 4; s.out_elems = (int64_t)B * Cin * H * W; }
  else if (op == 1) { s.M = Cin; s.N = (int64_t)B * H * W; s.K = 9ll * Cout; s.out_elems = (int64_t)B * Cin * H * W; }
  else { s.M = Cout; s.N = (int64_t)Cin * T; s.K = (int64_t)B * Ho * Wo; s.out_elems = (int64_t)Cout * Cin * T; }
  s.nsplit = ig_nsplit(s.M, s.N, s.K, s.ncls, ig_wide(kind, op, s.M));
  return s;
}

static int64_t ig_ws_elems(int kind, int op, int B, int Cin, int H, int W, int Cout) {
  const IgShape s = ig_shape(kind, op, B, Cin, H, W, Cout);
  int64_t n = s.nsplit > 1 ? s.nsplit * s.out_elems : 0;
  if (kind == 4 && op == 1) n += 16ll * Cin * Cout;          // the parity-class weight pack
  return n > 0 ? n : 1;
}

template <int MODE, bool WIDE = false>
static int ig_launch(IgArgs a, const IgShape& sh, float* slabs, float* out, hipStream_t s, const char* what) {
  if (sh.M >= (1ll << 31) || sh.N >= (1ll << 31) || sh.K >= (1ll << 31) || sh.out_elems >= (1ll << 31)) return TGSR_EUNSUPPORTED;
  a.M = (int)sh.M; a.N = (int)sh.N; a.K = (int)sh.K;
  a.nsplit = sh.nsplit;
  const int chunks = (int)((sh.K + kIgKC - 1) / kIgKC);
  a.chunks_per_split = (chunks + sh.nsplit - 1) / sh.nsplit;
  a.nsplit = (chunks + a.chunks_per_split - 1) / a.chunks_per_split;     // no empty splits
  a.slab_stride = a.nsplit > 1 ? sh.out_elems : 0;
  a.out = a.nsplit > 1 ? slabs : out;
  if (a.nsplit > 1) a.act = 0;                                           // the slab sum applies it
  const dim3 grid(WIDE ? (unsigned)((sh.N + 255) / 256) : (unsigned)((sh.N + 127) / 128),
                  WIDE ? (unsigned)((sh.M + 63) / 64) : (unsigned)((sh.M + 127) / 128), (unsigned)(a.nsplit * sh.ncls));
  hipLaunchKernelGGL((dconv_igemm_kernel<MODE, WIDE>), grid, dim3(256), 0, s, a);
  return note_launch(hipGetLastError(), what);
}

static int ig_finish(const IgArgs& a, int nsplit_used, int act, const float* slabs, float* out, int64_t n, hipStream_t s) {
  if (nsplit_used <= 1) return TGSR_OK;
  const int rb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(slab_sum_kernel, dim3(rb), dim3(256), 0, s, slabs, out, n, nsplit_used, act);
  return note_launch(hipGetLastError(), "slab_sum_kernel");
}

static int ig_used_splits(const IgShape& sh) {
  const int chunks = (int)((sh.K + kIgKC - 1) / kIgKC);
  const int cps = (chunks + sh.nsplit - 1) / sh.nsplit;
  return (chunks + cps - 1) / cps;
}

static int dconv_fwd(int kind, const float* x, int B, int Cin, int H, int W, const float* w, int Cout, int act, float* ws,
                     float* out, void* stream) {
  if (!x || !w || !out || !ws || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return TGSR_EINVAL;
  if (kind == 4 && ((H | W) & 1)) return TGSR_EUNSUPPORTED;
  if ((int64_t)B * Cin * H * W >= (1ll << 31)) return TGSR_EUNSUPPORTED;        // 32-bit gather offsets
  const IgShape sh = ig_shape(kind, 0, B, Cin, H, W, Cout);
  IgArgs a = {};
  a.A = w; a.S = x; a.C = Cin; a.Hs = H; a.Ws = W;
  a.PH = kind == 4 ? H / 2 : H; a.PW = kind == 4 ? W / 2 : W; a.OH = a.PH; a.OW = a.PW; a.act = act ? 1 : 0;
  hipStream_t s = as_stream(stream);
  const int rc = kind == 4 ? (ig_wide(4, 0, sh.M) ? ig_launch<kFwd4, true>(a, sh, ws, out, s, "dconv_igemm_kernel<fwd4, wide>")
                                                  : ig_launch<kFwd4>(a, sh, ws, out, s, "dconv_igemm_kernel<fwd4>"))
                           : ig_launch<kFwd3>(a, sh, ws, out, s, "dconv_igemm_kernel<fwd3>");
  if (rc) return rc;
  return ig_finish(a, ig_used_splits(sh), act ? 1 : 0, ws, out, sh.out_elems, s);
}

static int dconv_dgrad(int kind, const float* dy, int B, int Cin, int H, int W, const float* w, int Cout, float* ws,
                       float* dx, void* stream) {
  if (!dy || !w || !ws || !dx || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return TGSR_EINVAL;
  if (kind == 4 && ((H | W) & 1)) return TGSR_EUNSUPPORTED;
  if ((int64_t)B * Cout * H * W >= (1ll << 31)) return TGSR_EUNSUPPORTED;       // 32-bit gather offsets (dy is at most this)
  const IgShape sh = ig_shape(kind, 1, B, Cin, H, W, Cout);
  hipStream_t s = as_stream(stream);
  IgArgs a = {};
  a.S = dy; a.C = Cout; a.OH = H; a.OW = W;
  int rc;
  float* slabs = ws;
  if (kind == 4 && Cin <= 4 && (int64_t)Cout * Cin * 64 <= 64 * 1024) {
    // the image layer: M = Cin rows of a 128-row GEMM tile would idle 97 % of the matrix pipe (dconv_dgrad4_image_kernel)
    const dim3 grid((unsigned)((W / 2 + 63) / 64), (unsigned)((H / 2 + 3) / 4), (unsigned)B);
    const size_t lds = (size_t)Cout * Cin * 16 * sizeof(float);
    switch (Cin) {
      case 1: hipLaunchKernelGGL(dconv_dgrad4_image_kernel<1>, grid, dim3(256), lds, s, dy, w, dx, Cout, H / 2, W / 2); break;
      case 2: hipLaunchKernelGGL(dconv_dgrad4_image_kernel<2>, grid, dim3(256), lds, s, dy, w, dx, Cout, H / 2, W / 2); break;
      case 3: hipLaunchKernelGGL(dconv_dgrad4_image_kernel<3>, grid, dim3(256), lds, s, dy, w, dx, Cout, H / 2, W / 2); break;
      default: hipLaunchKernelGGL(dconv_dgrad4_image_kernel<4>, grid, dim3(256), lds, s, dy, w, dx, Cout, H / 2, W / 2); break;
    }
    return note_launch(hipGetLastError(), "dconv_dgrad4_image_kernel");
  }
  if (kind == 4) {
    const int64_t total = 16ll * Cin * Cout;
    const int64_t nf = (int64_t)Cin * Cout;
    const int pb = (int)((nf + 255) / 256 < 4096 ? (nf + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv4x4s2_pack_dgrad_kernel, dim3(pb), dim3(256), 0, s, w, ws, Cout, Cin, total);
    slabs = ws + total;
    a.A = ws; a.Hs = H / 2; a.Ws = W / 2; a.PH = H / 2; a.PW = W / 2;
    rc = ig_wide(4, 1, sh.M) ? ig_launch<kDgrad4, true>(a, sh, slabs, dx, s, "dconv_igemm_kernel<dgrad4, wide>")
                             : ig_launch<kDgrad4>(a, sh, slabs, dx, s, "dconv_igemm_kernel<dgrad4>");
  } else {
    a.A = w; a.Hs = H; a.Ws = W; a.PH = H; a.PW = W;
    rc = ig_launch<kDgrad3>(a, sh, slabs, dx, s, "dconv_igemm_kernel<dgrad3>");
  }
  if (rc) return rc;
  return ig_finish(a, ig_used_splits(sh), 0, slabs, dx, sh.out_elems, s);
}

static int dconv_wgrad(int kind, const float* dy, const float* x, int B, int Cin, int H, int W, int Cout, float* ws,
                       float* dw, void* stream) {
  if (!dy || !x || !ws || !dw || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return TGSR_EINVAL;
  if (kind == 4 && ((H | W) & 1)) return TGSR_EUNSUPPORTED;
  if ((int64_t)B * Cin * H * W >= (1ll << 31)) return TGSR_EUNSUPPORTED;        // 32-bit gather offsets
  const IgShape sh = ig_shape(kind, 2, B, Cin, H, W, Cout);
  IgArgs a = {};
  a.A = dy; a.S = x; a.C = Cin; a.Hs = H; a.Ws = W;
  a.PH = kind == 4 ? H / 2 : H; a.PW = kind == 4 ? W / 2 : W; a.OH = a.PH; a.OW = a.PW;
  hipStream_t s = as_stream(stream);
  const int rc = kind == 4 ? ig_launch<kWgrad4>(a, sh, ws, dw, s, "dconv_igemm_kernel<wgrad4>")
                           : ig_launch<kWgrad3>(a, sh, ws, dw, s, "dconv_igemm_kernel<wgrad3>");
  if (rc) return rc;
  return ig_finish(a, ig_used_splits(sh), 0, ws, dw, sh.out_elems, s);
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_conv4x4s2_ws_elems(int op, int B, int Cin, int H, int W, int Cout) {
  return (op < 0 || op > 2) ? 0 : ig_ws_elems(4, op, B, Cin, H, W, Cout);
}
extern "C" int tgsr_conv4x4s2_fwd(const float* x, int B, int Cin, int H, int W, const float* w, int Cout, int act,
                                  float* ws, float* out, void* stream) {
  return dconv_fwd(4, x, B, Cin, H, W, w, Cout, act, ws, out, stream);
}
extern "C" int tgsr_conv4x4s2_dgrad(const float* dy, int B, int Cin, int H, int W, const float* w, int Cout, float* ws,
                                    float* dx, void* stream) {
  return dconv_dgrad(4, dy, B, Cin, H, W, w, Cout, ws, dx, stream);
}
extern "C" int tgsr_conv4x4s2_wgrad(const float* dy, const float* x, int B, int Cin, int H, int W, int Cout, float* ws,
                                    float* dw, void* stream) {
  return dconv_wgrad(4, dy, x, B, Cin, H, W, Cout, ws, dw, stream);
}

extern "C" int64_t tgsr_conv3x3_gemm_ws_elems(int op, int B, int Cin, int H, int W, int Cout) {
  return (op < 0 || op > 2) ? 0 : ig_ws_elems(3, op, B, Cin, H, W, Cout);
}
extern "C" int tgsr_conv3x3_gemm_fwd(const float* x, int B, int Cin, int H, int W, const float* w, int Cout, float* ws,
                                     float* out, void* stream) {
  return dconv_fwd(3, x, B, Cin, H, W, w, Cout, 0, ws, out, stream);
}
extern "C" int tgsr_conv3x3_gemm_dgrad(const float* dy, int B, int Cin, int H, int W, const float* w, int Cout, float* ws,
                                       float* dx, void* stream) {
  return dconv_dgrad(3, dy, B, Cin, H, W, w, Cout, ws, dx, stream);
}
extern "C" int tgsr_conv3x3_gemm_wgrad(const float* dy, const float* x, int B, int Cin, int H, int W, int Cout, float* ws,
                                       float* dw, void* stream) {
  return dconv_wgrad(3, dy, x, B, Cin, H, W, Cout, ws, dw, stream);
}

extern "C" int tgsr_leaky_relu(const float* x, const float* y_for_bwd, float* out, int64_t n, void* stream) {
  if (!x || !out || n < 1) return TGSR_EINVAL;
  const int b = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(leaky_kernel, dim3(b), dim3(256), 0, as_stream(stream), x, y_for_bwd, out, n);
  return note_launch(hipGetLastError(), "leaky_kernel");
}
