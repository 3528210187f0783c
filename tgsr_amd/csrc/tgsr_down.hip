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

#include <type_traits>

namespace tgsr {

enum { kFwd4 = 0, kFwd3 = 1, kDgrad4 = 2, kDgrad3 = 3, kWgrad4 = 4, kWgrad3 = 5,
       // generic taps (CNN_ENCODER's frozen trunk, tgsr_igemm.hip): any KH x KW <= 32 taps, forward stride 1 | 2, data gradient stride 1
       kFwdG = 6, kDgradG = 7 };

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
  int64_t a_bytes, s_bytes;   // dconv_igemm6_kernel: sizes of A and S (buffer descriptors; < 4 GB, host-checked)
  // kFwdG / kDgradG only
  int gKH, gKW, gKK, gst, gpadh, gpadw;
  unsigned gmKK, gmKW;        // q / KK = (q * gmKK) >> 16 for q < 64;  t / KW likewise
  int64_t s_bstride, o_bstride;          // batch strides (elements) of S and of the output (channel slices of wider tensors)
  const float* gbias;
  const float* gmask;         // nullable, laid out like the output: the contribution is kept where gmask > 0
  int grelu, gacc;
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
  if (MODE == kFwdG) {                      // where tap (0, 0) of the window sits
    iy0 = a.gst * py - a.gpadh;
    ix0 = a.gst * px - a.gpadw;
    off = (int)(b * a.s_bstride);
    return;
  }
  if (MODE == kDgradG) {                    // tap (ky, kx) reads g at (iy0 - ky, ix0 - kx)
    iy0 = py + a.gpadh;
    ix0 = px + a.gpadw;
    off = (int)(b * a.s_bstride);
    return;
  }
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

// ---------------------------------------------------------------------------------------------------------------------
// The same GEMMs on the bf16 matrix pipe, fp32 in and out ("x6", profiles/HISTORY.md 3.18).  `v_mfma_f32_32x32x2_f32` runs at the fp32
// VECTOR rate (64 FLOP/clk/SIMD, 1/16 of the bf16 MFMA) and every VALU instruction beside it costs its full issue time.  An
// fp32 value is exactly x0 + x1 + x2 with three bf16 pieces (round-to-nearest split: 9 + 9 + 9 >= 24 significand bits), a
// bf16 x bf16 product is exact in fp32 and the bf16 MFMA accumulates in fp32, so
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0) + [a1 b2 + a2 b1 + a2 b2],   [..] <= 2^-26 |a b|:
// six `v_mfma_f32_32x32x16_bf16` replace eight `32x32x2_f32` of the same K = 16 at 192 instead of 512 cycles, with the loaders'
// VALU work in the issue slots the bf16 MFMA leaves free.  Measured (tools/diag/exp_split_mfma.hip): 374 against 157
// fp32-equivalent TFLOP/s in registers, and - the a0 b0 products in their own accumulator, one rounding per 16 k instead of eight -
// 2.5-3x LESS error against fp64 than the fp32 MFMA (which equals a sequential fmaf chain bit for bit).
// Modes: the 4x4 stride-2 convolution's forward, data gradient and weight gradient with K % 16 == 0 (a K-chunk = one input
// channel's 16 taps | four output channels x 4 taps | 16 pixels); everything else stays on the kernel above.
// LDS tile per operand and piece: [k half 2][rows][8 k] bf16 (16 B per slot): item q -> (k half q / rows, row q % rows) loads 8
// consecutive k, splits them and writes three 16-byte runs at q * 16: linear, conflict-free; a lane's MFMA operand (row l & 31,
// k 8 (l >> 5) ..) is one ds_read_b128, 32 consecutive slots per half wave ([row][k half] gave every read a 2-way bank conflict).
typedef unsigned int u32x4d __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned ig6_pack2(float lo, float hi) {       // v_cvt_pk_bf16_f32: round to nearest even, lo in bits 0..15
  bf16x2d v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
// eight consecutive k of one row -> the three bf16 pieces (k even in the low half of a word)
__device__ __forceinline__ void ig6_split8(const float (&v)[8], u32x4d& p0, u32x4d& p1, u32x4d& p2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    const unsigned u0 = ig6_pack2(a, b);
    const float ra = a - __uint_as_float(u0 << 16), rb = b - __uint_as_float(u0 & 0xffff0000u);       // exact
    const unsigned u1 = ig6_pack2(ra, rb);
    const float sa = ra - __uint_as_float(u1 << 16), sb = rb - __uint_as_float(u1 & 0xffff0000u);     // exact
    p0[j] = u0;
    p1[j] = u1;
    p2[j] = ig6_pack2(sa, sb);
  }
}
typedef unsigned int u32x2d __attribute__((ext_vector_type(2)));
template <int NV>   // NV = 8 | 4 consecutive k -> NV / 2 words per piece
__device__ __forceinline__ void ig6_split(const float (&v)[NV], unsigned (&p0)[NV / 2], unsigned (&p1)[NV / 2], unsigned (&p2)[NV / 2]) {
#pragma unroll
  for (int j = 0; j < NV / 2; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    const unsigned u0 = ig6_pack2(a, b);
    const float ra = a - __uint_as_float(u0 << 16), rb = b - __uint_as_float(u0 & 0xffff0000u);       // exact
    const unsigned u1 = ig6_pack2(ra, rb);
    const float sa = ra - __uint_as_float(u1 << 16), sb = rb - __uint_as_float(u1 & 0xffff0000u);     // exact
    p0[j] = u0;
    p1[j] = u1;
    p2[j] = ig6_pack2(sa, sb);
  }
}
__device__ __forceinline__ f32x16 ig6_mfma(u32x4d a, u32x4d b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8d, a), __builtin_bit_cast(bf16x8d, b), c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) void* ig6_lds_t;

// fp32 rows [M][K] (K % 16 == 0) -> the LDS images of dconv_igemm6_kernel's A operand: [m tile][chunk][piece 3][MB rows][16 k] bf16,
// rows >= M zero.  One thread per (row, 8 consecutive k).
template <int MB>
__global__ __launch_bounds__(256) void ig6_split_rows_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, int M, int K,
                                                               int64_t total) {
  const int kg = K >> 3;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % kg), m = (int)(idx / kg);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)m * K + 8 * g);
      const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)m * K + 8 * g + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    }
    u32x4d p0, p1, p2;
    ig6_split8(v, p0, p1, p2);
    const int mt = m / MB, r = m - mt * MB;
    unsigned* d = dst + ((int64_t)mt * (K >> 4) + (g >> 1)) * (3 * MB * 8) + ((g & 1) * MB + r) * 4;
    *reinterpret_cast<u32x4d*>(d) = p0;
    *reinterpret_cast<u32x4d*>(d + MB * 8) = p1;
    *reinterpret_cast<u32x4d*>(d + 2 * MB * 8) = p2;
  }
}

// The data gradient's A operand from the forward weight w [Cout][Cin][4][4] (Cout % 4 == 0), regrouped by parity class and split:
// A[cls = 2p + q][m = ci][k = 4 co + 2a + c] = w[co][ci][p + 2a][q + 2c]  ->  images [cls][m tile][chunk][piece][MB][16], rows >= Cin zero.
// One thread per (ci, pair of output channels): two 64-byte filter reads, four classes x three 16-byte stores.
template <int MB>
__global__ __launch_bounds__(256) void ig6_pack_dgrad_kernel(const float* __restrict__ w, unsigned* __restrict__ dst, int Cout, int Cin,
                                                               int mtiles, int64_t total) {
  const int half_co = Cout >> 1, K = 4 * Cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j2 = (int)(idx % half_co), ci = (int)(idx / half_co);
    float f[2][16];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const float4* fp = reinterpret_cast<const float4*>(w + ((int64_t)(2 * j2 + o) * Cin + ci) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t = ci < Cin ? fp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        f[o][4 * i] = t.x; f[o][4 * i + 1] = t.y; f[o][4 * i + 2] = t.z; f[o][4 * i + 3] = t.w;
      }
    }
    const int mt = ci / MB, r = ci - mt * MB;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int p = cls >> 1, q = cls & 1;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f[e >> 2][(p + 2 * ((e >> 1) & 1)) * 4 + q + 2 * (e & 1)];
      u32x4d p0, p1, p2;
      ig6_split8(v, p0, p1, p2);
      unsigned* d = dst + (((int64_t)cls * mtiles + mt) * (K >> 4) + (j2 >> 1)) * (3 * MB * 8) + ((j2 & 1) * MB + r) * 4;
      *reinterpret_cast<u32x4d*>(d) = p0;
      *reinterpret_cast<u32x4d*>(d + MB * 8) = p1;
      *reinterpret_cast<u32x4d*>(d + 2 * MB * 8) = p2;
    }
  }
}

// APRE: the A operand arrives pre-split as the LDS images (forward / data gradient; ig6_split_rows_kernel / ig6_pack_dgrad_kernel).
template <int MODE, bool WIDE = false, bool APRE = false>
__global__ __launch_bounds__(256, 2) void dconv_igemm6_kernel(IgArgs a) {
  constexpr bool WG = MODE == kWgrad4 || MODE == kWgrad3;
  constexpr bool KG = MODE == kFwdG || MODE == kDgradG;    // generic taps: k = (channel, tap) in KKs, KK a launch argument
  constexpr bool K3 = MODE == kFwd3 || MODE == kDgrad3 || KG;    // 3x3: k = (channel, tap) in 9s - a chunk of 16 straddles channels
  constexpr int ST = (MODE == kFwd4 || MODE == kWgrad4) ? 2 : 1;       // stride of the gathered tensor's pixel grid
  static_assert(!(WG && WIDE) && !(WG && APRE) && !(K3 && APRE) && !(K3 && !KG && WIDE) && !(MODE == kWgrad3 && WIDE),
                "the wide tile and the pre-split A serve the 4x4 convolution's pixel-column modes (and the generic-tap ones)");
  constexpr int MB = WIDE ? 64 : 128, NB = WIDE ? 256 : 128;
  constexpr int APL = MB * 8, BPL = NB * 8;                // 32-bit words per piece of a tile
  constexpr int NBI = NB / 128;                            // B items per thread
  constexpr int AG = MB / 16;                              // A values per thread and chunk: 8 (one item) | 4 (half an item)
  constexpr int NAR = APRE ? 3 * AG / 2 : AG;              // staged A registers per thread: three pieces | AG fp32 values
  __shared__ __attribute__((aligned(16))) unsigned a_s[2][3 * APL];
  __shared__ __attribute__((aligned(16))) unsigned b_s[2][3 * BPL];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5, wave = tid >> 6;
  const int wm = WIDE ? 0 : wave >> 1, wn = WIDE ? wave : wave & 1;
  const int ncls = MODE == kDgrad4 ? 4 : 1;
  const int cls = MODE == kDgrad4 ? (int)blockIdx.z % ncls : 0;
  const int z = (int)blockIdx.z / ncls;
  const int m0 = blockIdx.y * MB, n0 = blockIdx.x * NB;
  const int kbeg = z * a.chunks_per_split * kIgKC;         // multiples of 16; K % 16 == 0 (host-checked): whole chunks only
  const int kend = min(a.K, kbeg + a.chunks_per_split * kIgKC);
  const int nchunks = (kend - kbeg) / kIgKC;
  const int HsWs = a.Hs * a.Ws, hw = a.PH * a.PW;

  // Every load is a buffer load whose descriptor is rebased per chunk (base and size are workgroup-uniform: scalar work) and
  // whose per-thread byte offsets are computed ONCE; what lies outside the image - or a row >= M, a column >= N - gets an offset
  // the range check rejects and reads as zero.  No branch guards a load: hipcc can count them, so the loads of chunk c + 2 stay in
  // flight while chunk c + 1 is stored (exec-masked loads made it wait for `vmcnt(0)` at the top of every trip).
  constexpr unsigned kOut = 0xffffffffu;
  // ---- A item of this thread: (k half, row) = (tid / MB, tid % MB), tid < 2 MB.  APRE: image [class][m tile][chunk][piece]
  // [k half][MB rows][8 k], rows >= M zero: this thread's 16 bytes of each piece.  Otherwise rows of K-contiguous fp32: forward
  // w [Cout][K], data gradient the fp32 class pack [Cin][4 Cout], weight gradient dy [b][Cout][hw] with k = b hw + r (hw % 16
  // == 0: a chunk's 16 pixels never leave an image, so b and r are uniform).
  // (the wide tile has 128 A items: a thread takes half of one - AG = 4 values - so that no branch guards the A path)
  const int aq = tid / (8 / AG), asub = tid % (8 / AG);    // item, and which AG-run of its 8 k
  const int arow = aq % MB, ahalf = aq / MB;
  const bool aok = m0 + arow < a.M;
  unsigned avo;                                            // byte offset of this thread's A data inside a chunk's descriptor
  int64_t abase;                                           // element offset of chunk 0's descriptor base (uniform)
  if (APRE) {
    abase = ((int64_t)(cls * (int)gridDim.y + (int)blockIdx.y) * (a.K / kIgKC) + kbeg / kIgKC) * (3 * APL);
    avo = (unsigned)(tid * (2 * AG));
  } else if (MODE == kDgrad3) {
    // A[m = ci][k = 9 co + t] = w[co][ci][8 - t] (the filter read transposed and flipped): row part here, (co, t) part per trip
    abase = 0;
    avo = aok ? (unsigned)((m0 + arow) * 36) : kOut;
  } else if (!WG) {
    abase = (MODE == kDgrad4 ? (int64_t)cls * a.M * a.K : 0) + kbeg;
    avo = aok ? (unsigned)(((int64_t)(m0 + arow) * a.K + 8 * ahalf + AG * asub) * 4) : kOut;
  } else {
    abase = 0;
    avo = aok ? (unsigned)(((int64_t)(m0 + arow) * hw + 8 * ahalf + AG * asub) * 4) : kOut;
  }
  // ---- B items: q = tid + 256 t -> (k half q / NB, row q % NB).  Forward / data gradient: row = a pixel, the 8 k of an item are
  // window taps: their byte offsets relative to (sample 0, the chunk's first channel) never change.  Weight gradient: row = a
  // column j = (ci, ky, kx), the 8 k are pixels.
  unsigned bvo[NBI][8];
  int wj[NBI], wsh[NBI];
#pragma unroll
  for (int t = 0; t < NBI; ++t) {
    const int q = tid + 256 * t, half = q / NB, n = n0 + q % NB;
    wj[t] = -1; wsh[t] = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) bvo[t][e] = kOut;
    if (n < a.N) {
      if (KG) {
        // bvo[t][0] = byte offset (mod 2^32: the anchor may lie outside the image) of the window's anchor in channel 0 of its
        // sample; wsh[t] = which of the KK <= 32 taps fall inside the image
        int poff, piy, pix_;
        ig_decode_p<MODE>(a, n, poff, piy, pix_);
        bvo[t][0] = (unsigned)(poff + piy * a.Ws + pix_) * 4u;
        for (int tp = 0; tp < a.gKK; ++tp) {
          const int ky = (int)(((unsigned)tp * a.gmKW) >> 16), kx = tp - ky * a.gKW;
          const int iy = MODE == kFwdG ? piy + ky : piy - ky, ix = MODE == kFwdG ? pix_ + kx : pix_ - kx;
          if ((unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws) wsh[t] |= 1 << tp;
        }
      } else if (K3) {
        // bvo[t][0] = byte offset of the pixel in channel 0 of its sample; wsh[t] = which of the nine taps fall inside the image
        int poff, piy, pix_;
        ig_decode_p<MODE>(a, n, poff, piy, pix_);
        bvo[t][0] = (unsigned)(poff + piy * a.Ws + pix_) * 4u;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
          if ((unsigned)(piy + tp / 3 - 1) < (unsigned)a.Hs && (unsigned)(pix_ + tp % 3 - 1) < (unsigned)a.Ws) wsh[t] |= 1 << tp;
      } else if (!WG) {
        int poff, piy, pix_;
        ig_decode_p<MODE>(a, n, poff, piy, pix_);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          int iy, ix, ch;
          if (MODE == kFwd4) {             // taps (ky = 2 half + (e >> 2), kx = e & 3) of channel k0 / 16
            iy = piy + 2 * half + (e >> 2) - 1;
            ix = pix_ + (e & 3) - 1;
            ch = 0;
          } else {                         // output channel k0 / 4 + 2 half + (e >> 2), tap (a = (e >> 1) & 1, c = e & 1)
            iy = piy + 1 - (cls >> 1) - ((e >> 1) & 1);
            ix = pix_ + 1 - (cls & 1) - (e & 1);
            ch = 2 * half + (e >> 2);
          }
          if ((unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws) bvo[t][e] = (unsigned)(poff + ch * HsWs + iy * a.Ws + ix) * 4u;
        }
      } else {
        int off, dyk, dxk;
        ig_decode_j<MODE>(a, n, 0, off, dyk, dxk);
        wj[t] = off;
        wsh[t] = (dyk + 4) | ((dxk + 4) << 4) | (half << 8);
      }
    }
  }

  // Two register sets: trip c starts with the loads of chunk c + 2 into set c % 2, multiplies chunk c and splits and stores chunk
  // c + 1 (the other set).  A chunk past the end gets a descriptor of zero records: its loads touch no memory and return zeros,
  // so no trip needs a branch.  (Three sets - two trips of cover - spilled 30-270 registers in every form tried; refilling a set
  // right behind its split - 1.25 trips of cover - measured the same as this.)
  constexpr int NSETS = 2;
  unsigned ra[NSETS][NAR];
  float rb[NSETS][NBI][8];
  auto load = [&](int c, unsigned (&qa)[NAR], float (&qb)[NBI][8]) {
    const int k0 = kbeg + c * kIgKC;
    const bool live = c < nchunks;
    // A
    int64_t aoff;                                          // uniform
    if (APRE) aoff = (abase + (int64_t)c * (3 * APL)) * 4;
    else if (MODE == kDgrad3) aoff = 0;
    else if (!WG) aoff = (abase + (int64_t)c * kIgKC) * 4;
    else {
      const int b = k0 / hw, r = k0 - b * hw;
      aoff = ((int64_t)b * a.M * hw + r) * 4;
    }
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.A)) + (live ? aoff : 0), 0, live ? (int)(unsigned)(a.a_bytes - aoff) : 0, 0x00020000);
    if constexpr (APRE) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if constexpr (AG == 8) {
          const u32x4d v = __builtin_amdgcn_raw_buffer_load_b128(ar, avo + p * (APL * 4), 0, 0);
          qa[4 * p] = v[0]; qa[4 * p + 1] = v[1]; qa[4 * p + 2] = v[2]; qa[4 * p + 3] = v[3];
        } else {
          const u32x2d v = __builtin_amdgcn_raw_buffer_load_b64(ar, avo + p * (APL * 4), 0, 0);
          qa[2 * p] = v[0]; qa[2 * p + 1] = v[1];
        }
      }
    } else if constexpr (MODE == kDgrad3) {
      // the eight (co, t) of this item are wave-uniform (the k half is): scalar arithmetic, one add per load
      const int kh = __builtin_amdgcn_readfirstlane(ahalf);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + 8 * kh + e, co = k / 9, tp = k - 9 * co;
        qa[e] = __builtin_amdgcn_raw_buffer_load_b32(ar, avo == kOut ? kOut : avo + (unsigned)((co * a.M * 9 + 8 - tp) * 4), 0, 0);
      }
    } else {
      const u32x4d v0 = __builtin_amdgcn_raw_buffer_load_b128(ar, avo, 0, 0);
      qa[0] = v0[0]; qa[1] = v0[1]; qa[2] = v0[2]; qa[3] = v0[3];
      if constexpr (AG == 8) {
        const u32x4d v1 = __builtin_amdgcn_raw_buffer_load_b128(ar, avo == kOut ? kOut : avo + 16u, 0, 0);
        qa[4] = v1[0]; qa[5] = v1[1]; qa[6] = v1[2]; qa[7] = v1[3];
      }
    }
    // B
    if constexpr (KG) {
      // as the 3x3 path below with KK = KH KW taps per channel (launch arguments; the small divisions by multiply-shift)
      const int KK = a.gKK;
      const int c0 = k0 / KK, r0 = k0 - KK * c0;
      const int64_t soff = (int64_t)c0 * HsWs * 4;
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.S)) + (live ? soff : 0), 0, live ? (int)(unsigned)(a.s_bytes - soff) : 0, 0x00020000);
#pragma unroll
      for (int t = 0; t < NBI; ++t) {
        const int kh = __builtin_amdgcn_readfirstlane((tid + 256 * t) / NB);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int q = r0 + 8 * kh + e;
          const int cl = (int)(((unsigned)q * a.gmKK) >> 16), tp = q - KK * cl;
          const int ky = (int)(((unsigned)tp * a.gmKW) >> 16), kx = tp - ky * a.gKW;
          const int tapo = ky * a.Ws + kx;
          const int off = (cl * HsWs + (MODE == kFwdG ? tapo : -tapo)) * 4;
          qb[t][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sr, ((wsh[t] >> tp) & 1) ? bvo[t][0] + (unsigned)off : kOut, 0, 0));
        }
      }
    } else if constexpr (K3) {
      // k = 9 channel + tap: the chunk starts in channel c0 at tap r0 (scalars); element j = 8 (k half) + e of it is tap
      // (r0 + j) % 9 of channel c0 + (r0 + j) / 9 - wave-uniform, since the k half of an item is: scalar arithmetic, and per
      // thread one mask test, one add and one select per load
      const int c0 = k0 / 9, r0 = k0 - 9 * c0;
      const int64_t soff = (int64_t)c0 * HsWs * 4;
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.S)) + (live ? soff : 0), 0, live ? (int)(unsigned)(a.s_bytes - soff) : 0, 0x00020000);
#pragma unroll
      for (int t = 0; t < NBI; ++t) {
        const int kh = __builtin_amdgcn_readfirstlane((tid + 256 * t) / NB);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int q = r0 + 8 * kh + e, cl = q / 9, tp = q - 9 * cl;
          const int off = (cl * HsWs + (tp / 3 - 1) * a.Ws + tp % 3 - 1) * 4;
          qb[t][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sr, ((wsh[t] >> tp) & 1) ? bvo[t][0] + (unsigned)off : kOut, 0, 0));
        }
      }
    } else if (!WG) {
      const int64_t soff = (int64_t)(MODE == kFwd4 ? (k0 >> 4) : (k0 >> 2)) * HsWs * 4;      // the chunk's first channel
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.S)) + (live ? soff : 0), 0, live ? (int)(unsigned)(a.s_bytes - soff) : 0, 0x00020000);
#pragma unroll
      for (int t = 0; t < NBI; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) qb[t][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sr, bvo[t][e], 0, 0));
    } else {
      // the chunk's 16 pixels lie in one image b (uniform); pixel r0 + 8 half + e -> (py, px)
      const int b = k0 / hw, r0 = k0 - b * hw;
      const int64_t soff = (int64_t)b * a.C * HsWs * 4;
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.S)) + (live ? soff : 0), 0, live ? (int)(unsigned)(a.s_bytes - soff) : 0, 0x00020000);
      const int py0 = r0 / a.PW, px0 = r0 - py0 * a.PW;
#pragma unroll
      for (int t = 0; t < NBI; ++t) {
        const int dyk = (wsh[t] & 15) - 4, dxk = ((wsh[t] >> 4) & 15) - 4;
        int py = py0, px = px0 + 8 * (wsh[t] >> 8);
        while (px >= a.PW) { px -= a.PW; ++py; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int iy = ST * py + dyk, ix = ST * px + dxk;
          const bool ok = wj[t] >= 0 && (unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws;
          qb[t][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sr, ok ? (unsigned)(wj[t] + iy * a.Ws + ix) * 4u : kOut, 0, 0));
          ++px;
          if (px == a.PW) { px = 0; ++py; }
        }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  const int fa = (hh * MB + wm * 64 + l31) * 4, fb = (hh * NB + wn * 64 + l31) * 4;       // word offset of this lane's fragment, block 0

  // split (or, APRE, pass on) a register set and write it into LDS buffer `ob`: branch-free
  auto stage = [&](int ob, const unsigned (&qa)[NAR], const float (&qb)[NBI][8]) {
    unsigned p[3][AG / 2];
    if constexpr (APRE) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < AG / 2; ++j) p[i][j] = qa[i * (AG / 2) + j];
    } else {
      float v[AG];
#pragma unroll
      for (int e = 0; e < AG; ++e) v[e] = __uint_as_float(qa[e]);
      ig6_split<AG>(v, p[0], p[1], p[2]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if constexpr (AG == 8) *reinterpret_cast<u32x4d*>(&a_s[ob][i * APL + tid * 4]) = u32x4d{p[i][0], p[i][1], p[i][2], p[i][3]};
      else *reinterpret_cast<u32x2d*>(&a_s[ob][i * APL + tid * 2]) = u32x2d{p[i][0], p[i][1]};
    }
#pragma unroll
    for (int t = 0; t < NBI; ++t) {
      u32x4d p0, p1, p2;
      ig6_split8(qb[t], p0, p1, p2);
      *reinterpret_cast<u32x4d*>(&b_s[ob][(tid + 256 * t) * 4]) = p0;
      *reinterpret_cast<u32x4d*>(&b_s[ob][BPL + (tid + 256 * t) * 4]) = p1;
      *reinterpret_cast<u32x4d*>(&b_s[ob][2 * BPL + (tid + 256 * t) * 4]) = p2;
    }
  };

  // One trip (C = c mod 6: LDS buffer C % 2, register sets by C % NSETS): the loads of chunk c + NSETS into set c % NSETS - the 24
  // MFMAs of chunk c out of LDS buffer c % 2, with the split of chunk c + 1 (set (c + 1) % NSETS: 44 VALU operations per item) in
  // the issue slots the MFMAs leave free, ~6 per MFMA, and its LDS writes into the other buffer (everybody finished reading that
  // one before the barrier of the last trip) - barrier.  The interleave is spelled out (sched_group_barrier): left to itself
  // hipcc emits the MFMAs as one block and the split as another, and the two waves of a SIMD fall into step - both multiply,
  // then both split: 45 % of the matrix pipe's cycles.
  auto trip = [&](auto cc, int c) {
    constexpr int PAR = decltype(cc)::value, SL = PAR, SS = PAR ^ 1;
    load(c + 2, ra[SL], rb[SL]);
    // fragment reads in the order the MFMAs consume them: the first product needs two reads, not nine
    u32x4d af[2][3], bf[2][3];
    auto rda = [&](int blk, int p) { af[blk][p] = *reinterpret_cast<const u32x4d*>(&a_s[PAR][p * APL + fa + blk * 128]); };
    auto rdb = [&](int blk, int p) { bf[blk][p] = *reinterpret_cast<const u32x4d*>(&b_s[PAR][p * BPL + fb + blk * 128]); };
    rda(0, 2); rdb(0, 0); rda(0, 0); rdb(0, 2); rda(0, 1); rdb(0, 1);
    rdb(1, 0); rdb(1, 2); rdb(1, 1);
    rda(1, 2); rda(1, 0); rda(1, 1);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        f32x16 s = acc[mb][nb];                            // the five small products first, smallest first
        s = ig6_mfma(af[mb][2], bf[nb][0], s);
        s = ig6_mfma(af[mb][0], bf[nb][2], s);
        s = ig6_mfma(af[mb][1], bf[nb][1], s);
        s = ig6_mfma(af[mb][1], bf[nb][0], s);
        s = ig6_mfma(af[mb][0], bf[nb][1], s);
        acc[mb][nb] = ig6_mfma(af[mb][0], bf[nb][0], s);
      }
    stage(PAR ^ 1, ra[SS], rb[SS]);
    // the order of the region: masks 0x008 MFMA, 0x002 VALU, 0x100 LDS read, 0x200 LDS write, 0x020 VMEM read
    __builtin_amdgcn_sched_group_barrier(0x020, 2 + 8 * NBI, 0);   // the loads first
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      if (i == 9 || i == 17 || i == 23) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
    }
    __syncthreads();
  };

  // prologue: chunks 0 and 1 into the register sets, chunk 0 into LDS buffer 0
  load(0, ra[0], rb[0]);
  load(1, ra[1], rb[1]);
  stage(0, ra[0], rb[0]);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    trip(std::integral_constant<int, 0>{}, c);
    if (c + 1 >= nchunks) break;
    trip(std::integral_constant<int, 1>{}, c + 1);
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
      const int b = n / hw, r = n - b * hw;
      if (MODE == kDgrad4) {
        const int py = r / a.PW, px = r - py * a.PW;
        obase = (int64_t)b * a.M * a.OH * a.OW + (2 * py + 1 - (cls >> 1)) * a.OW + 2 * px + 1 - (cls & 1);
        mstride = (int64_t)a.OH * a.OW;
      } else {
        obase = (int64_t)b * a.M * hw + r;
        mstride = hw;
      }
    }
    if (KG && a.nsplit <= 1) {               // straight into the output's channel slice: + shift, ReLU, += as asked
      const int b = n / hw;
      obase = (int64_t)b * a.o_bstride + (n - b * hw);
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm * 64 + mb * 32 + acc_row(i, hh);
        if (m >= a.M) continue;
        float v = acc[mb][nb][i];
        if (KG) {
          if (a.nsplit <= 1) {
            if (a.gbias) v += a.gbias[m];
            if (a.grelu) v = v > 0.f ? v : 0.f;
            if (a.gmask && !(a.gmask[obase + m * mstride] > 0.f)) v = 0.f;
            if (a.gacc) v += ob[obase + m * mstride];
          }
        } else if (a.act) v = v > 0.f ? v : 0.2f * v;
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

// 1 (default): the 4x4 convolution's GEMMs run on the bf16 matrix pipe with three-piece operands (dconv_igemm6_kernel) where
// the shape allows; 0: fp32 MFMA everywhere.  TGSR_DCONV_SPLIT=0 | tgsr_dconv_set_split(0).
// Bits: 1 = on; 2 = the weights pre-split into LDS images by a pass of their own (forward / data gradient).  (The a0 b0 products in
// accumulators of their own - HILO, 2.5-3x less error than the fp32 MFMA - need 64 registers more than two waves per SIMD leave.)
static int g_ig_split = [] {
  const char* e = getenv("TGSR_DCONV_SPLIT");
  return e ? (atoi(e) & 3) : 1;
}();

// floats of workspace the split images of the A operand take (dconv_igemm6_kernel: forward = the weight, data gradient = its four
// parity-class regroupings; 6 bytes per element, rows padded to whole tiles); 0 for shapes the split form does not take
static int64_t ig6_image_elems(int kind, int op, int64_t M, int64_t K) {
  if (!(g_ig_split & 2) || kind != 4 || op == 2 || K % kIgKC != 0) return 0;       // (only the pre-split variant needs them)
  const int64_t MB = ig_wide(kind, op, M) ? 64 : 128, mt = (M + MB - 1) / MB;
  return (op == 1 ? 4 : 1) * mt * MB * K * 3 / 2;
}
// the head of the workspace: the A images, or (data gradient on the fp32 MFMA) the fp32 parity-class pack; slabs follow
static int64_t ig_ws_head(int kind, int op, const IgShape& s, int Cin, int Cout) {
  const int64_t img = ig6_image_elems(kind, op, s.M, s.K), pack = (kind == 4 && op == 1) ? 16ll * Cin * Cout : 0;
  return img > pack ? img : pack;
}

static int64_t ig_ws_elems(int kind, int op, int B, int Cin, int H, int W, int Cout) {
  const IgShape s = ig_shape(kind, op, B, Cin, H, W, Cout);
  const int64_t n = (s.nsplit > 1 ? s.nsplit * s.out_elems : 0) + ig_ws_head(kind, op, s, Cin, Cout);
  return n > 0 ? n : 1;
}


template <int MODE, bool WIDE = false>
static int ig_launch(IgArgs a, const IgShape& sh, float* slabs, float* out, hipStream_t s, const char* what, bool split = false) {
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
  if (split) {
    constexpr bool CANPRE = MODE == kFwd4 || MODE == kDgrad4;
    const bool pre = CANPRE && (g_ig_split & 2);
    if constexpr (CANPRE) {
      if (pre) hipLaunchKernelGGL((dconv_igemm6_kernel<MODE, WIDE, true>), grid, dim3(256), 0, s, a);
    }
    if (!pre) hipLaunchKernelGGL((dconv_igemm6_kernel<MODE, WIDE, false>), grid, dim3(256), 0, s, a);
    return note_launch(hipGetLastError(), "dconv_igemm6_kernel");
  }
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
  float* slabs = ws + ig_ws_head(kind, 0, sh, Cin, Cout);
  const bool wide = ig_wide(kind, 0, sh.M);
  const int64_t img = ig6_image_elems(kind, 0, sh.M, sh.K);
  a.s_bytes = (int64_t)B * Cin * H * W * 4;
  a.a_bytes = (g_ig_split & 2) ? img * 4 : (int64_t)Cout * sh.K * 4;
  if (kind == 3) a.a_bytes = (int64_t)Cout * sh.K * 4;
  // split form: whole chunks (3x3: 9 Cin % 16 == 0), 16-byte rows of w, both tensors inside a 4 GB descriptor
  const bool split = g_ig_split && sh.K % kIgKC == 0 && !(reinterpret_cast<uintptr_t>(ws) & 15) &&
                     !(reinterpret_cast<uintptr_t>(w) & 15) && a.s_bytes < (1ll << 32) && a.a_bytes < (1ll << 32);
  if (split && (g_ig_split & 2) && kind == 4) {   // the weight, split into three bf16 pieces, as the kernel's LDS images
    const int64_t MB = wide ? 64 : 128, total = ((sh.M + MB - 1) / MB) * MB * (sh.K / 8);
    const int pb = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (wide) hipLaunchKernelGGL(ig6_split_rows_kernel<64>, dim3(pb), dim3(256), 0, s, w, reinterpret_cast<unsigned*>(ws), (int)sh.M, (int)sh.K, total);
    else hipLaunchKernelGGL(ig6_split_rows_kernel<128>, dim3(pb), dim3(256), 0, s, w, reinterpret_cast<unsigned*>(ws), (int)sh.M, (int)sh.K, total);
    const int prc = note_launch(hipGetLastError(), "ig6_split_rows_kernel");
    if (prc) return prc;
    a.A = ws;
  }
  const int rc = kind == 4 ? (wide ? ig_launch<kFwd4, true>(a, sh, slabs, out, s, "dconv_igemm_kernel<fwd4, wide>", split)
                                   : ig_launch<kFwd4>(a, sh, slabs, out, s, "dconv_igemm_kernel<fwd4>", split))
                           : ig_launch<kFwd3>(a, sh, slabs, out, s, "dconv_igemm_kernel<fwd3>", split);
  if (rc) return rc;
  return ig_finish(a, ig_used_splits(sh), act ? 1 : 0, slabs, out, sh.out_elems, s);
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
    const bool wide = ig_wide(4, 1, sh.M);
    const int64_t img = ig6_image_elems(4, 1, sh.M, sh.K);
    a.s_bytes = (int64_t)B * Cout * (H / 2) * (W / 2) * 4;
    a.a_bytes = (g_ig_split & 2) ? img * 4 : 64ll * Cin * Cout;
    const bool split = g_ig_split && sh.K % kIgKC == 0 && !(reinterpret_cast<uintptr_t>(ws) & 15) && !(reinterpret_cast<uintptr_t>(w) & 15) &&
                       a.s_bytes < (1ll << 32) && a.a_bytes < (1ll << 32);
    if (split && (g_ig_split & 2)) {   // regrouped by parity class AND split into three bf16 pieces, as the kernel's LDS images
      const int64_t MB = wide ? 64 : 128, mt = (sh.M + MB - 1) / MB, total = mt * MB * (Cout / 2);
      const int pb = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
      if (wide) hipLaunchKernelGGL(ig6_pack_dgrad_kernel<64>, dim3(pb), dim3(256), 0, s, w, reinterpret_cast<unsigned*>(ws), Cout, Cin, (int)mt, total);
      else hipLaunchKernelGGL(ig6_pack_dgrad_kernel<128>, dim3(pb), dim3(256), 0, s, w, reinterpret_cast<unsigned*>(ws), Cout, Cin, (int)mt, total);
      rc = note_launch(hipGetLastError(), "ig6_pack_dgrad_kernel");
    } else {
      const int64_t total = 16ll * Cin * Cout;
      const int64_t nf = (int64_t)Cin * Cout;
      const int pb = (int)((nf + 255) / 256 < 4096 ? (nf + 255) / 256 : 4096);
      hipLaunchKernelGGL(conv4x4s2_pack_dgrad_kernel, dim3(pb), dim3(256), 0, s, w, ws, Cout, Cin, total);
      rc = note_launch(hipGetLastError(), "conv4x4s2_pack_dgrad_kernel");
    }
    if (rc) return rc;
    slabs = ws + ig_ws_head(4, 1, sh, Cin, Cout);
    a.A = ws; a.Hs = H / 2; a.Ws = W / 2; a.PH = H / 2; a.PW = W / 2;
    rc = wide ? ig_launch<kDgrad4, true>(a, sh, slabs, dx, s, "dconv_igemm_kernel<dgrad4, wide>", split)
              : ig_launch<kDgrad4>(a, sh, slabs, dx, s, "dconv_igemm_kernel<dgrad4>", split);
  } else {
    a.A = w; a.Hs = H; a.Ws = W; a.PH = H; a.PW = W;
    a.s_bytes = (int64_t)B * Cout * H * W * 4;
    a.a_bytes = 36ll * Cin * Cout;
    const bool split = g_ig_split && sh.K % kIgKC == 0 && a.s_bytes < (1ll << 32) && a.a_bytes < (1ll << 32);
    rc = ig_launch<kDgrad3>(a, sh, slabs, dx, s, "dconv_igemm_kernel<dgrad3>", split);
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
  // split form: 16-byte rows of dy; a chunk's sixteen pixels inside one image
  a.a_bytes = (int64_t)B * Cout * a.PH * a.PW * 4;
  a.s_bytes = (int64_t)B * Cin * H * W * 4;
  const bool split = g_ig_split && (a.PH * a.PW) % kIgKC == 0 && !(reinterpret_cast<uintptr_t>(dy) & 15) &&
                     a.s_bytes < (1ll << 32) && a.a_bytes < (1ll << 32);
  const int rc = kind == 4 ? ig_launch<kWgrad4>(a, sh, ws, dw, s, "dconv_igemm_kernel<wgrad4>", split)
                           : ig_launch<kWgrad3>(a, sh, ws, dw, s, "dconv_igemm_kernel<wgrad3>", split);
  if (rc) return rc;
  return ig_finish(a, ig_used_splits(sh), 0, ws, dw, sh.out_elems, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Generic-tap convolutions (CNN_ENCODER's frozen trunk: tgsr_igemm.hip owns the entry point, the fp32-MFMA fallback and the slab
// finish) on dconv_igemm6_kernel: forward (stride 1 | 2) and stride-1 data gradient of any KH x KW <= 32 taps with K % 16 == 0.
// Returns TGSR_EUNSUPPORTED for shapes this form does not take - the caller then uses its own kernel.
int ig6_gconv_launch(int dgrad, const float* A, const float* S, int64_t s_bstride, int64_t s_bytes, int B, int Hs, int Ws, int M, int K,
                     int PH, int PW, int KH, int KW, int stride, int padh, int padw, const float* bias, int relu, int accumulate,
                     const float* mask, float* out, int64_t o_bstride, float* slabs, int nsplit, int chunks_per_split, hipStream_t s) {
  const int KK = KH * KW;
  if (!g_ig_split || KK > 25 || K % kIgKC != 0 || (dgrad && stride != 1) || (stride != 1 && stride != 2)) return TGSR_EUNSUPPORTED;
  const int64_t a_bytes = (int64_t)M * K * 4;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || s_bytes >= (1ll << 31) || a_bytes >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  const int64_t N = (int64_t)B * PH * PW;
  if (N >= (1ll << 31)) return TGSR_EUNSUPPORTED;
  IgArgs a = {};
  a.A = A; a.S = S; a.out = nsplit > 1 ? slabs : out;
  a.M = M; a.N = (int)N; a.K = K; a.C = K / KK; a.Hs = Hs; a.Ws = Ws; a.PH = PH; a.PW = PW; a.OH = PH; a.OW = PW;
  a.nsplit = nsplit; a.chunks_per_split = chunks_per_split;
  a.slab_stride = nsplit > 1 ? (int64_t)B * M * PH * PW : 0;
  a.act = 0;
  a.a_bytes = a_bytes; a.s_bytes = s_bytes;
  a.gKH = KH; a.gKW = KW; a.gKK = KK; a.gst = stride; a.gpadh = padh; a.gpadw = padw;
  a.gmKK = (unsigned)((65536 + KK - 1) / KK); a.gmKW = (unsigned)((65536 + KW - 1) / KW);
  a.s_bstride = s_bstride; a.o_bstride = o_bstride;
  a.gbias = bias; a.gmask = mask; a.grelu = relu; a.gacc = accumulate;
  const bool wide = M <= 64;
  const dim3 grid(wide ? (unsigned)((N + 255) / 256) : (unsigned)((N + 127) / 128), wide ? (unsigned)((M + 63) / 64) : (unsigned)((M + 127) / 128),
                  (unsigned)nsplit);
  if (dgrad) {
    if (wide) hipLaunchKernelGGL((dconv_igemm6_kernel<kDgradG, true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dconv_igemm6_kernel<kDgradG, false, false>), grid, dim3(256), 0, s, a);
  } else {
    if (wide) hipLaunchKernelGGL((dconv_igemm6_kernel<kFwdG, true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dconv_igemm6_kernel<kFwdG, false, false>), grid, dim3(256), 0, s, a);
  }
  return note_launch(hipGetLastError(), "dconv_igemm6_kernel<generic taps>");
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_dconv_set_split(int on) {
  const int was = g_ig_split;
  g_ig_split = on & 3;
  return was;
}

extern "C" int tgsr_conv4x4s2_split_form(int op, int B, int Cin, int H, int W, int Cout) {
  if (!g_ig_split || op < 0 || op > 2 || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2 || ((H | W) & 1)) return 0;
  const IgShape sh = ig_shape(4, op, B, Cin, H, W, Cout);
  const int64_t xb = (int64_t)B * Cin * H * W * 4, yb = (int64_t)B * Cout * (H / 2) * (W / 2) * 4, lim = 1ll << 32;
  if (op == 0) return sh.K % kIgKC == 0 && xb < lim && (int64_t)Cout * sh.K * 4 < lim;
  if (op == 1) return !(Cin <= 4 && (int64_t)Cout * Cin * 64 <= 64 * 1024) && sh.K % kIgKC == 0 && yb < lim;
  return ((H / 2) * (W / 2)) % kIgKC == 0 && xb < lim && yb < lim;
}

extern "C" int tgsr_conv3x3_gemm_split_form(int op, int B, int Cin, int H, int W, int Cout) {
  if (!g_ig_split || op < 0 || op > 2 || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return 0;
  const int64_t xb = (int64_t)B * Cin * H * W * 4, yb = (int64_t)B * Cout * H * W * 4, wb = 36ll * Cin * Cout, lim = 1ll << 32;
  if (xb >= lim || yb >= lim || wb >= lim) return 0;
  return op == 0 ? (9 * Cin) % kIgKC == 0 : (op == 1 ? (9 * Cout) % kIgKC == 0 : (H * W) % kIgKC == 0);
}

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
