// CNN_ENCODER's frozen trunk (util.py:263-368: the sixteen Inception-v3 blocks the reference copies out of torchvision, eval mode,
// `requires_grad = False` on every parameter, util.py:274-275) on the library's own kernels: what generator_loss needs of it
// (losses.py:375-389) is the forward and the gradient with respect to the IMAGE - never a weight gradient.
//
//   conv (any KH x KW, stride 1 | 2, zero padding) + eval-mode BatchNorm + ReLU, forward:
//       y[b][coff + co][p] = relu(sum_j w'[co][j] S(p, j) + shift[co]),  w' = w * gamma / sqrt(var + eps) folded once (frozen)
//       one implicit GEMM  M = Cout, N = B OH OW, K = Cin KH KW  on the fp32 MFMA (32x32x2), operands gathered from NCHW by index
//       arithmetic (no im2col buffer), the output written straight into its channel slice of the block's concatenation
//       (torch.cat never runs);
//   its data gradient:  dx[b][ci][q] (+)= sum_k w'T[ci][k] G(q, k),  k = (co, ky, kx),  G = g[b][co][(q + pad - k) / stride]
//       the same kernel with the gather turned around (M = Cin, K = Cout KH KW); stride 2: taps whose source would fall
//       between output pixels read zero;
//   the block plumbing around them: 3x3 / stride-2 max pool, 3x3 / stride-1 average pool (count_include_pad), the 8x8 global
//   average, the ReLU mask g = dy * (y > 0), the bilinear 299 x 299 resize (nn.Upsample(size=(299, 299), mode='bilinear'),
//   util.py:310) - each forward and backward, all deterministic (gather form, no atomics).
// Few output pixels against a long reduction (the 8 x 8 stage: N = 1024, K up to 18 432) split K over blockIdx.z into slabs
// that a finishing kernel sums in a fixed order (+ shift, ReLU, the slice write).
#include "tgsr_common.h"

namespace tgsr {

struct GcArgs {
  const float* A;        // forward: w' [Cout][Cin KH KW]; data gradient: w'T [Cin][Cout KH KW]
  const float* S;        // the gathered tensor (forward: x, data gradient: g), based at its channel slice
  const float* bias;     // forward: shift [Cout]; nullptr: none
  float* out;            // output based at its channel slice (or the slabs when nsplit > 1)
  int M, N, K;
  int Hs, Ws;            // spatial size of S
  int PH, PW;            // the pixel grid N runs over (forward: output pixels; data gradient: input pixels)
  int64_t s_bstride, o_bstride;      // batch strides (elements) of S and out
  int KH, KW, SH, PADH, PADW;
  int relu, accumulate;
  const float* mask;     // nullable; laid out like `out`: the contribution is kept where mask > 0 (the ReLU of the tensor whose gradient this is)
  int nsplit, chunks_per_split;
  int64_t slab_stride;
};

constexpr int kGcKC = 16;

// DG = false: forward gather; true: data-gradient gather.  WIDE: a 64 (M) x 256 (N) tile for M <= 64, else 128 x 128.
template <bool DG, bool WIDE>
__global__ __launch_bounds__(256) void gconv_igemm_kernel(GcArgs a) {
  constexpr int MB = WIDE ? 64 : 128, NB = WIDE ? 256 : 128, PA = MB + 4, PB = NB + 4, NLA = MB / 16, NLB = WIDE ? 16 : 8;
  __shared__ float a_s[2][kGcKC * PA];
  __shared__ float b_s[2][kGcKC * PB];
  __shared__ int tap_s[64];                   // tap t -> (ky << 8) | kx
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5, wave = tid >> 6;
  const int wm = WIDE ? 0 : wave >> 1, wn = WIDE ? wave : wave & 1;
  const int z = blockIdx.z;
  const int m0 = blockIdx.y * MB, n0 = blockIdx.x * NB;
  const int KK = a.KH * a.KW;
  if (tid < KK) tap_s[tid] = ((tid / a.KW) << 8) | (tid % a.KW);
  const int kbeg = z * a.chunks_per_split * kGcKC;
  const int kend = min(a.K, kbeg + a.chunks_per_split * kGcKC);
  const int nchunks = kend > kbeg ? (kend - kbeg + kGcKC - 1) / kGcKC : 0;
  const int HsWs = a.Hs * a.Ws;

  // A tile [MB m][16 k]: k = tid & 15, m = (tid >> 4) + 16 i (consecutive lanes walk consecutive k of a row)
  const int ak = tid & 15, am = tid >> 4;
  // B tile [16 k][NB n]: this thread's pixel n and the k rows bk(i) it loads
  const int bn = WIDE ? tid : tid & 127;
  const int bk0 = WIDE ? 0 : tid >> 7, bks = WIDE ? 1 : 2;          // k row of item i: bk0 + bks * i
  const int n = n0 + bn;
  const bool pok = n < a.N;
  int64_t poff = 0;
  int py = 0, px = 0;
  if (pok) {
    const int hw = a.PH * a.PW, b = n / hw, r = n - b * hw;
    py = r / a.PW;
    px = r - py * a.PW;
    poff = (int64_t)b * a.s_bstride;
  }
  // where tap (0, 0) of this pixel's window sits (forward), or the numerator of the source pixel (data gradient)
  const int y0 = DG ? py + a.PADH : py * a.SH - a.PADH;
  const int x0 = DG ? px + a.PADW : px * a.SH - a.PADW;
  // (channel, tap) of this thread's NLB rows, advanced by 16 per chunk without a division
  int jc[NLB], jt[NLB];
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int j = kbeg + bk0 + bks * i;
    jc[i] = j / KK;
    jt[i] = j - jc[i] * KK;
  }
  const int q16 = kGcKC / KK, r16 = kGcKC - q16 * KK;
  __syncthreads();                                       // tap_s

  float ra[NLA], rb[NLB];
  auto load = [&](int c) {
    const int k0 = kbeg + c * kGcKC;
    {
      const int k = k0 + ak;
      const bool kok = k < kend;
#pragma unroll
      for (int i = 0; i < NLA; ++i) {
        const int m = m0 + am + 16 * i;
        ra[i] = (kok && m < a.M) ? a.A[(int64_t)m * a.K + k] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int j = k0 + bk0 + bks * i;
      float v = 0.f;
      if (pok && j < kend) {
        const int tp = tap_s[jt[i]], ky = tp >> 8, kx = tp & 255;
        int iy, ix;
        bool ok;
        if (DG) {
          const int ty = y0 - ky, tx = x0 - kx;
          if (a.SH == 1) {
            iy = ty; ix = tx;
            ok = true;
          } else {
            ok = ty >= 0 && tx >= 0 && ((ty | tx) & 1) == 0;
            iy = ty >> 1; ix = tx >> 1;
          }
        } else {
          iy = y0 + ky; ix = x0 + kx;
          ok = true;
        }
        if (ok && (unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws)
          v = a.S[poff + (int64_t)jc[i] * HsWs + iy * a.Ws + ix];
      }
      rb[i] = v;
      // advance (channel, tap) by one chunk
      jc[i] += q16;
      jt[i] += r16;
      if (jt[i] >= KK) { jt[i] -= KK; ++jc[i]; }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) a_s[buf][ak * PA + am + 16 * i] = ra[i];
#pragma unroll
    for (int i = 0; i < NLB; ++i) b_s[buf][(bk0 + bks * i) * PB + bn] = rb[i];
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
    for (int s = 0; s < kGcKC / 2; ++s) {
      const int kra = (2 * s + hh) * PA, krb = (2 * s + hh) * PB;
      // (wide tile: wm = 0, the wave's two 32-row blocks are rows 0..31 and 32..63 of the one 64-row m tile)
      const float a0 = as[kra], a1 = as[kra + 32], b0 = bs[krb], b1 = bs[krb + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // ---- epilogue: D[row = acc_row(i, hh)][col = lane & 31]; consecutive lanes = consecutive pixels: coalesced
  const int hw = a.PH * a.PW;
  const bool slabs = a.nsplit > 1;
  float* ob = a.out + (slabs ? (int64_t)z * a.slab_stride : 0);
  const int64_t obs = slabs ? (int64_t)a.M * hw : a.o_bstride;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int nn = n0 + wn * 64 + nb * 32 + l31;
    if (nn >= a.N) continue;
    const int b = nn / hw, r = nn - b * hw;
    const int64_t obase = (int64_t)b * obs + r;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm * 64 + mb * 32 + acc_row(i, hh);
        if (m >= a.M) continue;
        float v = acc[mb][nb][i];
        float* o = ob + obase + (int64_t)m * hw;
        if (!slabs) {
          if (a.bias) v += a.bias[m];
          if (a.relu) v = v > 0.f ? v : 0.f;
          if (a.mask && !(a.mask[o - a.out] > 0.f)) v = 0.f;
          if (a.accumulate) v += *o;
        }
        *o = v;
      }
  }
}

// Data gradient with respect to an IMAGE (Cin <= 4: Conv2d_1a_3x3, 3 -> 32, 3x3 / stride 2): as a GEMM it is M = 3 rows of a 64-row
// tile (0.8 ms at 299^2 and batch 16).  Here a thread owns one input pixel and its CIN channels and walks the output channels: of
// the KH x KW taps only those whose source (q + pad - k) / stride is a whole output pixel contribute.  The filter w'T [CIN][Cout KK]
// sits in LDS.
template <int CIN>
__global__ __launch_bounds__(256) void gconv_image_dgrad_kernel(const float* __restrict__ wT, const float* __restrict__ g, float* __restrict__ dx,
                                                                int Cout, int Hs, int Ws, int PH, int PW, int KH, int KW, int SH, int padh,
                                                                int padw, int64_t g_bstride, int64_t o_bstride, int B, int accumulate) {
  extern __shared__ float w_s[];
  const int KK = KH * KW, nW = CIN * Cout * KK;
  for (int i = threadIdx.x; i < nW; i += 256) w_s[i] = wT[i];
  __syncthreads();
  const int64_t total = (int64_t)B * PH * PW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int px = (int)(e % PW), py = (int)((e / PW) % PH), b = (int)(e / ((int64_t)PW * PH));
    float acc[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
    const float* gb = g + (int64_t)b * g_bstride;
    for (int ky = 0; ky < KH; ++ky) {
      const int ty = py + padh - ky;
      if (ty < 0 || (SH == 2 && (ty & 1))) continue;
      const int oy = SH == 2 ? ty >> 1 : ty;
      if (oy >= Hs) continue;
      for (int kx = 0; kx < KW; ++kx) {
        const int tx = px + padw - kx;
        if (tx < 0 || (SH == 2 && (tx & 1))) continue;
        const int ox = SH == 2 ? tx >> 1 : tx;
        if (ox >= Ws) continue;
        const float* gp = gb + oy * Ws + ox;
        const int t = ky * KW + kx;
        for (int co = 0; co < Cout; ++co) {
          const float gv = gp[(int64_t)co * Hs * Ws];
#pragma unroll
          for (int c = 0; c < CIN; ++c) acc[c] = fmaf(gv, w_s[(c * Cout + co) * KK + t], acc[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      float* o = dx + (int64_t)b * o_bstride + ((int64_t)c * PH + py) * PW + px;
      *o = accumulate ? *o + acc[c] : acc[c];
    }
  }
}

// out[b][m][r] (+)= act(sum_z slab[z][b][m][r] + bias[m]), slabs dense [B][M][hw], the destination strided by o_bstride
__global__ __launch_bounds__(256) void gconv_finish_kernel(const float* __restrict__ slabs, int nsplit, int64_t slab_stride,
                                                           const float* __restrict__ bias, float* __restrict__ out, int M, int hw,
                                                           int64_t total, int64_t o_bstride, int relu, int accumulate,
                                                           const float* __restrict__ mask) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    float v = 0.f;
    for (int zz = 0; zz < nsplit; ++zz) v += slabs[(int64_t)zz * slab_stride + e];
    const int64_t mhw = (int64_t)M * hw;
    const int b = (int)(e / mhw);
    const int64_t r = e - (int64_t)b * mhw;
    const int m = (int)(r / hw);
    if (bias) v += bias[m];
    if (relu) v = v > 0.f ? v : 0.f;
    float* o = out + (int64_t)b * o_bstride + r;
    if (mask && !(mask[(int64_t)b * o_bstride + r] > 0.f)) v = 0.f;
    if (accumulate) v += *o;
    *o = v;
  }
}

// w [Cout][Cin][KK] * scale[co] -> fwd pack [Cout][Cin KK] (mode 0) or data-gradient pack [Cin][Cout KK] (mode 1)
__global__ void gconv_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out, int Cout,
                                  int Cin, int KK, int mode) {
  const int64_t total = (int64_t)Cout * Cin * KK;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e % KK);
    const int ci = (int)((e / KK) % Cin);
    const int co = (int)(e / ((int64_t)KK * Cin));
    const float v = w[e] * (scale ? scale[co] : 1.f);
    if (mode == 0) out[e] = v;
    else out[((int64_t)ci * Cout + co) * KK + t] = v;
  }
}

// ------------------------------------------------------------------------------------------------ pooling, mask, resize
// x [B][C][H][W] dense -> out based at a channel slice (batch stride o_bstride): 3x3 / stride 2 / no padding max pool
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W,
                                                         int OH, int OW, int64_t x_bstride, int64_t o_bstride, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int ox = (int)(e % OW), oy = (int)((e / OW) % OH), c = (int)((e / ((int64_t)OW * OH)) % C), b = (int)(e / ((int64_t)OW * OH * C));
    const float* p = x + (int64_t)b * x_bstride + ((int64_t)c * H + 2 * oy) * W + 2 * ox;
    float m = p[0];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float v = p[dy * W + dx];
        m = v > m ? v : m;
      }
    out[(int64_t)b * o_bstride + ((int64_t)c * OH + oy) * OW + ox] = m;
  }
}

// its backward in gather form: input pixel (iy, ix) collects dy of every window whose FIRST maximum (row-major scan, strict >:
// torch's rule) it is.  dx (+)= ...
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dx, int C, int H, int W, int OH, int OW,
                                                             int64_t x_bstride, int64_t dy_bstride, int64_t dx_bstride,
                                                             int64_t total, int accumulate, const float* __restrict__ mask) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int ix = (int)(e % W), iy = (int)((e / W) % H), c = (int)((e / ((int64_t)W * H)) % C), b = (int)(e / ((int64_t)W * H * C));
    const float* xp = x + (int64_t)b * x_bstride + (int64_t)c * H * W;
    const float* gp = dy + (int64_t)b * dy_bstride + (int64_t)c * OH * OW;
    float g = 0.f;
    const int oy_lo = iy >= 2 ? (iy - 1) >> 1 : 0, oy_hi = min(OH - 1, iy >> 1);
    const int ox_lo = ix >= 2 ? (ix - 1) >> 1 : 0, ox_hi = min(OW - 1, ix >> 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float* p = xp + (2 * oy) * W + 2 * ox;
        float m = p[0];
        int arg = 0;
#pragma unroll
        for (int t = 1; t < 9; ++t) {
          const float v = p[(t / 3) * W + t % 3];
          if (v > m) { m = v; arg = t; }
        }
        if (2 * oy + arg / 3 == iy && 2 * ox + arg % 3 == ix) g += gp[oy * OW + ox];
      }
    const int64_t oo = (int64_t)b * dx_bstride + ((int64_t)c * H + iy) * W + ix;
    if (mask && !(mask[oo] > 0.f)) g = 0.f;
    float* o = dx + oo;
    *o = accumulate ? *o + g : g;
  }
}

// 3x3 / stride 1 / padding 1 average pool with count_include_pad (F.avg_pool2d's default): sum of the in-image taps / 9.  The
// operator is symmetric, so the same kernel is its own backward (dx (+)= avgpool3(dy)).
__global__ __launch_bounds__(256) void avgpool3_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W,
                                                       int64_t x_bstride, int64_t o_bstride, int64_t total, int accumulate,
                                                       const float* __restrict__ mask) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int ix = (int)(e % W), iy = (int)((e / W) % H), c = (int)((e / ((int64_t)W * H)) % C), b = (int)(e / ((int64_t)W * H * C));
    const float* p = x + (int64_t)b * x_bstride + (int64_t)c * H * W;
    float s = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = iy + dy, xx = ix + dx;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) s += p[yy * W + xx];
      }
    s *= (1.f / 9.f);
    const int64_t oo = (int64_t)b * o_bstride + ((int64_t)c * H + iy) * W + ix;
    if (mask && !(mask[oo] > 0.f)) s = 0.f;
    float* o = out + oo;
    *o = accumulate ? *o + s : s;
  }
}

// dx[b][c][2Y + py][2X + px] (+)= t_{py px}[b][c][Y][X]: the four parity classes of a stride-2 convolution's data gradient (each one a
// stride-1 data gradient over its own taps, computed dense at half resolution) woven into the gradient of the H x W input; `mask`
// (shaped like dx): the value is kept where mask > 0.  Class (py, px) holds ceil((H - py) / 2) x ceil((W - px) / 2) pixels.
__global__ __launch_bounds__(256) void interleave2x2_kernel(const float* __restrict__ t00, const float* __restrict__ t01,
                                                            const float* __restrict__ t10, const float* __restrict__ t11,
                                                            float* __restrict__ dx, int H, int W, int64_t total, int accumulate,
                                                            const float* __restrict__ mask) {
  const int H0 = (H + 1) >> 1, H1 = H >> 1, W0 = (W + 1) >> 1, W1 = W >> 1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W), y = (int)((e / W) % H);
    const int64_t pl = e / ((int64_t)W * H);
    const int py = y & 1, px = x & 1, Y = y >> 1, X = x >> 1;
    const float* t = py ? (px ? t11 : t10) : (px ? t01 : t00);
    const int Hc = py ? H1 : H0, Wc = px ? W1 : W0;
    float v = t[(pl * Hc + Y) * Wc + X];
    if (mask && !(mask[e] > 0.f)) v = 0.f;
    dx[e] = accumulate ? dx[e] + v : v;
  }
}

// out[k] = ((parts[0][k] + parts[1][k]) + parts[2][k]) + ...  over n dense tensors of m floats stacked back to back: the contributions
// to a Mixed block's input gradient, each computed on its own branch's stream into its own slot, summed in the order a one-stream walk
// would have accumulated them (the same bits).  m % 4 == 0, 16-byte aligned.
__global__ __launch_bounds__(256) void sum_stack_kernel(const float* __restrict__ parts, int n, int64_t m4, float* __restrict__ out) {
  const float4* p = reinterpret_cast<const float4*>(parts);
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < m4; k += (int64_t)gridDim.x * 256) {
    float4 s = p[k];
    for (int z = 1; z < n; ++z) {
      const float4 v = p[(int64_t)z * m4 + k];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(out)[k] = s;
  }
}

// mean over the HW pixels of every (b, c) plane: one wave per plane (the 8 x 8 global average pool); backward: broadcast / HW
__global__ __launch_bounds__(256) void plane_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t planes, int HW) {
  const int64_t pl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < HW; i += 64) s += x[pl * HW + i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[pl] = s / (float)HW;
}

__global__ __launch_bounds__(256) void plane_mean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t total, int HW) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) dx[e] = dy[e / HW] / (float)HW;
}

// g = dy * (y > 0) over a channel slice of two equally shaped buffers (batch stride bs), written in place of dy or to `out`
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                                                        int64_t per_sample, int64_t dy_bstride, int64_t y_bstride, int64_t o_bstride,
                                                        int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / per_sample, r = e - b * per_sample;
    const float v = dy[b * dy_bstride + r];
    out[b * o_bstride + r] = y[b * y_bstride + r] > 0.f ? v : 0.f;
  }
}

// F.interpolate(x, size=(OH, OW), mode='bilinear', align_corners=False): src = (dst + 0.5) * (H / OH) - 0.5 clamped at 0,
// the two taps (i0, min(i0 + 1, H - 1)) with weights (1 - l, l)
__device__ __forceinline__ void bil_tap(int o, float scale, int n, int& i0, int& i1, float& l) {
  float s = ((float)o + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 > n - 1 ? n - 1 : i0;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t planes, int H, int W,
                                                       int OH, int OW, float sh, float sw, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int ox = (int)(e % OW), oy = (int)((e / OW) % OH);
    const int64_t pl = e / ((int64_t)OW * OH);
    int y0, y1, x0, x1;
    float ly, lx;
    bil_tap(oy, sh, H, y0, y1, ly);
    bil_tap(ox, sw, W, x0, x1, lx);
    const float* p = x + pl * H * W;
    out[e] = (1.f - ly) * ((1.f - lx) * p[y0 * W + x0] + lx * p[y0 * W + x1]) + ly * ((1.f - lx) * p[y1 * W + x0] + lx * p[y1 * W + x1]);
  }
}

// its backward in gather form: source pixel (iy, ix) sums the weights of every destination pixel that taps it.  Destination rows
// tapping source row iy lie in a small range around iy / sh: scanned, deterministic.
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t planes, int H,
                                                           int W, int OH, int OW, float sh, float sw, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int ix = (int)(e % W), iy = (int)((e / W) % H);
    const int64_t pl = e / ((int64_t)W * H);
    const float* g = dy + pl * OH * OW;
    const float rh = 1.f / sh, rw = 1.f / sw;
    const int oy_lo = max(0, (int)(((float)iy - 1.f + 0.5f) * rh - 0.5f) - 1), oy_hi = min(OH - 1, (int)(((float)iy + 1.f + 0.5f) * rh - 0.5f) + 2);
    const int ox_lo = max(0, (int)(((float)ix - 1.f + 0.5f) * rw - 0.5f) - 1), ox_hi = min(OW - 1, (int)(((float)ix + 1.f + 0.5f) * rw - 0.5f) + 2);
    float s = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly;
      bil_tap(oy, sh, H, y0, y1, ly);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        bil_tap(ox, sw, W, x0, x1, lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx != 0.f) row += wx * g[oy * OW + ox];
      }
      s += wy * row;
    }
    dx[e] = s;
  }
}

// tgsr_down.hip: the same GEMMs on the bf16 matrix pipe with exact three-piece fp32 operands (dconv_igemm6_kernel, generic-tap modes)
int ig6_gconv_launch(int dgrad, const float* A, const float* S, int64_t s_bstride, int64_t s_bytes, int B, int Hs, int Ws, int M, int K,
                     int PH, int PW, int KH, int KW, int stride, int padh, int padw, const float* bias, int relu, int accumulate,
                     const float* mask, float* out, int64_t o_bstride, float* slabs, int nsplit, int chunks_per_split, hipStream_t s);

static int g_gconv_form = [] {              // TGSR_GCONV_SPLIT=0: the fp32-MFMA kernel of this file everywhere
  const char* e = getenv("TGSR_GCONV_SPLIT");
  return e ? atoi(e) : 1;
}();

static int gc_grid(int64_t total, int cap = 8192) {
  const int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace tgsr

using namespace tgsr;

// 1 (default): tgsr_gconv runs on the bf16 matrix pipe with exact three-piece operands where the shape qualifies; 0: fp32 MFMA
// everywhere.  Returns the previous setting.  (TGSR_GCONV_SPLIT sets the initial value.)
extern "C" int tgsr_gconv_set_form(int split) {
  const int was = g_gconv_form;
  g_gconv_form = split ? 1 : 0;
  return was;
}

// Workgroups a launch should reach before the reduction is split no further (TGSR_GCONV_FILL; default 224 = a little under one per
// CU: the trunk's branches run side by side on streams of their own, inception.py, so a launch need not fill the chip alone.
// Measured on the G/D + DAMSM step: 96 / 160 / 320 / 480 / 640 -> 34.8 / 31.8 / 31.2 / 31.0 / 31.1 ms with the branch heads behind the
// join; with the heads in slots 224 / 320 / 448 -> 29.19, 29.04 / 29.28, 29.24 / 29.31).
static int g_gconv_fill = [] {
  const char* e = getenv("TGSR_GCONV_FILL");
  const int v = e ? atoi(e) : 224;
  return v < 1 ? 224 : v;
}();

// How many K slabs a shape is split into (1 = none): fill ~g_gconv_fill workgroups when M x N alone cannot.
extern "C" int tgsr_gconv_nsplit(int M, int N, int K) {
  const bool wide = M <= 64;
  const int64_t tiles = (int64_t)((M + (wide ? 63 : 127)) / (wide ? 64 : 128)) * ((N + (wide ? 255 : 127)) / (wide ? 256 : 128));
  const int chunks = (K + kGcKC - 1) / kGcKC;
  if (tiles >= 192 || tiles * 5 >= g_gconv_fill * 3 || chunks < 16) return 1;   // (every split costs a finishing launch: ~7 us each)
  int64_t s = (g_gconv_fill + tiles - 1) / tiles;
  if (s > chunks / 8) s = chunks / 8;                      // at least 8 chunks (128 k) per slab
  if (s > 32) s = 32;
  return (int)(s < 1 ? 1 : s);
}

extern "C" int64_t tgsr_gconv_ws_elems(int B, int M, int PH, int PW, int K) {
  const int ns = tgsr_gconv_nsplit(M, B * PH * PW, K);
  return ns > 1 ? (int64_t)ns * B * M * PH * PW : 0;
}

// dgrad = 0: forward (S = x [.., Cin = K / (KH KW), Hs, Ws], pixel grid = output PH x PW); 1: data gradient (S = g [.., Cout, Hs, Ws]
// = the forward's OUTPUT grid, pixel grid = the forward's input PH x PW).  A = the matching pack of tgsr_gconv_pack.
extern "C" int tgsr_gconv(int dgrad, const float* A, const float* S, int64_t s_bstride, int B, int Hs, int Ws, int M, int K, int PH,
                          int PW, int KH, int KW, int stride, int padh, int padw, const float* bias, int relu, int accumulate,
                          const float* mask, float* out, int64_t o_bstride, float* ws, void* stream) {
  if (!A || !S || !out || B < 1 || M < 1 || K < 1 || Hs < 1 || Ws < 1 || PH < 1 || PW < 1) return TGSR_EINVAL;
  if (KH < 1 || KW < 1 || KH * KW > 64 || KH > 255 || KW > 255 || (stride != 1 && stride != 2) || padh < 0 || padw < 0) return TGSR_EUNSUPPORTED;
  if (K % (KH * KW)) return TGSR_EINVAL;
  if (dgrad && (bias || relu)) return TGSR_EINVAL;
  const int64_t N64 = (int64_t)B * PH * PW;
  if (N64 >= (1ll << 31) || (int64_t)M * K >= (1ll << 31)) return TGSR_EUNSUPPORTED;
  const int N = (int)N64;
  GcArgs a;
  a.A = A; a.S = S; a.bias = bias; a.out = out;
  a.M = M; a.N = N; a.K = K; a.Hs = Hs; a.Ws = Ws; a.PH = PH; a.PW = PW;
  a.s_bstride = s_bstride; a.o_bstride = o_bstride;
  a.KH = KH; a.KW = KW; a.SH = stride; a.PADH = padh; a.PADW = padw;
  a.relu = relu; a.accumulate = accumulate; a.mask = mask;
  if (dgrad && M <= 4 && !mask && (int64_t)M * K * 4 <= 48 * 1024) {
    // the gradient with respect to an image: one thread per pixel (gconv_image_dgrad_kernel)
    const size_t lds = (size_t)M * K * sizeof(float);
    const int Cout = K / (KH * KW);
    const dim3 g1(gc_grid(N64, 16384));
    hipStream_t s1 = as_stream(stream);
    switch (M) {
      case 1: hipLaunchKernelGGL(gconv_image_dgrad_kernel<1>, g1, dim3(256), lds, s1, A, S, out, Cout, Hs, Ws, PH, PW, KH, KW, stride, padh, padw, s_bstride, o_bstride, B, accumulate); break;
      case 2: hipLaunchKernelGGL(gconv_image_dgrad_kernel<2>, g1, dim3(256), lds, s1, A, S, out, Cout, Hs, Ws, PH, PW, KH, KW, stride, padh, padw, s_bstride, o_bstride, B, accumulate); break;
      case 3: hipLaunchKernelGGL(gconv_image_dgrad_kernel<3>, g1, dim3(256), lds, s1, A, S, out, Cout, Hs, Ws, PH, PW, KH, KW, stride, padh, padw, s_bstride, o_bstride, B, accumulate); break;
      default: hipLaunchKernelGGL(gconv_image_dgrad_kernel<4>, g1, dim3(256), lds, s1, A, S, out, Cout, Hs, Ws, PH, PW, KH, KW, stride, padh, padw, s_bstride, o_bstride, B, accumulate); break;
    }
    return note_launch(hipGetLastError(), "gconv_image_dgrad_kernel");
  }
  const int chunks = (K + kGcKC - 1) / kGcKC;
  int ns = tgsr_gconv_nsplit(M, N, K);
  if (ns > 1 && !ws) return TGSR_EINVAL;
  a.chunks_per_split = (chunks + ns - 1) / ns;
  ns = (chunks + a.chunks_per_split - 1) / a.chunks_per_split;
  a.nsplit = ns;
  a.slab_stride = (int64_t)B * M * PH * PW;
  if (ns > 1) a.out = ws;
  hipStream_t s = as_stream(stream);
  const bool wide = M <= 64;
  const dim3 grid((N + (wide ? 255 : 127)) / (wide ? 256 : 128), (M + (wide ? 63 : 127)) / (wide ? 64 : 128), ns);
  // the three-piece bf16 form where the shape qualifies (K % 16 == 0, <= 25 taps, stride-1 data gradient): ~2.4x the fp32 MFMA's rate
  int rc6 = TGSR_EUNSUPPORTED;
  if (g_gconv_form) {
    const int64_t s_bytes = ((int64_t)(B - 1) * s_bstride + (int64_t)(K / (KH * KW)) * Hs * Ws) * 4;
    rc6 = ig6_gconv_launch(dgrad, A, S, s_bstride, s_bytes, B, Hs, Ws, M, K, PH, PW, KH, KW, stride, padh, padw, bias, relu, accumulate,
                           mask, out, o_bstride, ws, ns, a.chunks_per_split, s);
    if (rc6 != TGSR_OK && rc6 != TGSR_EUNSUPPORTED) return rc6;
  }
  if (rc6 == TGSR_OK) {
    // (falls through to the slab finish below)
  } else if (dgrad) {
    if (wide) hipLaunchKernelGGL((gconv_igemm_kernel<true, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gconv_igemm_kernel<true, false>), grid, dim3(256), 0, s, a);
  } else {
    if (wide) hipLaunchKernelGGL((gconv_igemm_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gconv_igemm_kernel<false, false>), grid, dim3(256), 0, s, a);
  }
  int rc = note_launch(hipGetLastError(), "gconv_igemm_kernel");
  if (rc || ns == 1) return rc;
  const int64_t total = a.slab_stride;
  hipLaunchKernelGGL(gconv_finish_kernel, dim3(gc_grid(total)), dim3(256), 0, s, ws, ns, a.slab_stride, bias, out, M, PH * PW, total,
                     o_bstride, relu, accumulate, mask);
  return note_launch(hipGetLastError(), "gconv_finish_kernel");
}

extern "C" int tgsr_gconv_pack(const float* w, const float* scale, float* out, int Cout, int Cin, int KK, int dgrad, void* stream) {
  if (!w || !out || Cout < 1 || Cin < 1 || KK < 1) return TGSR_EINVAL;
  const int64_t total = (int64_t)Cout * Cin * KK;
  hipLaunchKernelGGL(gconv_pack_kernel, dim3(gc_grid(total, 2048)), dim3(256), 0, as_stream(stream), w, scale, out, Cout, Cin, KK,
                     dgrad ? 1 : 0);
  return note_launch(hipGetLastError(), "gconv_pack_kernel");
}

extern "C" int tgsr_maxpool3s2_fwd(const float* x, int64_t x_bstride, int B, int C, int H, int W, float* out, int64_t o_bstride,
                                   void* stream) {
  if (!x || !out || B < 1 || C < 1 || H < 3 || W < 3) return TGSR_EINVAL;
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  const int64_t total = (int64_t)B * C * OH * OW;
  hipLaunchKernelGGL(maxpool3s2_kernel, dim3(gc_grid(total)), dim3(256), 0, as_stream(stream), x, out, C, H, W, OH, OW, x_bstride,
                     o_bstride, total);
  return note_launch(hipGetLastError(), "maxpool3s2_kernel");
}

extern "C" int tgsr_maxpool3s2_bwd(const float* x, int64_t x_bstride, const float* dy, int64_t dy_bstride, int B, int C, int H, int W,
                                   float* dx, int64_t dx_bstride, int accumulate, const float* mask, void* stream) {
  if (!x || !dy || !dx || B < 1 || C < 1 || H < 3 || W < 3) return TGSR_EINVAL;
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  const int64_t total = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(gc_grid(total)), dim3(256), 0, as_stream(stream), x, dy, dx, C, H, W, OH, OW, x_bstride,
                     dy_bstride, dx_bstride, total, accumulate, mask);
  return note_launch(hipGetLastError(), "maxpool3s2_bwd_kernel");
}

extern "C" int tgsr_avgpool3(const float* x, int64_t x_bstride, int B, int C, int H, int W, float* out, int64_t o_bstride,
                             int accumulate, const float* mask, void* stream) {
  if (!x || !out || B < 1 || C < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  const int64_t total = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(avgpool3_kernel, dim3(gc_grid(total)), dim3(256), 0, as_stream(stream), x, out, C, H, W, x_bstride, o_bstride, total,
                     accumulate, mask);
  return note_launch(hipGetLastError(), "avgpool3_kernel");
}

extern "C" int tgsr_interleave2x2(const float* t00, const float* t01, const float* t10, const float* t11, float* dx, int64_t planes,
                                  int H, int W, int accumulate, const float* mask, void* stream) {
  if (!t00 || !dx || planes < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if ((W > 1 && !t01) || (H > 1 && !t10) || (H > 1 && W > 1 && !t11)) return TGSR_EINVAL;
  const int64_t total = planes * H * W;
  hipLaunchKernelGGL(interleave2x2_kernel, dim3(gc_grid(total)), dim3(256), 0, as_stream(stream), t00, t01, t10, t11, dx, H, W, total,
                     accumulate, mask);
  return note_launch(hipGetLastError(), "interleave2x2_kernel");
}

extern "C" int tgsr_sum_stack(const float* parts, int n, int64_t m, float* out, void* stream) {
  if (!parts || !out || n < 1 || m < 1) return TGSR_EINVAL;
  if ((m & 3) || ((reinterpret_cast<uintptr_t>(parts) | reinterpret_cast<uintptr_t>(out)) & 15)) return TGSR_EUNSUPPORTED;
  hipLaunchKernelGGL(sum_stack_kernel, dim3(gc_grid(m >> 2)), dim3(256), 0, as_stream(stream), parts, n, m >> 2, out);
  return note_launch(hipGetLastError(), "sum_stack_kernel");
}

extern "C" int tgsr_plane_mean(const float* x, float* out, int64_t planes, int HW, void* stream) {
  if (!x || !out || planes < 1 || HW < 1) return TGSR_EINVAL;
  hipLaunchKernelGGL(plane_mean_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, as_stream(stream), x, out, planes, HW);
  return note_launch(hipGetLastError(), "plane_mean_kernel");
}

extern "C" int tgsr_plane_mean_bwd(const float* dy, float* dx, int64_t planes, int HW, void* stream) {
  if (!dy || !dx || planes < 1 || HW < 1) return TGSR_EINVAL;
  const int64_t total = planes * HW;
  hipLaunchKernelGGL(plane_mean_bwd_kernel, dim3(gc_grid(total)), dim3(256), 0, as_stream(stream), dy, dx, total, HW);
  return note_launch(hipGetLastError(), "plane_mean_bwd_kernel");
}

extern "C" int tgsr_relu_mask(const float* dy, int64_t dy_bstride, const float* y, int64_t y_bstride, float* out, int64_t o_bstride,
                              int B, int64_t per_sample, void* stream) {
  if (!dy || !y || !out || B < 1 || per_sample < 1) return TGSR_EINVAL;
  const int64_t total = (int64_t)B * per_sample;
  hipLaunchKernelGGL(relu_mask_kernel, dim3(gc_grid(total)), dim3(256), 0, as_stream(stream), dy, y, out, per_sample, dy_bstride,
                     y_bstride, o_bstride, total);
  return note_launch(hipGetLastError(), "relu_mask_kernel");
}

extern "C" int tgsr_bilinear_fwd(const float* x, int64_t planes, int H, int W, int OH, int OW, float* out, void* stream) {
  if (!x || !out || planes < 1 || H < 1 || W < 1 || OH < 1 || OW < 1) return TGSR_EINVAL;
  const int64_t total = planes * OH * OW;
  hipLaunchKernelGGL(bilinear_kernel, dim3(gc_grid(total)), dim3(256), 0, as_stream(stream), x, out, planes, H, W, OH, OW,
                     (float)H / (float)OH, (float)W / (float)OW, total);
  return note_launch(hipGetLastError(), "bilinear_kernel");
}

extern "C" int tgsr_bilinear_bwd(const float* dy, int64_t planes, int H, int W, int OH, int OW, float* dx, void* stream) {
  if (!dy || !dx || planes < 1 || H < 1 || W < 1 || OH < 1 || OW < 1) return TGSR_EINVAL;
  const int64_t total = planes * H * W;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(gc_grid(total)), dim3(256), 0, as_stream(stream), dy, dx, planes, H, W, OH, OW,
                     (float)H / (float)OH, (float)W / (float)OW, total);
  return note_launch(hipGetLastError(), "bilinear_bwd_kernel");
}
