// downBlock's strided convolution for the discriminators (util.py:92-98: Conv2d(Cin, Cout, 4, 2, 1, bias=False) ->
// BatchNorm2d -> LeakyReLU(0.2)): forward, data gradient and weight gradient as fp32 MFMA (32x32x2) implicit GEMMs that
// gather their "B" operand straight from the NCHW tensors - no im2col buffer.  BatchNorm (batch statistics) + LeakyReLU
// is tgsr_bn_train_fwd / _bwd with act = 2 (tgsr_bn.hip); the first discriminator layer (no BatchNorm) takes the
// LeakyReLU in this kernel's epilogue.  SURVEY.md 8(f)1; the discriminator ARCHITECTURE is the build's (AttnGAN-style,
// no class exists in the reference), these kernels are pinned against a torch fp32 restatement in the oracle.
//
//   forward : out[b][co][oy][ox] = sum_{ci,ky,kx} w[co][ci][ky][kx] x[b][ci][2oy+ky-1][2ox+kx-1]
//             GEMM M = Cout, K = 16 Cin, N = B Ho Wo; A = w (already [Cout][16 Cin] row-major), B gathered.
//   dgrad   : dx[b][ci][iy][ix] = sum over the (co, ky, kx) that reach it.  Input pixels of parity class (p, q) =
//             ((iy+1)&1, (ix+1)&1) only see taps ky in {p, p+2}, kx in {q, q+2}: four GEMMs M = Cin, K = 4 Cout,
//             N = B (H/2)(W/2), A = the class's re-grouped weights (tgsr_conv4x4s2_pack_dgrad), B gathered from dy.
//   wgrad   : dw[co][ci][ky][kx] = sum_n dy[co][n] x_gather[k][n]: reduction over N = B Ho Wo split over blockIdx.z
//             into slabs that a second kernel sums in a fixed order (bitwise reproducible, no float atomics).
#include "tgsr_common.h"

namespace tgsr {

struct DownArgs {
  const float* A;        // forward: w [Cout][16 Cin]; dgrad: packed [4][Cin][4 Cout]
  const float* X;        // forward: x [B][Cin][H][W]; dgrad: dy [B][Cout][Ho][Wo]
  float* out;
  int B, Cin, H, W, Cout, Ho, Wo;
  int act;               // forward only: 1 = LeakyReLU(0.2) epilogue
};

// C[m][n] = sum_k A[m][k] G(k, n): workgroup = 32 m x 128 n (wave = 32 n), A chunk [32][64] through LDS (pitch 65,
// conflict free for the lane = m reads), G gathered per lane straight from global memory.
template <int MODE>   // 0 = forward, 1 = dgrad (blockIdx.z = parity class)
__global__ __launch_bounds__(256) void conv4x4s2_gemm_kernel(DownArgs a) {
  constexpr int KC = 64, P = KC + 1;
  __shared__ float a_s[32 * P];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5, wave = tid >> 6;
  const int cls = MODE == 1 ? blockIdx.z : 0, pp = cls >> 1, qq = cls & 1;
  const int M = MODE == 0 ? a.Cout : a.Cin;
  const int K = MODE == 0 ? a.Cin * 16 : a.Cout * 4;
  const int nH = MODE == 0 ? a.Ho : a.H / 2, nW = MODE == 0 ? a.Wo : a.W / 2;
  const int64_t N = (int64_t)a.B * nH * nW;
  const float* Ab = a.A + (MODE == 1 ? (int64_t)cls * M * K : 0);
  const int m0 = blockIdx.y * 32;
  const int64_t n = ((int64_t)blockIdx.x * 4 + wave) * 32 + l31;
  const bool nok = n < N;
  int nb = 0, ny = 0, nx = 0;
  if (nok) {
    nx = (int)(n % nW);
    const int64_t t = n / nW;
    ny = (int)(t % nH);
    nb = (int)(t / nH);
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();
    for (int idx = tid; idx < 32 * KC; idx += 256) {
      const int r = idx >> 6, k = idx & 63;
      a_s[r * P + k] = (m0 + r < M && k0 + k < K) ? Ab[(int64_t)(m0 + r) * K + k0 + k] : 0.f;
    }
    __syncthreads();
    float bv[KC / 2];
#pragma unroll
    for (int j = 0; j < KC / 2; ++j) {
      const int k = k0 + 2 * j + hh;
      float v = 0.f;
      if (nok && k < K) {
        if (MODE == 0) {
          const int ci = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
          const int iy = 2 * ny + ky - 1, ix = 2 * nx + kx - 1;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
            v = a.X[(((int64_t)nb * a.Cin + ci) * a.H + iy) * a.W + ix];
        } else {
          const int co = k >> 2, ta = (k >> 1) & 1, tc = k & 1;
          const int oy = ny + 1 - pp - ta, ox = nx + 1 - qq - tc;
          if ((unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo)
            v = a.X[(((int64_t)nb * a.Cout + co) * a.Ho + oy) * a.Wo + ox];
        }
      }
      bv[j] = v;
    }
#pragma unroll
    for (int j = 0; j < KC / 2; ++j)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s[l31 * P + 2 * j + hh], bv[j], acc, 0, 0, 0);
  }
  if (nok) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + acc_row(i, hh);
      if (m >= M) continue;
      float v = acc[i];
      if (MODE == 0) {
        if (a.act) v = v > 0.f ? v : 0.2f * v;
        a.out[(((int64_t)nb * a.Cout + m) * a.Ho + ny) * a.Wo + nx] = v;
      } else {
        a.out[(((int64_t)nb * a.Cin + m) * a.H + 2 * ny + 1 - pp) * a.W + 2 * nx + 1 - qq] = v;
      }
    }
  }
}

// packed[cls = 2p + q][ci][co * 4 + 2a + c] = w[co][ci][p + 2a][q + 2c]
__global__ void conv4x4s2_pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                            int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % (4 * Cout));
    int64_t t = i / (4 * Cout);
    const int ci = (int)(t % Cin), cls = (int)(t / Cin);
    const int co = k >> 2, ta = (k >> 1) & 1, tc = k & 1, pp = cls >> 1, qq = cls & 1;
    out[i] = w[(((int64_t)co * Cin + ci) * 4 + pp + 2 * ta) * 4 + qq + 2 * tc];
  }
}

struct DownWgradArgs {
  const float* dy;       // [B][Cout][Ho][Wo]
  const float* x;        // [B][Cin][H][W]
  float* ws;             // [nsplit][Cout][16 Cin]
  int B, Cin, H, W, Cout, Ho, Wo, nsplit;
};

// slab[z][co][k] = sum over the n range of split z of dy[co][n] * x_gather[k][n]: workgroup = 32 co x 128 k
// (wave = 32 k); both operands staged through LDS in [row][64 n] tiles (n contiguous in dy -> coalesced).
__global__ __launch_bounds__(256) void conv4x4s2_wgrad_kernel(DownWgradArgs a) {
  constexpr int RC = 64, P = RC + 1;
  __shared__ float a_s[32 * P];
  __shared__ float b_s[4 * 32 * P];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5, wave = tid >> 6;
  const int K = a.Cin * 16, HW = a.Ho * a.Wo;
  const int64_t N = (int64_t)a.B * HW;
  const int m0 = blockIdx.y * 32, kb = (blockIdx.x * 4 + wave) * 32;
  const int64_t per = ((N + a.nsplit - 1) / a.nsplit + RC - 1) / RC * RC;
  const int64_t lo = (int64_t)blockIdx.z * per, hi = lo + per < N ? lo + per : N;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float* mine = b_s + wave * 32 * P;
  for (int64_t r0 = lo; r0 < hi; r0 += RC) {
    __syncthreads();
    for (int idx = tid; idx < 32 * RC; idx += 256) {
      const int row = idx >> 6, rr = idx & 63;
      const int64_t n = r0 + rr;
      float v = 0.f;
      if (m0 + row < a.Cout && n < hi) {
        const int b = (int)(n / HW), pix = (int)(n - (int64_t)b * HW);
        v = a.dy[((int64_t)b * a.Cout + m0 + row) * HW + pix];
      }
      a_s[row * P + rr] = v;
    }
    for (int idx = lane; idx < 32 * RC; idx += 64) {
      const int row = idx >> 6, rr = idx & 63;
      const int64_t n = r0 + rr;
      const int k = kb + row;
      float v = 0.f;
      if (k < K && n < hi) {
        const int b = (int)(n / HW), pix = (int)(n - (int64_t)b * HW);
        const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
        const int ci = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
        const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          v = a.x[(((int64_t)b * a.Cin + ci) * a.H + iy) * a.W + ix];
      }
      mine[row * P + rr] = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < RC; r += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s[l31 * P + r + hh], mine[l31 * P + r + hh], acc, 0, 0, 0);
  }
  const int k = kb + l31;
  if (k < K) {
    float* slab = a.ws + (int64_t)blockIdx.z * a.Cout * K;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + acc_row(i, hh);
      if (m < a.Cout) slab[(int64_t)m * K + k] = acc[i];
    }
  }
}

__global__ void slab_sum_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n, int nsplit) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += ws[(int64_t)z * n + i];
    out[i] = s;
  }
}

// y = x > 0 ? x : 0.2 x (forward)  |  dx = dy * (y > 0 ? 1 : 0.2) (backward, from the OUTPUT y: same sign as x)
__global__ void leaky_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = g[i];
    out[i] = y ? (y[i] > 0.f ? v : 0.2f * v) : (v > 0.f ? v : 0.2f * v);
  }
}

static int down_nsplit(int B, int Cin, int Cout, int Ho, int Wo) {
  const int64_t tiles = (int64_t)((Cout + 31) / 32) * ((Cin * 16 + 127) / 128);
  const int64_t N = (int64_t)B * Ho * Wo;
  int64_t s = (1024 + tiles - 1) / tiles;          // aim at >= 1024 workgroups
  const int64_t cap = (N + 255) / 256;             // at least 256 reduction elements per split
  if (s > cap) s = cap;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_conv4x4s2_fwd(const float* x, int B, int Cin, int H, int W, const float* w, int Cout, int act,
                                  float* out, void* stream) {
  if (!x || !w || !out || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return TGSR_EINVAL;
  if ((H | W) & 1) return TGSR_EUNSUPPORTED;
  DownArgs a;
  a.A = w; a.X = x; a.out = out; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Ho = H / 2; a.Wo = W / 2;
  a.act = act ? 1 : 0;
  const int64_t N = (int64_t)B * a.Ho * a.Wo;
  const dim3 grid((unsigned)((N + 127) / 128), (unsigned)((Cout + 31) / 32), 1);
  hipLaunchKernelGGL(conv4x4s2_gemm_kernel<0>, grid, dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "conv4x4s2_gemm_kernel<fwd>");
}

extern "C" int tgsr_conv4x4s2_dgrad(const float* dy, int B, int Cin, int H, int W, const float* w, int Cout,
                                    float* wpack_ws, float* dx, void* stream) {
  if (!dy || !w || !wpack_ws || !dx || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return TGSR_EINVAL;
  if ((H | W) & 1) return TGSR_EUNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const int64_t total = (int64_t)16 * Cin * Cout;
  const int pb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(conv4x4s2_pack_dgrad_kernel, dim3(pb), dim3(256), 0, s, w, wpack_ws, Cout, Cin, total);
  DownArgs a;
  a.A = wpack_ws; a.X = dy; a.out = dx; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Ho = H / 2;
  a.Wo = W / 2; a.act = 0;
  const int64_t N = (int64_t)B * (H / 2) * (W / 2);
  const dim3 grid((unsigned)((N + 127) / 128), (unsigned)((Cin + 31) / 32), 4);
  hipLaunchKernelGGL(conv4x4s2_gemm_kernel<1>, grid, dim3(256), 0, s, a);
  return note_launch(hipGetLastError(), "conv4x4s2_gemm_kernel<dgrad>");
}

extern "C" int64_t tgsr_conv4x4s2_wgrad_ws_elems(int B, int Cin, int Cout, int H, int W) {
  return (int64_t)down_nsplit(B, Cin, Cout, H / 2, W / 2) * Cout * Cin * 16;
}

extern "C" int tgsr_conv4x4s2_wgrad(const float* dy, const float* x, int B, int Cin, int H, int W, int Cout, float* ws,
                                    float* dw, void* stream) {
  if (!dy || !x || !ws || !dw || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return TGSR_EINVAL;
  if ((H | W) & 1) return TGSR_EUNSUPPORTED;
  hipStream_t s = as_stream(stream);
  DownWgradArgs a;
  a.dy = dy; a.x = x; a.ws = ws; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Ho = H / 2; a.Wo = W / 2;
  a.nsplit = down_nsplit(B, Cin, Cout, a.Ho, a.Wo);
  const dim3 grid((unsigned)((Cin * 16 + 127) / 128), (unsigned)((Cout + 31) / 32), (unsigned)a.nsplit);
  hipLaunchKernelGGL(conv4x4s2_wgrad_kernel, grid, dim3(256), 0, s, a);
  const int64_t n = (int64_t)Cout * Cin * 16;
  const int rb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(slab_sum_kernel, dim3(rb), dim3(256), 0, s, ws, dw, n, a.nsplit);
  return note_launch(hipGetLastError(), "conv4x4s2_wgrad_kernel");
}

extern "C" int tgsr_leaky_relu(const float* x, const float* y_for_bwd, float* out, int64_t n, void* stream) {
  if (!x || !out || n < 1) return TGSR_EINVAL;
  const int b = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(leaky_kernel, dim3(b), dim3(256), 0, as_stream(stream), x, y_for_bwd, out, n);
  return note_launch(hipGetLastError(), "leaky_kernel");
}
