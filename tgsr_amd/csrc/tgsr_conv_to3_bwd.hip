// Backward of the image heads (tgsr_conv_to3_fwd): KxK conv to 3 channels [+ tanh, + alpha * addend].
//   g        = dy * (1 - t^2) with t = out - alpha * addend   (tanh heads; plain heads: g = dy)
//   dx[ci]   = sum_{co,ky,kx} w[co][ci][ky][kx] * g[co][y + P - ky][x + P - kx]          (3 -> Cin, VALU: K = 75)
//   dw[co][ci][ky][kx] = sum_{b,y,x} g[co][y][x] * x[ci][y + ky - P][x + kx - P]         (reduction over ~1M pixels)
//   d(addend) = alpha * dy (done by the caller: one scalar multiply)
// Both kernels tile 16 x 64 pixels like the forward.  dgrad: thread = 4 pixels x 16 input channels per pass, g tile
// (3 channels + halo) in LDS, weights as scalar loads.  wgrad: thread = (channel, ky, group of 4 rows) with a sliding
// 5-wide window along x, 15 accumulators (3 co x K kx); partial slabs per workgroup, summed in a fixed order.
#include "tgsr_common.h"

namespace tgsr {

struct To3BwdArgs {
  const float* dy;      // [B][3][H][W]
  const float* out;     // forward output (tanh heads) or null
  const float* addend;  // or null
  float alpha;
  const float* x;       // [B][Cin][H][W] (wgrad)
  int64_t xbs;
  const float* w;       // [3][Cin][K][K] (dgrad)
  int B, Cin, H, W, tiles_x, tiles_y;
  float* dx;            // [B][Cin][H][W]
  float* part;          // [nwg][3][Cin][K][K]
};

template <int K, bool TANH>
__device__ __forceinline__ void stage_g(const To3BwdArgs& a, float* g_s, int b, int y0, int x0, int halo) {
  // g_s[co][r][j]: row r <-> y0 - halo + r, column j <-> x0 - 4 + j; pitch 72
  const int TR = 16 + 2 * halo;
  const int64_t HW = (int64_t)a.H * a.W;
  for (int idx = threadIdx.x; idx < 3 * TR * 72; idx += 256) {
    const int co = idx / (TR * 72);
    const int rem = idx - co * (TR * 72);
    const int r = rem / 72, j = rem - r * 72;
    const int gy = y0 - halo + r, gx = x0 - 4 + j;
    float v = 0.f;
    if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
      const int64_t o = ((int64_t)b * 3 + co) * HW + (int64_t)gy * a.W + gx;
      v = a.dy[o];
      if (TANH) {
        const float t = a.out[o] - (a.addend ? a.alpha * a.addend[o] : 0.f);
        v *= 1.f - t * t;
      }
    }
    g_s[idx] = v;
  }
}

template <int K, bool TANH>
__global__ __launch_bounds__(256) void conv_to3_dgrad_kernel(To3BwdArgs a) {
  constexpr int P = K / 2, TR = 16 + 2 * P, MAXC = 64;
  __shared__ __attribute__((aligned(16))) float g_s[3 * TR * 72];
  __shared__ float w_s[3 * MAXC * K * K];
  const int tid = threadIdx.x, txi = tid & 15, tyi = tid >> 4;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * 16, x0 = tx * 64;
  stage_g<K, TANH>(a, g_s, b, y0, x0, P);
  for (int i = tid; i < 3 * a.Cin * K * K; i += 256) w_s[i] = a.w[i];
  __syncthreads();
  const int64_t HW = (int64_t)a.H * a.W;
  const int y = y0 + tyi, xx = x0 + 4 * txi;
  for (int c0 = 0; c0 < a.Cin; c0 += 16) {
    float acc[16][4];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[c][p] = 0.f;
#pragma unroll 1
    for (int ck = 0; ck < 3 * K; ++ck) {     // not unrolled: keeps at most 16 x K weights live (no spills)
      const int co = ck / K, ky = ck - co * K;
      const float* row = g_s + (co * TR + tyi + (K - 1 - ky)) * 72 + 4 * txi;
      const float4 v0 = *reinterpret_cast<const float4*>(row);
      const float4 v1 = *reinterpret_cast<const float4*>(row + 4);
      const float4 v2 = *reinterpret_cast<const float4*>(row + 8);
      const float in[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
      const float* wrow = w_s + ((co * a.Cin + c0) * K + ky) * K;   // + c*K*K + kx  (wave-uniform: LDS broadcast)
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c0 + c < a.Cin) {
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const float wv = wrow[c * K * K + kx];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = fmaf(wv, in[4 + p + P - kx], acc[c][p]);
          }
        }
      }
    }
    if (y < a.H) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c0 + c < a.Cin) {
          float* o = a.dx + ((int64_t)b * a.Cin + c0 + c) * HW + (int64_t)y * a.W + xx;
          if (xx + 3 < a.W && (a.W & 3) == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
          } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
              if (xx + p < a.W) o[p] = acc[c][p];
          }
        }
      }
    }
  }
}

// thread = (ci local 0..7, ky 0..K-1, row group 0..3); 8*K*4 <= 160 threads active
template <int K, bool TANH>
__global__ __launch_bounds__(256) void conv_to3_wgrad_kernel(To3BwdArgs a) {
  constexpr int P = K / 2, CK = 8, TR = 16 + K - 1, XP = 73;   // x tile: [8][TR][64 + K - 1 (+pad)], pitch 73
  __shared__ float g_s[3 * 16 * 72];
  __shared__ float x_s[CK * TR * XP];
  __shared__ float red_s[3 * 32 * K * K];                       // [co][ci 0..31][ky][kx] accumulated over chunks/groups
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * 16, x0 = tx * 64;
  const int64_t HW = (int64_t)a.H * a.W;
  stage_g<K, TANH>(a, g_s, b, y0, x0, 0);      // no halo: rows y0..y0+15, column j <-> x0 - 4 + j
  const int cil = tid & 7, ky = (tid >> 3) % K, rg = tid / (8 * K);
  const bool active = tid < 8 * K * 4;
  float* part = a.part + (int64_t)blockIdx.x * 3 * a.Cin * K * K;
  for (int c0 = 0; c0 < a.Cin; c0 += CK) {
    __syncthreads();
    // x tile column j <-> x0 - P + j, row r <-> y0 - P + r
    for (int idx = tid; idx < CK * TR * (64 + K - 1); idx += 256) {
      const int c = idx / (TR * (64 + K - 1));
      const int rem = idx - c * (TR * (64 + K - 1));
      const int r = rem / (64 + K - 1), j = rem - r * (64 + K - 1);
      const int gy = y0 - P + r, gx = x0 - P + j;
      float v = 0.f;
      if (c0 + c < a.Cin && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
        v = a.x[(int64_t)b * a.xbs + (int64_t)(c0 + c) * HW + (int64_t)gy * a.W + gx];
      x_s[(c * TR + r) * XP + j] = v;
    }
    __syncthreads();
    float acc[3][K];
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) acc[co][kx] = 0.f;
    if (active) {
      for (int r = rg * 4; r < rg * 4 + 4; ++r) {
        const float* xr = x_s + (cil * TR + r + ky) * XP;       // input row y0 + r + ky - P
        float win[K];
#pragma unroll
        for (int kx = 0; kx < K - 1; ++kx) win[kx + 1] = xr[kx];
        for (int col = 0; col < 64; ++col) {
#pragma unroll
          for (int kx = 0; kx < K - 1; ++kx) win[kx] = win[kx + 1];
          win[K - 1] = xr[col + K - 1];
          const float g0 = g_s[(0 * 16 + r) * 72 + 4 + col], g1 = g_s[(1 * 16 + r) * 72 + 4 + col],
                      g2 = g_s[(2 * 16 + r) * 72 + 4 + col];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            acc[0][kx] = fmaf(g0, win[kx], acc[0][kx]);
            acc[1][kx] = fmaf(g1, win[kx], acc[1][kx]);
            acc[2][kx] = fmaf(g2, win[kx], acc[2][kx]);
          }
        }
      }
    }
    // combine the 4 row groups of each (ci, ky) in a fixed order through LDS (barriers outside divergent code)
    for (int g = 0; g < 4; ++g) {
      if (active && rg == g) {
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            float* d = &red_s[((co * 32 + (c0 % 32) + cil) * K + ky) * K + kx];
            *d = (g == 0 ? 0.f : *d) + acc[co][kx];
          }
      }
      __syncthreads();
    }
    // flush every 32 channels (or at the end)
    if (((c0 + CK) % 32 == 0) || c0 + CK >= a.Cin) {
      __syncthreads();
      const int cbase = (c0 / 32) * 32;
      for (int e = tid; e < 3 * 32 * K * K; e += 256) {
        const int kk = e % (K * K);
        const int ci = (e / (K * K)) % 32, co = e / (K * K * 32);
        if (cbase + ci < a.Cin && cbase + ci < c0 + CK) part[((int64_t)co * a.Cin + cbase + ci) * K * K + kk] = red_s[e];
      }
    }
  }
}

// out[i] = sum_slot part[slot][i]: block = 32 outputs x 8 slot lanes, combined in a fixed order (reproducible)
__global__ __launch_bounds__(256) void to3_wgrad_reduce_kernel(const float* __restrict__ part, int nslots, int n,
                                                               float* __restrict__ dw) {
  __shared__ float red[8][32];
  const int o = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  float s = 0.f;
  if (i < n)
    for (int k = sg; k < nslots; k += 8) s += part[(int64_t)k * n + i];
  red[sg][o] = s;
  __syncthreads();
  if (sg == 0 && i < n) {
    float v = red[0][o];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += red[k][o];
    dw[i] = v;
  }
}

template <int K, bool TANH>
static int launch_to3_bwd(To3BwdArgs a, float* dw, hipStream_t s) {
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + 15) / 16;
  const int nwg = a.B * a.tiles_x * a.tiles_y;
  if (a.dx) hipLaunchKernelGGL((conv_to3_dgrad_kernel<K, TANH>), dim3(nwg), dim3(256), 0, s, a);
  if (dw) {
    hipLaunchKernelGGL((conv_to3_wgrad_kernel<K, TANH>), dim3(nwg), dim3(256), 0, s, a);
    const int n = 3 * a.Cin * K * K;
    hipLaunchKernelGGL(to3_wgrad_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, s, a.part, nwg, n, dw);
  }
  return note_launch(hipGetLastError(), "conv_to3_bwd");
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_conv_to3_bwd_ws_elems(int B, int Cin, int H, int W, int K) {
  return (int64_t)B * ((W + 63) / 64) * ((H + 15) / 16) * 3 * Cin * K * K;
}

extern "C" int tgsr_conv_to3_bwd(const float* dy, const float* out, const float* addend, float alpha, const float* x,
                                 int64_t x_bstride, const float* w, int B, int Cin, int H, int W, int K, int act,
                                 float* dx, float* ws, float* dw, void* stream) {
  if (!dy || B < 1 || Cin < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if ((K != 3 && K != 5) || Cin > 64) return TGSR_EUNSUPPORTED;
  if (act == TGSR_ACT_TANH_AXPY && !out) return TGSR_EINVAL;
  if (dx && !w) return TGSR_EINVAL;
  if (dw && (!x || !ws)) return TGSR_EINVAL;
  To3BwdArgs a;
  a.dy = dy; a.out = out; a.addend = addend; a.alpha = alpha; a.x = x; a.xbs = x_bstride; a.w = w;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.tiles_x = a.tiles_y = 0; a.dx = dx; a.part = ws;
  hipStream_t s = as_stream(stream);
  const bool th = act == TGSR_ACT_TANH_AXPY;
  if (K == 3) return th ? launch_to3_bwd<3, true>(a, dw, s) : launch_to3_bwd<3, false>(a, dw, s);
  return th ? launch_to3_bwd<5, true>(a, dw, s) : launch_to3_bwd<5, false>(a, dw, s);
}
