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


// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient on the matrix cores (fp32 MFMA 16x16x4, exact fmaf chains).  With u = x + kx - P the input column and
// y' = y + ky - P the input row,
//     dw[co][ci][ky][kx] = sum_{b,y',u} g[co][y' - ky + P][u - kx + P] * x[ci][y'][u]
// which, per ky, is a 16 x 16 product accumulated over pixels: rows r = co * K + kx (3K <= 15 of 16 used) take shifted
// reads of the small g tile (3 channels, in LDS, halo on g instead of on x), columns are 16 input channels whose
// values come straight from HBM - every x element is read exactly once, by float4 loads a lane uses for 4 successive
// MFMA k-steps (the 4 pixels of one step are u0 + 4 kq + s, kq = lane >> 4: any 4 pixels do as long as A reads the same).
// A B fragment feeds K MFMAs (one per ky), an A fragment Cin / 16.  Wave = RPW rows of a 64-column tile, workgroup = 4
// waves; the 4 waves' accumulators are summed through LDS in a fixed order into one slab per workgroup.
template <int K, bool TANH, int NCG>
__global__ __launch_bounds__(256) void conv_to3_wgrad_mfma_kernel(To3BwdArgs a, int rpw) {
  constexpr int P = K / 2, GP = 72, MAXR = 4 * 8 + 2 * P;          // g tile pitch; rows of the largest tile (rpw <= 8)
  constexpr int GS = 3 * MAXR * GP + 16, RS = 3 * 16 * NCG * K * 16;
  __shared__ float smem_f[GS > RS ? GS : RS];
  float* g_s = smem_f;
  float* red_s = smem_f;                                            // [wave 1..3][ky][cg][row 16][col 16], after the main loop
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int TRows = 4 * rpw, TRg = TRows + 2 * P;
  const int y0 = ty * TRows, x0 = tx * 64;
  const int64_t HW = (int64_t)a.H * a.W;
  // g tile: row r <-> y0 - P + r, column j <-> x0 - 4 + j (zero outside the image); one zero word for the unused rows
  for (int idx = tid; idx < 3 * TRg * GP; idx += 256) {
    const int co = idx / (TRg * GP);
    const int rem = idx - co * (TRg * GP);
    const int r = rem / GP, j = rem - r * GP;
    const int gy = y0 - P + r, gx = x0 - 4 + j;
    float v = 0.f;
    if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
      const int64_t o = ((int64_t)b * 3 + co) * HW + (int64_t)gy * a.W + gx;
      v = a.dy[o];
      if (TANH) {
        const float th = a.out[o] - (a.addend ? a.alpha * a.addend[o] : 0.f);
        v *= 1.f - th * th;
      }
    }
    g_s[(co * MAXR + r) * GP + j] = v;
  }
  if (tid < 16) g_s[3 * MAXR * GP + tid] = 0.f;
  __syncthreads();
  // A-fragment base of this lane: row r16 = co * K + kx -> g_s[co][.][4 + P - kx + 4 kq + ...]; rows >= 3K read the zero word
  const int co = r16 / K, kx = r16 - co * K;
  const bool arow = r16 < 3 * K;
  const int abase = arow ? co * MAXR * GP + 4 + P - kx + 4 * kq : 3 * MAXR * GP;
  const int amul = arow ? 1 : 0;                                     // zero rows: every offset collapses onto the zero word
  f32x4 acc[K][NCG];
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) acc[ky][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xb = a.x + (int64_t)b * a.xbs + (int64_t)r16 * HW + x0 + 4 * kq;
  const int ngrp = min(4, (a.W - x0) / 16);                          // 16-pixel groups of this tile inside the image (W % 16 == 0)
  const int nit = rpw * ngrp;                                        // (row, group) steps of this wave
  auto load_x = [&](int it, float4* xv) {
    const int rr = it / ngrp, gi = it - rr * ngrp;
    const int y = y0 + wave * rpw + rr;
    const bool ok = it < nit && y < a.H;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg)
      xv[cg] = ok ? *reinterpret_cast<const float4*>(xb + (int64_t)cg * 16 * HW + (int64_t)y * a.W + gi * 16)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float4 xcur[NCG], xnext[NCG];
  load_x(0, xcur);
  for (int it = 0; it < nit; ++it) {
    load_x(it + 1, xnext);
    const int rr = it / ngrp, gi = it - rr * ngrp;
    // g row for input row y' = y0 + wave*rpw + rr and tap ky: tile row (y' - y0) - ky + 2P
    const int arow0 = abase + amul * (((wave * rpw + rr + 2 * P) * GP) + gi * 16);
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const float av = g_s[arow0 + amul * (sidx - ky * GP)];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
          const float bv = sidx == 0 ? xcur[cg].x : sidx == 1 ? xcur[cg].y : sidx == 2 ? xcur[cg].z : xcur[cg].w;
          acc[ky][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ky][cg], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) xcur[cg] = xnext[cg];
  }
  // fixed-order sum of the 4 waves: waves 1..3 park their accumulators in LDS, wave 0 adds them 1, 2, 3 and writes the slab
  __syncthreads();                                                  // everybody is done reading the g tile
  if (wave > 0) {
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          red_s[((((wave - 1) * K + ky) * NCG + cg) * 16 + 4 * kq + j) * 16 + r16] = acc[ky][cg][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* part = a.part + (int64_t)blockIdx.x * 3 * a.Cin * K * K;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[ky][cg][j];
#pragma unroll
          for (int w = 0; w < 3; ++w) v += red_s[(((w * K + ky) * NCG + cg) * 16 + 4 * kq + j) * 16 + r16];
          const int row = 4 * kq + j;                                 // D: row = 4 (lane >> 4) + reg, column = lane & 15
          if (row < 3 * K) {
            const int oco = row / K, okx = row - oco * K;
            part[(((int64_t)oco * a.Cin + cg * 16 + r16) * K + ky) * K + okx] = v;
          }
        }
  }
}

// rows per wave of the MFMA weight-gradient tile: the largest of 8, 4, 2, 1 that still gives >= 1024 workgroups
static int to3_wgrad_rpw(int B, int H, int W) {
  const int tx = (W + 63) / 64;
  for (int rpw = 8; rpw > 1; rpw >>= 1)
    if ((int64_t)B * tx * ((H + 4 * rpw - 1) / (4 * rpw)) >= 1024) return rpw;
  return 1;
}
static bool to3_wgrad_mfma_ok(int Cin, int W) { return W % 16 == 0 && (Cin == 16 || Cin == 32 || Cin == 48 || Cin == 64); }

// out[i] = sum_slot part[slot][i]: block = 32 outputs x 32 slot lanes, combined in a fixed order (reproducible)
__global__ __launch_bounds__(1024) void to3_wgrad_reduce_kernel(const float* __restrict__ part, int nslots, int n,
                                                                float* __restrict__ dw) {
  __shared__ float red[32][32];
  const int o = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  float s = 0.f;
  if (i < n)
    for (int k = sg; k < nslots; k += 32) s += part[(int64_t)k * n + i];
  red[sg][o] = s;
  __syncthreads();
  if (sg == 0 && i < n) {
    float v = red[0][o];
#pragma unroll
    for (int k = 1; k < 32; ++k) v += red[k][o];
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
    const int n = 3 * a.Cin * K * K;
    int nslab = nwg;
    if (to3_wgrad_mfma_ok(a.Cin, a.W)) {
      const int rpw = to3_wgrad_rpw(a.B, a.H, a.W);
      a.tiles_y = (a.H + 4 * rpw - 1) / (4 * rpw);
      nslab = a.B * a.tiles_x * a.tiles_y;
      switch (a.Cin / 16) {
        case 1: hipLaunchKernelGGL((conv_to3_wgrad_mfma_kernel<K, TANH, 1>), dim3(nslab), dim3(256), 0, s, a, rpw); break;
        case 2: hipLaunchKernelGGL((conv_to3_wgrad_mfma_kernel<K, TANH, 2>), dim3(nslab), dim3(256), 0, s, a, rpw); break;
        case 3: hipLaunchKernelGGL((conv_to3_wgrad_mfma_kernel<K, TANH, 3>), dim3(nslab), dim3(256), 0, s, a, rpw); break;
        default: hipLaunchKernelGGL((conv_to3_wgrad_mfma_kernel<K, TANH, 4>), dim3(nslab), dim3(256), 0, s, a, rpw); break;
      }
    } else {
      hipLaunchKernelGGL((conv_to3_wgrad_kernel<K, TANH>), dim3(nwg), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(to3_wgrad_reduce_kernel, dim3((n + 31) / 32), dim3(1024), 0, s, a.part, nslab, n, dw);
  }
  return note_launch(hipGetLastError(), "conv_to3_bwd");
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_conv_to3_bwd_ws_elems(int B, int Cin, int H, int W, int K) {
  int64_t slabs = (int64_t)B * ((W + 63) / 64) * ((H + 15) / 16);
  if (to3_wgrad_mfma_ok(Cin, W)) {
    const int rpw = to3_wgrad_rpw(B, H, W);
    slabs = (int64_t)B * ((W + 63) / 64) * ((H + 4 * rpw - 1) / (4 * rpw));
  }
  return slabs * 3 * Cin * K * K;
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
