// Weight gradient of the fused 3x3 convolution (training path) for gfx950, fp32 on the MFMA units.
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} g[b][co][y][x] * xin[b][ci][y+ky-1][x+kx-1]
// where g is the gradient wrt the raw conv output and xin the conv input (for the upBlock form: the nearest-x2
// up-sampled input, read through the same `>> 1` LDS indexing as the forward, util.py:74-80).
// GEMM view: M = co, N = ci, K = pixels (up to B*H*W = 1M), one 32x32 accumulator per tap:
//   * wave <-> (block of 32 co, block of 32 ci), 9 accumulators (the taps) = 144 VGPRs; a workgroup is NCOB x NCIB
//     such waves (128x64 channels -> 8 waves) sharing one LDS image of g [co][64 px] (pitch 65) and of the input
//     halo tile [ci][4 rows][34] (plane 141): both MFMA operands are conflict-free ds_read_b32 (lane = channel);
//   * a k-step is two horizontally adjacent pixels: 1 g fragment + 9 shifted input fragments -> 9 MFMAs;
//   * each workgroup walks a contiguous run of 2x32-pixel tiles, keeps its accumulators in registers, and writes ONE
//     partial slab [tap][co][ci] at the end; a second kernel sums the slabs in a fixed order (bitwise reproducible,
//     no float atomics) into the torch layout [co][ci][3][3].
#include "tgsr_common.h"

namespace tgsr {

struct WgradArgs {
  const float* g;     // [B][Cout][Ho][Wo]
  const float* x;     // [B][Cin][H][W]   (H = Ho/2 when up)
  int64_t xbs;
  int B, Cin, Cout, H, W, Ho, Wo;
  int tiles_x, tiles_y, ntiles, tiles_per_wg, cgroups_i;  // channel groups along ci per co group
  float* partial;     // [nslots][9][Cout][CinPad]
  int CinPad;
};

template <int NCOB, int NCIB, bool UP>
__global__ __launch_bounds__(64 * NCOB * NCIB) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr int NT = 64 * NCOB * NCIB;
  constexpr int PA = 65;                       // g image: [co][2 rows x 32 cols] pitch
  constexpr int TR = UP ? 3 : 4, TC = UP ? 18 : 34;
  constexpr int PITCH = TC + 1;
  constexpr int PLANE = TR * PITCH + ((TR * PITCH) % 2 == 0 ? 1 : 0);   // odd: lanes (= ci) hit distinct banks
  __shared__ float g_s[NCOB * 32 * PA];
  __shared__ float x_s[NCIB * 32 * PLANE];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = wave % NCOB, cib = wave / NCOB;
  const int grp = blockIdx.y;
  const int co0 = (grp / a.cgroups_i) * NCOB * 32, ci0 = (grp % a.cgroups_i) * NCIB * 32;
  const int64_t HWo = (int64_t)a.Ho * a.Wo, HW = (int64_t)a.H * a.W;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int t_lo = blockIdx.x * a.tiles_per_wg;
  const int t_hi = t_lo + a.tiles_per_wg < a.ntiles ? t_lo + a.tiles_per_wg : a.ntiles;
  for (int tile = t_lo; tile < t_hi; ++tile) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = ty * 2, x0 = tx * 32;
    const int sy0 = (UP ? (y0 >> 1) : y0) - 1, sx0 = (UP ? (x0 >> 1) : x0) - 1;
    __syncthreads();
    // g tile: [NCOB*32][2][32]
    for (int idx = tid; idx < NCOB * 32 * 64; idx += NT) {
      const int c = idx >> 6, p = idx & 63;
      const int y = y0 + (p >> 5), x = x0 + (p & 31);
      float v = 0.f;
      if (co0 + c < a.Cout && y < a.Ho && x < a.Wo) v = a.g[((int64_t)b * a.Cout + co0 + c) * HWo + (int64_t)y * a.Wo + x];
      g_s[c * PA + p] = v;
    }
    // input halo tile: [NCIB*32][TR][TC]
    for (int idx = tid; idx < NCIB * 32 * TR * TC; idx += NT) {
      const int c = idx / (TR * TC);
      const int rem = idx - c * (TR * TC);
      const int r = rem / TC, cc = rem - r * TC;
      const int gy = sy0 + r, gx = sx0 + cc;
      float v = 0.f;
      if (ci0 + c < a.Cin && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
        v = a.x[(int64_t)b * a.xbs + (int64_t)(ci0 + c) * HW + (int64_t)gy * a.W + gx];
      x_s[c * PLANE + r * PITCH + cc] = v;
    }
    __syncthreads();
    const float* gw = g_s + (cob * 32 + l31) * PA;
    const float* xw = x_s + (cib * 32 + l31) * PLANE;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll 4
      for (int kc = 0; kc < 16; ++kc) {
        const int xo = 2 * kc + hh;                 // this lane half's pixel column inside the tile
        const float av = gw[r * 32 + xo];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int sr = UP ? (((r + ky - 1) >> 1) + 1) : (r + ky);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int sc = UP ? (((xo + kx - 1) >> 1) + 1) : (xo + kx);
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xw[sr * PITCH + sc], acc[ky * 3 + kx], 0, 0, 0);
          }
        }
      }
    }
  }
  // one slab per workgroup: partial[slot][tap][co][ci]; lane = ci (coalesced), register rows = co
  float* ps = a.partial + (int64_t)blockIdx.x * 9 * a.Cout * a.CinPad;
  const int ci = ci0 + cib * 32 + l31;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + cob * 32 + acc_row(i, hh);
      if (co < a.Cout && ci < a.CinPad) ps[((int64_t)t * a.Cout + co) * a.CinPad + ci] = acc[t][i];
    }
}

// dw[co][ci][tap] = sum_slot partial[slot][tap][co][ci]: block = 32 slab elements x 8 slot lanes, fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nslots, int Cout,
                                                           int Cin, int CinPad, float* __restrict__ dw) {
  __shared__ float red[8][32];
  const int64_t n = (int64_t)9 * Cout * CinPad;
  const int o = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + o;
  float s = 0.f;
  if (i < n)
    for (int k = sg; k < nslots; k += 8) s += partial[(int64_t)k * n + i];
  red[sg][o] = s;
  __syncthreads();
  if (sg == 0 && i < n) {
    const int ci = (int)(i % CinPad);
    const int64_t t = i / CinPad;
    const int co = (int)(t % Cout), tap = (int)(t / Cout);
    if (ci < Cin) {
      float v = red[0][o];
#pragma unroll
      for (int k = 1; k < 8; ++k) v += red[k][o];
      dw[((int64_t)co * Cin + ci) * 9 + tap] = v;
    }
  }
}

template <int NCOB, int NCIB>
static int launch_wgrad(const WgradArgs& a, int nslots, int groups, bool up, hipStream_t s) {
  dim3 grid(nslots, groups), block(64 * NCOB * NCIB);
  if (up) hipLaunchKernelGGL((conv3x3_wgrad_kernel<NCOB, NCIB, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((conv3x3_wgrad_kernel<NCOB, NCIB, false>), grid, block, 0, s, a);
  return note_launch(hipGetLastError(), "conv3x3_wgrad_kernel");
}

}  // namespace tgsr

using namespace tgsr;

static void wgrad_plan(int B, int Cin, int Cout, int Ho, int Wo, int* ncob, int* ncib, int* groups, int* gi,
                       int* nslots, int* tiles_per_wg, int* ntiles, int* cinpad) {
  const int cb = (Cout + 31) / 32, ib = (Cin + 31) / 32;
  *ncob = cb % 4 == 0 ? 4 : (cb % 2 == 0 ? 2 : 1);
  *ncib = ib % 2 == 0 ? 2 : 1;
  *gi = ib / *ncib;
  *groups = (cb / *ncob) * *gi;
  *cinpad = ib * 32;
  *ntiles = B * ((Ho + 1) / 2) * ((Wo + 31) / 32);
  // one partial slab per workgroup: ~256 CUs x 8 waves of workgroups in flight keeps the chip full while the slabs
  // (nslots x |dW|) stay ~75 MB for every layer shape
  int want = 2048 / (*ncob * *ncib) / *groups;
  want = want * wgrad_split_pct() / 100;
  if (want < 1) want = 1;
  if (want > *ntiles) want = *ntiles;
  *tiles_per_wg = (*ntiles + want - 1) / want;
  *nslots = (*ntiles + *tiles_per_wg - 1) / *tiles_per_wg;
}

extern "C" int64_t tgsr_conv3x3_wgrad_ws_elems(int B, int Cin, int Cout, int H, int W, int upsample) {
  int ncob, ncib, groups, gi, nslots, tpw, ntiles, cinpad;
  const int Ho = upsample ? 2 * H : H, Wo = upsample ? 2 * W : W;
  wgrad_plan(B, Cin, Cout, Ho, Wo, &ncob, &ncib, &groups, &gi, &nslots, &tpw, &ntiles, &cinpad);
  return (int64_t)nslots * 9 * Cout * cinpad;
}

extern "C" int tgsr_conv3x3_wgrad(const float* grad_out, const float* x, int64_t x_bstride, int B, int Cin, int H,
                                  int W, int Cout, int upsample, float* ws, float* dw, void* stream) {
  if (!grad_out || !x || !ws || !dw || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (Cout % 32 != 0) return TGSR_EUNSUPPORTED;
  WgradArgs a;
  a.g = grad_out; a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.Ho = upsample ? 2 * H : H; a.Wo = upsample ? 2 * W : W;
  int ncob, ncib, groups, gi, nslots, tpw, ntiles, cinpad;
  wgrad_plan(B, Cin, Cout, a.Ho, a.Wo, &ncob, &ncib, &groups, &gi, &nslots, &tpw, &ntiles, &cinpad);
  a.tiles_x = (a.Wo + 31) / 32; a.tiles_y = (a.Ho + 1) / 2; a.ntiles = ntiles; a.tiles_per_wg = tpw;
  a.cgroups_i = gi; a.partial = ws; a.CinPad = cinpad;
  hipStream_t s = as_stream(stream);
  const bool up = upsample != 0;
  int rc;
  if (ncob == 4 && ncib == 2) rc = launch_wgrad<4, 2>(a, nslots, groups, up, s);
  else if (ncob == 4) rc = launch_wgrad<4, 1>(a, nslots, groups, up, s);
  else if (ncob == 2 && ncib == 2) rc = launch_wgrad<2, 2>(a, nslots, groups, up, s);
  else if (ncob == 2) rc = launch_wgrad<2, 1>(a, nslots, groups, up, s);
  else if (ncib == 2) rc = launch_wgrad<1, 2>(a, nslots, groups, up, s);
  else rc = launch_wgrad<1, 1>(a, nslots, groups, up, s);
  if (rc) return rc;
  const int64_t n = (int64_t)9 * Cout * cinpad;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, ws, nslots, Cout, Cin,
                     cinpad, dw);
  return note_launch(hipGetLastError(), "wgrad_reduce_kernel");
}
