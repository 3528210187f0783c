// Weight gradient of the plain 3x3 convolution (stride 1, pad 1) in the Winograd F(2x2, 3x3) domain.  With
// Y = A^T (U (.) V) A per 2x2 output tile t,
//     dU[p][co][ci] = sum_t dM[p][co][t] * V[p][ci][t],   dM = A dY A^T (4x4 from the tile's 2x2 output gradients),
//     V = B^T d B (4x4 from its 4x4 input patch),           p = one of the 16 positions,
// and dW = G^T dU G afterwards: 16 products per tile and channel pair instead of 36 (9 taps x 4 pixels) - 2.25x fewer
// multiplies than tgsr_conv3x3_wgrad.hip.  fp32 throughout; the transforms only add / subtract, G holds {0, 1, +-1/2}.
//
// GEMM view per position: M = co, N = ci, K = tiles.  Workgroup = 2 co blocks x NCI ci blocks x 2 position halves of
// waves, wave = (32 co, 32 ci, 8 positions) = 8 MFMA 32x32x2 accumulators (128 VGPRs); it walks over chunks of 8
// consecutive tiles of one tile row: every thread transforms one (co, tile) item of dM and one (ci, tile) item of V
// straight from global memory into LDS images [p][channel][8 tiles] (pitch 9), then 8 x 4 MFMAs per wave consume the
// chunk.  One partial slab [16][Cout][Cin] per workgroup; wino_wgrad_reduce_kernel sums the slabs in a fixed order,
// applies G^T . G and writes the torch layout [Cout][Cin][3][3] (bitwise reproducible, no float atomics).
#include "tgsr_common.h"

namespace tgsr {

struct WinoWgradArgs {
  const float* g;     // [B][Cout][H][W]  gradient w.r.t. the raw convolution output
  const float* x;     // [B][Cin][H][W]
  int64_t xbs;
  int B, Cin, Cout, H, W;
  int tiles_y, chunks_x, nchunks, chunks_per_wg, cgroups_i;
  float* partial;     // [nslots][16][Cout][Cin]
};

constexpr int kWWT = 8, kWWP = kWWT + 1;    // tiles per chunk, LDS pitch

template <int NCI>
__global__ __launch_bounds__(256 * NCI) void wino_wgrad_kernel(WinoWgradArgs a) {
  constexpr int NT = 256 * NCI, NCO = 64, NCIN = 32 * NCI;
  __shared__ float m_s[16 * NCO * kWWP];     // dM [p][co][tile]
  __shared__ float v_s[16 * NCIN * kWWP];    // V  [p][ci][tile]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = wave & 1, ph = (wave >> 1) & 1, cib = wave >> 2;        // co block, position half, ci block
  const int grp = blockIdx.y;
  const int co0 = (grp / a.cgroups_i) * NCO, ci0 = (grp % a.cgroups_i) * NCIN;
  const int64_t HW = (int64_t)a.H * a.W;

  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;

  const int c_lo = blockIdx.x * a.chunks_per_wg;
  const int c_hi = c_lo + a.chunks_per_wg < a.nchunks ? c_lo + a.chunks_per_wg : a.nchunks;
  for (int chunk = c_lo; chunk < c_hi; ++chunk) {
    int t = chunk;
    const int cx = t % a.chunks_x;
    t /= a.chunks_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = 2 * ty, tx0 = cx * kWWT;
    __syncthreads();                          // the previous chunk's MFMAs are done with the LDS images
    // dM: item = (co, tile); dY = the tile's 2x2 output gradients (zero outside the image)
    for (int item = tid; item < NCO * kWWT; item += NT) {
      const int c = item / kWWT, tl = item - c * kWWT;
      const int x = 2 * (tx0 + tl);
      float d[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
      if (co0 + c < a.Cout) {
        const float* gp = a.g + ((int64_t)b * a.Cout + co0 + c) * HW;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (y0 + r < a.H && x + q < a.W) d[r][q] = gp[(int64_t)(y0 + r) * a.W + x + q];
      }
      // A dY: rows (d0), (d0 + d1), (d0 - d1), (-d1); then the same along the columns
      const float r4[4][2] = {{d[0][0], d[0][1]}, {d[0][0] + d[1][0], d[0][1] + d[1][1]},
                              {d[0][0] - d[1][0], d[0][1] - d[1][1]}, {-d[1][0], -d[1][1]}};
      float* mp = m_s + c * kWWP + tl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mp[(i * 4 + 0) * NCO * kWWP] = r4[i][0];
        mp[(i * 4 + 1) * NCO * kWWP] = r4[i][0] + r4[i][1];
        mp[(i * 4 + 2) * NCO * kWWP] = r4[i][0] - r4[i][1];
        mp[(i * 4 + 3) * NCO * kWWP] = -r4[i][1];
      }
    }
    // V: item = (ci, tile); d = the tile's 4x4 input patch (rows y0-1 .. y0+2, cols x-1 .. x+2)
    for (int item = tid; item < NCIN * kWWT; item += NT) {
      const int c = item / kWWT, tl = item - c * kWWT;
      const int x = 2 * (tx0 + tl);
      float d[4][4];
      const bool cok = ci0 + c < a.Cin;
      const float* xp = a.x + (int64_t)b * a.xbs + (int64_t)(ci0 + c) * HW;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gy = y0 - 1 + rr, gx = x - 1 + q;
          d[rr][q] = (cok && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) ? xp[(int64_t)gy * a.W + gx] : 0.f;
        }
      float tr[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        tr[0][q] = d[0][q] - d[2][q];
        tr[1][q] = d[1][q] + d[2][q];
        tr[2][q] = d[2][q] - d[1][q];
        tr[3][q] = d[1][q] - d[3][q];
      }
      float* vp = v_s + c * kWWP + tl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vp[(i * 4 + 0) * NCIN * kWWP] = tr[i][0] - tr[i][2];
        vp[(i * 4 + 1) * NCIN * kWWP] = tr[i][1] + tr[i][2];
        vp[(i * 4 + 2) * NCIN * kWWP] = tr[i][2] - tr[i][1];
        vp[(i * 4 + 3) * NCIN * kWWP] = tr[i][1] - tr[i][3];
      }
    }
    __syncthreads();
    const float* mw = m_s + (ph * 8 * NCO + cob * 32 + l31) * kWWP + hh;
    const float* vw = v_s + (ph * 8 * NCIN + cib * 32 + l31) * kWWP + hh;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
      for (int k = 0; k < kWWT / 2; ++k)
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(mw[p * NCO * kWWP + 2 * k], vw[p * NCIN * kWWP + 2 * k], acc[p], 0, 0, 0);
    }
  }
  // one slab per workgroup: partial[slot][p][co][ci]; lane = ci (coalesced), register rows = co
  float* ps = a.partial + (int64_t)blockIdx.x * 16 * a.Cout * a.Cin;
  const int ci = ci0 + cib * 32 + l31;
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + cob * 32 + acc_row(i, hh);
      if (co < a.Cout && ci < a.Cin) ps[((int64_t)(ph * 8 + p) * a.Cout + co) * a.Cin + ci] = acc[p][i];
    }
}

// dU[p][co][ci] = sum_slot partial (fixed order), then dW = G^T dU G with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ partial, int nslots, int Cout,
                                                                int Cin, float* __restrict__ dw) {
  __shared__ float red[8][32][17];
  const int64_t n = (int64_t)Cout * Cin;
  const int o = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int64_t e = (int64_t)blockIdx.x * 32 + o;         // (co, ci) pair
  float s[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) s[p] = 0.f;
  if (e < n) {
    // 2 slots x 16 positions = 32 independent loads in flight per trip (written out: without the SLP vectorizer hipcc issued
    // them one dependent add at a time and the kernel took 135 instead of 29 us); the adds keep the slot order
    int k = sg;
    for (; k + 8 < nslots; k += 16) {
      float v0[16], v1[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        v0[p] = partial[((int64_t)k * 16 + p) * n + e];
        v1[p] = partial[((int64_t)(k + 8) * 16 + p) * n + e];
      }
#pragma unroll
      for (int p = 0; p < 16; ++p) s[p] = (s[p] + v0[p]) + v1[p];
    }
    for (; k < nslots; k += 8)
#pragma unroll
      for (int p = 0; p < 16; ++p) s[p] += partial[((int64_t)k * 16 + p) * n + e];
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) red[sg][o][p] = s[p];
  __syncthreads();
  if (sg == 0 && e < n) {
    float u[4][4];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      float v = red[0][o][p];
#pragma unroll
      for (int k = 1; k < 8; ++k) v += red[k][o][p];
      u[p >> 2][p & 3] = v;
    }
    // columns of G: g0 = (1, .5, .5, 0), g1 = (0, .5, -.5, 0), g2 = (0, .5, .5, 1)
    float w1[3][4];   // G^T dU: rows a, columns j
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w1[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
      w1[1][j] = 0.5f * (u[1][j] - u[2][j]);
      w1[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
    float* out = dw + e * 9;
#pragma unroll
    for (int aa = 0; aa < 3; ++aa) {
      out[aa * 3 + 0] = w1[aa][0] + 0.5f * (w1[aa][1] + w1[aa][2]);
      out[aa * 3 + 1] = 0.5f * (w1[aa][1] - w1[aa][2]);
      out[aa * 3 + 2] = 0.5f * (w1[aa][1] + w1[aa][2]) + w1[aa][3];
    }
  }
}

}  // namespace tgsr

using namespace tgsr;

static void wwgrad_plan(int B, int Cin, int Cout, int H, int W, int* nci, int* groups, int* gi, int* nslots, int* cpw,
                        int* nchunks, int* tiles_y, int* chunks_x) {
  *nci = (Cin % 64 == 0) ? 2 : 1;
  *gi = Cin / (32 * *nci);
  *groups = (Cout / 64) * *gi;
  *tiles_y = (H + 1) / 2;
  *chunks_x = ((W + 1) / 2 + kWWT - 1) / kWWT;
  *nchunks = B * *tiles_y * *chunks_x;
  int want = 256 / *groups;                  // one 8-wave workgroup per CU: fewer, longer K walks keep the slabs small
  want = want * wgrad_split_pct() / 100;
  if (want < 1) want = 1;
  if (want > *nchunks) want = *nchunks;
  *cpw = (*nchunks + want - 1) / want;
  *nslots = (*nchunks + *cpw - 1) / *cpw;
}

extern "C" int64_t tgsr_wino_wgrad_ws_elems(int B, int Cin, int Cout, int H, int W) {
  int nci, groups, gi, nslots, cpw, nchunks, ty, cx;
  wwgrad_plan(B, Cin, Cout, H, W, &nci, &groups, &gi, &nslots, &cpw, &nchunks, &ty, &cx);
  return (int64_t)nslots * 16 * Cout * Cin;
}

extern "C" int tgsr_wino_wgrad(const float* grad_out, const float* x, int64_t x_bstride, int B, int Cin, int H, int W,
                               int Cout, float* ws, float* dw, void* stream) {
  if (!grad_out || !x || !ws || !dw || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (Cout % 64 != 0 || Cin % 32 != 0) return TGSR_EUNSUPPORTED;
  WinoWgradArgs a;
  a.g = grad_out; a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  int nci, groups, gi, nslots, cpw, nchunks;
  wwgrad_plan(B, Cin, Cout, H, W, &nci, &groups, &gi, &nslots, &cpw, &nchunks, &a.tiles_y, &a.chunks_x);
  a.nchunks = nchunks; a.chunks_per_wg = cpw; a.cgroups_i = gi; a.partial = ws;
  hipStream_t s = as_stream(stream);
  dim3 grid(nslots, groups);
  if (nci == 2) hipLaunchKernelGGL(wino_wgrad_kernel<2>, grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL(wino_wgrad_kernel<1>, grid, dim3(256), 0, s, a);
  int rc = note_launch(hipGetLastError(), "wino_wgrad_kernel");
  if (rc) return rc;
  const int64_t n = (int64_t)Cout * Cin;
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, ws, nslots, Cout, Cin,
                     dw);
  return note_launch(hipGetLastError(), "wino_wgrad_reduce_kernel");
}
