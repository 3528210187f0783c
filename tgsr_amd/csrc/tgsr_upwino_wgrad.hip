// Weight gradient of the upBlock convolution (Upsample(x2, nearest) -> conv3x3, util.py:74-80) in the domain of the
// up-sample-aware Winograd form of tgsr_upwino.hip.  With Y = A'^T (U' (.) V) A' per low-resolution pixel t,
//     dU'[p][co][ci] = sum_t dM[p][co][t] * V[p][ci][t],   dM = A' dY A'^T (3x3 from the pixel's 2x2 output gradients),
//     V = T d T^T (3x3 from its 3x3 input neighbourhood),   p = one of the 9 positions,
// and dW = G'^T dU' G' afterwards (G' = [[1,0,0],[1,1,1],[0,0,1]]: sums of the 9 positions).  That is 9 products per
// LOW-resolution pixel and channel pair instead of 9 taps on each of its 4 output pixels: 4x fewer multiplies than the
// direct form (tgsr_conv3x3_wgrad.hip with up = 1).  fp32 throughout, transforms only add / subtract.
//
// GEMM view per position: M = co, N = ci, K = low-res pixels.  Workgroup = 2 x NCI waves, wave = (32 co, 32 ci) with
// the 9 position accumulators (MFMA 32x32x2, 144 VGPRs); it walks over chunks of 16 consecutive pixels of one row:
// every thread transforms a few (channel, pixel) items of dM and V straight from global memory into LDS images
// [p][channel][16 px] (pitch 17: lanes = channels hit distinct banks), then 9 x 8 MFMAs per wave consume the chunk.
// One partial slab [9][Cout][Cin] per workgroup, summed in a fixed order by upwino_wgrad_reduce_kernel, which also
// applies G'^T . G' and writes the torch layout [Cout][Cin][3][3] (bitwise reproducible, no float atomics).
#include "tgsr_common.h"

namespace tgsr {

struct UpWgradArgs {
  const float* g;     // [B][Cout][2H][2W]  gradient w.r.t. the raw convolution output
  const float* x;     // [B][Cin][H][W]     low-resolution input
  int64_t xbs;
  int B, Cin, Cout, H, W;
  int chunks_x, nchunks, chunks_per_wg, cgroups_i;
  float* partial;     // [nslots][9][Cout][Cin]
};

constexpr int kUWT = 16, kUWP = kUWT + 1;   // pixels per chunk, LDS pitch

template <int NCI>
__global__ __launch_bounds__(128 * NCI) void upwino_wgrad_kernel(UpWgradArgs a) {
  constexpr int NT = 128 * NCI, NCO = 64, NCIN = 32 * NCI;
  __shared__ float m_s[9 * NCO * kUWP];      // dM [p][co][px]
  __shared__ float v_s[9 * NCIN * kUWP];     // V  [p][ci][px]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = wave & 1, cib = wave >> 1;
  const int grp = blockIdx.y;
  const int co0 = (grp / a.cgroups_i) * NCO, ci0 = (grp % a.cgroups_i) * NCIN;
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const int64_t HWo = (int64_t)Ho * Wo, HW = (int64_t)a.H * a.W;

  f32x16 acc[9];
#pragma unroll
  for (int p = 0; p < 9; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;

  const int c_lo = blockIdx.x * a.chunks_per_wg;
  const int c_hi = c_lo + a.chunks_per_wg < a.nchunks ? c_lo + a.chunks_per_wg : a.nchunks;
  // Software pipeline (round 4, as tgsr_wino_wgrad.hip): the raw values of the NEXT chunk - the 2x2 output gradients of a
  // thread's dM items, the 3x3 neighbourhoods of its V items - are fetched into registers right before the MFMAs of the
  // current chunk, so their latency runs under the matrix work instead of opening every chunk.
  constexpr int ND = NCO * kUWT / NT, NV = NCIN * kUWT / NT;
  static_assert(ND * NT == NCO * kUWT && NV * NT == NCIN * kUWT, "items divide evenly");
  float gd[ND][4], xd[NV][9];
  // Branch-free (see tgsr_wino_wgrad.hip): unconditional loads from clamped addresses, out-of-image values zeroed by a select;
  // the row is wave-uniform, only the ends of an image row need a per-lane select.
  int dpx[ND], vpx[NV];
  const float* gplane[ND];
  const float* xplane[NV];
#pragma unroll
  for (int n = 0; n < ND; ++n) {
    const int item = tid + n * NT, c = item / kUWT;
    dpx[n] = item - c * kUWT;
    gplane[n] = a.g + (int64_t)(co0 + c < a.Cout ? co0 + c : 0) * HWo;
  }
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const int item = tid + n * NT, c = item / kUWT;
    vpx[n] = item - c * kUWT;
    xplane[n] = a.x + (int64_t)(ci0 + c < a.Cin ? ci0 + c : 0) * HW;
  }
  const int Hm1 = a.H - 1, Wm1 = a.W - 1;
  // validity of the fetched values, applied at transform time (a select here would make the wave wait for its loads at once)
  unsigned dok[ND], vcm[NV], vrm = 0;
  auto fetch = [&](int chunk) {
    int t = chunk;
    const int cx = t % a.chunks_x;
    t /= a.chunks_x;
    const int y = t % a.H;
    const int b = t / a.H;
    const int x0 = cx * kUWT;
    const int64_t gb = (int64_t)b * a.Cout * HWo, xb = (int64_t)b * a.xbs;
#pragma unroll
    for (int n = 0; n < ND; ++n) {           // dM item = (co, px): the 2x2 output gradients of low-res pixel (y, x0 + px)
      const int item = tid + n * NT, c = item / kUWT;
      const int x = x0 + dpx[n];
      dok[n] = (x < a.W && co0 + c < a.Cout) ? 1u : 0u;
      const float* gp = gplane[n] + gb + (int64_t)(2 * y) * Wo + 2 * (x < Wm1 ? x : Wm1);   // 8-byte aligned: Wo, 2x even
      const float2 r0 = *reinterpret_cast<const float2*>(gp), r1 = *reinterpret_cast<const float2*>(gp + Wo);
      gd[n][0] = r0.x; gd[n][1] = r0.y; gd[n][2] = r1.x; gd[n][3] = r1.y;
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {           // V item = (ci, px): the 3x3 neighbourhood of the low-res input
      const int item = tid + n * NT, c = item / kUWT;
      const int x = x0 + vpx[n];
      const bool cok = x < a.W && ci0 + c < a.Cin;
      int gxc[3];
      vcm[n] = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int gx = x - 1 + q;
        vcm[n] |= ((cok && (unsigned)gx < (unsigned)a.W) ? 1u : 0u) << q;
        gxc[q] = gx < 0 ? 0 : (gx < Wm1 ? gx : Wm1);
      }
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const int gy = y - 1 + rr;
        const float* rowp = xplane[n] + xb + (int64_t)(gy < 0 ? 0 : (gy < Hm1 ? gy : Hm1)) * a.W;
#pragma unroll
        for (int q = 0; q < 3; ++q) xd[n][3 * rr + q] = rowp[gxc[q]];
      }
    }
    vrm = 0;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) vrm |= ((unsigned)(y - 1 + rr) < (unsigned)a.H ? 1u : 0u) << rr;
  };
  if (c_lo < c_hi) fetch(c_lo);
  // (barriers as s_waitcnt lgkmcnt(0) + s_barrier: the release fence of __syncthreads() waits for vmcnt(0) on gfx9 - loads and
  // stores share the counter - which would park every wave on its own prefetch)
  for (int chunk = c_lo; chunk < c_hi; ++chunk) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // the previous chunk's MFMAs are done with the LDS images
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      const int item = tid + n * NT, c = item / kUWT, px = item - c * kUWT;
      const float d00 = dok[n] ? gd[n][0] : 0.f, d01 = dok[n] ? gd[n][1] : 0.f, d10 = dok[n] ? gd[n][2] : 0.f,
                  d11 = dok[n] ? gd[n][3] : 0.f;
      // rows of A' dY: (d0), (d0 + d1), (-d1);  then the same along the columns
      const float r[3][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {-d10, -d11}};
      float* mp = m_s + c * kUWP + px;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        mp[(i * 3 + 0) * NCO * kUWP] = r[i][0];
        mp[(i * 3 + 1) * NCO * kUWP] = r[i][0] + r[i][1];
        mp[(i * 3 + 2) * NCO * kUWP] = -r[i][1];
      }
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int item = tid + n * NT, c = item / kUWT, px = item - c * kUWT;
      float d[9], tr[3][3];
#pragma unroll
      for (int e = 0; e < 9; ++e) d[e] = (((vrm >> (e / 3)) & 1u) && ((vcm[n] >> (e % 3)) & 1u)) ? xd[n][e] : 0.f;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        tr[0][q] = d[q] - d[3 + q];
        tr[1][q] = d[3 + q];
        tr[2][q] = d[3 + q] - d[6 + q];
      }
      float* vp = v_s + c * kUWP + px;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        vp[(i * 3 + 0) * NCIN * kUWP] = tr[i][0] - tr[i][1];
        vp[(i * 3 + 1) * NCIN * kUWP] = tr[i][1];
        vp[(i * 3 + 2) * NCIN * kUWP] = tr[i][1] - tr[i][2];
      }
    }
    if (chunk + 1 < c_hi) fetch(chunk + 1);   // in flight under this chunk's MFMAs
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float* mw = m_s + (cob * 32 + l31) * kUWP + hh;
    const float* vw = v_s + (cib * 32 + l31) * kUWP + hh;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
#pragma unroll
      for (int k = 0; k < kUWT / 2; ++k)
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(mw[p * NCO * kUWP + 2 * k], vw[p * NCIN * kUWP + 2 * k], acc[p], 0, 0, 0);
    }
  }
  // one slab per workgroup: partial[slot][p][co][ci]; lane = ci (coalesced), register rows = co
  float* ps = a.partial + (int64_t)blockIdx.x * 9 * a.Cout * a.Cin;
  const int ci = ci0 + cib * 32 + l31;
#pragma unroll
  for (int p = 0; p < 9; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + cob * 32 + acc_row(i, hh);
      if (co < a.Cout && ci < a.Cin) ps[((int64_t)p * a.Cout + co) * a.Cin + ci] = acc[p][i];
    }
}

// dU'[p][co][ci] = sum_slot partial (fixed order: 8 slot lanes, then a tree of 8), then dW = G'^T dU' G':
// dw[co][ci][a][b] = sum_{i in I(a)} sum_{j in I(b)} dU'[3i + j],  I(0) = {0,1}, I(1) = {1}, I(2) = {1,2}
__global__ __launch_bounds__(256) void upwino_wgrad_reduce_kernel(const float* __restrict__ partial, int nslots,
                                                                  int Cout, int Cin, float* __restrict__ dw) {
  __shared__ float red[8][32][9];
  const int64_t n = (int64_t)Cout * Cin;
  const int o = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int64_t e = (int64_t)blockIdx.x * 32 + o;         // (co, ci) pair
  float s[9];
#pragma unroll
  for (int p = 0; p < 9; ++p) s[p] = 0.f;
  if (e < n)
    for (int k = sg; k < nslots; k += 8)
#pragma unroll
      for (int p = 0; p < 9; ++p) s[p] += partial[((int64_t)k * 9 + p) * n + e];
#pragma unroll
  for (int p = 0; p < 9; ++p) red[sg][o][p] = s[p];
  __syncthreads();
  if (sg == 0 && e < n) {
    float u[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      float v = red[0][o][p];
#pragma unroll
      for (int k = 1; k < 8; ++k) v += red[k][o][p];
      u[p] = v;
    }
    float* out = dw + e * 9;
#pragma unroll
    for (int aa = 0; aa < 3; ++aa)
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const bool ia = aa == 0 ? i <= 1 : (aa == 1 ? i == 1 : i >= 1);
            const bool jb = bb == 0 ? j <= 1 : (bb == 1 ? j == 1 : j >= 1);
            if (ia && jb) v += u[i * 3 + j];
          }
        out[aa * 3 + bb] = v;
      }
  }
}

}  // namespace tgsr

using namespace tgsr;

static void upwgrad_plan(int B, int Cin, int Cout, int H, int W, int* nci, int* groups, int* gi, int* nslots, int* cpw,
                         int* nchunks) {
  *nci = (Cin % 64 == 0) ? 2 : 1;
  *gi = Cin / (32 * *nci);
  *groups = (Cout / 64) * *gi;
  *nchunks = B * H * ((W + kUWT - 1) / kUWT);
  int want = 512 / *groups;                  // two workgroups per CU in flight; fewer, longer K walks keep the slabs small
  if (*nchunks <= 1024 && want >= 64) want /= 2;   // the 32 x 32 upBlocks are slab-bound (tools/exp_wgrad.py: 54 -> 41 us, 46 -> 42)
  want = want * wgrad_split_pct() / 100;
  if (want < 1) want = 1;
  if (want > *nchunks) want = *nchunks;
  *cpw = (*nchunks + want - 1) / want;
  *nslots = (*nchunks + *cpw - 1) / *cpw;
}

extern "C" int64_t tgsr_upwino_wgrad_ws_elems(int B, int Cin, int Cout, int H, int W) {
  if (B < 1 || H < 1 || W < 1 || Cout < 64 || Cout % 64 != 0 || Cin < 32 || Cin % 32 != 0) return 0;   // shapes tgsr_upwino_wgrad refuses
  int nci, groups, gi, nslots, cpw, nchunks;
  upwgrad_plan(B, Cin, Cout, H, W, &nci, &groups, &gi, &nslots, &cpw, &nchunks);
  return (int64_t)nslots * 9 * Cout * Cin;
}

extern "C" int tgsr_upwino_wgrad(const float* grad_out, const float* x, int64_t x_bstride, int B, int Cin, int H, int W,
                                 int Cout, float* ws, float* dw, void* stream) {
  if (!grad_out || !x || !ws || !dw || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (Cout % 64 != 0 || Cin % 32 != 0) return TGSR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(grad_out) & 7) != 0) return TGSR_EUNSUPPORTED;
  UpWgradArgs a;
  a.g = grad_out; a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  int nci, groups, gi, nslots, cpw, nchunks;
  upwgrad_plan(B, Cin, Cout, H, W, &nci, &groups, &gi, &nslots, &cpw, &nchunks);
  a.chunks_x = (W + kUWT - 1) / kUWT; a.nchunks = nchunks; a.chunks_per_wg = cpw; a.cgroups_i = gi; a.partial = ws;
  hipStream_t s = as_stream(stream);
  dim3 grid(nslots, groups);
  if (nci == 2) hipLaunchKernelGGL(upwino_wgrad_kernel<2>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(upwino_wgrad_kernel<1>, grid, dim3(128), 0, s, a);
  int rc = note_launch(hipGetLastError(), "upwino_wgrad_kernel");
  if (rc) return rc;
  const int64_t n = (int64_t)Cout * Cin;
  hipLaunchKernelGGL(upwino_wgrad_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, ws, nslots, Cout, Cin,
                     dw);
  return note_launch(hipGetLastError(), "upwino_wgrad_reduce_kernel");
}
