// Weight gradient of the plain 3x3 convolution (stride 1, pad 1) in the Winograd F(2x2, 3x3) domain.  With
// Y = A^T (U (.) V) A per 2x2 output tile t,
//     dU[p][co][ci] = sum_t dM[p][co][t] * V[p][ci][t],   dM = A dY A^T (4x4 from the tile's 2x2 output gradients),
//     V = B^T d B (4x4 from its 4x4 input patch),           p = one of the 16 positions,
// and dW = G^T dU G afterwards: 16 products per tile and channel pair instead of 36 (9 taps x 4 pixels) - 2.25x fewer
// multiplies than tgsr_conv3x3_wgrad.hip.  fp32 throughout; the transforms only add / subtract, G holds {0, 1, +-1/2}.
//
// GEMM view per position: M = co, N = ci, K = tiles.  Workgroup = NCOB co blocks x NCI ci blocks x 2 position halves of
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

// NCOB = 32-channel co blocks per workgroup (2: Cout % 64 == 0; 1: the 32-channel layers of NetG_highweight - model.py:258-262
// ResBlock(32), residual24 / 48 - which the direct-form kernel of tgsr_conv3x3_wgrad.hip served at 9 TFLOP/s), NCI = ci blocks.
// Waves = NCOB x 2 position halves x NCI.
//
// Software pipeline (round 4): a chunk's raw values (the 2x2 output gradients of a thread's dM items, the 4x4 input patches
// of its V items: up to 8 + 32 registers) are fetched one chunk AHEAD, right before the MFMAs of the current chunk, so that
// their latency runs under the matrix work.  Before, every chunk began with those loads exposed (~2 us of a ~7 us chunk on
// the 128^2 layers: the kernel sat at 0.24 of the fp32 MFMA peak with its MFMA phases 93 % dense).
template <int NCI, int NCOB>
__global__ __launch_bounds__(128 * NCI * NCOB) void wino_wgrad_kernel(WinoWgradArgs a) {
  constexpr int NT = 128 * NCI * NCOB, NCO = 32 * NCOB, NCIN = 32 * NCI;
  constexpr int ND = NCO * kWWT / NT, NV = NCIN * kWWT / NT;      // dM / V items per thread (1 or 2)
  static_assert(ND * NT == NCO * kWWT && NV * NT == NCIN * kWWT, "items divide evenly");
  constexpr int NBUF = (NCI == 2 && NCOB == 2) ? 2 : 1;   // 8-wave workgroups double-buffer (below); 144 of the CU's 160 KB
  __shared__ float m_s[NBUF * 16 * NCO * kWWP];     // dM [buffer][p][co][tile]
  __shared__ float v_s[NBUF * 16 * NCIN * kWWP];    // V  [buffer][p][ci][tile]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = wave % NCOB, ph = (wave / NCOB) & 1, cib = wave / (2 * NCOB);   // co block, position half, ci block
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
  float gd[ND][4], xd[NV][16];
  // Branch-free fetch: every load is unconditional from a CLAMPED address, out-of-image values are zeroed by a select.  The
  // first version wrote `in range ? load : 0` per element: hipcc made each of the 20 loads of a thread its own exec-masked
  // branch with 64-bit address arithmetic - ~1 200 VALU instructions per wave and chunk, 4.8 k cycles against the 2 k of the
  // chunk's MFMAs (the kernel's real bound: 0.24-0.37 of the MFMA peak however the phases were overlapped).  Row validity is
  // wave-uniform (a chunk is one tile row), only the first / last column of an image row needs a per-lane select.
  int dtl[ND], vtl[NV];
  const float* gplane[ND];
  const float* xplane[NV];
#pragma unroll
  for (int n = 0; n < ND; ++n) {
    const int item = tid + n * NT, c = item / kWWT;
    dtl[n] = item - c * kWWT;
    gplane[n] = a.g + (int64_t)(co0 + c) * HW;               // Cout % 32 == 0, Cin % 32 == 0 (launcher): every channel exists
  }
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const int item = tid + n * NT, c = item / kWWT;
    vtl[n] = item - c * kWWT;
    xplane[n] = a.x + (int64_t)(ci0 + c) * HW;
  }
  const int Hm1 = a.H - 1, Wm1 = a.W - 1;
  // validity of the fetched values (applied by transform(): a select at fetch time would make the wave wait for its loads
  // right there instead of under the MFMAs): per-lane column bits, wave-uniform row bits
  unsigned dcm[ND], vcm[NV], drm = 0, vrm = 0;
  auto fetch = [&](int chunk) {
    int t = chunk;
    const int cx = t % a.chunks_x;
    t /= a.chunks_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = 2 * ty, tx0 = cx * kWWT;
    const int64_t gb = (int64_t)b * a.Cout * HW, xb = (int64_t)b * a.xbs;
#pragma unroll
    for (int n = 0; n < ND; ++n) {           // dM item = (co, tile): the tile's 2x2 output gradients (zero outside the image)
      const int x = 2 * (tx0 + dtl[n]);
      const int xc0 = x < Wm1 ? x : Wm1, xc1 = x + 1 < Wm1 ? x + 1 : Wm1;
      dcm[n] = (x < a.W ? 1u : 0u) | (x + 1 < a.W ? 2u : 0u);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int gy = y0 + r;
        const float* rowp = gplane[n] + gb + (int64_t)(gy < a.H ? gy : Hm1) * a.W;
        gd[n][2 * r] = rowp[xc0];
        gd[n][2 * r + 1] = rowp[xc1];
      }
    }
    drm = 1u | (y0 + 1 < a.H ? 2u : 0u);
#pragma unroll
    for (int n = 0; n < NV; ++n) {           // V item = (ci, tile): its 4x4 input patch (rows y0-1 .. y0+2, cols x-1 .. x+2)
      const int x = 2 * (tx0 + vtl[n]);
      int gxc[4];
      vcm[n] = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gx = x - 1 + q;
        vcm[n] |= ((unsigned)gx < (unsigned)a.W ? 1u : 0u) << q;
        gxc[q] = gx < 0 ? 0 : (gx < Wm1 ? gx : Wm1);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int gy = y0 - 1 + rr;
        const float* rowp = xplane[n] + xb + (int64_t)(gy < 0 ? 0 : (gy < Hm1 ? gy : Hm1)) * a.W;
#pragma unroll
        for (int q = 0; q < 4; ++q) xd[n][4 * rr + q] = rowp[gxc[q]];
      }
    }
    vrm = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) vrm |= ((unsigned)(y0 - 1 + rr) < (unsigned)a.H ? 1u : 0u) << rr;
  };
  // registers (the fetched raw values) -> the transformed LDS images of buffer `buf`
  auto transform = [&](int buf) {
    float* mb = m_s + buf * (16 * NCO * kWWP);
    float* vb = v_s + buf * (16 * NCIN * kWWP);
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      const int item = tid + n * NT, c = item / kWWT, tl = item - c * kWWT;
      const float d00 = (dcm[n] & 1u) ? gd[n][0] : 0.f, d01 = (dcm[n] & 2u) ? gd[n][1] : 0.f;
      const float d10 = ((drm & 2u) && (dcm[n] & 1u)) ? gd[n][2] : 0.f, d11 = ((drm & 2u) && (dcm[n] & 2u)) ? gd[n][3] : 0.f;
      // A dY: rows (d0), (d0 + d1), (d0 - d1), (-d1); then the same along the columns
      const float r4[4][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {d00 - d10, d01 - d11}, {-d10, -d11}};
      float* mp = mb + c * kWWP + tl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mp[(i * 4 + 0) * NCO * kWWP] = r4[i][0];
        mp[(i * 4 + 1) * NCO * kWWP] = r4[i][0] + r4[i][1];
        mp[(i * 4 + 2) * NCO * kWWP] = r4[i][0] - r4[i][1];
        mp[(i * 4 + 3) * NCO * kWWP] = -r4[i][1];
      }
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int item = tid + n * NT, c = item / kWWT, tl = item - c * kWWT;
      float d[16], tr[4][4];
#pragma unroll
      for (int e = 0; e < 16; ++e) d[e] = (((vrm >> (e >> 2)) & 1u) && ((vcm[n] >> (e & 3)) & 1u)) ? xd[n][e] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        tr[0][q] = d[q] - d[8 + q];
        tr[1][q] = d[4 + q] + d[8 + q];
        tr[2][q] = d[8 + q] - d[4 + q];
        tr[3][q] = d[4 + q] - d[12 + q];
      }
      float* vp = vb + c * kWWP + tl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vp[(i * 4 + 0) * NCIN * kWWP] = tr[i][0] - tr[i][2];
        vp[(i * 4 + 1) * NCIN * kWWP] = tr[i][1] + tr[i][2];
        vp[(i * 4 + 2) * NCIN * kWWP] = tr[i][2] - tr[i][1];
        vp[(i * 4 + 3) * NCIN * kWWP] = tr[i][1] - tr[i][3];
      }
    }
  };
  auto mfmas = [&](int buf) {
    const float* mw = m_s + buf * (16 * NCO * kWWP) + (ph * 8 * NCO + cob * 32 + l31) * kWWP + hh;
    const float* vw = v_s + buf * (16 * NCIN * kWWP) + (ph * 8 * NCIN + cib * 32 + l31) * kWWP + hh;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
      for (int k = 0; k < kWWT / 2; ++k)
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(mw[p * NCO * kWWP + 2 * k], vw[p * NCIN * kWWP + 2 * k], acc[p], 0, 0, 0);
    }
  };
  // (barriers as s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads()' release fence waits for vmcnt(0) on gfx9 - loads and
  // stores share the counter - which would park every wave on its own prefetch)
  auto barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  if constexpr (NBUF == 2) {
    // Two LDS image sets: chunk k's MFMAs read one while chunk k + 1 is transformed into the other, ONE barrier per chunk - and
    // the two waves that share a SIMD take the two halves of an iteration in OPPOSITE order, so that one transforms (VALU, LDS stores) while the other multiplies: an fp32 MFMA and the same wave's
    // other instructions do not overlap on gfx950, two waves' do.  Single-buffered, every wave of the
    // workgroup - the only one a CU holds at 72 KB of LDS per set and 176 registers - transformed, then multiplied, in step:
    // 4.6 us per chunk on the 128^2 layers against 1.95 us of matrix work.
    //   Which waves share a SIMD is the dispatcher's business: every wave publishes its SIMD id (HW_ID[5:4]) and takes the
    //   parity of its rank among the workgroup's waves on the same SIMD.
    __shared__ int simd_of[2 * NCOB * NCI];
    const int simd = (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3u);
    if (lane == 0) simd_of[wave] = simd;
    __syncthreads();
    int rank = 0;
    for (int w = 0; w < wave; ++w) rank += simd_of[w] == simd ? 1 : 0;
    const bool mfma_first = rank & 1;
    if (c_lo < c_hi) {
      fetch(c_lo);
      transform(0);
      if (c_lo + 1 < c_hi) fetch(c_lo + 1);
      barrier();
    }
    for (int chunk = c_lo; chunk < c_hi; ++chunk) {
      const int buf = (chunk - c_lo) & 1;
      const bool more = chunk + 1 < c_hi;                       // workgroup-uniform
      if (mfma_first) mfmas(buf);
      if (more) {
        transform(buf ^ 1);
        if (chunk + 2 < c_hi) fetch(chunk + 2);
      }
      if (!mfma_first) mfmas(buf);
      barrier();
    }
  } else {
    if (c_lo < c_hi) fetch(c_lo);
    for (int chunk = c_lo; chunk < c_hi; ++chunk) {
      barrier();                                // the previous chunk's MFMAs are done with the LDS images
      transform(0);
      if (chunk + 1 < c_hi) fetch(chunk + 1);   // in flight under this chunk's MFMAs
      barrier();
      mfmas(0);
    }
  }
  // one slab per workgroup: partial[slot][p][co][ci]; lane = ci (coalesced), register rows = co.  Every (co, ci) of the
  // workgroup's blocks exists (Cout % 32 == 0, Cin % 32 == 0): unconditional stores off one base pointer.
  const int64_t CC = (int64_t)a.Cout * a.Cin;
  float* ps = a.partial + (int64_t)blockIdx.x * 16 * CC + (int64_t)(ph * 8) * CC + (int64_t)(co0 + cob * 32 + 4 * hh) * a.Cin +
              ci0 + cib * 32 + l31;
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) ps[p * CC + (int64_t)((i & 3) + 8 * (i >> 2)) * a.Cin] = acc[p][i];   // acc_row(i, hh) - 4 hh
}

// ---------------------------------------------------------------------------------------------------------------------
// The 64 co x 64 ci instance with the raw rows STAGED THROUGH LDS by LDS-DMA (round 4, W % 4 == 0, 16-byte aligned tensors).
// In-kernel stamps of the register-fetch kernel above (per wave and 8-tile chunk, 64 -> 128 @128^2): MFMA phase 2 180 cycles
// (2 048 of matrix work: dense), barrier 730, transform 2 250 - and 2 650 cycles just ISSUING the chunk's 20 scattered dword
// loads per thread: 160 wave-level load instructions per chunk and CU, each touching 8-16 cache lines for 64 useful dwords,
// back up in the vector-memory pipe while the wave sits at the issue.  The prefetched values had always arrived (vmcnt wait:
// ~0) - it was never latency.  Here a chunk's input is 32 DMA instructions per workgroup instead of 160 loads:
//   X  tile [4 rows][64 ci][24 cols]  (cols 2 tx0 - 4 .. 2 tx0 + 19: six aligned 16-byte pieces per (row, channel))
//   dY tile [2 rows][64 co][16 cols]  (four pieces per (row, channel))
// 2 048 pieces, 4 per thread, each wholly inside the image or fetched from a zero block (no bounds handling, no selects
// afterwards); ONE raw buffer (32 KB: chunk k + 1 is copied while chunk k is multiplied) and one set of transformed images
// (72 KB): two barriers per chunk, 104 KB of LDS.
typedef __attribute__((address_space(3))) void* lds_ptrg_t;
__device__ __attribute__((aligned(16))) float g_wgrad_zero[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int kWGXC = 24;                               // X tile columns
constexpr int kWGY = 2 * 64 * 16;                       // 2048 floats of dY

// NCI = 2: 64 co x 64 ci per workgroup (8 waves, 104 KB of LDS: one workgroup per CU); NCI = 1 (TGSR_WGRAD_TILE=32): 64 co x 32 ci
// (4 waves, 75 KB: TWO workgroups per CU, whose transform and MFMA phases can overlap each other's).
template <int NCI>
__global__ __launch_bounds__(256 * NCI) void wino_wgrad_dma_kernel(WinoWgradArgs a) {
  constexpr int NT = 256 * NCI, NW = 4 * NCI, NCO = 64, NCIN = 32 * NCI;
  constexpr int XF = 4 * NCIN * kWGXC;                  // floats of the X tile [4 rows][NCIN][24]
  constexpr int NPX = XF / 4, NPIECE = NPX + 512;       // 16-byte pieces: X, then dY [2 rows][64 co][16]
  constexpr int NK = NPIECE / NT;                       // pieces per thread: 4 | 5
  static_assert(NK * NT == NPIECE, "pieces divide evenly");
  constexpr int ND = NCO * kWWT / NT;                   // dM items per thread: 1 | 2
  __shared__ __attribute__((aligned(16))) float raw_s[XF + kWGY];
  __shared__ float m_s[16 * NCO * kWWP];      // dM [p][co][tile]
  __shared__ float v_s[16 * NCIN * kWWP];     // V  [p][ci][tile]
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

  // DMA plan: instruction q of a chunk is issued by wave q % NW as its (q / NW)-th; lane l of instruction q copies piece
  // P = 64 q + l.  X pieces P < NPX: (row = P / (6 NCIN), ci = (P % (6 NCIN)) / 6, j = P % 6); dY pieces P - NPX: (row = / 256,
  // co = (% 256) / 4, j = % 4).  The LDS image of a chunk is simply the pieces in order: X [row][ci][24], dY [row][co][16].
  int prow[NK], pcol[NK], pisx[NK];           // image row offset (relative to y0), first column (relative to 2 tx0), X or dY
  const float* pbase[NK];                     // channel plane of the piece (batch 0)
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int P = 64 * (wave + NW * k) + lane;
    if (P < NPX) {
      const int row = P / (6 * NCIN), r = P - row * (6 * NCIN), ci = r / 6, j = r - ci * 6;
      prow[k] = row - 1; pcol[k] = 4 * j - 4; pisx[k] = 1;
      pbase[k] = a.x + (int64_t)(ci0 + ci) * HW;
    } else {
      const int Q = P - NPX, row = Q >> 8, r = Q & 255, co = r >> 2, j = r & 3;
      prow[k] = row; pcol[k] = 4 * j; pisx[k] = 0;
      pbase[k] = a.g + (int64_t)(co0 + co) * HW;
    }
  }
  const int64_t gbs = (int64_t)a.Cout * HW;
  auto issue = [&](int chunk) {
    int t = chunk;
    const int cx = t % a.chunks_x;
    t /= a.chunks_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = 2 * ty, xo = 2 * cx * kWWT;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int gy = y0 + prow[k], gx = xo + pcol[k];
      const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;          // W % 4 == 0: a piece is all in or all out
      const float* src = ok ? pbase[k] + (int64_t)b * (pisx[k] ? a.xbs : gbs) + (int64_t)gy * a.W + gx : g_wgrad_zero;
      const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptrg_t)(raw_s + (wave + NW * k) * 256));
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory");
    }
  };
  // ONE raw buffer: a chunk's copies are issued right behind the barrier that ends the transform of the previous chunk (the
  // last reader of the buffer) and fly under that chunk's MFMAs.
  if (c_lo < c_hi) issue(c_lo);
  for (int chunk = c_lo; chunk < c_hi; ++chunk) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of `chunk` landed; its MFMA operands are read
    __builtin_amdgcn_s_barrier();             // every wave's pieces landed AND the previous chunk's MFMAs are done with the images
#pragma unroll
    for (int n = 0; n < ND; ++n) {            // dM item (co, tile)
      const int item = tid + n * NT, c = item >> 3, tl = item & 7;
      const float* yr = raw_s + XF + c * 16 + 2 * tl;
      const float2 g0 = *reinterpret_cast<const float2*>(yr), g1 = *reinterpret_cast<const float2*>(yr + 64 * 16);
      // A dY: rows (d0), (d0 + d1), (d0 - d1), (-d1); then the same along the columns
      const float r4[4][2] = {{g0.x, g0.y}, {g0.x + g1.x, g0.y + g1.y}, {g0.x - g1.x, g0.y - g1.y}, {-g1.x, -g1.y}};
      float* mp = m_s + c * kWWP + tl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mp[(i * 4 + 0) * NCO * kWWP] = r4[i][0];
        mp[(i * 4 + 1) * NCO * kWWP] = r4[i][0] + r4[i][1];
        mp[(i * 4 + 2) * NCO * kWWP] = r4[i][0] - r4[i][1];
        mp[(i * 4 + 3) * NCO * kWWP] = -r4[i][1];
      }
    }
    {                                         // V item (ci, tile): one per thread
      const int tc = tid >> 3, tl = tid & 7;
      const float* xr = raw_s + tc * kWGXC + 2 * tl + 3;          // row 0, column x - 1 of this item's 4x4 patch
      float d[4][4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float* rp = xr + rr * (NCIN * kWGXC);
        d[rr][0] = rp[0];
        const float2 mid = *reinterpret_cast<const float2*>(rp + 1);            // columns x, x + 1: 8-byte aligned (2 tl + 4)
        d[rr][1] = mid.x; d[rr][2] = mid.y;
        d[rr][3] = rp[3];
      }
      float tr[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        tr[0][q] = d[0][q] - d[2][q];
        tr[1][q] = d[1][q] + d[2][q];
        tr[2][q] = d[2][q] - d[1][q];
        tr[3][q] = d[1][q] - d[3][q];
      }
      float* vp = v_s + tc * kWWP + tl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vp[(i * 4 + 0) * NCIN * kWWP] = tr[i][0] - tr[i][2];
        vp[(i * 4 + 1) * NCIN * kWWP] = tr[i][1] + tr[i][2];
        vp[(i * 4 + 2) * NCIN * kWWP] = tr[i][2] - tr[i][1];
        vp[(i * 4 + 3) * NCIN * kWWP] = tr[i][1] - tr[i][3];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // images complete; the raw buffer is free
    if (chunk + 1 < c_hi) issue(chunk + 1);   // in flight under this chunk's MFMAs
    const float* mw = m_s + (ph * 8 * NCO + cob * 32 + l31) * kWWP + hh;
    const float* vw = v_s + (ph * 8 * NCIN + cib * 32 + l31) * kWWP + hh;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
      for (int k = 0; k < kWWT / 2; ++k)
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(mw[p * NCO * kWWP + 2 * k], vw[p * NCIN * kWWP + 2 * k], acc[p], 0, 0, 0);
    }
  }
  const int64_t CC = (int64_t)a.Cout * a.Cin;
  float* ps = a.partial + (int64_t)blockIdx.x * 16 * CC + (int64_t)(ph * 8) * CC + (int64_t)(co0 + cob * 32 + 4 * hh) * a.Cin +
              ci0 + cib * 32 + l31;
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) ps[p * CC + (int64_t)((i & 3) + 8 * (i >> 2)) * a.Cin] = acc[p][i];
}

// (Tried on top of this kernel and measured SLOWER, 283 vs 252 us on 64 -> 128 @128^2, so not kept: both phases overlapped again -
// V images double-buffered, the dM images dropped (the A operand computed by the lane that needs it from the raw dY tile in LDS: 138
// KB in all), SIMD partners taking transform and MFMAs in opposite order, one barrier per chunk.  A wave's VALU / LDS work does not
// run under its SIMD partner's fp32 MFMAs any better than under its own: the stamps of the register-fetch kernel had already shown
// a transform phase of 100 instructions taking 2 250 cycles beside a multiplying partner.  Fewer non-MFMA instructions per chunk
// is the only lever on this pipe.)

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

// TGSR_WGRAD_TILE = 32 | 64: force the DMA-staged kernel's 64 co x 32 ci (two workgroups per CU) or 64 x 64 form on the
// Cout % 64 == 0, Cin % 64 == 0 layers (0 / unset: by layer size, see wwgrad_plan)
static int wgrad_tile() {
  static const int v = [] { const char* e = getenv("TGSR_WGRAD_TILE"); return e ? atoi(e) : 0; }();
  return v;
}

static void wwgrad_plan(int B, int Cin, int Cout, int H, int W, int* nci, int* groups, int* gi, int* nslots, int* cpw,
                        int* nchunks, int* tiles_y, int* chunks_x) {
  *tiles_y = (H + 1) / 2;
  *chunks_x = ((W + 1) / 2 + kWWT - 1) / kWWT;
  *nchunks = B * *tiles_y * *chunks_x;
  // The 64 x 32 tile (two 4-wave workgroups per CU) measured 3-20 % faster than the 64 x 64 one on the 32^2 and 64^2 layers
  // (<= 2 048 chunks at batch 16: 47 -> 37, 35 -> 31, 83 -> 79, 57 -> 54 us) and 3 % slower on the 128^2 ones
  // (tools/exp_wgrad.py; TGSR_WGRAD_TILE=32 | 64 forces one of them).
  const bool can32 = Cin % 64 == 0 && Cout % 64 == 0 && W % 4 == 0;
  const int force = wgrad_tile();
  const bool use32 = can32 && (force == 32 || (force == 0 && *nchunks <= 2048));
  *nci = (Cin % 64 == 0 && !use32) ? 2 : 1;
  *gi = Cin / (32 * *nci);
  const int ncob = (Cout % 64 == 0) ? 2 : 1;
  *groups = (Cout / (32 * ncob)) * *gi;
  int want = 256 / *groups;                  // one 8-wave workgroup per CU: fewer, longer K walks keep the slabs small
  if (use32) want = 512 / *groups;           // two 4-wave workgroups per CU
  // the 32 x 32 layers (<= 512 chunks at batch 16) are slab-bound - 2-4 chunks of work per workgroup against a 64-KB..512-KB
  // slab written and re-read: half the split measured 5-20 % faster there, slower everywhere else (tools/exp_wgrad.py)
  if (*nchunks <= 512 && want >= 64) want /= 2;
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
  if (Cout % 32 != 0 || Cin % 32 != 0) return TGSR_EUNSUPPORTED;
  WinoWgradArgs a;
  a.g = grad_out; a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  int nci, groups, gi, nslots, cpw, nchunks;
  wwgrad_plan(B, Cin, Cout, H, W, &nci, &groups, &gi, &nslots, &cpw, &nchunks, &a.tiles_y, &a.chunks_x);
  a.nchunks = nchunks; a.chunks_per_wg = cpw; a.cgroups_i = gi; a.partial = ws;
  hipStream_t s = as_stream(stream);
  dim3 grid(nslots, groups);
  const bool co64 = Cout % 64 == 0;
  // the DMA-staged instance: aligned planes and rows (TGSR_WGRAD_DMA=0 keeps the register-fetch kernel for A/B runs)
  static const bool dma_on = [] { const char* e = getenv("TGSR_WGRAD_DMA"); return !(e && e[0] == '0'); }();
  const bool dma_ok = dma_on && W % 4 == 0 && ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 &&
                      x_bstride % 4 == 0;
  if (nci == 2 && co64 && dma_ok) hipLaunchKernelGGL(wino_wgrad_dma_kernel<2>, grid, dim3(512), 0, s, a);
  else if (nci == 1 && co64 && dma_ok && Cin % 64 == 0)      // the plan chose the 64 x 32 tile for a 64-ci-multiple layer
    hipLaunchKernelGGL(wino_wgrad_dma_kernel<1>, grid, dim3(256), 0, s, a);
  else if (nci == 2 && co64) hipLaunchKernelGGL((wino_wgrad_kernel<2, 2>), grid, dim3(512), 0, s, a);
  else if (nci == 2) hipLaunchKernelGGL((wino_wgrad_kernel<2, 1>), grid, dim3(256), 0, s, a);
  else if (co64) hipLaunchKernelGGL((wino_wgrad_kernel<1, 2>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((wino_wgrad_kernel<1, 1>), grid, dim3(128), 0, s, a);
  int rc = note_launch(hipGetLastError(), "wino_wgrad_kernel");
  if (rc) return rc;
  const int64_t n = (int64_t)Cout * Cin;
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, ws, nslots, Cout, Cin,
                     dw);
  return note_launch(hipGetLastError(), "wino_wgrad_reduce_kernel");
}
