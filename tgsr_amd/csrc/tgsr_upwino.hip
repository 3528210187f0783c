// upBlock (util.py:74-80: Upsample(x2, nearest) -> conv3x3 -> BatchNorm(eval) -> GLU) by Winograd F(2x2, 3x3) applied
// to the UP-SAMPLED image, with the up-sampling folded into the input transform.
//
// A 2x2 output tile at (2y, 2x) reads the 4x4 patch rows 2y-1 .. 2y+2 of the up-sampled image = low-resolution rows
// (y-1, y, y, y+1): the two middle rows (and columns) are equal.  With B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],
// [0,1,0,-1]] the row transform of (a, b, b, c) is (a - b, 2b, 0, b - c): the third transformed row and column vanish,
// so only 9 of the 16 Winograd positions carry a product:
//     2.25 multiplies per output instead of 4 (sub-pixel form, tgsr_upconv.hip) or 9 (direct).
// The factor 2 moves into the weights (exact), which leaves
//     V = T d T^T,  T = [[1,-1,0],[0,1,0],[0,1,-1]]   on the 3x3 low-resolution neighbourhood d of pixel (y, x)
//     U = G' g G'^T, G' = [[1,0,0],[1,1,1],[0,0,1]]   (tap sums: U[1][1] is the sum of all nine taps, ...)
//     Y = A'^T (U (.) V) A',  A'^T = [[1,1,0],[0,1,-1]]
// Everything fp32; U only adds weights, V only subtracts neighbours.
//
// Kernel structure = tgsr_winograd.hip (see there for the measurements behind it): MFMA 16x16x4, a wave owns 16 tiles
// (16 consecutive low-res pixels of a row = 2 x 32 outputs) x 32 couts (16 value + their 16 gate channels) with all 9
// positions live (72 accumulators), workgroup = 4 waves = 2 low-res rows x 2 cout halves; 37 KB LDS and 126 VGPRs
// put FOUR workgroups on a CU (4 waves per SIMD), which hides a copy's latency well enough that U (9 KB per stage,
// double buffered) is fetched one stage ahead and a stage simply ends on vmcnt(0) (3-5 % faster than three U buffers at
// 3 workgroups per CU); the raw rows [4 ci][4 rows][24 cols] (three buffers) still come two stages ahead.  Copies are
// LDS-DMA issued from inline asm (see tgsr_winograd.hip); V [3 rows i][4 ci][16 tiles][4] is computed one stage
// ahead by the two cout-half waves of a row (h = 0: rows i = 0, 1; h = 1: row i = 2) and shared through LDS.
#include "tgsr_common.h"

#include <type_traits>

namespace tgsr {

typedef __attribute__((address_space(3))) void* lds_ptru_t;
__device__ __attribute__((aligned(16))) float g_upw_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct UpwArgs {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;        // low-resolution input
  const float* upack;      // [stage][group][ A: [3 i][4 ci][2 halves][16][4] | B: [3 i][2 halves][4 ci][16][2] ]
  int Cout;
  const float* scale;
  const float* shift;
  float* out;              // [B][Cout/2][2H][2W]
  int64_t obs;
  int tiles_x, tiles_y, nstages, ngroups;
};

constexpr int kUCK = 4;                                   // input channels per stage
constexpr int kUTC = 24, kUTR = 4;                        // raw tile: 4 low-res rows x (16 + 8) columns per channel
// channel planes 112 words apart (96 used + 16 pad): a ds_read_b32 serves lanes 0-31 = channels lg, lg + 1 x 16 tiles in
// one cycle only if the two channels sit 16 banks apart (112 = 16 mod 32); at 96 = 0 mod 32 every raw read of the
// input transform was a 2-way conflict (profiles/r02i_fp32_pmc.csv: 19 % of this kernel's LDS cycles)
constexpr int kUPLANE = kUTR * kUTC + 16;
constexpr int kUA = 3 * kUCK * 2 * 16 * 4;                // 1536 floats: (j0cb0, j0cb1, j1cb0, j1cb1) per (i, ci, half, l)
constexpr int kUB = 3 * kUCK * 2 * 16 * 2;                // 768 floats: (j2cb0, j2cb1)
constexpr int kUU = kUA + kUB;                            // 2304 floats of U per stage = 9 DMA pieces
constexpr int kURawN = kUCK * kUPLANE;                    // 448 floats of raw input per stage (64 of them padding)
constexpr int kURaw = 512;                                // = 2 DMA pieces (waves 0 and 1)
constexpr int kUV = 3 * kUCK * 16 * 4;                    // V image of one tile row: 768 floats
constexpr int kUNB = 2;                                    // U buffers: the next stage's U is fetched during this one
constexpr int kUSmem = kUNB * kUU + 3 * kURaw + 2 * 2 * kUV + 2 * 64;

typedef float f32x4u __attribute__((ext_vector_type(4)));

template <bool GLU>   // GLU: value/gate blocks + sigmoid gate, out [B][Cout/2]; else plain affine, out [B][Cout]
__global__ __launch_bounds__(256, 4) void upwino_kernel(UpwArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[kUSmem];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wave >> 1, h = wave & 1;                  // low-res row / cout half of this wave
  // 1-D grid over (tile, cout group) with the group fastest INSIDE an XCD's contiguous run of ids: the groups of a tile
  // run back to back on the same XCD, so the second read of the tile's input rows hits that XCD's L2 (with the groups
  // on grid.y the re-read came from HBM: measured 1.2x the algorithmic traffic, profiles/r02a_fp32_pmc.csv)
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = t % a.ngroups;
  t /= a.ngroups;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * 2, x0 = tx * 16;                    // low-res origin of the workgroup tile
  const float* xb = a.x + (int64_t)b * a.xbs;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;
  float* us = smem;                                       // 3 x U stage
  float* raws = smem + kUNB * kUU;                        // 3 x raw stage
  float* vs = smem + kUNB * kUU + 3 * kURaw + w * 2 * kUV;   // this row's 2 V images
  float* aff_s = smem + kUNB * kUU + 3 * kURaw + 2 * 2 * kUV;

  auto dma16 = [&](const float* g, float* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptru_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory");
  };
  // raw: waves 0 and 1 copy floats [wave*256, wave*256 + 256) of the stage tile (384 used); running source pointer per
  // lane, out-of-image / past-the-tile lanes read the zero block with stride 0.  Cin % 4 == 0 (host-checked).
  const float* rptr = g_upw_zero;
  int rstep = 0;
  if (wave < 2) {
    const int e = (wave * 64 + lane) * 4;
    const int c = e / kUPLANE;
    const int rem = e - c * kUPLANE;
    const int r = rem / kUTC, j = rem - r * kUTC;
    const int gy = y0 - 1 + r, gx = x0 - 4 + j;
    const bool ok = e < kURawN && r < kUTR && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   // W % 4 == 0
    if (ok) {
      rptr = xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(gy * a.W + gx);
      rstep = (int)(kUCK * HW);
    }
  }
  auto issue_raw = [&](int buf) {
    if (wave < 2) {                                       // wave-uniform
      dma16(rptr, raws + buf * kURaw + wave * 256);
      rptr += rstep;
    }
  };
  // U: kUU / 256 = 9 pieces per stage: waves take pieces wave, wave + 4 and wave 0 also piece 8
  const float* ubase = a.upack + (int64_t)grp * kUU;
  const int64_t ustride = (int64_t)a.ngroups * kUU;
  const unsigned uoff = (unsigned)(lane * 16);
  auto issue_u = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k < 2 || wave == 0) {                           // wave-uniform
        const int piece = wave + 4 * k;
        const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptru_t)(us + buf * kUU + piece * 256));
        const float* g = ubase + piece * 256;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(uoff), "s"(g), "s"(l) : "memory");
      }
    }
    ubase += ustride;
  };

  // ---- input transform: lane = (tile l15, channel lg).  d = 3x3 low-res neighbourhood (rows y-1, y, y+1 of this wave's
  // row y = y0 + w; cols x-1, x, x+1).  Row pass t0 = a - b, t1 = b, t2 = b - c; column pass the same on each row.
  // h = 0 writes rows i = 0, 1 of V, h = 1 writes row i = 2; V[i][ci][tile][4] = (v_i0, v_i1, v_i2, 0).
  const int rlane = lg * kUPLANE + w * kUTC + 3 + l15;
  const int vwl = (lg * 16 + l15) * 4;
  auto t_read = [&](auto hc, const float* rawb, float (&d)[3][3]) {
    constexpr int H_ = decltype(hc)::value;
    const float* rp = rawb + rlane;
#pragma unroll
    for (int r = H_; r < 3; ++r)                          // h = 0 needs rows a, b (and b for i = 1); h = 1 rows b, c
#pragma unroll
      for (int q = 0; q < 3; ++q) d[r][q] = (H_ == 0 && r == 2) ? 0.f : rp[r * kUTC + q];
    if (H_ == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) d[0][q] = rp[q];
    }
  };
  auto t_write = [&](auto hc, const float (&d)[3][3], float* vdst) {
    constexpr int H_ = decltype(hc)::value;
    auto put = [&](int i, const float (&tr)[3]) {
      f32x4u v;
      v[0] = tr[0] - tr[1];
      v[1] = tr[1];
      v[2] = tr[1] - tr[2];
      v[3] = 0.f;
      *reinterpret_cast<f32x4u*>(vdst + i * (kUCK * 64) + vwl) = v;
    };
    if (H_ == 0) {
      float t0[3], t1[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        t0[q] = d[0][q] - d[1][q];
        t1[q] = d[1][q];
      }
      put(0, t0);
      put(1, t1);
    } else {
      float t2[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) t2[q] = d[1][q] - d[2][q];
      put(2, t2);
    }
  };

  if (tid < 128) {   // logical column lc = half*32 + block*16 + l -> global cout; [0,64) scale, [64,128) shift
    const int lc = tid & 63, hh = lc >> 5, cb = (lc >> 4) & 1, l = lc & 15;
    const int col = GLU ? (cb ? (a.Cout >> 1) : 0) + grp * 32 + hh * 16 + l : grp * 64 + lc;
    aff_s[tid] = a.scale ? (tid < 64 ? a.scale[col] : a.shift[col]) : (tid < 64 ? 1.f : 0.f);
  }

  f32x4u M[9][2];
#pragma unroll
  for (int p = 0; p < 9; ++p)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) M[p][cb][i] = 0.f;

  // ---- prologue: raw(0..2), U(0..1); transform raw(0) -> V[0]
  issue_raw(0);
  issue_u(0);
  if (a.nstages > 1) { issue_raw(1); }
  if (a.nstages > 2) issue_raw(2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  {
    float d[3][3];
    if (h) { t_read(std::integral_constant<int, 1>{}, raws, d); t_write(std::integral_constant<int, 1>{}, d, vs); }
    else { t_read(std::integral_constant<int, 0>{}, raws, d); t_write(std::integral_constant<int, 0>{}, d, vs); }
  }
  __syncthreads();

  const int ualane = ((lg * 2 + h) * 16 + l15) * 4;       // A part: [i][ci = lg][h][l15][4]
  // B part: [i][h][ci = lg][l15][2] - the two channels of a 32-lane ds_read_b64 group 32 words apart = all 64 banks (with
  // the half inside the channel, [ci][h], they were 64 words apart: the same banks, a 2-way conflict on every read)
  const int ublane = kUA + ((h * kUCK + lg) * 16 + l15) * 2;
  const int vlane = (lg * 16 + l15) * 4;                  // V: [i][ci = lg][l15][4]

  int b3 = 0;                                             // st % 3
  auto stage = [&](auto hc, auto more_c, auto more3_c, int st) {
    constexpr bool MORE = decltype(more_c)::value, MORE3 = decltype(more3_c)::value;
    const int par = st & 1;
    const int b3n = b3 == 2 ? 0 : b3 + 1;
    const float* ub = us + par * kUU;
    const float* vb = vs + par * kUV + vlane;
    f32x4u af[3], bf[3];
    float2 ag[3];
    float d[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      af[i] = *reinterpret_cast<const f32x4u*>(ub + ualane + i * (kUCK * 2 * 64));
      ag[i] = *reinterpret_cast<const float2*>(ub + ublane + i * (kUCK * 2 * 32));
      bf[i] = *reinterpret_cast<const f32x4u*>(vb + i * (kUCK * 64));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MORE) t_read(hc, raws + b3n * kURaw, d);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      M[i * 3 + 0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][0], bf[i][0], M[i * 3 + 0][0], 0, 0, 0);
      M[i * 3 + 0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][1], bf[i][0], M[i * 3 + 0][1], 0, 0, 0);
      M[i * 3 + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][2], bf[i][1], M[i * 3 + 1][0], 0, 0, 0);
      M[i * 3 + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][3], bf[i][1], M[i * 3 + 1][1], 0, 0, 0);
      M[i * 3 + 2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[i].x, bf[i][2], M[i * 3 + 2][0], 0, 0, 0);
      M[i * 3 + 2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[i].y, bf[i][2], M[i * 3 + 2][1], 0, 0, 0);
      if (i == 0) {                                       // the copies go where no LDS reads are queued
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) issue_u(par ^ 1);
        if (MORE3) issue_raw(b3);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 1) {
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) t_write(hc, d, vs + (par ^ 1) * kUV);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // U(st+1) (this stage's copy), raw(st+2) landed, V(st+1) written -> barrier
    if (MORE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    b3 = b3n;
  };
  auto run = [&](auto hc) {
    using T = std::true_type;
    using F = std::false_type;
    int st = 0;
    for (; st + 3 < a.nstages; ++st) stage(hc, T{}, T{}, st);
    for (; st + 1 < a.nstages; ++st) stage(hc, T{}, F{}, st);
    stage(hc, F{}, F{}, st);
  };
  if (h) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});

  // ---- output transform Y = A'^T M A' (A'^T = [[1,1,0],[0,1,-1]]) + affine + GLU; lane = tile (l15), register r of
  // block cb = cout cb-block channel 4*lg + r of this wave's half; the two column phases leave as one float2
  const int oy = 2 * (y0 + w), ox = 2 * (x0 + l15);
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const int64_t HWo = (int64_t)Ho * Wo;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  auto ytile = [&](int cb, int r, float (&y)[2][2]) {
    float rr[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float m0 = M[i * 3 + 0][cb][r], m1 = M[i * 3 + 1][cb][r], m2 = M[i * 3 + 2][cb][r];
      rr[i][0] = m0 + m1;
      rr[i][1] = m1 - m2;
    }
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      y[0][dx] = rr[0][dx] + rr[1][dx];
      y[1][dx] = rr[1][dx] - rr[2][dx];
    }
  };
  if (y0 + w < a.H && x0 + l15 < a.W) {
    if (GLU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lc = h * 32 + 4 * lg + r;               // value column; its gate is lc + 16
        float yv[2][2], yg[2][2];
        ytile(0, r, yv);
        ytile(1, r, yg);
        const float sv = aff_s[lc], tv = aff_s[64 + lc], sg = aff_s[lc + 16], tg = aff_s[64 + lc + 16];
        const int c = grp * 32 + h * 16 + 4 * lg + r;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const float o0 = (yv[dy][0] * sv + tv) * __builtin_amdgcn_rcpf(1.f + __expf(-(yg[dy][0] * sg + tg)));
          const float o1 = (yv[dy][1] * sv + tv) * __builtin_amdgcn_rcpf(1.f + __expf(-(yg[dy][1] * sg + tg)));
#ifdef TGSR_UPW_NOSTORE   // diagnostic build (tools/upw_store_cost.py): the epilogue's arithmetic without its stores (a.B is never negative)
          if (a.B < 0)
#endif
          *reinterpret_cast<float2*>(ob + (int64_t)c * HWo + (int64_t)(oy + dy) * Wo + ox) = make_float2(o0, o1);
        }
      }
    } else {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lc = h * 32 + cb * 16 + 4 * lg + r;
          float yv[2][2];
          ytile(cb, r, yv);
          const float sv = aff_s[lc], tv = aff_s[64 + lc];
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
            *reinterpret_cast<float2*>(ob + (int64_t)(grp * 64 + lc) * HWo + (int64_t)(oy + dy) * Wo + ox) =
                make_float2(yv[dy][0] * sv + tv, yv[dy][1] * sv + tv);
        }
    }
  }
}

// upack[stage][group][ A | B ]:  A [3 i][4 ci][2 halves][16 l][4] = (U[i][0] cb0, U[i][0] cb1, U[i][1] cb0, U[i][1] cb1),
// B [3 i][2 halves][4 ci][16 l][2] = (U[i][2] cb0, U[i][2] cb1);  U = G' g G'^T with G' = [[1,0,0],[1,1,1],[0,0,1]];
// cb 0 = value channel grp*32 + half*16 + l, cb 1 = its gate Cout/2 + grp*32 + half*16 + l.
__global__ void pack_upwino_weight_kernel(const float* __restrict__ wt, float* __restrict__ up, int Cout, int Cin,
                                          int glu, int64_t total) {
  const int ngrp = Cout / 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx % kUU);
    int64_t t = idx / kUU;
    const int grp = (int)(t % ngrp);
    const int st = (int)(t / ngrp);
    int i, ci, hh, l, j, cb;
    if (e < kUA) {
      const int q = e & 3;
      l = (e >> 2) & 15; hh = (e >> 6) & 1; ci = (e >> 7) & 3; i = e >> 9;
      j = q >> 1; cb = q & 1;
    } else {
      const int f = e - kUA;
      cb = f & 1; l = (f >> 1) & 15; ci = (f >> 5) & 3; hh = (f >> 7) & 1; i = f >> 8;
      j = 2;
    }
    const int co = glu ? (cb ? (Cout >> 1) : 0) + grp * 32 + hh * 16 + l : grp * 64 + hh * 32 + cb * 16 + l;
    const int c = st * kUCK + ci;
    float u = 0.f;
    if (c < Cin) {
      const float* g = wt + ((int64_t)co * Cin + c) * 9;
      float gi[3];   // row i of G' applied to the filter rows
      for (int k = 0; k < 3; ++k) {
        const float g0 = g[0 * 3 + k], g1 = g[1 * 3 + k], g2 = g[2 * 3 + k];
        gi[k] = i == 0 ? g0 : (i == 1 ? g0 + g1 + g2 : g2);
      }
      u = j == 0 ? gi[0] : (j == 1 ? gi[0] + gi[1] + gi[2] : gi[2]);
    }
    up[idx] = u;
  }
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_packed_upwino_weight_elems(int Cout, int Cin) {
  return (int64_t)((Cin + kUCK - 1) / kUCK) * (Cout / 64) * kUU;
}

extern "C" int tgsr_pack_upwino_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream) {
  if (!w || !upack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  if (Cout % 64 != 0) return TGSR_EUNSUPPORTED;
  const int64_t total = tgsr_packed_upwino_weight_elems(Cout, Cin);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_upwino_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, upack, Cout, Cin,
                     glu ? 1 : 0, total);
  return note_launch(hipGetLastError(), "pack_upwino_weight_kernel");
}

static int upwino_launch(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                         const float* scale, const float* shift, float* out, int64_t out_bstride, bool glu,
                         void* stream) {
  if (!x || !upack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  if (Cout % 64 != 0 || Cin % kUCK != 0) return TGSR_EUNSUPPORTED;
  if ((int64_t)H * W >= (1 << 26) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  if ((W & 3) || (x_bstride & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 7) ||
      (out_bstride & 1))
    return TGSR_EUNSUPPORTED;
  UpwArgs a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.upack = upack; a.Cout = Cout;
  a.scale = scale; a.shift = shift; a.out = out; a.obs = out_bstride;
  a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 1) / 2; a.nstages = (Cin + kUCK - 1) / kUCK;
  a.ngroups = Cout / 64;
  dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y * a.ngroups));
  if (glu) hipLaunchKernelGGL(upwino_kernel<true>, grid, dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(upwino_kernel<false>, grid, dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "upwino_kernel");
}

extern "C" int tgsr_upwino_glu_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack,
                                   int Cout, const float* scale, const float* shift, float* out, int64_t out_bstride,
                                   void* stream) {
  return upwino_launch(x, x_bstride, B, Cin, H, W, upack, Cout, scale, shift, out, out_bstride, true, stream);
}

extern "C" int tgsr_upwino_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack,
                               int Cout, const float* scale, const float* shift, float* out, int64_t out_bstride,
                               void* stream) {
  return upwino_launch(x, x_bstride, B, Cin, H, W, upack, Cout, scale, shift, out, out_bstride, false, stream);
}
