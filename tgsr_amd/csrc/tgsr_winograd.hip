// 3x3 convolution (stride 1, pad 1) by Winograd F(2x2, 3x3) in fp32 on the MFMA units, with the same fused epilogues
// as tgsr_conv3x3.hip (BatchNorm-eval affine, GLU | residual).  Inference path of ResBlock.block / residual24/48
// (util.py:110-130, model.py:229-232).
//
//   Y(2x2) = A^T [ (G g G^T) (.) (B^T d B) ] A     per 4x4 input tile d, 3x3 filter g:
// 16 multiplies per 4 outputs instead of 36 -> 2.25x fewer MFMAs than the direct form.  All transform matrices are
// {0, +-1, +-1/2}: measured end to end on the shipped checkpoint (oracle with this algorithm in fp32 vs fp64) the
// error is indistinguishable from the direct fp32 convolution (fine2: 2.95e-5 vs 2.92e-5 max, 5.4e-7 vs 4.8e-7 mean).
//
// Mapping: see the geometry comment above the kernel.
#include "tgsr_common.h"

#include <cstdlib>
#include <type_traits>

namespace tgsr {

typedef __attribute__((address_space(3))) void* lds_ptrw_t;
typedef const __attribute__((address_space(1))) void* glb_ptrw_t;
__device__ __attribute__((aligned(16))) float g_wino_zero[4] = {0.f, 0.f, 0.f, 0.f};

#ifdef TGSR_WINO_STAMPS
// Diagnostic build only (tools/wino_stamps.py): per-workgroup cycle stamps, never compiled into the shipped library.
__device__ unsigned long long g_wstamps[8 * 8192];
#define TGSR_WSTAMP(k)                                                                                  \
  do {                                                                                                  \
    const int bid_ = blockIdx.x;                                                                        \
    if (threadIdx.x == 0 && bid_ < 8192) {                                                              \
      g_wstamps[bid_ * 8 + (k)] = __builtin_amdgcn_s_memtime();                                         \
      if ((k) == 0) g_wstamps[bid_ * 8 + 6] = __builtin_amdgcn_s_memrealtime();                         \
      if ((k) == 3) g_wstamps[bid_ * 8 + 7] = __builtin_amdgcn_s_memrealtime();                         \
    }                                                                                                   \
  } while (0)
#else
#define TGSR_WSTAMP(k)
#endif

struct WinoArgs {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;
  const float* upack;     // [stage][group][pos pair 8][ci 4][cout half 2][16][4]
  int Cout;
  const float* scale;
  const float* shift;
  const float* res;
  int64_t rbs;
  float* out;
  int64_t obs;
  int tiles_x, tiles_y, nstages, ngroups;
  float* stat;            // STATS: [Cout][nslots][2] per-wave (sum, sum of squares) of the outputs, else unused
  int nslots;
};

// Geometry.  MFMA 16x16x4 with ALL 16 transformed positions live: a wave owns 16 tiles (one tile row = 2 x 32 output
// pixels) x 32 output channels (2 blocks of 16: GLU value block + its gate block, or 2 plain blocks)
// = 16 pos x 2 x 4 = 128 accumulator registers, so TWO workgroups (8 waves) fit a CU.  That matters: measured on
// gfx950, a wave's own VALU / LDS / DMA instructions do not overlap its own MFMAs (a variant of this kernel with one
// 256-accumulator wave per SIMD lost ~7 cycles per non-MFMA instruction however they were scheduled), only another
// wave's do.  A workgroup = 4 waves = 2 tile rows x 2 cout halves (4 x 32 outputs x 64 couts):
//   stage = 4 input channels = one MFMA k-step per position.
//   U   [8 pos pairs][4 ci][2 cout halves][16][4] (16 KB, triple buffered): a linear LDS-DMA copy of the
//       pre-transformed pack; one ds_read_b128 yields a lane's A fragments of 2 positions x 2 cout blocks.
//   raw [4 ci][6 rows][40 cols] (3.75 KB, triple buffered, LDS-DMA in 16-byte pieces; the tile starts 4 columns left
//       of the outputs so every piece is aligned and wholly inside or outside the image).
//   V   per tile row [8 pos pairs][4 ci][16 tiles][2] (4 KB, double buffered): the input transform B^T d B, shared
//       by the two cout-half waves of the row; each of them computes two of the four rows of V for one
//       (tile, channel) per lane, one stage ahead.
// One barrier per stage publishes U(st+1), raw(st+2) and V(st+1).  LDS 77 KB per workgroup.
// A second shape of the same kernel (NH = 1) serves layers with 32 output channels per group (Cout % 64 != 0): the 4
// waves are 4 tile rows x ONE cout half (8 x 32 outputs x 32 couts), each wave computes its own V (all four rows).
constexpr int kWCK = 4;                                  // input channels per stage
constexpr int kWTC = 40;                                 // raw tile columns: 32 + 8
constexpr int kWV = 8 * kWCK * 32;                       // V image of one tile row: 1024 floats
template <int NH>
struct WinoGeo {
  static constexpr int ROWS = NH == 2 ? 2 : 4;           // tile rows per workgroup
  static constexpr int TR = 2 * ROWS + 2;                // raw rows: 6 | 10
  static constexpr int PLANE = TR * kWTC;
  static constexpr int U = 8 * kWCK * NH * 64;           // floats of U per stage: 4096 | 2048
  static constexpr int UK = U / 256 / 4;                 // U DMA pieces (1 KB) per wave per stage: 4 | 2
  static constexpr int RAWN = kWCK * PLANE;              // 960 | 1600 floats of raw input per stage
  static constexpr int RP = (RAWN + 255) / 256;          // raw DMA pieces per stage: 4 | 7
  static constexpr int RAW = RP * 256;                   // floats per raw buffer (the last piece partly used)
  static constexpr int RK = (RP + 3) / 4;                // raw DMA slots per wave: 1 | 2
  static constexpr int SMEM = 3 * U + 3 * RAW + ROWS * 2 * kWV + 2 * 64;
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool GLU, int NH, bool STATS = false>
__global__ __launch_bounds__(256, 2) void wino_conv3x3_kernel(WinoArgs a) {
  static_assert(!(GLU && STATS), "batch statistics are taken of the raw (plain-epilogue) convolution output");
  using Geo = WinoGeo<NH>;
  constexpr int kWPLANE = Geo::PLANE, kWU = Geo::U, kWUK = Geo::UK, kWRawN = Geo::RAWN, kWRaw = Geo::RAW;
  __shared__ __attribute__((aligned(16))) float smem[Geo::SMEM];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = NH == 2 ? wave >> 1 : wave, h = NH == 2 ? wave & 1 : 0;     // tile row / cout half of this wave
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
  const int y0 = ty * (2 * Geo::ROWS), x0 = tx * 32;     // output origin of the workgroup tile
  const float* xb = a.x + (int64_t)b * a.xbs;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;
  float* us = smem;                                      // 3 x U stage
  float* raws = smem + 3 * kWU;                          // 3 x raw stage
  float* vs = smem + 3 * kWU + 3 * kWRaw + w * 2 * kWV;  // this tile row's 2 V images
  float* aff_s = smem + 3 * kWU + 3 * kWRaw + Geo::ROWS * 2 * kWV;
  TGSR_WSTAMP(0);

  // ---- DMA plan.  raw: wave q copies floats [q*256, q*256 + 256) of the stage tile.  Per lane a running source
  // pointer and its per-stage stride; out-of-image (or past-the-tile) lanes read the zero block with stride 0, so
  // issuing a piece is branch-free.  Cin % 4 == 0 (host-checked): a stage never reads past the last channel.
  const float* rptr[Geo::RK];
  int rstep[Geo::RK];
#pragma unroll
  for (int k = 0; k < Geo::RK; ++k) {                    // piece wave + 4k of the stage tile
    const int e = ((wave + 4 * k) * 64 + lane) * 4;      // first float of this lane's 16-byte piece
    const int c = e / kWPLANE;
    const int rem = e - c * kWPLANE;
    const int r = rem / kWTC, j = rem - r * kWTC;
    const int gy = y0 - 1 + r, gx = x0 - 4 + j;
    const bool ok = e < kWRawN && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   // W % 4 == 0
    rptr[k] = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(gy * a.W + gx) : g_wino_zero;
    rstep[k] = ok ? (int)(kWCK * HW) : 0;
  }
  const int nraw = (Geo::RP - 1 - wave) / 4 + 1;         // raw pieces this wave copies per stage (wave-uniform)
  // The copies are issued from inline assembly: hipcc cannot tell which LDS reads a global_load_lds may alias and
  // puts s_waitcnt vmcnt(0) in front of the next ds_read, which would serialize every stage on its own prefetch.
  // All vmcnt / lgkmcnt synchronization of the copies is therefore explicit (the stage-end barrier below).
  auto dma16 = [&](const float* g, float* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptrw_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory");
  };
  auto issue_raw = [&](int buf) {                        // issued for stages 0, 1, 2, ... in order
#pragma unroll
    for (int k = 0; k < Geo::RK; ++k) {
      if (4 * k + 3 < Geo::RP || wave + 4 * k < Geo::RP) {   // wave-uniform; only the last slot can be missing
        dma16(rptr[k], raws + buf * kWRaw + (wave + 4 * k) * 256);
        rptr[k] += rstep[k];
      }
    }
  };
  // U: scalar base (advances one stage per call) + four per-lane byte offsets, so a stage's copies cost no VALU work
  const float* ubase = a.upack + (int64_t)grp * kWU;     // stage 0 of this group
  const int64_t ustride = (int64_t)a.ngroups * kWU;
  unsigned uoff[kWUK];
#pragma unroll
  for (int k = 0; k < kWUK; ++k) uoff[k] = (unsigned)(((wave + 4 * k) * 64 + lane) * 16);
  auto issue_u = [&](int buf) {                          // issued for stages 0, 1, 2, ... in order
#pragma unroll
    for (int k = 0; k < kWUK; ++k) {
      const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptrw_t)(us + buf * kWU + (wave + 4 * k) * 256));
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(uoff[k]), "s"(ubase), "s"(l)
                   : "memory");
    }
    ubase += ustride;
  };

  // ---- input transform: lane = (tile l15, channel lg); this wave computes rows i = 2h, 2h+1 of V = B^T d B
  //   h = 0: i=0: d0 - d2, i=1: d1 + d2        h = 1: i=2: d2 - d1, i=3: d1 - d3     (d_r = raw row r of the 4x4 patch)
  // as  first = A - B,  second = C +- D  with the rows A..D picked at compile time (the stage loop is instantiated per
  // h), then the column pass, written as float2 (positions 4i + {0,1} and 4i + {2,3}).
  const int rlane = lg * kWPLANE + (2 * w) * kWTC + 2 * l15 + 3;
  const int vwl0 = lg * 32 + l15 * 2;                                  // + (4 * half) * (kWCK * 32): pos pair (2 half + ii) * 2
  auto t_read = [&](auto hc, const float* rawb, float (&d)[4][4]) {    // rows A, B, C, D
    constexpr int H = decltype(hc)::value;
    constexpr int row[4] = {H ? 2 : 0, H ? 1 : 2, 1, H ? 3 : 2};
    const float* rp = rawb + rlane;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) d[r][q] = rp[row[r] * kWTC + q];
  };
  auto t_write = [&](auto hc, const float (&d)[4][4], float* vdst) {
    constexpr int H = decltype(hc)::value;
    float tr[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tr[0][q] = d[0][q] - d[1][q];
      tr[1][q] = H ? d[2][q] - d[3][q] : d[2][q] + d[3][q];
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      float* vp = vdst + vwl0 + (4 * H + 2 * ii) * (kWCK * 32);
      *reinterpret_cast<float2*>(vp) = make_float2(tr[ii][0] - tr[ii][2], tr[ii][1] + tr[ii][2]);
      *reinterpret_cast<float2*>(vp + kWCK * 32) = make_float2(tr[ii][2] - tr[ii][1], tr[ii][1] - tr[ii][3]);
    }
  };

  if (tid < 128) {   // logical column lc = half*32 + block*16 + l of this workgroup -> global cout; [0,64) scale, [64,128) shift
    const int lc = tid & 63, hh = lc >> 5, cb = (lc >> 4) & 1, l = lc & 15;
    // (NH = 1: only columns [0, 32) of the "half 0" rows are used)
    int col = GLU ? (cb ? (a.Cout >> 1) : 0) + grp * (16 * NH) + hh * 16 + l : grp * (32 * NH) + lc;
    if (col >= a.Cout) col = 0;
    aff_s[tid] = a.scale ? (tid < 64 ? a.scale[col] : a.shift[col]) : (tid < 64 ? 1.f : 0.f);
  }

  f32x4v M[16][2];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) M[p][cb][i] = 0.f;

  // ---- prologue: raw(0..2), U(0..1); transform raw(0) -> V[0]
  issue_raw(0);
  issue_u(0);
  if (a.nstages > 1) { issue_raw(1); issue_u(1); }
  if (a.nstages > 2) issue_raw(2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everything issued so far is visible
  {
    float d[4][4];
    if (NH == 1 || !h) { t_read(std::integral_constant<int, 0>{}, raws, d); t_write(std::integral_constant<int, 0>{}, d, vs); }
    if (NH == 1 || h) { t_read(std::integral_constant<int, 1>{}, raws, d); t_write(std::integral_constant<int, 1>{}, d, vs); }
  }
  __syncthreads();                                       // V(0) visible to the other cout half
  TGSR_WSTAMP(1);

  const int ulane = (lg * NH + h) * 64 + l15 * 4;        // A: U[pp][ci = lg][h][l15][4]
  const int vlane = lg * 32 + l15 * 2;                   // B: V[pp][ci = lg][l15][2]

  // One stage: MFMAs on U(st), V(st); transform raw(st+1) -> V(st+1) [MORE]; fetch U(st+2) [MORE2] and raw(st+3)
  // [MORE3].  A stage is only ~1-2k cycles, shorter than a DMA round trip, so the copies run TWO stages ahead (three
  // U / raw buffers) and the wait that ends a stage is counted: it leaves this stage's own copies in flight.
  // Order inside a stage (pinned with sched_barrier: left alone, hipcc re-uses the fragment registers and so puts
  // every ds_read right in front of its MFMA): all 12 fragment reads, the 5 copies, the 8 raw reads of the transform;
  // 16 MFMAs; the transform arithmetic + 2 V writes; 16 MFMAs; wait + barrier.
  const int oy = y0 + 2 * w, ox = x0 + 2 * l15;
  const int64_t HWo = (int64_t)a.H * a.W;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  const float* __restrict__ rb = a.res ? a.res + (int64_t)b * a.rbs : nullptr;
  float2 rres[GLU ? 1 : 2][4][2];                        // residual tile of this lane (plain epilogue)
  auto load_res = [&]() {                                // issued before the last stage, consumed in the epilogue
    if (!GLU) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            rres[cb][i][dy] = make_float2(0.f, 0.f);
            if (rb && ox < a.W && oy + dy < a.H)
              rres[cb][i][dy] = *reinterpret_cast<const float2*>(
                  rb + (int64_t)(grp * (32 * NH) + h * 32 + cb * 16 + 4 * lg + i) * HWo + (int64_t)(oy + dy) * a.W + ox);
          }
    }
  };

  int b3 = 0;                                            // st % 3
  auto stage = [&](auto hc, auto more_c, auto more2_c, auto more3_c, int st) {
    constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value, MORE3 = decltype(more3_c)::value;
    const int par = st & 1;
    const int b3n = b3 == 2 ? 0 : b3 + 1, b3p = b3 == 0 ? 2 : b3 - 1;     // (st+1) % 3, (st+2) % 3
    const float* ub = us + b3 * kWU + ulane;
    const float* vb = vs + par * kWV + vlane;
    f32x4v af[8];
    float2 bf[8];
    float d[4][4], d2[4][4];                             // d2: NH = 1, this wave transforms both row halves
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      af[pp] = *reinterpret_cast<const f32x4v*>(ub + pp * kWCK * NH * 64);
      bf[pp] = *reinterpret_cast<const float2*>(vb + pp * kWCK * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MORE) {
      if (NH == 2) {
        t_read(hc, raws + b3n * kWRaw, d);
      } else {
        t_read(H0{}, raws + b3n * kWRaw, d);
        t_read(H1{}, raws + b3n * kWRaw, d2);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      M[2 * pp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pp][0], bf[pp].x, M[2 * pp][0], 0, 0, 0);
      M[2 * pp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pp][1], bf[pp].x, M[2 * pp][1], 0, 0, 0);
      M[2 * pp + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pp][2], bf[pp].y, M[2 * pp + 1][0], 0, 0, 0);
      M[2 * pp + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pp][3], bf[pp].y, M[2 * pp + 1][1], 0, 0, 0);
      if (pp == 1 || pp == 5) {                          // the copies go where no LDS reads are queued
        __builtin_amdgcn_sched_barrier(0);
        if (pp == 1 && MORE2) issue_u(b3p);              // U(st+2) replaces U(st-1)
        if (pp == 5 && MORE3) issue_raw(b3);             // raw(st+3) replaces raw(st), consumed one stage ago
        __builtin_amdgcn_sched_barrier(0);
      }
      if (pp == 3) {
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) {
          if (NH == 2) {
            t_write(hc, d, vs + (par ^ 1) * kWV);
          } else {
            t_write(H0{}, d, vs + (par ^ 1) * kWV);
            t_write(H1{}, d2, vs + (par ^ 1) * kWV);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // U(st+1) and raw(st+2) (issued a stage ago) landed, V(st+1) written -> barrier; this stage's copies stay in flight
    // (no LDS is reused after the last stage: no barrier there, and the residual tile prefetched before it stays in flight)
    if (MORE) {
      if (NH == 2) {
        constexpr int kInFlight = (MORE2 ? kWUK : 0) + (MORE3 ? 1 : 0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kInFlight) : "memory");
      } else {                                           // NH = 1: wave 3 copies one raw piece, the others two
        constexpr int kU = MORE2 ? kWUK : 0;
        if (!MORE3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kU) : "memory");
        else if (nraw == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kU + 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kU + 1) : "memory");
      }
    }
    b3 = b3n;
  };
  auto run = [&](auto hc) {
    using T = std::true_type;
    using F = std::false_type;
    int st = 0;
#if defined(TGSR_WINO_EXP)   // diagnostic builds (wrong results): 1 = MFMAs + fragment reads only, 2 = + transform, 3 = + copies
    for (; st + 3 < a.nstages; ++st)
      stage(hc, std::integral_constant<bool, TGSR_WINO_EXP == 2>{}, std::integral_constant<bool, TGSR_WINO_EXP == 3>{},
            std::integral_constant<bool, TGSR_WINO_EXP == 3>{}, st);
#endif
    for (; st + 3 < a.nstages; ++st) stage(hc, T{}, T{}, T{}, st);
    if (st + 2 < a.nstages) stage(hc, T{}, T{}, F{}, st++);
    if (st + 1 < a.nstages) stage(hc, T{}, F{}, F{}, st++);
    load_res();
    stage(hc, F{}, F{}, F{}, st);
  };
  if (NH == 2 && h) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
  TGSR_WSTAMP(2);

  // ---- output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) + epilogue; lane = tile (l15), register i of
  // block cb = cout cb*16 + 4*lg + i of this wave's half; the two column phases of a tile leave as one float2
  auto ytile = [&](int cb, int i, float (&y)[2][2]) {
    float rr[4][2];                                      // M A: per transformed row r, the two output columns
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m0 = M[r * 4 + 0][cb][i], m1 = M[r * 4 + 1][cb][i], m2 = M[r * 4 + 2][cb][i], m3 = M[r * 4 + 3][cb][i];
      rr[r][0] = m0 + m1 + m2;
      rr[r][1] = m1 - m2 - m3;
    }
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      y[0][dx] = rr[0][dx] + rr[1][dx] + rr[2][dx];
      y[1][dx] = rr[1][dx] - rr[2][dx] - rr[3][dx];
    }
  };
  float ssum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ssq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (ox < a.W) {
    if (GLU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int lc = h * 32 + 4 * lg + i;              // value column; its gate is lc + 16
        float yv[2][2], yg[2][2];
        ytile(0, i, yv);
        ytile(1, i, yg);
        const float sv = aff_s[lc], tv = aff_s[64 + lc], sg = aff_s[lc + 16], tg = aff_s[64 + lc + 16];
        const int c = grp * (16 * NH) + h * 16 + 4 * lg + i;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          if (oy + dy >= a.H) continue;
          const float o0 = (yv[dy][0] * sv + tv) * __builtin_amdgcn_rcpf(1.f + __expf(-(yg[dy][0] * sg + tg)));
          const float o1 = (yv[dy][1] * sv + tv) * __builtin_amdgcn_rcpf(1.f + __expf(-(yg[dy][1] * sg + tg)));
          *reinterpret_cast<float2*>(ob + (int64_t)c * HWo + (int64_t)(oy + dy) * a.W + ox) = make_float2(o0, o1);
        }
      }
    } else {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int lc = h * 32 + cb * 16 + 4 * lg + i;
          float yv[2][2];
          ytile(cb, i, yv);
          const float sv = aff_s[lc], tv = aff_s[64 + lc];
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            if (oy + dy >= a.H) continue;
            const float o0 = yv[dy][0] * sv + tv + rres[cb][i][dy].x, o1 = yv[dy][1] * sv + tv + rres[cb][i][dy].y;
            *reinterpret_cast<float2*>(ob + (int64_t)(grp * (32 * NH) + lc) * HWo + (int64_t)(oy + dy) * a.W + ox) =
                make_float2(o0, o1);
            if (STATS) {
              ssum[cb][i] += o0 + o1;
              ssq[cb][i] += o0 * o0 + o1 * o1;
            }
          }
        }
      }
    }
  }
  if (STATS) {
    // BatchNorm's batch statistics ride the epilogue: this wave's 2 x 32 outputs of each of its 32 channels are summed
    // over the 16 tiles (lanes l15 of a lane group; out-of-image lanes hold zeros) and leave as ONE (sum, sumsq) pair per
    // channel and wave - slot = (sample, tile row, 32-column chunk); the normalise pass combines the slots in a fixed order
    const int slot = ((b * a.tiles_y + ty) * a.tiles_x + tx) * Geo::ROWS + w;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = ssum[cb][i], q = ssq[cb][i];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
          s += __shfl_xor(s, o);
          q += __shfl_xor(q, o);
        }
        if (l15 == 0) {
          const int c = grp * (32 * NH) + h * 32 + cb * 16 + 4 * lg + i;
          *reinterpret_cast<float2*>(a.stat + ((int64_t)c * a.nslots + slot) * 2) = make_float2(s, q);
        }
      }
    }
  }
  TGSR_WSTAMP(3);
}

// upack[stage][group][pos pair 8][ci 4][cout half nh][16][4] <- U = G g G^T.  A group is the 32 * nh output channels
// of one workgroup (nh = 2 when Cout % 64 == 0, else 1); element q of a 4-vector = position 2*pp + (q >> 1), cout block
// q & 1 of the half (GLU: block 0 = value channels grp*16*nh + half*16 + l, block 1 = their gates Cout/2 + ...;
// plain: grp*32*nh + half*32 + block*16 + l).
// tr != 0: the source is the FORWARD conv's weight [Cin][Cout][3][3]; the pack is of the data-gradient conv
// w'[co][c][r][k] = w[c][co][2 - r][2 - k]
__global__ void pack_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int glu,
                                        int nh, int tr, int64_t total) {
  const int ngrp = Cout / (32 * nh);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx & 3), l = (int)((idx >> 2) & 15), hh = nh == 2 ? (int)((idx >> 6) & 1) : 0;
    int64_t t = idx >> (nh == 2 ? 7 : 6);
    const int ci = (int)(t % kWCK);
    t /= kWCK;
    const int pp = (int)(t % 8);
    t /= 8;
    const int grp = (int)(t % ngrp);
    const int st = (int)(t / ngrp);
    const int pos = 2 * pp + (q >> 1), cb = q & 1;
    const int co = glu ? (cb ? (Cout >> 1) : 0) + grp * (16 * nh) + hh * 16 + l
                       : grp * (32 * nh) + hh * 32 + cb * 16 + l;
    const int c = st * kWCK + ci;
    float u = 0.f;
    if (c < Cin) {
      const int i = pos >> 2, j = pos & 3;
      const float* g = tr ? w + ((int64_t)c * Cout + co) * 9 : w + ((int64_t)co * Cin + c) * 9;
      float gi[3];   // row i of G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] applied to the filter rows
      for (int k = 0; k < 3; ++k) {
        const float g0 = tr ? g[8 - k] : g[0 * 3 + k], g1 = tr ? g[5 - k] : g[1 * 3 + k], g2 = tr ? g[2 - k] : g[2 * 3 + k];
        gi[k] = i == 0 ? g0 : (i == 1 ? 0.5f * (g0 + g1 + g2) : (i == 2 ? 0.5f * (g0 - g1 + g2) : g2));
      }
      u = j == 0 ? gi[0] : (j == 1 ? 0.5f * (gi[0] + gi[1] + gi[2]) : (j == 2 ? 0.5f * (gi[0] - gi[1] + gi[2]) : gi[2]));
    }
    up[idx] = u;
  }
}

}  // namespace tgsr

using namespace tgsr;

#ifdef TGSR_WINO_STAMPS
extern "C" int tgsr_debug_read_wstamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(tgsr::g_wstamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int64_t tgsr_packed_wino_weight_elems(int Cout, int Cin) {
  return (int64_t)((Cin + kWCK - 1) / kWCK) * 16 * kWCK * Cout;
}

static int pack_wino_weight(const float* w, float* upack, int Cout, int Cin, int glu, int tr, void* stream) {
  if (!w || !upack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  if (Cout % 32 != 0) return TGSR_EUNSUPPORTED;
  const int64_t total = tgsr_packed_wino_weight_elems(Cout, Cin);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_wino_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, upack, Cout, Cin,
                     glu ? 1 : 0, Cout % 64 == 0 ? 2 : 1, tr, total);
  return note_launch(hipGetLastError(), "pack_wino_weight_kernel");
}

extern "C" int tgsr_pack_wino_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream) {
  return pack_wino_weight(w, upack, Cout, Cin, glu, 0, stream);
}

extern "C" int tgsr_pack_wino_weight_dgrad(const float* w, float* upack, int Cout, int Cin, void* stream) {
  return pack_wino_weight(w, upack, Cout, Cin, 0, 1, stream);
}

static void wino_geometry(int B, int Cin, int H, int W, int Cout, WinoArgs& a) {
  const int nh = Cout % 64 == 0 ? 2 : 1;                 // cout halves per workgroup: 64- or 32-channel groups
  a.tiles_x = (W + 31) / 32; a.tiles_y = nh == 2 ? (H + 3) / 4 : (H + 7) / 8; a.nstages = (Cin + kWCK - 1) / kWCK;
  a.ngroups = Cout / (32 * nh);
  a.nslots = B * a.tiles_y * a.tiles_x * (nh == 2 ? 2 : 4);
}

extern "C" int tgsr_wino_stats_nslots(int B, int H, int W, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Cout < 1 || Cout % 32 != 0) return 0;
  WinoArgs a;
  wino_geometry(B, 4, H, W, Cout, a);
  return a.nslots;
}

static int wino_conv3x3(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                        const float* scale, const float* shift, const float* residual, int64_t res_bstride, float* out,
                        int64_t out_bstride, int epilogue, float* stat, void* stream) {
  if (!x || !upack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  const bool glu = epilogue == TGSR_EPI_AFFINE_GLU;
  if (!glu && epilogue != TGSR_EPI_AFFINE) return TGSR_EINVAL;
  if (glu && residual) return TGSR_EINVAL;
  if (Cout % 32 != 0 || Cin % kWCK != 0) return TGSR_EUNSUPPORTED;
  if ((int64_t)H * W >= (1 << 28) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  if ((W & 3) || (x_bstride & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 7) || (out_bstride & 1) ||
      (residual && ((reinterpret_cast<uintptr_t>(residual) & 7) || (res_bstride & 1))))
    return TGSR_EUNSUPPORTED;
  WinoArgs a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.upack = upack; a.Cout = Cout;
  a.scale = scale; a.shift = shift; a.res = residual; a.rbs = res_bstride; a.out = out; a.obs = out_bstride;
  const int nh = Cout % 64 == 0 ? 2 : 1;
  wino_geometry(B, Cin, H, W, Cout, a);
  a.stat = stat;
  if (stat && (glu || residual || scale)) return TGSR_EINVAL;     // statistics are of the raw convolution output
  dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y * a.ngroups));
  size_t dyn = 0;
#ifdef TGSR_WINO_STAMPS
  if (const char* e = getenv("TGSR_WINO_DYN")) dyn = (size_t)atoi(e);   // diagnostic: extra LDS to force 1 workgroup per CU
#endif
  if (nh == 2) {
    if (glu) hipLaunchKernelGGL((wino_conv3x3_kernel<true, 2>), grid, dim3(256), dyn, as_stream(stream), a);
    else if (stat) hipLaunchKernelGGL((wino_conv3x3_kernel<false, 2, true>), grid, dim3(256), dyn, as_stream(stream), a);
    else hipLaunchKernelGGL((wino_conv3x3_kernel<false, 2>), grid, dim3(256), dyn, as_stream(stream), a);
  } else {
    if (glu) hipLaunchKernelGGL((wino_conv3x3_kernel<true, 1>), grid, dim3(256), dyn, as_stream(stream), a);
    else if (stat) hipLaunchKernelGGL((wino_conv3x3_kernel<false, 1, true>), grid, dim3(256), dyn, as_stream(stream), a);
    else hipLaunchKernelGGL((wino_conv3x3_kernel<false, 1>), grid, dim3(256), dyn, as_stream(stream), a);
  }
  return note_launch(hipGetLastError(), "wino_conv3x3_kernel");
}

extern "C" int tgsr_wino_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W,
                                     const float* upack, int Cout, const float* scale, const float* shift,
                                     const float* residual, int64_t res_bstride, float* out, int64_t out_bstride,
                                     int epilogue, void* stream) {
  return wino_conv3x3(x, x_bstride, B, Cin, H, W, upack, Cout, scale, shift, residual, res_bstride, out, out_bstride,
                      epilogue, nullptr, stream);
}

extern "C" int tgsr_wino_conv3x3_stats_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W,
                                           const float* upack, int Cout, float* out, int64_t out_bstride,
                                           float* stat_partial, void* stream) {
  if (!stat_partial) return TGSR_EINVAL;
  return wino_conv3x3(x, x_bstride, B, Cin, H, W, upack, Cout, nullptr, nullptr, nullptr, 0, out, out_bstride,
                      TGSR_EPI_AFFINE, stat_partial, stream);
}
