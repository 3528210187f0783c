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
    const int bid_ = blockIdx.y * gridDim.x + blockIdx.x;                                               \
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
  const float* upack;     // [stage][group][pos 16][ci 8][64 interleaved]
  int Cout;
  const float* scale;
  const float* shift;
  const float* res;
  int64_t rbs;
  float* out;
  int64_t obs;
  int tiles_x, tiles_y, nstages;
};

// Geometry: MFMA 16x16x4 (4 accumulator registers per 16 couts x 16 tiles), ALL 16 transformed positions live at
// once: a wave owns 16 tiles (one tile row = 2 x 32 output pixels) x 64 output channels (4 blocks of 16: GLU value
// blocks 0,1 and gate blocks 2,3, or 4 plain blocks) = 16 x 4 x 4 = 256 accumulator registers, one wave per SIMD; the
// 4 waves of a workgroup take 4 consecutive tile rows (8 x 32 outputs) and share the transformed weights and the
// raw input rows:
//   stage = 8 input channels.
//   U   [16 pos][8 ci][16 x 4 couts] (32 KB, double buffered): a linear LDS-DMA copy of the pre-transformed pack,
//       whose 64 columns are stored interleaved (position 4*l + cb = column cb*16 + l) so one ds_read_b128 yields a
//       lane's A fragments of all four cout blocks.
//   raw [8 ci][10 rows][40 cols] (12.5 KB, double buffered, LDS-DMA in 16-byte pieces; the tile starts 4 columns
//       left of the outputs so every piece is aligned and wholly inside or outside the image).
//   V   per wave [16 pos][8 ci][16 tiles] (8 KB, double buffered): the wave's own input transform B^T d B.
// One wave per SIMD means nothing hides a stall, so everything that is not an MFMA is cut into micro-operations of
// <= 8 instructions and placed in the gaps BETWEEN the 128 MFMAs of a stage (an instruction issued right after an
// MFMA runs under its 32 pipe cycles): the fragments of the next position, then the DMA of raw(st+2) and U(st+1),
// then the transform raw(st+1) -> V(st+1) as reads / row sums / column sums + writes, each a few gaps apart so that
// no LDS or DMA latency is ever waited for.  The only waits are the vmcnt(0) + barrier that ends a stage.
constexpr int kWCK = 8;                                  // input channels per stage
constexpr int kWTC = 40, kWTR = 10, kWPLANE = kWTR * kWTC;   // raw tile: 10 rows x (32 + 8) columns per channel
constexpr int kWU = 16 * kWCK * 64;                      // floats of U per stage: 8192
constexpr int kWUK = kWU / 256 / 4;                      // U DMA pieces (1 KB) per wave per stage: 8
constexpr int kWRawN = kWCK * kWPLANE;                   // 3200 floats of raw input per stage
constexpr int kWRawP = (kWRawN + 255) / 256;             // = 13 DMA pieces of 256 floats (the last one half used)
constexpr int kWRaw = kWRawP * 256;                      // LDS floats per raw stage (padded to whole pieces)
constexpr int kWIK = (kWRawP + 3) / 4;                   // raw DMA pieces per wave per stage: <= 4
constexpr int kWV = 16 * kWCK * 16;                      // V image: [pos 16][ci 8][16 tiles]
constexpr int kWSmem = 2 * kWU + 2 * kWRaw + 4 * 2 * kWV + 2 * 64;

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool GLU>
__global__ __launch_bounds__(256, 1) void wino_conv3x3_kernel(WinoArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[kWSmem];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int grp = blockIdx.y;
  const int y0 = ty * 8, x0 = tx * 32;                   // output origin of the workgroup tile
  const float* xb = a.x + (int64_t)b * a.xbs;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;
  float* us = smem;                                      // 2 x U stage
  float* raws = smem + 2 * kWU;                          // 2 x raw stage
  float* vs = smem + 2 * kWU + 2 * kWRaw + wave * 2 * kWV;   // this wave's 2 V images
  float* aff_s = smem + 2 * kWU + 2 * kWRaw + 4 * 2 * kWV;
  TGSR_WSTAMP(0);

  // ---- DMA plan.  raw: piece q = wave + 4*k covers floats [q*256, q*256 + 256) of the stage tile.  Per lane a running
  // source pointer and its per-stage stride; out-of-image (or past-the-tile) lanes read the zero block with stride 0,
  // so issuing a piece is branch-free.  Cin % 8 == 0 (host-checked): a stage never reads past the last channel.
  const float* rptr[kWIK];
  int rstep[kWIK];
#pragma unroll
  for (int k = 0; k < kWIK; ++k) {
    const int e = ((wave + 4 * k) * 64 + lane) * 4;      // first float of this lane's 16-byte piece
    const int c = e / kWPLANE;
    const int rem = e - c * kWPLANE;
    const int r = rem / kWTC, j = rem - r * kWTC;
    const int gy = y0 - 1 + r, gx = x0 - 4 + j;
    const bool ok = e < kWRawN && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   // W % 4 == 0
    rptr[k] = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(gy * a.W + gx) : g_wino_zero;
    rstep[k] = ok ? (int)(kWCK * HW) : 0;
  }
  auto issue_raw = [&](int k, int buf) {                 // pieces are issued for stages 0, 1, 2, ... in order
    if (4 * k + 3 < kWRawP || wave + 4 * k < kWRawP) {   // wave-uniform; only the last k can fail
      __builtin_amdgcn_global_load_lds((glb_ptrw_t)rptr[k], (lds_ptrw_t)(raws + buf * kWRaw + (wave + 4 * k) * 256), 16,
                                       0, 0);
      rptr[k] += rstep[k];
    }
  };
  const float* uptr = a.upack + (int64_t)grp * kWU + (wave * 64 + lane) * 4;    // this wave's pieces of stage 0
  const int64_t ustride = (int64_t)gridDim.y * kWU;
  auto issue_u = [&](int k, int buf) {                   // k = 0..7 in order; the pointer moves on after the last
    __builtin_amdgcn_global_load_lds((glb_ptrw_t)(uptr + k * 1024), (lds_ptrw_t)(us + buf * kWU + (wave + 4 * k) * 256),
                                     16, 0, 0);
    if (k == kWUK - 1) uptr += ustride;
  };

  // ---- input transform V = B^T d B of this lane's tile (l15) for channels 2*lg + cc, in micro-operations:
  //   rd(cc, r): the 4 raw floats of patch row r      rs(cc, i): row i of B^T d (4 sums)
  //   cw(cc, i): row i of (B^T d) B (4 sums) + its 4 LDS writes
  float d[2][4][4], tr[2][4][4];
  const int rlane = (2 * lg) * kWPLANE + (2 * wave) * kWTC + 2 * l15 + 3;   // patch origin inside the raw stage tile
  const int vwl = (2 * lg) * 16 + l15;
  auto t_rd = [&](const float* rawb, int cc, int r) {
    const float* rp = rawb + rlane + cc * kWPLANE + r * kWTC;
#pragma unroll
    for (int q = 0; q < 4; ++q) d[cc][r][q] = rp[q];
  };
  auto t_rs = [&](int cc, int i) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      tr[cc][i][q] = i == 0 ? d[cc][0][q] - d[cc][2][q]
                            : (i == 1 ? d[cc][1][q] + d[cc][2][q]
                                      : (i == 2 ? d[cc][2][q] - d[cc][1][q] : d[cc][1][q] - d[cc][3][q]));
  };
  auto t_cw = [&](float* vdst, int cc, int i, int half) {
    float* vp = vdst + vwl + cc * 16 + (i * 4) * 128;    // V[pos = i*4 + jj][c][tile], pos stride 8 * 16
    if (half == 0) {
      vp[0 * 128] = tr[cc][i][0] - tr[cc][i][2];
      vp[1 * 128] = tr[cc][i][1] + tr[cc][i][2];
    } else {
      vp[2 * 128] = tr[cc][i][2] - tr[cc][i][1];
      vp[3 * 128] = tr[cc][i][1] - tr[cc][i][3];
    }
  };
  // transform micro-operation m (0..31), <= 4 instructions each: 8 reads, 8 row sums, 16 column sums + writes
  auto t_op = [&](int m, const float* rawb, float* vdst) {
    if (m < 8) t_rd(rawb, (m >> 2) & 1, m & 3);
    else if (m < 16) t_rs((m >> 2) & 1, m & 3);
    else t_cw(vdst, (m >> 3) & 1, (m >> 1) & 3, m & 1);
  };

  if (tid < 64) {                                        // logical column tid of this workgroup -> global cout
    int col;
    if (GLU) col = (tid < 32 ? grp * 32 : (a.Cout >> 1) + grp * 32 - 32) + tid;
    else col = grp * 64 + tid;
    aff_s[tid] = a.scale ? a.scale[col] : 1.f;
    aff_s[64 + tid] = a.scale ? a.shift[col] : 0.f;
  }

  f32x4v M[16][4];
#if defined(TGSR_WINO_EXP) && TGSR_WINO_EXP == 6
  typedef float f32x16v __attribute__((ext_vector_type(16)));
  f32x16v MM[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) MM[p][i] = 0.f;
#endif
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) M[p][cb][i] = 0.f;

  // ---- prologue: raw(0), raw(1), U(0); transform raw(0) -> V[0]
#pragma unroll
  for (int k = 0; k < kWIK; ++k) issue_raw(k, 0);
#pragma unroll
  for (int k = 0; k < kWUK; ++k) issue_u(k, 0);
  if (a.nstages > 1) {
#pragma unroll
    for (int k = 0; k < kWIK; ++k) issue_raw(k, 1);
  }
  __syncthreads();                                       // vmcnt(0) + barrier: raw(0), raw(1), U(0) visible
#pragma unroll
  for (int m = 0; m < 32; ++m) t_op(m, raws, vs);
  TGSR_WSTAMP(1);

  const int oy = y0 + 2 * wave, ox = x0 + 2 * l15;
  const int64_t HWo = (int64_t)a.H * a.W;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  const float* __restrict__ rb = a.res ? a.res + (int64_t)b * a.rbs : nullptr;
  const bool inx = ox < a.W;

  const int ulane = lg * 64 + l15 * 4;                   // A fragments: U[(pos*8 + 4*ks + lg)][l15*4 .. +3]
  const int vlane = lg * 16 + l15;                       // B fragment:  V[(pos*8 + 4*ks + lg)][l15]

  // One stage; MORE = a next stage exists (its U is fetched, its input transformed), MORE2 = so does the one after
  // (its raw rows are fetched).  Compile-time so that the gaps hold straight-line code.
  auto stage = [&](auto more_c, auto more2_c, int st) {
    constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
    const int par = st & 1;
    const float* ub = us + par * kWU + ulane;
    const float* vb = vs + par * kWV + vlane;
    const float* rawn = raws + (par ^ 1) * kWRaw;
    float* vnxt = vs + (par ^ 1) * kWV;
    f32x4v a0[3], a1[3];                                  // A fragments (4 cout blocks) of ks = 0 / 1; positions p, p+1, p+2
    float b0[3], b1[3];
    auto frag = [&](int p, int slot_) {
      a0[slot_] = *reinterpret_cast<const f32x4v*>(ub + p * 8 * 64);
      a1[slot_] = *reinterpret_cast<const f32x4v*>(ub + (p * 8 + 4) * 64);
      b0[slot_] = vb[p * 8 * 16];
      b1[slot_] = vb[(p * 8 + 4) * 16];
    };
    frag(0, 0);
    frag(1, 1);
#if defined(TGSR_WINO_EXP) && TGSR_WINO_EXP == 9
    float dummy[4] = {1.f, 2.f, 3.f, 4.f};
#endif
    // Gaps of a position (after its MFMA g), at most ~4 instructions each so that the next MFMA issues on time:
    //   1, 3: one transform micro-operation (LDS traffic)   4: the fragments of position p + 2
    //   5, 7: one DMA piece (positions 0-5)
    // LDS operations complete in order and the fragments are fetched two positions (12 MFMAs, ~400 cycles) ahead, so
    // the counted lgkmcnt wait at a position's first MFMA never stalls, even with the 4 waves reading in lockstep.
    auto tslot = [&](int n) {                            // transform(st+1): 32 micro-operations
#if !defined(TGSR_WINO_EXP) || TGSR_WINO_EXP != 8
      if (MORE && n < 32) t_op(n, rawn, vnxt);
#endif
    };
    auto dslot = [&](int n) {                            // raw(st+2): kWIK pieces, U(st+1): kWUK pieces
#if defined(TGSR_WINO_EXP) && TGSR_WINO_EXP == 7
      if (false) {
#else
      if (MORE) {
#endif
        if (n < kWIK) { if (MORE2) issue_raw(n, par); }
        else if (n < kWIK + kWUK) issue_u(n - kWIK, par ^ 1);
      }
    };
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int cur = p % 3, nxt = (p + 2) % 3;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int ks = g >> 2, cb = g & 3;
#if defined(TGSR_WINO_EXP) && TGSR_WINO_EXP == 6   // diagnostic: same stream with 4 MFMA 32x32x2 per position (wrong results)
        if ((g & 1) == 0) MM[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks ? a1[cur][cb] : a0[cur][cb], ks ? b1[cur] : b0[cur], MM[p], 0, 0, 0);
#else
        M[p][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ks ? a1[cur][cb] : a0[cur][cb], ks ? b1[cur] : b0[cur], M[p][cb],
                                                        0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
#if defined(TGSR_WINO_EXP) && TGSR_WINO_EXP == 9   // diagnostic: 4 independent VALU instructions in every gap
        asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3"
                     : "+v"(dummy[0]), "+v"(dummy[1]), "+v"(dummy[2]), "+v"(dummy[3]));
#endif
        if (g == 1 || g == 3) {
          tslot(p * 2 + (g >> 1));
        } else if (g == 4) {
          if (p + 2 < 16) frag(p + 2, nxt);
        } else if (g == 5 || g == 7) {
          dslot(p * 2 + ((g - 5) >> 1));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // U(st+1), raw(st+2) landed (vmcnt(0)) and visible; U(st) / V(st) / raw(st+1) may be overwritten
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    int st = 0;
#if defined(TGSR_WINO_EXP) && (TGSR_WINO_EXP == 1 || TGSR_WINO_EXP == 9)   // diagnostic: MFMAs + fragment reads only (results are wrong)
    for (; st + 1 < a.nstages; ++st) stage(F{}, F{}, st);
#endif
    for (; st + 2 < a.nstages; ++st) stage(T{}, T{}, st);
    if (st + 1 < a.nstages) stage(T{}, F{}, st++);
    stage(F{}, F{}, st);
  }
  TGSR_WSTAMP(2);
#if defined(TGSR_WINO_EXP) && TGSR_WINO_EXP == 6
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) M[p][i >> 2][i & 3] = MM[p][i];
#endif

  // ---- output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) + epilogue; lane = tile (l15), registers of
  // block cb = couts cb*16 + 4*lg + i; the two column phases of a tile leave as one float2
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
  if (inx) {
#pragma unroll
    for (int cb = 0; cb < (GLU ? 2 : 4); ++cb) {
      float2 rres[4][2];                                 // residual of this block: all 8 loads in flight together
      if (!GLU) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            rres[i][dy] = make_float2(0.f, 0.f);
            if (rb && oy + dy < a.H)
              rres[i][dy] = *reinterpret_cast<const float2*>(rb + (int64_t)(grp * 64 + cb * 16 + 4 * lg + i) * HWo +
                                                             (int64_t)(oy + dy) * a.W + ox);
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int lc = cb * 16 + 4 * lg + i;             // logical column of the value (or plain) channel
        float yv[2][2], yg[2][2];
        ytile(cb, i, yv);
        if (GLU) ytile(cb + 2, i, yg);
        const float sv = aff_s[lc], tv = aff_s[64 + lc];
        const float sg = GLU ? aff_s[32 + lc] : 0.f, tg = GLU ? aff_s[64 + 32 + lc] : 0.f;
        const int c = GLU ? grp * 32 + lc : grp * 64 + lc;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          if (oy + dy >= a.H) continue;
          float o0 = yv[dy][0] * sv + tv, o1 = yv[dy][1] * sv + tv;
          if (GLU) {
            o0 *= __builtin_amdgcn_rcpf(1.f + __expf(-(yg[dy][0] * sg + tg)));
            o1 *= __builtin_amdgcn_rcpf(1.f + __expf(-(yg[dy][1] * sg + tg)));
          } else {
            o0 += rres[i][dy].x;
            o1 += rres[i][dy].y;
          }
          *reinterpret_cast<float2*>(ob + (int64_t)c * HWo + (int64_t)(oy + dy) * a.W + ox) = make_float2(o0, o1);
        }
      }
    }
  }
  TGSR_WSTAMP(3);
}

// upack[stage][group][pos 16][ci 8][64] <- U = G g G^T, pos = i * 4 + j of the 4x4 transformed filter; a group is
// the 64 output channels of one workgroup (GLU: 32 value + the 32 matching gate channels), stored interleaved:
// position 4*l + cb holds the group's logical column cb*16 + l.
__global__ void pack_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int glu,
                                        int64_t total) {
  const int ngrp = Cout / 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx & 63);
    int64_t t = idx >> 6;
    const int ci = (int)(t % kWCK);
    t /= kWCK;
    const int pos = (int)(t % 16);
    t /= 16;
    const int grp = (int)(t % ngrp);
    const int st = (int)(t / ngrp);
    const int lc = (q & 3) * 16 + (q >> 2);
    const int co = glu ? (lc < 32 ? grp * 32 + lc : (Cout >> 1) + grp * 32 + lc - 32) : grp * 64 + lc;
    const int c = st * kWCK + ci;
    float u = 0.f;
    if (c < Cin) {
      const int i = pos >> 2, j = pos & 3;
      const float* g = w + ((int64_t)co * Cin + c) * 9;
      float gi[3];   // row i of G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] applied to the filter rows
      for (int k = 0; k < 3; ++k) {
        const float g0 = g[0 * 3 + k], g1 = g[1 * 3 + k], g2 = g[2 * 3 + k];
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

extern "C" int tgsr_pack_wino_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream) {
  if (!w || !upack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  if (Cout % 64 != 0) return TGSR_EUNSUPPORTED;
  const int64_t total = tgsr_packed_wino_weight_elems(Cout, Cin);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_wino_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, upack, Cout, Cin,
                     glu ? 1 : 0, total);
  return note_launch(hipGetLastError(), "pack_wino_weight_kernel");
}

extern "C" int tgsr_wino_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W,
                                     const float* upack, int Cout, const float* scale, const float* shift,
                                     const float* residual, int64_t res_bstride, float* out, int64_t out_bstride,
                                     int epilogue, void* stream) {
  if (!x || !upack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  const bool glu = epilogue == TGSR_EPI_AFFINE_GLU;
  if (!glu && epilogue != TGSR_EPI_AFFINE) return TGSR_EINVAL;
  if (glu && residual) return TGSR_EINVAL;
  if (Cout % 64 != 0 || Cin % kWCK != 0) return TGSR_EUNSUPPORTED;
  if ((int64_t)H * W >= (1 << 28) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  if ((W & 3) || (x_bstride & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 7) || (out_bstride & 1) ||
      (residual && ((reinterpret_cast<uintptr_t>(residual) & 7) || (res_bstride & 1))))
    return TGSR_EUNSUPPORTED;
  WinoArgs a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.upack = upack; a.Cout = Cout;
  a.scale = scale; a.shift = shift; a.res = residual; a.rbs = res_bstride; a.out = out; a.obs = out_bstride;
  a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 7) / 8; a.nstages = (Cin + kWCK - 1) / kWCK;
  dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y), (unsigned)(Cout / 64));
  if (glu) hipLaunchKernelGGL(wino_conv3x3_kernel<true>, grid, dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(wino_conv3x3_kernel<false>, grid, dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "wino_conv3x3_kernel");
}
