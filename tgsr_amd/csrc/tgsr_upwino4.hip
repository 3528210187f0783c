// upBlock (util.py:74-80: Upsample(x2, nearest) -> conv3x3 -> BatchNorm -> GLU) by Winograd F(4x4, 3x3) applied to the
// UP-SAMPLED image with the nearest-x2 folded into the input transform - the F(4x4) counterpart of tgsr_upwino.hip.
//
// A 4x4 output tile at up-sampled rows 4Y .. 4Y+3 reads up-sampled rows 4Y-1 .. 4Y+4 = low-resolution rows
// (a, b, b, c, c, d) = L[2Y-1], L[2Y], L[2Y], L[2Y+1], L[2Y+1], L[2Y+2].  The input transform of that 6-vector,
//   B^T (a,b,b,c,c,d) = (4a - 5b + c,  -8b + 2c,  0,  3(c - b),  b - c,  4b - 5c + d),
// has a vanishing third entry and its fifth is -1/3 of its fourth: per dimension FOUR distinct values
//   s = T (a,b,c,d),  T = [4 -5 1 0; 0 -8 2 0; 0 -3 3 0; 0 4 -5 1],
// and FIVE live transformed rows / columns (0, 1, 3, 4, 5).  So 25 of the 36 positions carry a product - 25 multiplies per
// 16 outputs against 36 for the up-sample-aware F(2x2) form (9 per 4 outputs), 144 for the direct convolution on the
// up-sampled grid - and the B operands of the 25 MFMAs come from only 16 values S = T P T^T of the 4x4 low-resolution
// patch P (position (i, j) reads S[rho(i)][rho(j)], rho = 0,1,-,2,2,3; the -1/3 factors of row / column 4 are folded into
// U = G g G^T by the pack kernel, in double).  The input transform costs 16 VALU operations per wave and stage (one row
// of S each) where tgsr_winograd4.hip's costs 48 on three waves of four.
// Numerics: F(4x4)'s (tgsr_winograd4.hip) - the upBlocks produce 64^2 .. 256^2 (x16: 512^2) images, all inside the error
// study's ">= 64 x 64 pixels" (tools/exp_wino4_numerics.py routes the upBlock-shaped convolutions as well: 3.4e-5 on the
// finest image against fp64, stated tolerance 1e-4).
//
// Geometry.  A workgroup = 4 waves = ONE tile row (4 x 64 OUTPUT pixels) x 4 channel blocks (64 accumulator rows); a wave =
// 16 tiles x 16 rows x 25 positions = 100 accumulator registers; stage = 4 input channels = 25 MFMAs per wave; 168 registers:
// THREE workgroups per CU (three waves per SIMD; two: 177 instead of 170 us on the largest layer), independent of each other:
// one's prologue / epilogue runs under the others' main loops.
//   A   straight from L2 into registers, as in tgsr_winograd4.hip's wide form: the pack is in per-wave fragment order
//       [stage][group][cb 4][quad 7][lane 64][4] (positions p = 5 ri + cj over the live rows / columns 0,1,3,4,5, quad p / 4,
//       element p % 4, three pad slots); one global_load_dwordx4 per quad, issued behind the quad's MFMAs for the NEXT stage
//       into the same registers; counted waits: the VMEM stream of a wave is [raw copy of this stage, 0-1][7 fragment loads
//       of the next], so vmcnt(6) in front of quad q's MFMAs = "A(q) has arrived", vmcnt(7) at the stage's end = "my copy has
//       landed".  An EVEN number of stages (host-checked; see tgsr_winograd4.hip for why); the last stage prefetches stage 0
//       again and vmcnt(0) precedes the epilogue.  (With U through LDS - 28 KB per stage, 32 copy instructions per 8-wave
//       workgroup and stage - the 128^2 -> 256^2 upBlock took 187 us; this form: see profiles/HISTORY.md 3.1f.)
//   raw [4 ci][4 low-res rows][40 cols] in planes of 192 floats (3 KB per stage, double buffered, LDS-DMA by waves 1-3; the
//       tile starts 4 columns left of the first low-res column: every 16-byte piece is aligned and wholly in or out).
//   S   [4 rows of S = quads][4 ci][16 tiles][4] (4 KB, double buffered): wave cb computes row cb.
// LDS 15 KB.  GLU channel blocks as in tgsr_winograd4.hip (value, value, gate, gate per lane).
#include "tgsr_common.h"

#include <type_traits>

namespace tgsr {

typedef __attribute__((address_space(3))) void* lds_ptru4_t;
__device__ __attribute__((aligned(16))) float g_upw4_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct Upw4Args {
  const float* x;         // low-resolution input [B][Cin][H][W]
  int64_t xbs;
  int B, Cin, H, W;       // LOW-resolution size; the output is 2H x 2W
  const float* upack;     // [stage][group][cb 4][quad 7][lane 64][4]
  int Cout;
  const float* scale;
  const float* shift;
  float* out;
  int64_t obs;
  int tiles_x, tiles_y, nstages, ngroups;
};

constexpr int ku4CK = 4;
constexpr int ku4TC = 40;                                  // raw tile columns: 32 + 8 (low resolution)
constexpr int ku4TR = 4;                                   // raw tile rows: 2 + 2 (low resolution)
constexpr int ku4PLANE = 192;                              // floats per channel plane (160 used; 0 mod 64 banks)
constexpr int ku4RAW = ku4CK * ku4PLANE;                   // 768 floats = 3 DMA pieces of 1 KB
constexpr int ku4NQ = 7;                                   // A quads per stage (25 positions + 3 pad)
constexpr int ku4UW = ku4NQ * 256;                         // floats of U per wave and stage
constexpr int ku4V = 4 * ku4CK * 16 * 4;                   // 1024 floats of S
constexpr int ku4SMEM = 2 * ku4RAW + 2 * ku4V + 128;

// live transformed row / column k (0..4) -> index 0,1,3,4,5 of the 6 x 6 Winograd domain; -> row / column of S
__host__ __device__ constexpr int u4_live(int k) { return k < 2 ? k : k + 1; }
__host__ __device__ constexpr int u4_rho(int k) { return k < 2 ? k : (k < 4 ? 2 : 3); }

typedef float f32x4u4 __attribute__((ext_vector_type(4)));

// T (a, b, c, d) = (4a - 5b + c, -8b + 2c, 3(c - b), 4b - 5c + d)
__device__ __forceinline__ void u4_t(float a, float b, float c, float d, float (&s)[4]) {
  s[0] = fmaf(4.f, a, fmaf(-5.f, b, c));
  s[1] = fmaf(-8.f, b, c + c);
  s[2] = 3.f * (c - b);
  s[3] = fmaf(4.f, b, fmaf(-5.f, c, d));
}
template <int R>   // row R of T only
__device__ __forceinline__ float u4_t_row(float a, float b, float c, float d) {
  return R == 0 ? fmaf(4.f, a, fmaf(-5.f, b, c)) : (R == 1 ? fmaf(-8.f, b, c + c) : (R == 2 ? 3.f * (c - b) : fmaf(4.f, b, fmaf(-5.f, c, d))));
}
// A^T applied to one 6-vector whose third entry is zero
__device__ __forceinline__ void u4_at(float m0, float m1, float m3, float m4, float m5, float (&y)[4]) {
  const float d2 = m3 - m4, s2 = m3 + m4;
  y[0] = m0 + m1 + s2;
  y[1] = fmaf(2.f, d2, m1);
  y[2] = fmaf(4.f, s2, m1);
  y[3] = fmaf(8.f, d2, m1) + m5;
}

template <bool GLU>
__global__ __launch_bounds__(256, 3) void upwino4_kernel(Upw4Args a) {
  __shared__ __attribute__((aligned(16))) float smem[ku4SMEM];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int cb = __builtin_amdgcn_readfirstlane(tid >> 6);             // channel block of this wave
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = t % a.ngroups;
  t /= a.ngroups;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * 4, x0 = tx * 64;                   // OUTPUT origin of the workgroup tile
  const int yl = ty * 2, xl = tx * 32;                   // low-resolution origin
  const float* xb = a.x + (int64_t)b * a.xbs;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;
  float* raws = smem;
  float* vs = smem + 2 * ku4RAW;
  float* aff_s = smem + 2 * ku4RAW + 2 * ku4V;

  // ---- raw copies: 3 pieces per stage, waves 1, 2, 3 one each
  const float* rptr = g_upw4_zero;
  int rstep = 0;
  if (cb >= 1) {
    const int e = ((cb - 1) * 64 + lane) * 4;            // first float of this lane's 16-byte piece
    const int c = e / ku4PLANE;
    const int rem = e - c * ku4PLANE;
    const int r = rem / ku4TC, j = rem - r * ku4TC;
    const int gy = yl - 1 + r, gx = xl - 4 + j;
    const bool ok = rem < ku4TR * ku4TC && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   // W % 4 == 0
    rptr = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(gy * a.W + gx) : g_upw4_zero;
    rstep = ok ? (int)(ku4CK * HW) : 0;
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptru4_t)smem);
  const unsigned lds_raw = lds0 + (cb >= 1 ? cb - 1 : 0) * 1024;
  auto issue_raw = [&](int buf) {
    if (cb >= 1) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(rptr), "s"(lds_raw + buf * (ku4RAW * 4)) : "memory");
      rptr += rstep;
    }
  };
  // ---- A fragments: this wave's 7 KB of a stage, two per-lane offsets 4 KB apart + an immediate
  const float* ubase = a.upack + ((int64_t)grp * 4 + cb) * ku4UW;      // stage 0
  const float* const ubase0 = ubase;
  const int64_t ustride = (int64_t)a.ngroups * 4 * ku4UW;
  unsigned voff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) voff[k] = (unsigned)(lane * 16 + k * 4096);
  f32x4u4 af[ku4NQ];
#define TGSR_U4_LOAD(q)                                                                                                \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(af[q]) : "v"(voff[(q) / 4]), "s"(ubase), "n"(((q) % 4) * 1024) : "memory")
#define TGSR_U4_WAIT(q) asm volatile("s_waitcnt vmcnt(6)" : "+v"(af[q])::"memory")

  // ---- input transform: lane = (tile l15, channel lg); wave cb computes row cb of S = T P T^T, P = the 4 x 4 low-resolution
  // patch of tile l15: raw rows 0 .. 3, raw columns 2 l15 + 3 .. 2 l15 + 6
  const int rlane = lg * ku4PLANE + 2 * l15;
  const int vwl = (lg * 16 + l15) * 4;
  auto t_read = [&](auto rc, const float* rawb, float (&d)[4][4]) {
    constexpr int R = decltype(rc)::value;
    const float* rp = rawb + rlane;
#pragma unroll
    for (int p = (R == 0 ? 0 : 1); p < (R == 3 ? 4 : 3); ++p) {
      const float2 mid = *reinterpret_cast<const float2*>(rp + p * ku4TC + 4);
      d[p][0] = rp[p * ku4TC + 3];
      d[p][1] = mid.x; d[p][2] = mid.y;
      d[p][3] = rp[p * ku4TC + 6];
    }
  };
  auto t_write = [&](auto rc, const float (&d)[4][4], float* vdst) {
    constexpr int R = decltype(rc)::value;
    float v[4], s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)                            // row R of T P
      v[j] = u4_t_row<R>(R == 0 ? d[0][j] : 0.f, d[1][j], d[2][j], R == 3 ? d[3][j] : 0.f);
    u4_t(v[0], v[1], v[2], v[3], s);                       // (T P) T^T, row R
    *reinterpret_cast<f32x4u4*>(vdst + vwl + R * (ku4CK * 16 * 4)) = f32x4u4{s[0], s[1], s[2], s[3]};
  };

  if (tid < 128) {   // aff_s[cb * 16 + m] = scale, [64 + ...] = shift of accumulator row m of block cb
    const int lc = tid & 63, cbk = lc >> 4, m = lc & 15;
    int col = GLU ? ((m & 2) ? (a.Cout >> 1) : 0) + grp * 32 + cbk * 8 + 2 * (m >> 2) + (m & 1) : grp * 64 + lc;
    if (col >= a.Cout) col = 0;
    aff_s[tid] = a.scale ? (tid < 64 ? a.scale[col] : a.shift[col]) : (tid < 64 ? 1.f : 0.f);
  }

  f32x4u4 M[25];
#pragma unroll
  for (int p = 0; p < 25; ++p)
#pragma unroll
    for (int i = 0; i < 4; ++i) M[p][i] = 0.f;
  const int vlane = (lg * 16 + l15) * 4;                 // B: S[r][ci = lg][l15][4]

  // One stage: the raw copy of stage st+2 first; the raw reads of the transform of raw(st+1); the 25 MFMAs, each quad behind
  // the wait for its fragments and followed by the load of the next stage's; the transform's 16 operations behind the second
  // quad; wait for the copy + barrier.  The transform is not skipped in the last stage (stale raw in, an S image nobody reads).
  auto stage = [&](auto rc, auto parc, const bool MORE, const bool MORE2) {
    constexpr int R = decltype(rc)::value, PAR = decltype(parc)::value;
    if (MORE2) issue_raw(PAR);                           // raw(st+2) replaces raw(st), transformed one stage ago
    ubase = MORE ? ubase + ustride : ubase0;             // where the fragments loaded during this stage come from
    float d[4][4];
    t_read(rc, raws + (PAR ^ 1) * ku4RAW, d);
    const float* vb = vs + PAR * ku4V + vlane;
    f32x4u4 bf[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bf[r] = *reinterpret_cast<const f32x4u4*>(vb + r * (ku4CK * 16 * 4));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < ku4NQ; ++q) {
      TGSR_U4_WAIT(q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = 4 * q + e;
        if (p < 25)
          M[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][e], bf[u4_rho(p / 5)][u4_rho(p % 5)], M[p], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      TGSR_U4_LOAD(q);
      if (q == 1) {
        t_write(rc, d, vs + (PAR ^ 1) * ku4V);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MORE) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto run = [&](auto rc) {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    issue_raw(0);
    if (a.nstages > 1) issue_raw(1);
    TGSR_U4_LOAD(0); TGSR_U4_LOAD(1); TGSR_U4_LOAD(2); TGSR_U4_LOAD(3); TGSR_U4_LOAD(4); TGSR_U4_LOAD(5); TGSR_U4_LOAD(6);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
      float d[4][4];
      t_read(rc, raws, d);
      t_write(rc, d, vs);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int st = 0; st < a.nstages; st += 2) {          // an even number of stages: exactly two copies of the stage
      stage(rc, P0{}, true, st + 2 < a.nstages);
      stage(rc, P1{}, st + 2 < a.nstages, st + 3 < a.nstages);
    }
  };
  if (cb == 0) run(std::integral_constant<int, 0>{});
  else if (cb == 1) run(std::integral_constant<int, 1>{});
  else if (cb == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last stage's (unused) fragment loads
#undef TGSR_U4_LOAD
#undef TGSR_U4_WAIT

  // ---- output transform Y = A^T M A over the live rows / columns + epilogue; lane = tile l15, register i = accumulator
  // row 4 lg + i of block cb
  auto ytile = [&](int i, float (&y)[4][4]) {
    float c[4][5];                                       // A^T M: per live column cj, the four output rows
#pragma unroll
    for (int cj = 0; cj < 5; ++cj) {
      float col[4];
      u4_at(M[0 * 5 + cj][i], M[1 * 5 + cj][i], M[2 * 5 + cj][i], M[3 * 5 + cj][i], M[4 * 5 + cj][i], col);
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r][cj] = col[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) u4_at(c[r][0], c[r][1], c[r][2], c[r][3], c[r][4], y[r]);
  };
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const int oy = y0, ox = x0 + 4 * l15;
  const int64_t HWo = (int64_t)Ho * Wo;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  if (ox < Wo) {
    if (GLU) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float yv[4][4], yg[4][4];
        ytile(p, yv);
        ytile(p + 2, yg);
        const int m = 4 * lg + p;
        const float sv = aff_s[cb * 16 + m], tv = aff_s[64 + cb * 16 + m], sg = aff_s[cb * 16 + m + 2], tg = aff_s[64 + cb * 16 + m + 2];
        const int c = grp * 32 + cb * 8 + 2 * lg + p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (oy + r >= Ho) continue;
          f32x4u4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            o[k] = (yv[r][k] * sv + tv) * __builtin_amdgcn_rcpf(1.f + __expf(-(yg[r][k] * sg + tg)));
          *reinterpret_cast<f32x4u4*>(ob + (int64_t)c * HWo + (int64_t)(oy + r) * Wo + ox) = o;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 4 * lg + i, c = grp * 64 + cb * 16 + m;
        float yv[4][4];
        ytile(i, yv);
        const float sv = aff_s[cb * 16 + m], tv = aff_s[64 + cb * 16 + m];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (oy + r >= Ho) continue;
          f32x4u4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = yv[r][k] * sv + tv;
          *reinterpret_cast<f32x4u4*>(ob + (int64_t)c * HWo + (int64_t)(oy + r) * Wo + ox) = o;
        }
      }
    }
  }
}

// upack[stage][group][cb 4][quad 7][lane 64 = (ci lane >> 4, row lane & 15)][4] <- U'[i][j] = f(i) f(j) (G g G^T)[i][j] over the
// live rows / columns i, j in {0, 1, 3, 4, 5}, f(4) = -1/3 (the input transform's fifth entry is -1/3 of its fourth), else 1;
// position p = 5 ri + cj at quad p / 4, element p % 4 (p >= 25: zero).  Rows of a block as in pack_wino4_weight_kernel.
__global__ void pack_upwino4_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int glu,
                                           int64_t total) {
  const double G[6][3] = {{0.25, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  const int ngrp = Cout / 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 3), m = (int)((idx >> 2) & 15), ci = (int)((idx >> 6) & 3);
    int64_t t = idx >> 8;
    const int q = (int)(t % ku4NQ);
    t /= ku4NQ;
    const int cbk = (int)(t & 3);
    t >>= 2;
    const int grp = (int)(t % ngrp);
    const int st = (int)(t / ngrp);
    const int p = 4 * q + e;
    const int co = glu ? ((m & 2) ? (Cout >> 1) : 0) + grp * 32 + cbk * 8 + 2 * (m >> 2) + (m & 1) : grp * 64 + cbk * 16 + m;
    const int c = st * ku4CK + ci;
    double u = 0.0;
    if (c < Cin && p < 25) {
      const int i = u4_live(p / 5), j = u4_live(p % 5);
      const float* gw = w + ((int64_t)co * Cin + c) * 9;
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 3; ++l) u += G[i][k] * (double)gw[k * 3 + l] * G[j][l];
      if (i == 4) u *= -1.0 / 3;
      if (j == 4) u *= -1.0 / 3;
    }
    up[idx] = (float)u;
  }
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_packed_upwino4_weight_elems(int Cout, int Cin) {
  return (int64_t)((Cin + ku4CK - 1) / ku4CK) * (ku4NQ * 4) * ku4CK * Cout;
}

extern "C" int tgsr_pack_upwino4_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream) {
  if (!w || !upack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  if (Cout % 64 != 0) return TGSR_EUNSUPPORTED;
  const int64_t total = tgsr_packed_upwino4_weight_elems(Cout, Cin);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_upwino4_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, upack, Cout, Cin, glu ? 1 : 0,
                     total);
  return note_launch(hipGetLastError(), "pack_upwino4_weight_kernel");
}

extern "C" int tgsr_upwino4_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                                const float* scale, const float* shift, float* out, int64_t out_bstride, int glu, void* stream) {
  if (!x || !upack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  if (Cout % 64 != 0 || Cin % (2 * ku4CK) != 0) return TGSR_EUNSUPPORTED;          // an even number of 4-channel stages
  if ((int64_t)H * W >= (1 << 26) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  if ((W & 3) || (x_bstride & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(upack) & 15) || (out_bstride & 3))
    return TGSR_EUNSUPPORTED;
  Upw4Args a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.upack = upack; a.Cout = Cout;
  a.scale = scale; a.shift = shift; a.out = out; a.obs = out_bstride;
  a.tiles_x = (2 * W + 63) / 64; a.tiles_y = (2 * H + 3) / 4; a.nstages = Cin / ku4CK; a.ngroups = Cout / 64;
  const dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y * a.ngroups));
  if (glu) hipLaunchKernelGGL((upwino4_kernel<true>), grid, dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((upwino4_kernel<false>), grid, dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "upwino4_kernel");
}
