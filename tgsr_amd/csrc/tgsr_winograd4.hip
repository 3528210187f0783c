// 3x3 convolution (stride 1, pad 1) by Winograd F(4x4, 3x3) in fp32 on the MFMA units, same fused epilogues as
// tgsr_winograd.hip (BatchNorm-eval affine, GLU | residual).  Serves the LARGE layers of the inference path - the 128^2
// ResBlocks of G_SR_NET_low (util.py:110-130 behind NEXT_STAGE_G util.py:814-821), which carry 60 % of a forward's
// convolution time:
//
//   Y(4x4) = A^T [ (G g G^T) (.) (B^T d B) ] A     per 6x6 input tile d, 3x3 filter g, interpolation points 0, +-1, +-2, inf:
// 36 multiplies per 16 outputs instead of 144 - 1.78x fewer MFMAs than F(2x2, 3x3), 4x fewer than the direct form.
// Numerics: the transforms hold 4, 5, 8 and 1/24 where F(2x2) holds 1 and 1/2: per layer, on unit-scale data, 1.6e-5 .. 3.4e-5
// (max) against 7e-7 .. 1.8e-6.  Measured end to end on the shipped checkpoint (oracle with this algorithm in fp32 on chosen
// layers vs the fp64 oracle, profiles/HISTORY.md 3.1e; stated tolerance 1e-4): with the 128^2 layers on it the finest image stays at
// 3.0e-5 max (the direct fp32 form: 2.9e-5; the reference's own CPU path 3.7e-5), with the 64^2 layers as well the 256^2
// image of G_SR_NET_low moves 0.9e-5 -> 2.3e-5; with the 32^2 layers too the finest image is off by 2.1e-4 - the callers
// (ops.wino4_wanted) therefore route layers of >= 64 x 64 pixels here and nothing smaller.
//
// Geometry.  MFMA 16x16x4, one accumulator per transformed position: a wave owns 16 tiles (one tile row = 4 x 64 output
// pixels) x 16 output channels x 36 positions = 144 accumulator registers; a workgroup = 8 waves = 2 tile rows (g) x 4
// channel blocks (cb) = 8 x 64 outputs x 64 couts, one workgroup per CU (two waves per SIMD).
//   stage = 4 input channels = one MFMA k-step per position (36 MFMAs per wave).
//   U   [9 quads][4 ci][4 cb][16 couts][4 positions] (36 KB, double buffered): a linear LDS-DMA copy of the pre-transformed
//       pack; one ds_read_b128 = a lane's A operands of 4 positions.
//   raw [4 ci][10 rows][72 cols] in planes of 768 floats (12 KB, double buffered; LDS-DMA in 16-byte pieces, the tile
//       starts 4 columns left of the outputs so that every piece is aligned and wholly inside or outside the image;
//       768 = 0 mod 64 banks: the transform's ds_read_b128 of 16 neighbouring tiles x 4 channels is conflict free).
//   V   per tile row [9 quads][4 ci][16 tiles][4 positions] (9 KB, double buffered): the input transform B^T d B of one
//       (tile, channel) per lane, split over the waves cb = 0, 1, 2 of the tile row by ROWS of B^T - {0, 5}, {1, 2},
//       {3, 4}: 48 VALU operations each - one stage ahead; a quad = 4 consecutive of a wave's 12 values, which fixes the
//       order of the 36 positions everywhere (w4_pos).
// One barrier per stage; the copies of a stage (U(st+1), raw(st+2)) are issued at its start and have landed at its end:
// a stage is ~3 k cycles (2 x 36 MFMAs of 32 cycles per SIMD), longer than a DMA round trip.  LDS 133 KB.
// GLU: a channel block holds 8 value channels and their 8 gates, ordered so that a lane's four accumulator registers
// are (value, value, gate, gate) of two output channels - the gate never leaves the lane.
#include "tgsr_common.h"

#include <type_traits>

namespace tgsr {

typedef __attribute__((address_space(3))) void* lds_ptr4_t;
__device__ __attribute__((aligned(16))) float g_wino4_zero[4] = {0.f, 0.f, 0.f, 0.f};

#ifdef TGSR_WINO4_STAMPS
// Diagnostic build only (tools/wino4_stamps.py): per-workgroup cycle stamps, never compiled into the shipped library.
__device__ unsigned long long g_w4stamps[8 * 8192];
#define TGSR_W4STAMP(k)                                                                                 \
  do {                                                                                                  \
    const int bid_ = blockIdx.x;                                                                        \
    if (threadIdx.x == 0 && bid_ < 8192) {                                                              \
      g_w4stamps[bid_ * 8 + (k)] = __builtin_amdgcn_s_memtime();                                        \
      if ((k) == 0) g_w4stamps[bid_ * 8 + 6] = __builtin_amdgcn_s_memrealtime();                        \
      if ((k) == 3) g_w4stamps[bid_ * 8 + 7] = __builtin_amdgcn_s_memrealtime();                        \
    }                                                                                                   \
  } while (0)
#else
#define TGSR_W4STAMP(k)
#endif

struct Wino4Args {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;
  const float* upack;     // [stage][group][quad 9][ci 4][cb 4][16][4]
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

constexpr int k4CK = 4;                                   // input channels per stage
constexpr int k4TC = 72;                                  // raw tile columns: 64 + 8
constexpr int k4TR = 10;                                  // raw tile rows: 8 + 2
constexpr int k4PLANE = 768;                              // floats per channel plane (720 used)
constexpr int k4RAW = k4CK * k4PLANE;                     // 3072 floats = 12 DMA pieces of 1 KB
constexpr int k4U = 9 * k4CK * 64 * 4;                    // 9216 floats = 36 pieces
constexpr int k4V = 9 * k4CK * 16 * 4;                    // 2304 floats per tile row
constexpr int k4SMEM = 2 * k4U + 2 * k4RAW + 4 * k4V + 128;

// The order of the 36 transformed positions: the wave that computes rows (ra, rb) of V = B^T d B produces 12 values
// [ra][0..5], [rb][0..5] = 3 quads; waves 0, 1, 2 of a tile row take the row pairs (0, 5), (1, 2), (3, 4).
__host__ __device__ constexpr int w4_row(int q, int e) {
  return (q / 3 == 0) ? ((4 * (q % 3) + e) / 6 ? 5 : 0) : ((q / 3 == 1) ? ((4 * (q % 3) + e) / 6 ? 2 : 1) : ((4 * (q % 3) + e) / 6 ? 4 : 3));
}
__host__ __device__ constexpr int w4_col(int q, int e) { return (4 * (q % 3) + e) % 6; }
__host__ __device__ constexpr int w4_pos(int i, int j) {   // accumulator index (4 * quad + element) of position (i, j)
  return (3 * ((i == 0 || i == 5) ? 0 : ((i == 1 || i == 2) ? 1 : 2)) + (((i == 5 || i == 2 || i == 4) ? 6 : 0) + j) / 4) * 4 +
         (((i == 5 || i == 2 || i == 4) ? 6 : 0) + j) % 4;
}

typedef float f32x4w4 __attribute__((ext_vector_type(4)));

// Interpolation points 0, +-kP, +-kR, inf (profiles/HISTORY.md 3.1g).  Rounds 1-4 used Lavin's 0, +-1, +-2, inf; what dominates the error of
// F(4x4) in fp32 is the accumulation over the input channels of products whose magnitude - at the positions of the outermost
// points - is many times the output's (they cancel in A^T M A), and that ratio is a property of the points alone (row scalings
// between G, B^T and A^T leave it unchanged).  +-5/8, +-3/2 brings a layer's error on unit-scale data from 3.3e-5 max / 1.0e-6 mean
// to 0.9e-5 / 5.5e-7 (Cin 64; tools/exp_wino4_points.py) at the same instruction count: a symmetric pair keeps the even / odd
// sharing (P +- c Q), every constant is a dyadic rational (exact in fp32), and U = G g G^T is computed in double by the packs.
constexpr double k4P = 0.625, k4R = 1.5;
constexpr float k4P2 = (float)(k4P * k4P), k4R2 = (float)(k4R * k4R), k4PR2 = (float)(k4P * k4P * k4R * k4R),
                k4S2 = (float)(k4P * k4P + k4R * k4R), k4Pf = (float)k4P, k4Rf = (float)k4R,
                k4P3 = (float)(k4P * k4P * k4P), k4R3 = (float)(k4R * k4R * k4R);
static_assert((double)k4PR2 == k4P * k4P * k4R * k4R && (double)k4S2 == k4P * k4P + k4R * k4R && (double)k4P3 == k4P * k4P * k4P &&
              (double)k4R3 == k4R * k4R * k4R, "the interpolation points must give transform constants that are exact in fp32");
// B^T (rows: points 0, +P, -P, +R, -R, inf) = [PR2 0 -S2 0 1 0; 0 -P R2 -R2 P 1 0; 0 P R2 -R2 -P 1 0; 0 -R P2 -P2 R 1 0;
//   0 R P2 -P2 -R 1 0; 0 PR2 0 -S2 0 1]   (PR2 = P^2 R^2, S2 = P^2 + R^2);   G row of point x: (1, x, x^2) / prod(x - other points),
// G row 0 = (1 / PR2, 0, 0), G row inf = (0, 0, 1);   A^T = [1 1 1 1 1 0; 0 P -P R -R 0; 0 P2 P2 R2 R2 0; 0 P3 -P3 R3 -R3 1].
__host__ __device__ inline void w4_G(double (&G)[6][3]) {
  const double pt[4] = {k4P, -k4P, k4R, -k4R};
  G[0][0] = 1.0 / (k4P * k4P * k4R * k4R); G[0][1] = 0.0; G[0][2] = 0.0;
  for (int j = 0; j < 4; ++j) {
    double n = pt[j];                                      // (x - 0)
    for (int k = 0; k < 4; ++k)
      if (k != j) n *= pt[j] - pt[k];
    G[1 + j][0] = 1.0 / n; G[1 + j][1] = pt[j] / n; G[1 + j][2] = pt[j] * pt[j] / n;
  }
  G[5][0] = 0.0; G[5][1] = 0.0; G[5][2] = 1.0;
}

// one 1-D pass of the input transform restricted to a row pair: R = 0: rows 0, 5; 1: rows 1, 2; 2: rows 3, 4 of B^T
template <int R>
__device__ __forceinline__ void w4_bt_pair(float d0, float d1, float d2, float d3, float d4, float d5, float& oa, float& ob) {
  if (R == 0) {
    oa = fmaf(k4PR2, d0, fmaf(-k4S2, d2, d4));
    ob = fmaf(k4PR2, d1, fmaf(-k4S2, d3, d5));
  } else if (R == 1) {
    const float p = fmaf(-k4R2, d2, d4), q = fmaf(-k4R2, d1, d3);
    oa = fmaf(k4Pf, q, p);
    ob = fmaf(-k4Pf, q, p);
  } else {
    const float p = fmaf(-k4P2, d2, d4), q = fmaf(-k4P2, d1, d3);
    oa = fmaf(k4Rf, q, p);
    ob = fmaf(-k4Rf, q, p);
  }
}
// a full 1-D pass (all six rows of B^T) of one 6-vector
__device__ __forceinline__ void w4_bt_full(const float (&t)[6], float (&o)[6]) {
  o[0] = fmaf(k4PR2, t[0], fmaf(-k4S2, t[2], t[4]));
  const float p1 = fmaf(-k4R2, t[2], t[4]), q1 = fmaf(-k4R2, t[1], t[3]);
  o[1] = fmaf(k4Pf, q1, p1);
  o[2] = fmaf(-k4Pf, q1, p1);
  const float p2 = fmaf(-k4P2, t[2], t[4]), q2 = fmaf(-k4P2, t[1], t[3]);
  o[3] = fmaf(k4Rf, q2, p2);
  o[4] = fmaf(-k4Rf, q2, p2);
  o[5] = fmaf(k4PR2, t[1], fmaf(-k4S2, t[3], t[5]));
}
// A^T applied to one 6-vector
__device__ __forceinline__ void w4_at(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
  const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  y[0] = m0 + s1 + s2;
  y[1] = fmaf(k4Rf, d2, k4Pf * d1);
  y[2] = fmaf(k4R2, s2, k4P2 * s1);
  y[3] = fmaf(k4R3, d2, fmaf(k4P3, d1, m5));
}

template <bool GLU, bool STATS = false>
__global__ __launch_bounds__(512, 2) void wino4_conv3x3_kernel(Wino4Args a) {
  static_assert(!(GLU && STATS), "batch statistics are taken of the raw (plain-epilogue) convolution output");
  __shared__ __attribute__((aligned(16))) float smem[k4SMEM];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, cb = wave & 3;                // tile row / channel block of this wave
  int t = xcd_remap(blockIdx.x, gridDim.x);              // the cout groups of a tile run back to back on one XCD
  const int grp = t % a.ngroups;
  t /= a.ngroups;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * 8, x0 = tx * 64;
  const float* xb = a.x + (int64_t)b * a.xbs;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;
  float* us = smem;
  float* raws = smem + 2 * k4U;
  float* vs = smem + 2 * k4U + 2 * k4RAW + g * 2 * k4V;  // this tile row's two V images
  float* aff_s = smem + 2 * k4U + 2 * k4RAW + 4 * k4V;
  TGSR_W4STAMP(0);

  // ---- DMA plan: 48 pieces of 1 KB per stage.  The six transforming waves (cb < 3) copy the raw tile, two pieces each
  // (r6 = 3 g + cb and r6 + 6); the two others (cb = 3) copy U, 18 pieces each - what a stage costs beyond its MFMAs is the
  // instruction count of its busiest SIMD, and the transform already puts 2 x 58 on three of the four (stamps: all copies
  // spread evenly 3 733 cycles per stage, this split 3 651).  Out-of-image (and plane padding) lanes read the zero block, stride 0.
  const int r6 = 3 * g + cb;
  const float* rptr[2];
  int rstep[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = (((cb < 3 ? r6 : 0) + 6 * k) * 64 + lane) * 4;   // first float of this lane's 16-byte piece
    const int c = e / k4PLANE;
    const int rem = e - c * k4PLANE;
    const int r = rem / k4TC, j = rem - r * k4TC;
    const int gy = y0 - 1 + r, gx = x0 - 4 + j;
    const bool ok = rem < k4TR * k4TC && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   // W % 4 == 0
    rptr[k] = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(gy * a.W + gx) : g_wino4_zero;
    rstep[k] = ok ? (int)(k4CK * HW) : 0;
  }
  // LDS addresses of the copies as plain integers (wave-uniform: SGPR arithmetic, no generic-pointer casts in the loop)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr4_t)smem);
  const unsigned lds_raw = lds0 + (2 * k4U + r6 * 256) * 4;          // + buf * k4RAW * 4 (+ 6 KB: the second piece)
  const unsigned lds_u = lds0 + g * (18 * 1024);                     // + buf * k4U * 4 + k KB
  auto issue_raw = [&](int buf) {                        // waves cb < 3; stages 0, 1, 2, ... in order
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(rptr[k]), "s"(lds_raw + buf * (k4RAW * 4) + k * 6144) : "memory");
      rptr[k] += rstep[k];
    }
  };
  const float* ubase = a.upack + (int64_t)grp * k4U;     // stage 0 of this group
  const int64_t ustride = (int64_t)a.ngroups * k4U;
  const unsigned uoff0 = (unsigned)((g * 18 * 64 + lane) * 16);      // per-lane byte offset of this wave's first U piece
  auto issue_u = [&](int buf) {                          // waves cb = 3
#pragma unroll
    for (int k = 0; k < 18; ++k)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(uoff0 + k * 1024), "s"(ubase), "s"(lds_u + buf * (k4U * 4) + k * 1024) : "memory");
    ubase += ustride;
  };

  // ---- input transform of the row pair R = cb (cb = 3: none): lane = (tile l15, channel lg); the 6 x 6 patch of tile
  // l15 of tile row g = raw rows 4g .. 4g + 5, raw columns 4 l15 + 3 .. 4 l15 + 8
  const int rlane = lg * k4PLANE + (4 * g) * k4TC + 4 * l15;
  const int vwl = (lg * 16 + l15) * 4;
  auto t_read = [&](auto rc, const float* rawb, float (&d)[6][6]) {
    constexpr int R = decltype(rc)::value;
    const float* rp = rawb + rlane;
#pragma unroll
    for (int p = (R == 0 ? 0 : 1); p < (R == 0 ? 6 : 5); ++p) {
      const f32x4w4 mid = *reinterpret_cast<const f32x4w4*>(rp + p * k4TC + 4);
      d[p][0] = rp[p * k4TC + 3];
      d[p][1] = mid[0]; d[p][2] = mid[1]; d[p][3] = mid[2]; d[p][4] = mid[3];
      d[p][5] = rp[p * k4TC + 8];
    }
  };
  auto t_write = [&](auto rc, const float (&d)[6][6], float* vdst) {
    constexpr int R = decltype(rc)::value;
    float ta[6], tb[6], oa[6], ob[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)                           // column pass: rows (ra, rb) of B^T d
      w4_bt_pair<R>(R == 0 ? d[0][j] : 0.f, d[1][j], d[2][j], d[3][j], d[4][j], R == 0 ? d[5][j] : 0.f, ta[j], tb[j]);
    w4_bt_full(ta, oa);                                   // row pass: all six columns of (B^T d) B
    w4_bt_full(tb, ob);
    constexpr int kVQ = k4CK * 16 * 4;                     // floats per quad of a V image
    float* vp = vdst + vwl + (3 * R) * kVQ;
    *reinterpret_cast<f32x4w4*>(vp) = f32x4w4{oa[0], oa[1], oa[2], oa[3]};
    *reinterpret_cast<f32x4w4*>(vp + kVQ) = f32x4w4{oa[4], oa[5], ob[0], ob[1]};
    *reinterpret_cast<f32x4w4*>(vp + 2 * kVQ) = f32x4w4{ob[2], ob[3], ob[4], ob[5]};
  };

  if (tid < 128) {   // aff_s[cb * 16 + m] = scale, [64 + ...] = shift of accumulator row m of block cb
    const int lc = tid & 63, cbk = lc >> 4, m = lc & 15;
    int col = GLU ? ((m & 2) ? (a.Cout >> 1) : 0) + grp * 32 + cbk * 8 + 2 * (m >> 2) + (m & 1) : grp * 64 + lc;
    if (col >= a.Cout) col = 0;
    aff_s[tid] = a.scale ? (tid < 64 ? a.scale[col] : a.shift[col]) : (tid < 64 ? 1.f : 0.f);
  }

  f32x4w4 M[36];
#pragma unroll
  for (int p = 0; p < 36; ++p)
#pragma unroll
    for (int i = 0; i < 4; ++i) M[p][i] = 0.f;

  const int ulane = (lg * 64 + cb * 16 + l15) * 4;       // A: U[q][ci = lg][cb][l15][4]
  const int vlane = (lg * 16 + l15) * 4;                 // B: V[q][ci = lg][l15][4]

  // One stage: the copies U(st+1) [MORE] and raw(st+2) [MORE2] first - they have the whole stage to land; the raw reads of the
  // transform of raw(st+1); then ONE scheduling region with the 36 MFMAs on U(st), V(st), their 18 fragment reads two quads
  // ahead, and the transform's 48 VALU operations spread over the MFMA gaps, one or two per gap, with its three V writes behind
  // quad 7; the wait for this stage's copies and the barrier that publishes them and V(st+1).
  // What a stage costs (stamps + builds with parts compiled out, tools/wino4_stamps.py, 64 -> 128 @128^2): 2 304 cycles of MFMA
  // (2 waves x 36 x 32), + 150 for the fragment reads, + 900 for the transform, + 375 for the copies = 3 700.  The transform's
  // 900 are its 2 x 58 instructions on the busiest SIMD at ~7.7 cycles each: beside fp32 MFMAs a VALU / LDS instruction costs
  // that much of the SIMD's time wherever it sits - as one block behind quad 2, staggered between the two waves of a SIMD, or
  // one per MFMA gap as here, the stage took the same 3 650-3 730 cycles (the 24 free issue cycles per gap the guide measures
  // beside bf16 MFMAs do not exist beside v_mfma_f32_16x16x4_f32; tgsr_winograd.hip found the same in round 1).  Fewer
  // non-MFMA instructions per SIMD is the only lever: hence the copy split above (-80 cycles per stage).
  // The transform is NOT skipped in the last stage (a branch would split the region): it reads the stale raw buffer and writes
  // the V image nobody reads.  The buffer parity is compile-time (the loop runs two stages per trip), the two flags are
  // wave-uniform branches: with a stage instantiated per flag combination hipcc lost the in-place accumulators between the
  // copies (900 spilled registers).
  auto stage = [&](auto rc, auto parc, const bool MORE, const bool MORE2) {
    constexpr int R = decltype(rc)::value, PAR = decltype(parc)::value;
#if defined(TGSR_W4_EXP)   // diagnostic builds (wrong results): bit 0 = no U copies, bit 1 = no transform, bit 2 = no raw copies
    constexpr bool kNoU = TGSR_W4_EXP & 1, kNoT = TGSR_W4_EXP & 2, kNoRaw = TGSR_W4_EXP & 4;
#else
    constexpr bool kNoU = false, kNoT = false, kNoRaw = false;
#endif
    constexpr bool TR = R < 3 && !kNoT;
    if (R == 3 && MORE && !kNoU) issue_u(PAR ^ 1);       // U(st+1) replaces U(st-1)
    if (R < 3 && MORE2 && !kNoRaw) issue_raw(PAR);       // raw(st+2) replaces raw(st), transformed one stage ago
    float d[6][6];
    if (TR) t_read(rc, raws + (PAR ^ 1) * k4RAW, d);
    __builtin_amdgcn_sched_barrier(0);
    const float* ub = us + PAR * k4U + ulane;
    const float* vb = vs + PAR * k4V + vlane;
    f32x4w4 af[3], bf[3];
    af[0] = *reinterpret_cast<const f32x4w4*>(ub);
    bf[0] = *reinterpret_cast<const f32x4w4*>(vb);
    af[1] = *reinterpret_cast<const f32x4w4*>(ub + (k4CK * 64 * 4));
    bf[1] = *reinterpret_cast<const f32x4w4*>(vb + (k4CK * 16 * 4));
    if (TR) t_write(rc, d, vs + (PAR ^ 1) * k4V);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      if (q + 2 < 9) {
        af[(q + 2) % 3] = *reinterpret_cast<const f32x4w4*>(ub + (q + 2) * (k4CK * 64 * 4));
        bf[(q + 2) % 3] = *reinterpret_cast<const f32x4w4*>(vb + (q + 2) * (k4CK * 16 * 4));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        M[4 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q % 3][e], bf[q % 3][e], M[4 * q + e], 0, 0, 0);
    }
    // the order of the region: masks 0x008 MFMA, 0x002 VALU, 0x100 LDS read, 0x200 LDS write
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      if (q + 2 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (TR && q < 8) {
          if (e < 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
      }
      if (TR && q == 7) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MORE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto run = [&](auto rc) {
    constexpr int R = decltype(rc)::value;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // prologue: raw(0), U(0), raw(1); transform raw(0) -> V[0]
    if (R < 3) {
      issue_raw(0);
      if (a.nstages > 1) issue_raw(1);
    } else {
      issue_u(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (R < 3) {
      float d[6][6];
      t_read(rc, raws, d);
      t_write(rc, d, vs);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TGSR_W4STAMP(1);
    int st = 0;
    for (; st + 1 < a.nstages; st += 2) {
      stage(rc, P0{}, true, st + 2 < a.nstages);
      stage(rc, P1{}, st + 2 < a.nstages, st + 3 < a.nstages);
    }
    if (st < a.nstages) stage(rc, P0{}, false, false);   // odd stage count
  };
  if (cb == 0) run(std::integral_constant<int, 0>{});
  else if (cb == 1) run(std::integral_constant<int, 1>{});
  else if (cb == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});
  TGSR_W4STAMP(2);

  // ---- output transform Y = A^T M A + epilogue; lane = tile l15, register i = accumulator row 4 lg + i of block cb
  auto ytile = [&](int i, float (&y)[4][4]) {
    float c[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float col[4];
      w4_at(M[w4_pos(0, j)][i], M[w4_pos(1, j)][i], M[w4_pos(2, j)][i], M[w4_pos(3, j)][i], M[w4_pos(4, j)][i], M[w4_pos(5, j)][i], col);
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r][j] = col[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) w4_at(c[r][0], c[r][1], c[r][2], c[r][3], c[r][4], c[r][5], y[r]);
  };
  const int oy = y0 + 4 * g, ox = x0 + 4 * l15;
  const int64_t HWo = (int64_t)a.H * a.W;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  const float* __restrict__ rb = a.res ? a.res + (int64_t)b * a.rbs : nullptr;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  if (ox < a.W) {
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
          if (oy + r >= a.H) continue;
          f32x4w4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            o[k] = (yv[r][k] * sv + tv) * __builtin_amdgcn_rcpf(1.f + __expf(-(yg[r][k] * sg + tg)));
          *reinterpret_cast<f32x4w4*>(ob + (int64_t)c * HWo + (int64_t)(oy + r) * a.W + ox) = o;
        }
      }
    } else {
      // the residual tile of this lane (4 channels x 4 rows of 4 pixels): all 16 loads in flight before the first transform
      f32x4w4 rr[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = grp * 64 + cb * 16 + 4 * lg + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          rr[i][r] = f32x4w4{0.f, 0.f, 0.f, 0.f};
          if (rb && oy + r < a.H) rr[i][r] = *reinterpret_cast<const f32x4w4*>(rb + (int64_t)c * HWo + (int64_t)(oy + r) * a.W + ox);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 4 * lg + i, c = grp * 64 + cb * 16 + m;
        float yv[4][4];
        ytile(i, yv);
        const float sv = aff_s[cb * 16 + m], tv = aff_s[64 + cb * 16 + m];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (oy + r >= a.H) continue;
          f32x4w4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = yv[r][k] * sv + tv + rr[i][r][k];
          *reinterpret_cast<f32x4w4*>(ob + (int64_t)c * HWo + (int64_t)(oy + r) * a.W + ox) = o;
          if (STATS) {
            ssum[i] += (o[0] + o[1]) + (o[2] + o[3]);
            ssq[i] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
          }
        }
      }
    }
  }
  if (STATS) {
    // BatchNorm's batch statistics ride the epilogue (as in tgsr_winograd.hip): this wave's 4 x 64 outputs of each of its 16
    // channels are summed over the 16 tiles (lanes l15 of a lane group; out-of-image lanes hold zeros) and leave as ONE
    // (sum, sumsq) pair per channel and wave - slot = (sample, tile row pair, 64-column chunk, g); the normalise pass
    // combines the slots in a fixed order
    const int slot = ((b * a.tiles_y + ty) * a.tiles_x + tx) * 2 + g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sm = ssum[i], sq = ssq[i];
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        sm += __shfl_xor(sm, o);
        sq += __shfl_xor(sq, o);
      }
      if (l15 == 0) {
        const int c = grp * 64 + cb * 16 + 4 * lg + i;
        *reinterpret_cast<float2*>(a.stat + ((int64_t)c * a.nslots + slot) * 2) = make_float2(sm, sq);
      }
    }
  }
  TGSR_W4STAMP(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// The wide form for layers with Cout % 128 == 0 (the GLU convolutions 64 -> 128 of the ResBlocks): a workgroup = 8 waves =
// ONE tile row (4 x 64 outputs) x 8 channel blocks = 128 accumulator rows.  What a stage costs beyond its MFMAs is the
// non-MFMA instruction count of its busiest SIMD, and most of that is the input transform (kernel above: 900 of 1 400
// cycles): here one tile row's transform (3 waves) feeds 8 waves of MFMAs instead of 4, so a SIMD carries ONE transforming wave.
// Sixteen accumulator rows more per tile row need twice the U per stage - 74 KB, which double-buffered no longer fits LDS -
// so the A operands come straight from L2 into registers: the pack is in per-wave fragment order [stage][group][cb 8][quad 9]
// [lane 64][4], one global_load_dwordx4 per quad, issued behind the quad's MFMAs for the NEXT stage into the same registers
// (a rolling single buffer: the load has a whole stage to return; the registers' old contents were read by MFMAs issued
// before it).  The loads are inline assembly with explicit, counted waits: per wave the VMEM stream is [raw copies of this
// stage, 0-2][A loads of the next stage, 9], so `vmcnt(8)` in front of quad q's MFMAs means "A(q) has arrived" (eight younger
// A loads may be in flight; younger copies only make the wait stricter) and `vmcnt(9)` at the stage's end means "my copies
// have landed".  The last stage re-loads stage 0's fragments (harmless, keeps the counting uniform); `vmcnt(0)` in front of
// the epilogue keeps late arrivals out of registers the epilogue reuses.
// LDS: raw [4 ci][6 rows][72] in planes of 448 floats x 2 + V x 2 + affine table = 34 KB.
constexpr int k4wTR = 6, k4wPLANE = 448, k4wRAW = k4CK * k4wPLANE;     // 1792 floats = 7 DMA pieces
constexpr int k4wSMEM = 2 * k4wRAW + 2 * k4V + 256;
constexpr int k4wUW = 9 * 256;                                          // floats of U per wave and stage

// NB = channel blocks (waves) per workgroup: 8 = the wide form proper (128 rows); 4 = the same register-fed kernel for 64-row
// groups (Cout % 128 != 0: the +residual convolutions): 4-wave workgroups, two per CU that share nothing, so one's prologue and
// epilogue - the residual variant's 131 KB in / 131 KB out burst - run under the other's main loop; the transform is not amortised
// further than in the narrow form (three of four waves transform), what it saves is the 36 U copies and the A reads from LDS.
template <bool GLU, int NB, bool STATS = false>
__global__ __launch_bounds__(64 * NB, 2) void wino4w_conv3x3_kernel(Wino4Args a) {
  static_assert(!(GLU && STATS), "batch statistics are taken of the raw (plain-epilogue) convolution output");
  constexpr int NR = 16 * NB;                              // accumulator rows of a workgroup (128 | 64)
  __shared__ __attribute__((aligned(16))) float smem[k4wSMEM];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int cb = __builtin_amdgcn_readfirstlane(tid >> 6);             // channel block of this wave
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = t % a.ngroups;
  t /= a.ngroups;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * 4, x0 = tx * 64;
  const float* xb = a.x + (int64_t)b * a.xbs;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;
  float* raws = smem;
  float* vs = smem + 2 * k4wRAW;
  float* aff_s = smem + 2 * k4wRAW + 2 * k4V;

  // ---- raw copies: 7 pieces per stage on the five waves that do not transform - wave 3: pieces 0, 1; wave 7: 2, 3; waves
  // 4, 5, 6: pieces 4, 5, 6
  // (NB = 4: every wave copies - pieces 2 cb, 2 cb + 1 on waves 0-2, piece 6 on wave 3)
  const int npiece = NB == 8 ? ((cb == 3 || cb == 7) ? 2 : (cb >= 4 ? 1 : 0)) : (cb == 3 ? 1 : 2);
  const int piece0 = NB == 8 ? (cb == 3 ? 0 : (cb == 7 ? 2 : cb)) : 2 * cb;
  const float* rptr[2];
  int rstep[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = (((npiece ? piece0 : 0) + k) * 64 + lane) * 4;
    const int c = e / k4wPLANE;
    const int rem = e - c * k4wPLANE;
    const int r = rem / k4TC, j = rem - r * k4TC;
    const int gy = y0 - 1 + r, gx = x0 - 4 + j;
    const bool ok = e < k4wRAW && rem < k4wTR * k4TC && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    rptr[k] = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(gy * a.W + gx) : g_wino4_zero;
    rstep[k] = ok ? (int)(k4CK * HW) : 0;
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr4_t)smem);
  const unsigned lds_raw = lds0 + piece0 * 1024;
  auto issue_raw = [&](int buf) {
    if (npiece >= 1) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(rptr[0]), "s"(lds_raw + buf * (k4wRAW * 4)) : "memory");
      rptr[0] += rstep[0];
    }
    if (npiece >= 2) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(rptr[1]), "s"(lds_raw + buf * (k4wRAW * 4) + 1024) : "memory");
      rptr[1] += rstep[1];
    }
  };
  // ---- A fragments: this wave's 9 KB of a stage, three per-lane offsets 4 KB apart + an immediate
  const float* ubase = a.upack + ((int64_t)grp * NB + cb) * k4wUW;     // stage 0
  const float* const ubase0 = ubase;
  const int64_t ustride = (int64_t)a.ngroups * NB * k4wUW;
  unsigned voff[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) voff[k] = (unsigned)(lane * 16 + k * 4096);
  f32x4w4 af[9];
#define TGSR_W4W_LOAD(q)                                                                                               \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(af[q]) : "v"(voff[(q) / 4]), "s"(ubase), "n"(((q) % 4) * 1024) : "memory")
#define TGSR_W4W_WAIT(q) asm volatile("s_waitcnt vmcnt(8)" : "+v"(af[q])::"memory")

  // ---- input transform: waves 0, 1, 2 take the row pairs (0, 5), (1, 2), (3, 4) of B^T (as in the kernel above, one tile row)
  const int rlane = lg * k4wPLANE + 4 * l15;
  const int vwl = (lg * 16 + l15) * 4;
  auto t_read = [&](auto rc, const float* rawb, float (&d)[6][6]) {
    constexpr int R = decltype(rc)::value;
    const float* rp = rawb + rlane;
#pragma unroll
    for (int p = (R == 0 ? 0 : 1); p < (R == 0 ? 6 : 5); ++p) {
      const f32x4w4 mid = *reinterpret_cast<const f32x4w4*>(rp + p * k4TC + 4);
      d[p][0] = rp[p * k4TC + 3];
      d[p][1] = mid[0]; d[p][2] = mid[1]; d[p][3] = mid[2]; d[p][4] = mid[3];
      d[p][5] = rp[p * k4TC + 8];
    }
  };
  auto t_write = [&](auto rc, const float (&d)[6][6], float* vdst) {
    constexpr int R = decltype(rc)::value;
    float ta[6], tb[6], oa[6], ob[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
      w4_bt_pair<R>(R == 0 ? d[0][j] : 0.f, d[1][j], d[2][j], d[3][j], d[4][j], R == 0 ? d[5][j] : 0.f, ta[j], tb[j]);
    w4_bt_full(ta, oa);
    w4_bt_full(tb, ob);
    constexpr int kVQ = k4CK * 16 * 4;
    float* vp = vdst + vwl + (3 * R) * kVQ;
    *reinterpret_cast<f32x4w4*>(vp) = f32x4w4{oa[0], oa[1], oa[2], oa[3]};
    *reinterpret_cast<f32x4w4*>(vp + kVQ) = f32x4w4{oa[4], oa[5], ob[0], ob[1]};
    *reinterpret_cast<f32x4w4*>(vp + 2 * kVQ) = f32x4w4{ob[2], ob[3], ob[4], ob[5]};
  };

  if (tid < 2 * NR) {   // aff_s[cb * 16 + m] = scale, [NR + ...] = shift of accumulator row m of block cb
    const int lc = tid & (NR - 1), cbk = lc >> 4, m = lc & 15;
    int col = GLU ? ((m & 2) ? (a.Cout >> 1) : 0) + grp * (NR / 2) + cbk * 8 + 2 * (m >> 2) + (m & 1) : grp * NR + lc;
    if (col >= a.Cout) col = 0;
    aff_s[tid] = a.scale ? (tid < NR ? a.scale[col] : a.shift[col]) : (tid < NR ? 1.f : 0.f);
  }

  f32x4w4 M[36];
#pragma unroll
  for (int p = 0; p < 36; ++p)
#pragma unroll
    for (int i = 0; i < 4; ++i) M[p][i] = 0.f;
  const int vlane = (lg * 16 + l15) * 4;

  auto stage = [&](auto rc, auto parc, const bool MORE, const bool MORE2) {
    constexpr int R = decltype(rc)::value, PAR = decltype(parc)::value;
    if (MORE2) issue_raw(PAR);                           // raw(st+2) replaces raw(st), transformed one stage ago
    ubase = MORE ? ubase + ustride : ubase0;             // where the fragments loaded during this stage come from
    float d[6][6];
    if (R < 3) t_read(rc, raws + (PAR ^ 1) * k4wRAW, d);
    const float* vb = vs + PAR * k4V + vlane;
    f32x4w4 bf[3];
    bf[0] = *reinterpret_cast<const f32x4w4*>(vb);
    bf[1] = *reinterpret_cast<const f32x4w4*>(vb + (k4CK * 16 * 4));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      if (q + 2 < 9) bf[(q + 2) % 3] = *reinterpret_cast<const f32x4w4*>(vb + (q + 2) * (k4CK * 16 * 4));
      TGSR_W4W_WAIT(q);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        M[4 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][e], bf[q % 3][e], M[4 * q + e], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      TGSR_W4W_LOAD(q);
      if (q == 2 && R < 3) {
        t_write(rc, d, vs + (PAR ^ 1) * k4V);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MORE) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto run = [&](auto rc) {
    constexpr int R = decltype(rc)::value;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    issue_raw(0);
    if (a.nstages > 1) issue_raw(1);
    TGSR_W4W_LOAD(0); TGSR_W4W_LOAD(1); TGSR_W4W_LOAD(2); TGSR_W4W_LOAD(3); TGSR_W4W_LOAD(4);
    TGSR_W4W_LOAD(5); TGSR_W4W_LOAD(6); TGSR_W4W_LOAD(7); TGSR_W4W_LOAD(8);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (R < 3) {
      float d[6][6];
      t_read(rc, raws, d);
      t_write(rc, d, vs);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // an EVEN number of stages (host-checked): exactly two copies of the stage, P0 -> P1 -> back edge.  A third copy behind the
    // loop (an odd tail) got other physical registers for af[] and hipcc bridged them with v_mov - of fragments that had not
    // arrived yet (every output wrong at Cin = 12; the parity tests of the wide form would catch a recurrence)
    for (int st = 0; st < a.nstages; st += 2) {
      stage(rc, P0{}, true, st + 2 < a.nstages);
      stage(rc, P1{}, st + 2 < a.nstages, st + 3 < a.nstages);
    }
  };
  if (cb == 0) run(std::integral_constant<int, 0>{});
  else if (cb == 1) run(std::integral_constant<int, 1>{});
  else if (cb == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last stage's (unused) fragment loads: see the header
#undef TGSR_W4W_LOAD
#undef TGSR_W4W_WAIT

  auto ytile = [&](int i, float (&y)[4][4]) {
    float c[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float col[4];
      w4_at(M[w4_pos(0, j)][i], M[w4_pos(1, j)][i], M[w4_pos(2, j)][i], M[w4_pos(3, j)][i], M[w4_pos(4, j)][i], M[w4_pos(5, j)][i], col);
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r][j] = col[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) w4_at(c[r][0], c[r][1], c[r][2], c[r][3], c[r][4], c[r][5], y[r]);
  };
  const int oy = y0, ox = x0 + 4 * l15;
  const int64_t HWo = (int64_t)a.H * a.W;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  const float* __restrict__ rb = a.res ? a.res + (int64_t)b * a.rbs : nullptr;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  if (ox < a.W) {
    if (GLU) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float yv[4][4], yg[4][4];
        ytile(p, yv);
        ytile(p + 2, yg);
        const int m = 4 * lg + p;
        const float sv = aff_s[cb * 16 + m], tv = aff_s[NR + cb * 16 + m], sg = aff_s[cb * 16 + m + 2], tg = aff_s[NR + cb * 16 + m + 2];
        const int c = grp * (NR / 2) + cb * 8 + 2 * lg + p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (oy + r >= a.H) continue;
          f32x4w4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            o[k] = (yv[r][k] * sv + tv) * __builtin_amdgcn_rcpf(1.f + __expf(-(yg[r][k] * sg + tg)));
          *reinterpret_cast<f32x4w4*>(ob + (int64_t)c * HWo + (int64_t)(oy + r) * a.W + ox) = o;
        }
      }
    } else {
      f32x4w4 rr[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = grp * NR + cb * 16 + 4 * lg + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          rr[i][r] = f32x4w4{0.f, 0.f, 0.f, 0.f};
          if (rb && oy + r < a.H) rr[i][r] = *reinterpret_cast<const f32x4w4*>(rb + (int64_t)c * HWo + (int64_t)(oy + r) * a.W + ox);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 4 * lg + i, c = grp * NR + cb * 16 + m;
        float yv[4][4];
        ytile(i, yv);
        const float sv = aff_s[cb * 16 + m], tv = aff_s[NR + cb * 16 + m];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (oy + r >= a.H) continue;
          f32x4w4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = yv[r][k] * sv + tv + rr[i][r][k];
          *reinterpret_cast<f32x4w4*>(ob + (int64_t)c * HWo + (int64_t)(oy + r) * a.W + ox) = o;
          if (STATS) {
            ssum[i] += (o[0] + o[1]) + (o[2] + o[3]);
            ssq[i] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
          }
        }
      }
    }
  }
  if (STATS) {
    // BatchNorm's batch statistics (as in the narrow form): one (sum, sumsq) pair per channel and wave - slot = (sample, tile
    // row, 64-column chunk): the narrow form's pairs in another order
    const int slot = (b * a.tiles_y + ty) * a.tiles_x + tx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sm = ssum[i], sq = ssq[i];
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        sm += __shfl_xor(sm, o);
        sq += __shfl_xor(sq, o);
      }
      if (l15 == 0) {
        const int c = grp * NR + cb * 16 + 4 * lg + i;
        *reinterpret_cast<float2*>(a.stat + ((int64_t)c * a.nslots + slot) * 2) = make_float2(sm, sq);
      }
    }
  }
}

// wide pack: upack[stage][group of 16 nb rows][cb nb][quad 9][lane 64 = (ci lane >> 4, row lane & 15)][4], nb = 8 (Cout % 128 == 0)
// or 4; row m of block cb: plain = cout grp*16nb + cb*16 + m; GLU = value channel grp*8nb + cb*8 + 2 (m >> 2) + (m & 1) when
// m & 2 == 0, else its gate
// tr != 0: the source is the FORWARD conv's weight [Cin][Cout][3][3]; the pack is of the data-gradient conv (plain order)
__global__ void pack_wino4w_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int glu, int nb,
                                          int tr, int64_t total) {
  double G[6][3];
  w4_G(G);
  const int ngrp = Cout / (16 * nb);                       // nb = 8 | 4 blocks per group
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 3), m = (int)((idx >> 2) & 15), ci = (int)((idx >> 6) & 3);
    int64_t t = idx >> 8;
    const int q = (int)(t % 9);
    t /= 9;
    const int cbk = (int)(t % nb);
    t /= nb;
    const int grp = (int)(t % ngrp);
    const int st = (int)(t / ngrp);
    const int i = w4_row(q, e), j = w4_col(q, e);
    const int co = glu ? ((m & 2) ? (Cout >> 1) : 0) + grp * (8 * nb) + cbk * 8 + 2 * (m >> 2) + (m & 1) : grp * (16 * nb) + cbk * 16 + m;
    const int c = st * k4CK + ci;
    double u = 0.0;
    if (c < Cin) {
      const float* gw = tr ? w + ((int64_t)c * Cout + co) * 9 : w + ((int64_t)co * Cin + c) * 9;
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 3; ++l) u += G[i][k] * (double)(tr ? gw[8 - (k * 3 + l)] : gw[k * 3 + l]) * G[j][l];
    }
    up[idx] = (float)u;
  }
}

// upack[stage][group][quad 9][ci 4][cb 4][row 16][4] <- U = G g G^T (computed in double, rounded once); a group is the 64
// accumulator rows of one workgroup.  Row m of block cb: plain = cout grp*64 + cb*16 + m; GLU = value channel
// grp*32 + cb*8 + 2 (m >> 2) + (m & 1) when m & 2 == 0, its gate (+ Cout/2) otherwise.
// tr != 0: the source is the FORWARD conv's weight [Cin][Cout][3][3]; the pack is of the data-gradient conv
// w'[co][c][r][k] = w[c][co][2 - r][2 - k] (plain channel order)
__global__ void pack_wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int glu, int tr,
                                         int64_t total) {
  double G[6][3];
  w4_G(G);
  const int ngrp = Cout / 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 3), m = (int)((idx >> 2) & 15), cbk = (int)((idx >> 6) & 3), ci = (int)((idx >> 8) & 3);
    int64_t t = idx >> 10;
    const int q = (int)(t % 9);
    t /= 9;
    const int grp = (int)(t % ngrp);
    const int st = (int)(t / ngrp);
    const int i = w4_row(q, e), j = w4_col(q, e);
    const int co = glu ? ((m & 2) ? (Cout >> 1) : 0) + grp * 32 + cbk * 8 + 2 * (m >> 2) + (m & 1) : grp * 64 + cbk * 16 + m;
    const int c = st * k4CK + ci;
    double u = 0.0;
    if (c < Cin) {
      const float* gw = tr ? w + ((int64_t)c * Cout + co) * 9 : w + ((int64_t)co * Cin + c) * 9;
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 3; ++l) u += G[i][k] * (double)(tr ? gw[8 - (k * 3 + l)] : gw[k * 3 + l]) * G[j][l];
    }
    up[idx] = (float)u;
  }
}

}  // namespace tgsr

using namespace tgsr;

#ifdef TGSR_WINO4_STAMPS
extern "C" int tgsr_debug_read_w4stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(tgsr::g_w4stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int64_t tgsr_packed_wino4_weight_elems(int Cout, int Cin) {
  return (int64_t)((Cin + k4CK - 1) / k4CK) * 36 * k4CK * Cout;
}

static int pack_wino4_weight(const float* w, float* upack, int Cout, int Cin, int glu, int tr, void* stream) {
  if (!w || !upack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  if (Cout % 64 != 0) return TGSR_EUNSUPPORTED;
  const int64_t total = tgsr_packed_wino4_weight_elems(Cout, Cin);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_wino4_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, upack, Cout, Cin, glu ? 1 : 0, tr,
                     total);
  return note_launch(hipGetLastError(), "pack_wino4_weight_kernel");
}

extern "C" int tgsr_pack_wino4_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream) {
  return pack_wino4_weight(w, upack, Cout, Cin, glu, 0, stream);
}

extern "C" int tgsr_pack_wino4_weight_dgrad(const float* w, float* upack, int Cout, int Cin, void* stream) {
  return pack_wino4_weight(w, upack, Cout, Cin, 0, 1, stream);
}

static void wino4_geometry(int B, int Cin, int H, int W, int Cout, Wino4Args& a) {
  a.tiles_x = (W + 63) / 64; a.tiles_y = (H + 7) / 8; a.nstages = Cin / k4CK; a.ngroups = Cout / 64;
  a.nslots = B * a.tiles_y * a.tiles_x * 2;
}

extern "C" int tgsr_wino4_stats_nslots(int B, int H, int W, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Cout < 1 || Cout % 64 != 0) return 0;
  Wino4Args a;
  wino4_geometry(B, 4, H, W, Cout, a);
  return a.nslots;
}

static int wino4_conv3x3(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                         const float* scale, const float* shift, const float* residual, int64_t res_bstride, float* out,
                         int64_t out_bstride, int epilogue, float* stat, void* stream) {
  if (!x || !upack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  const bool glu = epilogue == TGSR_EPI_AFFINE_GLU;
  if (!glu && epilogue != TGSR_EPI_AFFINE) return TGSR_EINVAL;
  if (glu && residual) return TGSR_EINVAL;
  if (stat && (glu || residual || scale)) return TGSR_EINVAL;     // statistics are of the raw convolution output
  if (Cout % 64 != 0 || Cin % k4CK != 0) return TGSR_EUNSUPPORTED;
  if ((int64_t)H * W >= (1 << 28) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  if ((W & 3) || (x_bstride & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (out_bstride & 3) || (residual && ((reinterpret_cast<uintptr_t>(residual) & 15) || (res_bstride & 3))))
    return TGSR_EUNSUPPORTED;
  Wino4Args a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.upack = upack; a.Cout = Cout;
  a.scale = scale; a.shift = shift; a.res = residual; a.rbs = res_bstride; a.out = out; a.obs = out_bstride;
  wino4_geometry(B, Cin, H, W, Cout, a);
  a.stat = stat;
  const dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y * a.ngroups));
  if (glu) hipLaunchKernelGGL((wino4_conv3x3_kernel<true>), grid, dim3(512), 0, as_stream(stream), a);
  else if (stat) hipLaunchKernelGGL((wino4_conv3x3_kernel<false, true>), grid, dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((wino4_conv3x3_kernel<false>), grid, dim3(512), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "wino4_conv3x3_kernel");
}

extern "C" int tgsr_wino4_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack,
                                      int Cout, const float* scale, const float* shift, const float* residual,
                                      int64_t res_bstride, float* out, int64_t out_bstride, int epilogue, void* stream) {
  return wino4_conv3x3(x, x_bstride, B, Cin, H, W, upack, Cout, scale, shift, residual, res_bstride, out, out_bstride, epilogue,
                       nullptr, stream);
}

extern "C" int tgsr_wino4_conv3x3_stats_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack,
                                            int Cout, float* out, int64_t out_bstride, float* stat_partial, void* stream) {
  if (!stat_partial) return TGSR_EINVAL;
  return wino4_conv3x3(x, x_bstride, B, Cin, H, W, upack, Cout, nullptr, nullptr, nullptr, 0, out, out_bstride, TGSR_EPI_AFFINE,
                       stat_partial, stream);
}

static int pack_wino4_wide_weight(const float* w, float* upack, int Cout, int Cin, int glu, int tr, void* stream) {
  if (!w || !upack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  if (Cout % 64 != 0) return TGSR_EUNSUPPORTED;
  const int64_t total = tgsr_packed_wino4_weight_elems(Cout, Cin);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_wino4w_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, upack, Cout, Cin, glu ? 1 : 0,
                     Cout % 128 == 0 ? 8 : 4, tr, total);
  return note_launch(hipGetLastError(), "pack_wino4w_weight_kernel");
}

extern "C" int tgsr_pack_wino4_wide_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream) {
  return pack_wino4_wide_weight(w, upack, Cout, Cin, glu, 0, stream);
}

extern "C" int tgsr_pack_wino4_wide_weight_dgrad(const float* w, float* upack, int Cout, int Cin, void* stream) {
  return pack_wino4_wide_weight(w, upack, Cout, Cin, 0, 1, stream);
}

static int wino4_wide_conv3x3(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                              const float* scale, const float* shift, const float* residual, int64_t res_bstride, float* out,
                              int64_t out_bstride, int epilogue, float* stat, void* stream) {
  if (!x || !upack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  const bool glu = epilogue == TGSR_EPI_AFFINE_GLU;
  if (!glu && epilogue != TGSR_EPI_AFFINE) return TGSR_EINVAL;
  if (glu && residual) return TGSR_EINVAL;
  if (stat && (glu || residual || scale)) return TGSR_EINVAL;     // statistics are of the raw convolution output
  if (Cout % 64 != 0 || Cin % (2 * k4CK) != 0) return TGSR_EUNSUPPORTED;       // an even number of 4-channel stages
  if ((int64_t)H * W >= (1 << 28) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  if ((W & 3) || (x_bstride & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(upack) & 15) || (out_bstride & 3) ||
      (residual && ((reinterpret_cast<uintptr_t>(residual) & 15) || (res_bstride & 3))))
    return TGSR_EUNSUPPORTED;
  const int nb = Cout % 128 == 0 ? 8 : 4;                  // 128-row groups where the layer has them, else 64-row groups
  Wino4Args a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.upack = upack; a.Cout = Cout;
  a.scale = scale; a.shift = shift; a.res = residual; a.rbs = res_bstride; a.out = out; a.obs = out_bstride;
  a.tiles_x = (W + 63) / 64; a.tiles_y = (H + 3) / 4; a.nstages = Cin / k4CK; a.ngroups = Cout / (16 * nb);
  a.stat = stat; a.nslots = B * a.tiles_y * a.tiles_x;
  const dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y * a.ngroups));
  if (nb == 8) {
    if (glu) hipLaunchKernelGGL((wino4w_conv3x3_kernel<true, 8>), grid, dim3(512), 0, as_stream(stream), a);
    else if (stat) hipLaunchKernelGGL((wino4w_conv3x3_kernel<false, 8, true>), grid, dim3(512), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((wino4w_conv3x3_kernel<false, 8>), grid, dim3(512), 0, as_stream(stream), a);
  } else {
    if (glu) hipLaunchKernelGGL((wino4w_conv3x3_kernel<true, 4>), grid, dim3(256), 0, as_stream(stream), a);
    else if (stat) hipLaunchKernelGGL((wino4w_conv3x3_kernel<false, 4, true>), grid, dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((wino4w_conv3x3_kernel<false, 4>), grid, dim3(256), 0, as_stream(stream), a);
  }
  return note_launch(hipGetLastError(), "wino4w_conv3x3_kernel");
}

extern "C" int tgsr_wino4_wide_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack,
                                           int Cout, const float* scale, const float* shift, const float* residual,
                                           int64_t res_bstride, float* out, int64_t out_bstride, int epilogue, void* stream) {
  return wino4_wide_conv3x3(x, x_bstride, B, Cin, H, W, upack, Cout, scale, shift, residual, res_bstride, out, out_bstride,
                            epilogue, nullptr, stream);
}

extern "C" int tgsr_wino4_wide_stats_nslots(int B, int H, int W, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Cout < 1 || Cout % 64 != 0) return 0;
  return B * ((H + 3) / 4) * ((W + 63) / 64);
}

extern "C" int tgsr_wino4_wide_conv3x3_stats_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W,
                                                 const float* upack, int Cout, float* out, int64_t out_bstride,
                                                 float* stat_partial, void* stream) {
  if (!stat_partial) return TGSR_EINVAL;
  return wino4_wide_conv3x3(x, x_bstride, B, Cin, H, W, upack, Cout, nullptr, nullptr, nullptr, 0, out, out_bstride,
                            TGSR_EPI_AFFINE, stat_partial, stream);
}
