// Fused 3x3 convolution for gfx950: fp32 implicit GEMM on v_mfma_f32_32x32x2_f32 with the reference's
// epilogues (BatchNorm-eval affine, GLU, residual add) and the nearest-x2 up-sample folded into the LDS read.
//
// Replaces conv3x3 / upBlock / ResBlock.block of the reference (util.py:62-65, 74-80, 110-130) - see
// include/tgsr_hip.h.  Written for CDNA4 directly:
//   * GEMM orientation D[cout][pixel] = W[cout][k] * X[k][pixel]: the MFMA "A" operand is the weight
//     (lane = cout), "B" is the input (lane = pixel).  One accumulator register then holds 32 consecutive
//     pixels of one output channel across lanes 0..31 -> every store instruction writes two 128-B runs of
//     an NCHW plane, and the GLU pair (c, c + C/2) lives in the same lane/register of two accumulators.
//   * a workgroup = 4 waves = TH x 32 output pixels x (1..4) blocks of 32 output channels; each wave owns
//     R rows and ALL the channel blocks, so a weight fragment is reused R times and an input fragment NCB times
//     from registers, and both come from LDS as conflict-free ds_read_b32 (32 consecutive dwords per half-wave).
//   * K loop = input-channel stages of 4, double buffered in LDS: [4][TR][PITCH] halo tile + [9 taps][4][NCB*32]
//     weights (pre-packed on the device).  Stages are filled by LDS-DMA (global_load_lds, 16 B/lane for weights,
//     4 B/lane for the misaligned halo rows, out-of-image lanes read a zero word) issued one stage ahead, so the
//     copy of stage c+1 runs under the 9.2k MFMA cycles of stage c: one barrier per stage, no staging VGPRs.
//   * <= 48 KB LDS and ~128 accumulator VGPRs per workgroup -> 2-3 workgroups per CU.
#include "tgsr_common.h"

namespace tgsr {

struct ConvArgs {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;
  const float* wpack;
  int Cout;
  const float* scale;
  const float* shift;
  const float* res;
  int64_t rbs;
  float* out;
  int64_t obs;
  int Ho, Wo, tiles_x, tiles_y, nchunks;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// Source of every out-of-image / past-Cin element of a halo tile: the LDS-DMA has no predicate, so lanes that
// would read outside the tensor read this zero instead.
__device__ __attribute__((aligned(16))) float g_conv_zero[4] = {0.f, 0.f, 0.f, 0.f};

#ifdef TGSR_CONV_STAMPS
// Diagnostic build only (tools/conv_stamps.py): per-workgroup cycle stamps, never compiled into the shipped library.
__device__ unsigned long long g_stamps[8 * 8192];
#define TGSR_STAMP(k)                                                                          \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 8192) {                                               \
      g_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();                           \
      if ((k) == 0) g_stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();           \
      if ((k) == 3) g_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();           \
    }                                                                                          \
  } while (0)
#else
#define TGSR_STAMP(k)
#endif

template <int NOB, bool GLU, bool UP, int R, int WV>
struct ConvCfg {
  static constexpr int NCB = NOB * (GLU ? 2 : 1);  // accumulator blocks of 32 couts per row
  static constexpr int NCOL = NCB * 32;            // weight columns held in LDS
  static constexpr int TH = WV * R;                // output rows per workgroup (WV waves, R rows each)
  static constexpr int TR = UP ? TH / 2 + 2 : TH + 2;  // staged input rows (with halo)
  static constexpr int TC = UP ? 18 : 34;              // staged input cols (with halo)
  static constexpr int PITCH = TC;
  static constexpr int PLANE = TR * PITCH;
  static constexpr int IN_ELEMS = kConvCK * PLANE;
  static constexpr int IN_UNITS = (IN_ELEMS + 63) / 64;        // 64-dword LDS-DMA pieces
  static constexpr int W_ELEMS = 9 * kConvCK * NCOL;
  static constexpr int W_UNITS = (W_ELEMS + 255) / 256;        // 64 x 16-byte LDS-DMA pieces
  static constexpr int W_PAD = W_UNITS * 256;
  static constexpr int BUF = W_PAD + IN_UNITS * 64;            // floats per stage buffer
  static constexpr int SMEM = 2 * BUF + 2 * NCOL;              // two stages + the epilogue's scale/shift columns
};

// One stage (= kConvCK input channels): weights [9][CK][NCOL] then the input halo tile [CK][TR][PITCH], both
// written by LDS-DMA (global_load_lds): no VGPR round trip, the copy of stage c+1 flies under the MFMAs of stage c.
// The per-lane source offsets do not depend on the stage, so they are computed once (StagePlan) and a DMA piece
// costs ~8 instructions; the pieces are issued one per k-step INSIDE the MFMA stream, where VALU issue is free.
template <int NOB, bool GLU, bool UP, int R, int WV>
struct StagePlan {
  using C = ConvCfg<NOB, GLU, UP, R, WV>;
  static constexpr int WK = (C::W_UNITS + WV - 1) / WV;   // weight pieces per wave per stage (64 lanes x 16 B each)
  static constexpr int IK = (C::IN_UNITS + WV - 1) / WV;  // input pieces per wave per stage (64 lanes x 4 B each)
  int woff[WK];   // float offset inside the packed-weight stage block, -1 = padding lane
  int ioff[IK];   // (channel-in-stage << 28) | (gy * W + gx), -1 = outside the image / padding lane

  __device__ __forceinline__ void init(const ConvArgs& a, int wave, int lane, int grp, int sy0, int sx0) {
    constexpr int NCOL = C::NCOL, TR = C::TR, TC = C::TC;
#pragma unroll
    for (int k = 0; k < WK; ++k) {
      const int q = (wave + WV * k) * 64 + lane;  // float4 index inside the stage's weight block
      const int row = q / (NCOL / 4);
      const int c4 = q - row * (NCOL / 4);
      const int seg = c4 >> 3, f4 = c4 & 7;
      int col;
      if (GLU)
        col = (seg < NOB ? (grp * NOB + seg) * 32 : (a.Cout >> 1) + (grp * NOB + seg - NOB) * 32);
      else
        col = (grp * NOB + seg) * 32;
      woff[k] = (row < 9 * kConvCK) ? row * a.Cout + col + f4 * 4 : -1;
    }
#pragma unroll
    for (int k = 0; k < IK; ++k) {
      const int idx = (wave + WV * k) * 64 + lane;
      const int c = idx / (TR * TC);
      const int rem = idx - c * (TR * TC);
      const int r = rem / TC, cc = rem - r * TC;
      const int gy = sy0 + r, gx = sx0 + cc;
      const bool ok = c < kConvCK && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      ioff[k] = ok ? ((c << 28) | (gy * a.W + gx)) : -1;
    }
  }

  // piece p of stage `ch` -> buf;  p < WK: weights, else input.  p is a compile-time constant at every call.
  __device__ __forceinline__ void issue(int p, const ConvArgs& a, float* buf, int ch, int wave, const float* xb,
                                        uint32_t HW) const {
    if (p < WK) {
      const int u = wave + WV * p;
      if (WV * p + WV - 1 < C::W_UNITS || u < C::W_UNITS) {
        const float* wsrc = a.wpack + (int64_t)ch * 9 * kConvCK * a.Cout;
        const float* g = woff[p] >= 0 ? wsrc + woff[p] : g_conv_zero;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(buf + u * 256), 16, 0, 0);
      }
    } else {
      const int k = p - WK;
      const int u = wave + WV * k;
      if (WV * k + WV - 1 < C::IN_UNITS || u < C::IN_UNITS) {
        const int v = ioff[k];
        const int c = ch * kConvCK + (v >> 28);
        const bool ok = v >= 0 && c < a.Cin;
        const float* g = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(v & 0x0fffffff) : g_conv_zero;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(buf + C::W_PAD + u * 64), 4, 0, 0);
      }
    }
  }
};

template <int NOB, bool GLU, bool UP, int R, int WV>
__global__ __launch_bounds__(64 * WV, 2) void conv3x3_mfma_kernel(ConvArgs a) {
  using C = ConvCfg<NOB, GLU, UP, R, WV>;
  using Plan = StagePlan<NOB, GLU, UP, R, WV>;
  constexpr int NCB = C::NCB, NCOL = C::NCOL, TH = C::TH, PITCH = C::PITCH, PLANE = C::PLANE;
  constexpr int NS = 9 * (kConvCK / 2);  // k-steps (tap, channel pair) per stage
  constexpr int NP = Plan::WK + Plan::IK;  // DMA pieces per wave per stage
  static_assert(NP <= NS, "one DMA piece per k-step must cover a stage");
  __shared__ __attribute__((aligned(16))) float smem[C::SMEM];  // ONE array: two stage buffers + affine columns

  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int grp = blockIdx.y;
  const int ty0 = ty * TH, tx0 = tx * 32;
  const int sy0 = (UP ? (ty0 >> 1) : ty0) - 1, sx0 = (UP ? (tx0 >> 1) : tx0) - 1;
  const float* xb = a.x + (int64_t)b * a.xbs;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;

  TGSR_STAMP(0);
  Plan plan;
  plan.init(a, wave, lane, grp, sy0, sx0);
#pragma unroll
  for (int p = 0; p < NP; ++p) plan.issue(p, a, smem, 0, wave, xb, HW);

  f32x16 acc[NCB][R];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.f;

  // lane-dependent LDS offsets (floats) of the two MFMA operands inside a stage buffer
  const int a_off = h * NCOL + l31;                      // weights: [tap][ci][NCOL]
  const int wrow = wave * R;
  int b_off[3][R];                                       // input: [ci][row][col], per (ky, r); kx adds b_col
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int r = 0; r < R; ++r)
      b_off[ky][r] = C::W_PAD + h * PLANE + (UP ? (((wrow + r + ky - 1) >> 1) + 1) : (wrow + r + ky)) * PITCH;
  int b_col[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) b_col[kx] = UP ? (((l31 + kx - 1) >> 1) + 1) : (l31 + kx);

  // per-column affine (BN eval) of this workgroup's channels -> LDS, so the epilogue issues no global loads
  float* aff_s = smem + 2 * C::BUF;
  for (int cidx = tid; cidx < NCOL; cidx += 64 * WV) {
    const int seg = cidx >> 5, i = cidx & 31;
    int col;
    if (GLU)
      col = (seg < NOB ? (grp * NOB + seg) * 32 : (a.Cout >> 1) + (grp * NOB + seg - NOB) * 32) + i;
    else
      col = (grp * NOB + seg) * 32 + i;
    aff_s[cidx] = a.scale ? a.scale[col] : 1.f;
    aff_s[NCOL + cidx] = a.scale ? a.shift[col] : 0.f;
  }
  __syncthreads();  // drains the DMA (vmcnt(0)) and publishes the stage to every wave
  TGSR_STAMP(1);

  for (int ch = 0; ch < a.nchunks; ++ch) {
    const float* cur = smem + (ch & 1) * C::BUF;
    float* nxt = smem + ((ch + 1) & 1) * C::BUF;
    const bool more = ch + 1 < a.nchunks;

    // ---- NS k-steps, fragments fetched one step ahead of the MFMAs that use them
    float av[NCB], bv[R], an[NCB], bn[R];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) av[cb] = cur[a_off + cb * 32];
#pragma unroll
    for (int r = 0; r < R; ++r) bv[r] = cur[b_off[0][r] + b_col[0]];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      // MFMAs of step s in two halves with the ds_reads of step s+1 pinned between them: the reads are in flight
      // for half a step (>= 256 cycles) before the next step's first MFMA waits on them.  (Left to itself hipcc
      // sinks every read to just before its use and exposes the LDS latency 2x per step.)
      constexpr int NM = NCB * R, HALF = (NM + 1) / 2;
#pragma unroll
      for (int m = 0; m < HALF; ++m)
        acc[m / R][m % R] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m / R], bv[m % R], acc[m / R][m % R], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s < NP && more) plan.issue(s, a, nxt, ch + 1, wave, xb, HW);   // one DMA piece of the next stage
      if (s + 1 < NS) {
        const int s1 = s + 1;
        const int tap = s1 / (kConvCK / 2), kk = s1 % (kConvCK / 2);
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) an[cb] = cur[a_off + (tap * kConvCK + 2 * kk) * NCOL + cb * 32];
#pragma unroll
        for (int r = 0; r < R; ++r) bn[r] = cur[b_off[ky][r] + 2 * kk * PLANE + b_col[kx]];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = HALF; m < NM; ++m)
        acc[m / R][m % R] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m / R], bv[m % R], acc[m / R][m % R], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < NS) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) av[cb] = an[cb];
#pragma unroll
        for (int r = 0; r < R; ++r) bv[r] = bn[r];
      }
    }
    __syncthreads();  // next stage landed (vmcnt(0)) and this one is free to be overwritten
  }

  TGSR_STAMP(2);
  // ---- epilogue: affine (BN eval), GLU, residual; each register = 32 consecutive pixels of one channel.
  // __restrict__ locals: without them every store is followed by s_waitcnt vmcnt(0) (possible alias with the
  // next load) and the 64-128 outputs per lane serialise on full memory round trips.
  const int x = tx0 + l31;
  const int64_t HWo = (int64_t)a.Ho * a.Wo;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  const float* __restrict__ rb = a.res ? a.res + (int64_t)b * a.rbs : nullptr;
  const bool xok = x < a.Wo;
#pragma unroll
  for (int j = 0; j < NOB; ++j) {
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 4) {
      float resv[4][R];
      if (!GLU) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int cv = (grp * NOB + j) * 32 + acc_row(i0 + ii, h);
            const int y = ty0 + wrow + r;
            resv[ii][r] = (rb && xok && y < a.Ho) ? rb[(int64_t)cv * HWo + (int64_t)y * a.Wo + x] : 0.f;
          }
      }
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = i0 + ii;
        const int lc = j * 32 + acc_row(i, h);   // column inside this workgroup's NCOL
        const int cv = (grp * NOB + j) * 32 + acc_row(i, h);
        const float sv = aff_s[lc], tv = aff_s[NCOL + lc];
        float sg = 1.f, tg = 0.f;
        if (GLU) {
          sg = aff_s[lc + NOB * 32];
          tg = aff_s[NCOL + lc + NOB * 32];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int y = ty0 + wrow + r;
          float v = acc[j][r][i] * sv + tv;
          if (GLU) {
            const float g = acc[NOB + j][r][i] * sg + tg;
            v = v * (1.f / (1.f + __expf(-g)));
          } else {
            v += resv[ii][r];
          }
          if (y < a.Ho && xok) ob[(int64_t)cv * HWo + (int64_t)y * a.Wo + x] = v;
        }
      }
    }
  }
  TGSR_STAMP(3);
}

template <int NOB, bool GLU, bool UP, int R, int WV>
static int launch_conv(const ConvArgs& a, int groups, hipStream_t s) {
  using C = ConvCfg<NOB, GLU, UP, R, WV>;
  ConvArgs k = a;
  k.tiles_x = (a.Wo + 31) / 32;
  k.tiles_y = (a.Ho + C::TH - 1) / C::TH;
  dim3 grid((unsigned)(a.B * k.tiles_x * k.tiles_y), (unsigned)groups);
  hipLaunchKernelGGL((conv3x3_mfma_kernel<NOB, GLU, UP, R, WV>), grid, dim3(64 * WV), 0, s, k);
  return note_launch(hipGetLastError(), "conv3x3_mfma_kernel");
}

// rows per workgroup = WV * R: (4,4) 16 rows, (4,2) 8, (4,1) 4
template <int NOB, bool GLU, bool UP>
static int dispatch_r(const ConvArgs& a, int groups, int rows, hipStream_t s) {
  constexpr int NCB = NOB * (GLU ? 2 : 1);
  if constexpr (NCB <= 2) {
    if (rows == 16) return launch_conv<NOB, GLU, UP, 4, 4>(a, groups, s);
  }
  if (rows >= 8) return launch_conv<NOB, GLU, UP, 2, 4>(a, groups, s);
  return launch_conv<NOB, GLU, UP, 1, 4>(a, groups, s);
}

template <int NOB, bool GLU>
static int dispatch_up_r(const ConvArgs& a, int groups, bool up, int rows, hipStream_t s) {
  return up ? dispatch_r<NOB, GLU, true>(a, groups, rows, s) : dispatch_r<NOB, GLU, false>(a, groups, rows, s);
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* wpack,
                                int Cout, const float* scale, const float* shift, const float* residual,
                                int64_t res_bstride, float* out, int64_t out_bstride, int epilogue, int upsample,
                                void* stream) {
  if (!x || !wpack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  const bool glu = epilogue == TGSR_EPI_AFFINE_GLU;
  if (!glu && epilogue != TGSR_EPI_AFFINE) return TGSR_EINVAL;
  if (glu && residual) return TGSR_EINVAL;
  if (Cout % (glu ? 64 : 32) != 0) return TGSR_EUNSUPPORTED;
  if ((int64_t)H * W >= (1 << 28) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W;
  a.wpack = wpack; a.Cout = Cout; a.scale = scale; a.shift = shift;
  a.res = residual; a.rbs = res_bstride; a.out = out; a.obs = out_bstride;
  a.Ho = upsample ? 2 * H : H; a.Wo = upsample ? 2 * W : W;
  a.nchunks = (Cin + kConvCK - 1) / kConvCK;
  a.tiles_x = a.tiles_y = 0;
  // Tile choice.  Candidates from most operand reuse (2 output blocks per workgroup, 8 accumulators per wave) to
  // least (1 block, 4 rows); take the first that gives >= 2 workgroups per CU, else the first
  // with >= 1 per CU, else the smallest tile: at B = 16 an idle CU costs more than the lost reuse.  (2-wave
  // workgroups of 2 rows were measured: same MFMA chain per wave, no gain - the WV parameter stays for a K-split.)
  const int unit = glu ? 64 : 32;          // couts consumed per output block
  const int nob_max = (Cout % (2 * unit) == 0) ? 2 : 1;
  struct Cand { int nob, rows; };
  Cand cands[8];
  int nc = 0;
  for (int nb = nob_max; nb >= 1; --nb) {
    const int ncb = nb * (glu ? 2 : 1);
    if (ncb <= 2) cands[nc++] = {nb, 16};
    cands[nc++] = {nb, 8};
    cands[nc++] = {nb, 4};
  }
  auto ntiles = [&](const Cand& c) {
    return (int64_t)B * ((a.Ho + c.rows - 1) / c.rows) * ((a.Wo + 31) / 32) * (Cout / (unit * c.nob));
  };
  int pick = nc - 1;
  for (int pass = 0; pass < 2 && pick == nc - 1; ++pass)
    for (int i = 0; i < nc; ++i)
      if (ntiles(cands[i]) >= (pass == 0 ? 512 : 256)) { pick = i; break; }
  const int nob = cands[pick].nob, rows = cands[pick].rows;
  const int groups = Cout / (unit * nob);
  hipStream_t s = as_stream(stream);
  if (glu) return nob == 2 ? dispatch_up_r<2, true>(a, groups, upsample != 0, rows, s)
                           : dispatch_up_r<1, true>(a, groups, upsample != 0, rows, s);
  return nob == 2 ? dispatch_up_r<2, false>(a, groups, upsample != 0, rows, s)
                  : dispatch_up_r<1, false>(a, groups, upsample != 0, rows, s);
}

#ifdef TGSR_CONV_STAMPS
extern "C" int tgsr_debug_read_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(tgsr::g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -3;
}
#endif
