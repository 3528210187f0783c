// KxK convolution to 3 output channels (the image heads) for gfx950.
//
// Replaces GET_IMAGE_G_noAct.img (conv3x3 ngf->3, util.py:913-915) and NetG_highweight.conv_output
// (conv5x5 ngf->3 + Tanh, model.py:224) fused with `one * . + a * SRb` (model.py:280/288/297).
// Cout = 3 does not fill an MFMA tile (a 32-row tile would waste 90 % of it), and at 256x256 the op reads
// 8.4 MB and writes 0.8 MB per image against 0.1-0.3 GFLOP: it is a streaming kernel, so it runs on the VALU:
//   * workgroup = TH x 64 output pixels (TH = 16, or 8 / 4 on small images so that every CU gets work; there the
//     input channels are also split over 2 / 4 thread groups whose partial sums meet in LDS),
//     thread = 4 consecutive pixels x 3 channels (12 accumulators);
//   * the input is staged in LDS 2 channels at a time with its halo, double buffered, by LDS-DMA
//     (global_load_lds, 16 B per lane: the tile starts 4 columns left of the output tile so every piece is an
//     aligned float4; out-of-image pieces read a zero block) - the copy of stage c+1 runs under the FMAs of stage c;
//   * a thread reads its 12 input floats of a row as three aligned ds_read_b128 and reuses them for all K taps x
//     4 pixels x 3 channels; weights are wave-uniform -> scalar loads feeding the SGPR operand of v_fma.
// Widths that are not a multiple of 4 (never on the SR path) take the 4-byte DMA form of the same kernel.
#include "tgsr_common.h"

namespace tgsr {

typedef __attribute__((address_space(3))) void* lds_ptr3_t;
typedef const __attribute__((address_space(1))) void* glb_ptr1_t;

__device__ __attribute__((aligned(16))) float g_to3_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct To3Args {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;
  const float* w;  // [3][Cin][K][K]
  const float* addend;
  float alpha;
  float* out;
  int tiles_x, tiles_y;
};

__device__ __forceinline__ float fast_tanh(float v) {
  // tanh(v) = 1 - 2 / (exp(2v) + 1); absolute error ~1e-7 (what parity needs); saturates cleanly at +-1
  const float e = __expf(2.f * v);
  return 1.f - 2.f / (e + 1.f);
}

// TH output rows x 64 columns per workgroup; KS thread groups of 16 * TH threads each take 1/KS of the input channels
// (small images have too few tiles to fill the chip: the channel split gives every CU several waves and a quarter of
// the barrier-separated stages per wave) and are summed through LDS before the epilogue.
template <int K, int ACT, bool VEC4, int TH, int KS>
__global__ __launch_bounds__(16 * TH * KS) void conv_to3_kernel(To3Args a) {
  constexpr int P = K / 2, CK = 2, TW = 64, NG = 16 * TH, NW = NG / 64;
  constexpr int TR = TH + K - 1;
  constexpr int PITCH = 72;  // LDS column j = input column x0 - 4 + j; 72 = 64 + 4 left + 4 right
  constexpr int STAGE = CK * TR * PITCH;               // floats per stage (multiple of 4)
  constexpr int PIECE = VEC4 ? 256 : 64;               // floats per wave DMA instruction
  constexpr int UNITS = (STAGE + PIECE - 1) / PIECE;
  constexpr int BUF = UNITS * PIECE;
  constexpr int UK = (UNITS + NW - 1) / NW;            // pieces per wave per stage
  static_assert(NG % 64 == 0, "a channel group is a whole number of waves");
  static_assert((KS - 1) * 12 * NG <= KS * 2 * BUF, "reduction buffer fits the stage buffers");
  __shared__ __attribute__((aligned(16))) float smem_all[KS * 2 * BUF];

  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / NG);
  const int tid = threadIdx.x - grp * NG, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* smem = smem_all + grp * 2 * BUF;
  // TH x 16 threads, 4 pixels wide each.  A wave covers 4 rows x 16 column quads; WHICH lane takes which (row, quad)
  // follows the lane groups a ds_read_b128 is served in - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
  // (MI355X_MICROARCH.md, LDS): each group reads ONE row, i.e. 64 consecutive words = 64 distinct banks at any row
  // alignment.  With lane = 16 row + quad, a group straddled two rows whose pitch (72 words) is not a multiple of the 64
  // banks: SQ_LDS_BANK_CONFLICT was 72 % of the LDS cycles of this kernel (profiles/r02i_fp32_pmc.csv).
  const int lq = (lane >> 2) & 7;
  const int txi = (lq >> 1) * 4 + (lane & 3), tyi = 4 * wave + 2 * (lane >> 5) + ((0x96 >> lq) & 1);
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;
  const float* xb = a.x + (int64_t)b * a.xbs;

  // per-lane source offsets of this wave's DMA pieces: (channel-in-stage << 28) | (gy * W + gx), -1 = zero fill
  int off[UK];
#pragma unroll
  for (int k = 0; k < UK; ++k) {
    const int e = ((wave + NW * k) * 64 + lane) * (VEC4 ? 4 : 1);   // first float of this lane's piece
    const int c = e / (TR * PITCH);
    const int rem = e - c * (TR * PITCH);
    const int r = rem / PITCH, j = rem - r * PITCH;
    const int gy = y0 - P + r, gx = x0 - 4 + j;
    const bool ok = c < CK && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;  // VEC4: W % 4 == 0
    off[k] = ok ? ((c << 28) | (gy * a.W + gx)) : -1;
  }
  auto issue = [&](int k, float* buf, int c0) {
    const int u = wave + NW * k;
    if (u < UNITS) {
      const int v = off[k];
      const int c = c0 + (v >> 28);
      const bool ok = v >= 0 && c < a.Cin;
      const float* g = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(v & 0x0fffffff) : g_to3_zero;
      if (VEC4)
        __builtin_amdgcn_global_load_lds((glb_ptr1_t)g, (lds_ptr3_t)(buf + u * 256), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((glb_ptr1_t)g, (lds_ptr3_t)(buf + u * 64), 4, 0, 0);
    }
  };

  float acc[3][4];
#pragma unroll
  for (int co = 0; co < 3; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;

  // this group's stages: [s0, s1) of the ceil(Cin / CK) channel stages; every group runs the same number of trips
  const int nst = (a.Cin + CK - 1) / CK, per = (nst + KS - 1) / KS;
  const int s0 = grp * per, s1 = s0 + per < nst ? s0 + per : nst;
  if (s0 < s1) {
#pragma unroll
    for (int k = 0; k < UK; ++k) issue(k, smem, s0 * CK);
  }
  __syncthreads();

  for (int i = 0; i < per; ++i) {
    const int st = s0 + i;
    const float* cur = smem + (i & 1) * BUF;
    if (st + 1 < s1) {
#pragma unroll
      for (int k = 0; k < UK; ++k) issue(k, smem + ((i + 1) & 1) * BUF, (st + 1) * CK);
    }
    const int c0 = st * CK;
    if (st < s1) {   // group-uniform
#pragma unroll
      for (int c = 0; c < CK; ++c) {
        if (c0 + c < a.Cin) {   // uniform
          const float* wc = a.w + (int64_t)(c0 + c) * K * K;  // + co * Cin*K*K
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            const float* row = cur + (c * TR + tyi + ky) * PITCH + 4 * txi;
            float4 v0 = *reinterpret_cast<const float4*>(row);
            float4 v1 = *reinterpret_cast<const float4*>(row + 4);
            float4 v2 = *reinterpret_cast<const float4*>(row + 8);
            // Keep the three reads whole ds_read_b128s: of v0 / v2 only one (K = 3) or two (K = 5) components are used, and
            // hipcc then narrows them to ds_read_b32 / _b64 at a 4-word lane stride - 8 distinct banks for 32 lanes, a 4-way
            // conflict on every one of them (what SQ_LDS_BANK_CONFLICT = 72 % of this kernel's LDS cycles really was); a
            // whole b128 of a row is conflict free (lane mapping above) and costs 4 LDS cycles instead of 8.
            asm volatile("" : "+v"(v0.x), "+v"(v0.y), "+v"(v0.z), "+v"(v0.w));
            asm volatile("" : "+v"(v2.x), "+v"(v2.y), "+v"(v2.z), "+v"(v2.w));
            const float in[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
            // pixel p (column x0 + 4*txi + p) tap kx reads LDS column 4*txi + 4 + p + kx - P
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
#pragma unroll
              for (int co = 0; co < 3; ++co) {
                const float wv = wc[(int64_t)co * a.Cin * K * K + ky * K + kx];
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(wv, in[4 + p + kx - P], acc[co][p]);
              }
            }
          }
        }
      }
    }
    __syncthreads();  // next stage landed (vmcnt(0)); this one may be overwritten
  }

  if (KS > 1) {       // sum the channel groups (fixed order) into group 0
    float* red = smem_all;
    if (grp > 0) {
#pragma unroll
      for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int p = 0; p < 4; ++p) red[((grp - 1) * 12 + co * 4 + p) * NG + tid] = acc[co][p];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g)
#pragma unroll
      for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[co][p] += red[((g - 1) * 12 + co * 4 + p) * NG + tid];
  }

  const int y = y0 + tyi, xx = x0 + 4 * txi;
  float* __restrict__ outp = a.out;
  const float* __restrict__ addp = a.addend;
  if (y < a.H) {
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const int64_t o = ((int64_t)b * 3 + co) * HW + (int64_t)y * a.W + xx;
      float r[4];
      if (VEC4 && xx + 3 < a.W) {
        float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ACT == TGSR_ACT_TANH_AXPY && addp) ad = *reinterpret_cast<const float4*>(addp + o);
        const float adv[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
        for (int p = 0; p < 4; ++p)
          r[p] = ACT == TGSR_ACT_TANH_AXPY ? fast_tanh(acc[co][p]) + a.alpha * adv[p] : acc[co][p];
        *reinterpret_cast<float4*>(outp + o) = make_float4(r[0], r[1], r[2], r[3]);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (xx + p < a.W) {
            float v = acc[co][p];
            if (ACT == TGSR_ACT_TANH_AXPY) {
              v = fast_tanh(v);
              if (addp) v += a.alpha * addp[o + p];
            }
            outp[o + p] = v;
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The streaming kernel above, with its copies really in flight (round 6).  hipcc cannot tell which LDS bytes an LDS-DMA
// (`__builtin_amdgcn_global_load_lds`) writes, so it puts `s_waitcnt vmcnt(0)` in front of the first ds_read behind one: the "copy of
// stage c + 1 under the FMAs of stage c" of the kernel above never overlapped anything - every 2-channel stage paid a full HBM round
// trip (~1.5 us against ~0.3 us of FMAs; 2.7 TB/s on the 256^2 head, which sits ALONE at the end of the one-lane step) - and the
// filter taps, which it loads with vector loads right before their use (the kernel stores to global memory, so the loads are not
// provably invariant and do not become scalar loads), paid an L2 round trip per channel.  Here:
//   * the copies are issued in inline assembly (the compiler does not see them) into a ring of THREE stage buffers, two stages
//     ahead, with counted waits: every wave issues exactly UK pieces per stage (the buffer is padded to whole rounds of pieces;
//     the padding pieces read the zero block), so `vmcnt(UK)` in front of a stage means "this stage has landed, the next one may
//     still be in flight"; one barrier per stage, the copy of stage i + 2 is issued behind it (everybody has left stage i - 1);
//   * the whole filter sits in LDS ([Cin][3 K K, padded to 4]; dynamic shared memory), read as broadcasts.
// The FMA chains are those of the kernel above in the same order: the images are bit-identical.  Needs W % 4 == 0 (the 16-byte copy
// form) and a filter of <= 16 KB; everything else stays on the kernel above (TGSR_TO3_PIPE=0: everything).
// Measured (batch 16, the six stand-alone heads of an inference step, same box): 0.206 -> 0.189 ms per step, 1.87 -> 2.04 TB/s -
// 8 %, not the 2x the serialised copies suggested: four workgroups per CU already overlapped one another's round trips.
template <int K, int ACT, int TH, int KS>
__global__ __launch_bounds__(16 * TH * KS) void conv_to3_pipe_kernel(To3Args a) {
  constexpr int P = K / 2, CK = 2, TW = 64, NG = 16 * TH, NW = NG / 64, NT = 16 * TH * KS;
  constexpr int TR = TH + K - 1;
  constexpr int PITCH = 72;
  constexpr int STAGE = CK * TR * PITCH;
  constexpr int UK = ((STAGE + 255) / 256 + NW - 1) / NW;          // 1 KB pieces per wave and stage: the same for every wave
  constexpr int BUF = UK * NW * 256, D = 3;
  constexpr int WPC = (3 * K * K + 3) & ~3;                        // filter floats per input channel ([co][ky][kx], padded)
  static_assert(NG % 64 == 0, "a channel group is a whole number of waves");
  static_assert((KS - 1) * 12 * NG <= KS * D * BUF, "reduction buffer fits the stage buffers");
  __shared__ __attribute__((aligned(16))) float smem_all[KS * D * BUF];
  extern __shared__ __attribute__((aligned(16))) float w_s[];      // [Cin][WPC]

  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / NG);
  const int tid = threadIdx.x - grp * NG, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* smem = smem_all + grp * D * BUF;
  const int lq = (lane >> 2) & 7;                                  // (the lane -> (row, quad) mapping of the kernel above)
  const int txi = (lq >> 1) * 4 + (lane & 3), tyi = 4 * wave + 2 * (lane >> 5) + ((0x96 >> lq) & 1);
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;
  const float* xb = a.x + (int64_t)b * a.xbs;

  for (int o = threadIdx.x; o < a.Cin * 3 * K * K; o += NT) {      // the filter -> LDS, [c][co][tap]
    const int c = o / (3 * K * K), r = o - c * (3 * K * K);
    const int co = r / (K * K), tp = r - co * (K * K);
    w_s[c * WPC + r] = a.w[((int64_t)co * a.Cin + c) * (K * K) + tp];
  }

  int off[UK];                                                     // (channel-in-stage << 28) | (gy * W + gx), -1 = zero fill
#pragma unroll
  for (int k = 0; k < UK; ++k) {
    const int e = ((wave + NW * k) * 64 + lane) * 4;
    const int c = e / (TR * PITCH);
    const int rem = e - c * (TR * PITCH);
    const int r = rem / PITCH, j = rem - r * PITCH;
    const int gy = y0 - P + r, gx = x0 - 4 + j;
    const bool ok = c < CK && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    off[k] = ok ? ((c << 28) | (gy * a.W + gx)) : -1;
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr3_t)smem) + (unsigned)wave * 1024u;
  auto issue = [&](int bi, int c0) {                               // stage of channels c0, c0 + 1 -> ring buffer bi
#pragma unroll
    for (int k = 0; k < UK; ++k) {
      const int v = off[k];
      const int c = c0 + (v >> 28);
      const bool ok = v >= 0 && c < a.Cin;
      const float* g = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(v & 0x0fffffff) : g_to3_zero;
      const unsigned dst = lds0 + (unsigned)(bi * BUF + NW * k * 256) * 4u;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory");
    }
  };

  float acc[3][4];
#pragma unroll
  for (int co = 0; co < 3; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;

  const int nst = (a.Cin + CK - 1) / CK, per = (nst + KS - 1) / KS;
  const int s0 = grp * per, s1 = s0 + per < nst ? s0 + per : nst;
  if (s0 < s1) issue(0, s0 * CK);
  if (s0 + 1 < s1) issue(1, (s0 + 1) * CK);
  __syncthreads();                                                 // the filter is in LDS (the copies are not waited for here)

  int bi = 0;                                                      // ring buffer of stage i
  for (int i = 0; i < per; ++i) {
    const int st = s0 + i;
    // stage i has landed (the pieces of stage i + 1, issued behind it, may still be in flight); everybody has left stage i - 1
    if (st + 1 < s1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(UK) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + 2 < s1) issue(bi >= 1 ? bi - 1 : 2, (st + 2) * CK);   // (bi + 2) % 3: the buffer stage i - 1 was read from
    const float* cur = smem + bi * BUF;
    const int c0 = st * CK;
    if (st < s1) {   // group-uniform
#pragma unroll
      for (int c = 0; c < CK; ++c) {
        if (c0 + c < a.Cin) {   // uniform
          const float* wc = w_s + (c0 + c) * WPC;
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            const float* row = cur + (c * TR + tyi + ky) * PITCH + 4 * txi;
            float4 v0 = *reinterpret_cast<const float4*>(row);
            float4 v1 = *reinterpret_cast<const float4*>(row + 4);
            float4 v2 = *reinterpret_cast<const float4*>(row + 8);
            asm volatile("" : "+v"(v0.x), "+v"(v0.y), "+v"(v0.z), "+v"(v0.w));       // (whole ds_read_b128s: see the kernel above)
            asm volatile("" : "+v"(v2.x), "+v"(v2.y), "+v"(v2.z), "+v"(v2.w));
            const float in[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
#pragma unroll
              for (int co = 0; co < 3; ++co) {
                const float wv = wc[co * K * K + ky * K + kx];
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(wv, in[4 + p + kx - P], acc[co][p]);
              }
            }
          }
        }
      }
    }
    bi = bi == 2 ? 0 : bi + 1;
  }

  if (KS > 1) {       // sum the channel groups (fixed order) into group 0
    __syncthreads();  // every group has finished reading its stage buffers: the reduction image goes on top of them
    float* red = smem_all;
    if (grp > 0) {
#pragma unroll
      for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int p = 0; p < 4; ++p) red[((grp - 1) * 12 + co * 4 + p) * NG + tid] = acc[co][p];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g)
#pragma unroll
      for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[co][p] += red[((g - 1) * 12 + co * 4 + p) * NG + tid];
  }

  const int y = y0 + tyi, xx = x0 + 4 * txi;
  float* __restrict__ outp = a.out;
  const float* __restrict__ addp = a.addend;
  if (y < a.H && xx < a.W) {                                       // (W % 4 == 0: a thread's four pixels are inside or outside together)
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const int64_t o = ((int64_t)b * 3 + co) * HW + (int64_t)y * a.W + xx;
      float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ACT == TGSR_ACT_TANH_AXPY && addp) ad = *reinterpret_cast<const float4*>(addp + o);
      const float adv[4] = {ad.x, ad.y, ad.z, ad.w};
      float r[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) r[p] = ACT == TGSR_ACT_TANH_AXPY ? fast_tanh(acc[co][p]) + a.alpha * adv[p] : acc[co][p];
      *reinterpret_cast<float4*>(outp + o) = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same heads on the matrix cores (fp32 MFMA 16x16x4, exact fmaf chains) for the large images.  Three output channels
// cannot fill an MFMA tile, so the kernel column kx moves into the M dimension: rows r = co * K + kx (15 of 16 for the
// 5x5 heads) and
//     V[(co, kx)][u] = sum_{ky, ci} w[co][ci][ky][kx] * x[ci][y + ky - P][u],      out[co][y][x] = sum_kx V[(co, kx)][x + kx - P]
// i.e. K * Cin / 4 MFMAs per 16 input columns instead of K * K * 3 * Cin VALU FMAs per pixel; the K-term shift-sum over
// kx goes through a small per-wave LDS image of V.  Workgroup = 8 rows x 64 columns, wave = 2 rows x 5 column tiles of 16
// (80 columns from x0 - 4: the halo and float4 alignment); the input is staged 16 channels at a time [ch][row][80] with a
// channel stride = 16 mod 64 words, so the B-fragment read (4 channels x 16 columns per lane group) touches 64 banks.
template <int K, int ACT, int CH, int TRO>
__global__ __launch_bounds__(256) void conv_to3_mfma_kernel(To3Args a) {
  constexpr int RPW = TRO / 4;                                      // output rows per wave
  constexpr int NJ = CH / 4;                                       // MFMA k-steps (4 channels) per staged chunk
  constexpr int P = K / 2, ROWS = TRO + K - 1, PITCH = 80;
  constexpr int CS = (ROWS * PITCH) % 64 == 0 ? ROWS * PITCH + 16 : (ROWS * PITCH + 63) / 64 * 64 + 16 - ((ROWS * PITCH) % 64 <= 16 ? 64 : 0);
  static_assert(CS % 64 == 16 && CS >= ROWS * PITCH && CS % 4 == 0, "channel stride");
  constexpr int VP = 84;                                           // V image pitch
  constexpr int SM = CH * CS > 4 * 16 * VP ? CH * CS : 4 * 16 * VP;   // the V images go on top of the dead input tile
  __shared__ __attribute__((aligned(16))) float x_s[SM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TRO, x0 = tx * 64;
  const int64_t HW = (int64_t)a.H * a.W;
  const float* xb = a.x + (int64_t)b * a.xbs;
  const int co = col / K, kx = col - co * K;                      // this lane's A row (co, kx); rows >= 3K are zero
  const bool arow = col < 3 * K;
  f32x4 acc[RPW][5];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int ut = 0; ut < 5; ++ut) acc[r][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging: thread = one float4 column group (row, q) of the tile, looping over the chunk's 16 channels (consecutive
  // threads = consecutive float4s of a row: coalesced); the NEXT chunk's loads are issued before this chunk's MFMAs
  constexpr int NU = ROWS * (PITCH / 4), NUPT = (NU + 255) / 256;      // staging units (row, float4 column) and units per thread
  int soff[NUPT];
  bool sin[NUPT], suse[NUPT];
  const float* src[NUPT];
#pragma unroll
  for (int u = 0; u < NUPT; ++u) {
    const int id = tid + 256 * u;
    const int srow = id / (PITCH / 4), sq = id - srow * (PITCH / 4);
    const int gy = y0 - P + srow, gx = x0 - 4 + 4 * sq;
    suse[u] = id < NU;
    sin[u] = suse[u] && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    src[u] = xb + (sin[u] ? (int64_t)gy * a.W + gx : 0);
    soff[u] = srow * PITCH + 4 * sq;
  }
  float4 pre[NUPT][CH];
  float af[K][NJ], afn[K][NJ];
  auto wload = [&](int c0, float (&dst)[K][NJ]) {
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        dst[ky][j] = arow ? a.w[(((int64_t)co * a.Cin + c0 + 4 * j + kq) * K + ky) * K + kx] : 0.f;
  };
  auto gload = [&](int c0) {
#pragma unroll
    for (int u = 0; u < NUPT; ++u)
#pragma unroll
      for (int ch = 0; ch < CH; ++ch)
        pre[u][ch] = sin[u] ? *reinterpret_cast<const float4*>(src[u] + (int64_t)(c0 + ch) * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  gload(0);
  wload(0, afn);
  for (int c0 = 0; c0 < a.Cin; c0 += CH) {
    __syncthreads();                                               // everybody is done reading the previous chunk
#pragma unroll
    for (int u = 0; u < NUPT; ++u)
      if (suse[u]) {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) *reinterpret_cast<float4*>(x_s + ch * CS + soff[u]) = pre[u][ch];
      }
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int j = 0; j < NJ; ++j) af[ky][j] = afn[ky][j];
    __syncthreads();
    if (c0 + CH < a.Cin) {
      gload(c0 + CH);
      wload(c0 + CH, afn);
    }
    const float* bs = x_s + kq * CS + (RPW * wave) * PITCH + col;
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int ut = 0; ut < 5; ++ut)   // innermost: consecutive MFMAs go to different accumulators
            acc[r][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky][j], bs[4 * j * CS + (r + ky) * PITCH + ut * 16], acc[r][ut], 0, 0, 0);
  }
  __syncthreads();                                                 // the input tile is dead: V images go on top of it
  float* vs = x_s + wave * (16 * VP);                              // one row's V image per wave, reused row after row
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
#pragma unroll
    for (int ut = 0; ut < 5; ++ut)
#pragma unroll
      for (int q = 0; q < 4; ++q) vs[(4 * kq + q) * VP + ut * 16 + col] = acc[r][ut][q];   // D: row 4 (l >> 4) + reg
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this wave's own V image is written
    __builtin_amdgcn_wave_barrier();
    const int y = y0 + RPW * wave + r;
#pragma unroll
    for (int oc = 0; oc < 3; ++oc) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) v += vs[(oc * K + k) * VP + lane + k - P + 4];   // column j = x - x0 + 4
      const int64_t o = ((int64_t)b * 3 + oc) * HW + (int64_t)y * a.W + x0 + lane;
      if (ACT == TGSR_ACT_TANH_AXPY) {
        v = fast_tanh(v);
        if (a.addend) v += a.alpha * a.addend[o];
      }
      a.out[o] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // read before the next row overwrites it
    __builtin_amdgcn_wave_barrier();
  }
}

static int g_to3_pipe = [] {
  const char* e = getenv("TGSR_TO3_PIPE");
  return (e && e[0] == '0') ? 0 : 1;
}();
static bool to3_pipe() { return g_to3_pipe != 0; }

template <int K, int ACT, int TH, int KS>
static int launch_to3_th(To3Args a, hipStream_t s) {
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + TH - 1) / TH;
  const dim3 grid((unsigned)(a.B * a.tiles_x * a.tiles_y));
  const bool vec4 = (a.W % 4 == 0) && (a.xbs % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0) &&
                    (!a.addend || (reinterpret_cast<uintptr_t>(a.addend) & 15) == 0);
  constexpr int WPC = (3 * K * K + 3) & ~3;
  const size_t wbytes = (size_t)a.Cin * WPC * sizeof(float);
  if (vec4 && wbytes <= 16 * 1024 && to3_pipe())
    hipLaunchKernelGGL((conv_to3_pipe_kernel<K, ACT, TH, KS>), grid, dim3(16 * TH * KS), wbytes, s, a);
  else if (vec4)
    hipLaunchKernelGGL((conv_to3_kernel<K, ACT, true, TH, KS>), grid, dim3(16 * TH * KS), 0, s, a);
  else
    hipLaunchKernelGGL((conv_to3_kernel<K, ACT, false, TH, KS>), grid, dim3(16 * TH * KS), 0, s, a);
  return note_launch(hipGetLastError(), "conv_to3_kernel");
}

template <int K, int ACT>
static int launch_to3(To3Args a, hipStream_t s) {
  // 16-row tiles when they still give >= 2 workgroups per CU; smaller images take 8- or 4-row tiles (more workgroups)
  // and split the input channels over 2 / 4 thread groups (more waves per workgroup, fewer stages per wave)
  auto tiles = [&](int th) { return (int64_t)a.B * ((a.W + 63) / 64) * ((a.H + th - 1) / th); };
  // large images of the 5x5 heads: the MFMA form (measured at B = 16: 256^2 116 -> 83 us, 128^2 34 -> 25 us); the 3x3
  // heads fill 9 of 16 MFMA rows and gain nothing over the streaming kernel
  if (K == 5 && a.Cin % 16 == 0 && a.W % 64 == 0 && a.H % 8 == 0 && tiles(8) >= 512 && a.xbs % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(a.x) & 15) == 0) {
    a.tiles_x = a.W / 64;
    // 8-channel chunks (31 KB of LDS, 4 workgroups per CU): 83 us on the 256^2 head against 90 us with 16-channel chunks
    // (16-row tiles - 1.25x halo rows instead of 1.5x - measured 85 us: the re-read is not what bounds it)
    a.tiles_y = a.H / 8;
    hipLaunchKernelGGL((conv_to3_mfma_kernel<K, ACT, 8, 8>), dim3((unsigned)(a.B * a.tiles_x * a.tiles_y)), dim3(256), 0, s, a);
    return note_launch(hipGetLastError(), "conv_to3_mfma_kernel");
  }
  if (tiles(16) >= 512) return launch_to3_th<K, ACT, 16, 1>(a, s);
  if (tiles(8) >= 512) return launch_to3_th<K, ACT, 8, 2>(a, s);
  return launch_to3_th<K, ACT, 4, 4>(a, s);
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_conv_to3_set_pipe(int on) {
  const int was = g_to3_pipe;
  g_to3_pipe = on ? 1 : 0;
  return was;
}

extern "C" int tgsr_conv_to3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* w,
                                 int K, int act, const float* addend, float alpha, float* out, void* stream) {
  if (!x || !w || !out || B < 1 || Cin < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (K != 3 && K != 5) return TGSR_EUNSUPPORTED;
  if (act != TGSR_ACT_NONE && act != TGSR_ACT_TANH_AXPY) return TGSR_EINVAL;
  if (act == TGSR_ACT_NONE && addend) return TGSR_EINVAL;
  if ((int64_t)H * W >= (1 << 28) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  To3Args a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.w = w;
  a.addend = addend; a.alpha = alpha; a.out = out; a.tiles_x = a.tiles_y = 0;
  hipStream_t s = as_stream(stream);
  if (K == 3) return act == TGSR_ACT_NONE ? launch_to3<3, TGSR_ACT_NONE>(a, s) : launch_to3<3, TGSR_ACT_TANH_AXPY>(a, s);
  return act == TGSR_ACT_NONE ? launch_to3<5, TGSR_ACT_NONE>(a, s) : launch_to3<5, TGSR_ACT_TANH_AXPY>(a, s);
}
