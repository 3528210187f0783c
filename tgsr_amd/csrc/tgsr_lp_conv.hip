// Reduced-precision fused 3x3 convolution for the inference path (BASELINE.json configs[4]: "MFMA bf16 ... fused conv"):
// conv3x3 (stride 1, zero pad 1, no bias) [+ nearest x2 in front] + BatchNorm(eval) affine + GLU | + residual, on
// v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation.  Replaces the same reference blocks as tgsr_conv3x3_fwd:
// ResBlock.block util.py:110-130, upBlock util.py:74-80, residual24/48 model.py:229-232.
//
// Layout: activations are zero-bordered channels-last images [B][H+2][W+2][cpitch] of 2-byte elements
// (tgsr_lp_common.h).  Implicit GEMM D[cout][pixel] = sum_k W[cout][k] X[k][pixel], k = (tap, cin):
//   MFMA A = weights: lane l holds cout l&31, the 8 input channels 8(l>>5).. of a 16-channel k-step of one tap;
//   MFMA B = input  : lane l holds pixel l&31 (32 consecutive columns of one row, shifted by the tap), same 8 channels;
//   D: lane = pixel, registers = 16 of the 32 couts (4 runs of 4 consecutive couts) -> 8-byte channels-last stores, and
//      the GLU pair (c, c + Cout/2) sits in the same lane/register of two accumulators.
// Workgroup = 4 waves = TR rows x 32 columns of outputs x ALL Cout; a wave owns TR/4 rows.
//   input tile  [(TR+2) x 34 pixels][CIN] in LDS, fetched ONCE by LDS-DMA with the 16-byte slots of a pixel XOR-swizzled
//               through the per-lane SOURCE address (the LDS side of a DMA is linear), so that the B-fragment
//               ds_read_b128 of 32 neighbouring pixels is bank-conflict free (checked by simulation: 4 LDS cycles);
//               with `UP` the source address is the low-resolution pixel (y >> 1, x >> 1): the up-sampled tensor of
//               upBlock never exists;
//   weights     streamed through LDS in chunks of (one kernel row = 3 taps) x 16 input channels x Cout, packed on the
//               host side in exactly fragment order (a chunk is a linear copy, an A fragment a linear 1-KiB read),
//               triple buffered: chunks c+1 and c+2 are in flight while chunk c feeds >= 24 MFMAs per wave (counted
//               s_waitcnt vmcnt); one barrier per chunk.
// LDS <= 79 KB (CIN 64, Cout 128, TR 8: 43 KB tile + 3 x 12 KB weight chunks) -> two workgroups per CU;
// accumulators <= 128 registers.
#include "tgsr_lp_common.h"

namespace tgsr {

struct LpConvArgs {
  const char* x;       // input, element [b][0][0][0] of the padded image
  int xcp;             // input channel pitch (elements)
  int B, H, W;         // OUTPUT spatial size
  int Hi, Wi;          // input spatial size (H, W or H/2, W/2)
  const char* wpack;
  const float* scale;
  const float* shift;
  const char* res;
  int rcp, rco;
  char* out;           // may be null for the head-fused upBlock (the feature image is then never written)
  int ocp, oco;
  int tiles_x, tiles_y;
  const char* hw;      // head fusion (lp_upconv_glu_kernel<.., HK>): image-head filter, lp_pack_to3 layout [HK][lane 64][8]
  float* hpart;        // per-tile partial sums of the head [B][tiles_y][tiles_x][3][8 + 2P][64 + 2P]
  LpAttFuse att;       // lp_upconv_glu_kernel<.., ATT = true>: the next stage's word attention on the tile just produced
};

constexpr int kEpiAffine = 0, kEpiGlu = 1, kEpiRes = 2;

template <int CIN>
__device__ __forceinline__ int lp_swz(int c) {
  return CIN == 64 ? (c >> 1) & 7 : (c >> 2) & 3;
}

template <int CIN, int COUT, int TR>
constexpr int lp_conv_occ() {   // waves per SIMD to allocate registers for: the CIN = 64 tiles are LDS-limited to 2 workgroups
  return (CIN == 64 || (COUT / 32) * (TR / 4) * 16 >= 64) ? 2 : 3;
}

// (the timing experiments of profiles/HISTORY.md 3.8 - parts of this kernel compiled out, in-kernel clock stamps - were made on a diagnostic
// COPY of this file, tools/diag/tgsr_lp_conv_dbg.hip; a second copy of a kernel drifts, so it was removed in round 5 - it is in the
// history at 67ad6ee with its drivers tools/lp_conv_experiments.sh / lp_conv_clock.py)
// LDS bytes of one workgroup of lp_conv3x3_body (the constants are restated inside the body)
template <int CIN, int COUT, int TR>
constexpr int lp_conv_smem_bytes() {
  constexpr int NSL = CIN / 8, TILE_INSTR = ((TR + 2) * 34 * NSL + 63) / 64, NK16 = CIN / 16;
  constexpr int KC16_WANT = 128 / COUT > 1 ? 128 / COUT : 1, KC16 = KC16_WANT < NK16 ? KC16_WANT : NK16;
  return TILE_INSTR * 1024 + 3 * (3 * (COUT / 32) * KC16) * 1024;
}

// One workgroup tile (index t = (b * tiles_y + ty) * tiles_x + tx) of the convolution; `smem`: lp_conv_smem_bytes() bytes of LDS,
// 1024-byte aligned.  A device function so that lp_conv3x3_kernel (one tile per workgroup) and lp_resblocks_kernel (the four
// convolutions of two ResBlocks, one after the other in the same workgroup) run the SAME instructions per tile.
template <class T, int CIN, int COUT, int EPI, bool UP, int TR>
__device__ __forceinline__ void lp_conv3x3_body(const LpConvArgs& a, int t, char* smem) {
  constexpr int NCB = COUT / 32, RW = TR / 4, TC = 34, NPIX = (TR + 2) * TC;
  constexpr int PB = CIN * 2, NSL = CIN / 8;
  constexpr int TILE_SLOTS = NPIX * NSL, TILE_INSTR = (TILE_SLOTS + 63) / 64, TILE_BYTES = TILE_INSTR * 1024;
  // a weight chunk = one kernel row (3 taps) x KC16 k-steps of 16 input channels: sized so that a chunk feeds >= 24
  // MFMAs per wave (768+ matrix-pipe cycles) - the LDS-DMA of the NEXT chunk has to land within one chunk's compute
  // (measured: with 12-MFMA chunks the Cout = 64 layers spent 71 % of their wave cycles parked on that wait)
  constexpr int NK16 = CIN / 16;
  constexpr int KC16_WANT = 128 / COUT > 1 ? 128 / COUT : 1;
  constexpr int KC16 = KC16_WANT < NK16 ? KC16_WANT : NK16;
  // Three chunk buffers, chunk c+2 in flight while chunk c computes: an LDS-DMA takes ~1.1 us from issue to landed
  // (MI355X_MICROARCH.md, ldsdma-fill) but a 24-MFMA chunk shared by two waves of a SIMD is only ~0.7 us of matrix
  // pipe - with one chunk of look-ahead every chunk start waited for its weights (27 % of the wave cycles parked).
  constexpr int NBUF = 3;
  constexpr int WAVE_INSTR = (3 * NCB * KC16) / 4;               // weight DMAs per wave and chunk (at least)
  constexpr int STEP_BYTES = 3 * NCB * 1024;                     // one k16-step of a kernel row: [dx][cb][lane][8]
  constexpr int CHUNK_INSTR = 3 * NCB * KC16, CHUNK_BYTES = CHUNK_INSTR * 1024;
  constexpr int NKG = NK16 / KC16, NCH = 3 * NKG;
  static_assert(NK16 % KC16 == 0, "chunking");
  static_assert(TILE_BYTES + NBUF * CHUNK_BYTES == lp_conv_smem_bytes<CIN, COUT, TR>(), "LDS size");
  char* tile = smem;
  char* wbuf = smem + TILE_BYTES;
  float* aff = reinterpret_cast<float*>(smem + TILE_BYTES + NBUF * CHUNK_BYTES - COUT * 8);   // written after the main loop

  const int tid = threadIdx.x, lane = tid & 63, c0 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TR, x0 = tx * 32;                         // output origin (unpadded coordinates)

  // ---- input tile: all lanes of the 4 waves copy 16-byte slots; slot S of the tile = pixel S / NSL, physical slot
  // S % NSL holding the pixel's logical slot (S % NSL) ^ swz(column)
  {
    const int64_t img = (int64_t)(a.Hi + 2) * (a.Wi + 2) * a.xcp * 2;
    const char* xb = a.x + (int64_t)b * img;
    const int64_t rowb = (int64_t)(a.Wi + 2) * a.xcp * 2;
#pragma unroll
    for (int k = 0; k < (TILE_INSTR + 3) / 4; ++k) {
      const int ins = wave + 4 * k;
      if (ins < TILE_INSTR) {                                   // wave-uniform
        const int S = ins * 64 + lane;
        int pix = S / NSL;
        const int ps = S % NSL;
        if (pix >= NPIX) pix = 0;                               // tail lanes of the last piece: any valid address
        const int r = pix / TC, c = pix - r * TC;
        const int ls = ps ^ lp_swz<CIN>(c);
        const int sy = UP ? (y0 + 1 + r) >> 1 : y0 + r;         // padded source coordinates
        const int sx = UP ? (x0 + 1 + c) >> 1 : x0 + c;
        lds_dma16(xb + sy * rowb + (int64_t)sx * (a.xcp * 2) + ls * 16, tile + ins * 1024);
      }
    }
  }
  auto issue_w = [&](int ch, int buf) {
    const char* src = a.wpack + (int64_t)ch * CHUNK_BYTES + lane * 16;
#pragma unroll
    for (int k = 0; k < (CHUNK_INSTR + 3) / 4; ++k) {
      const int ins = wave + 4 * k;
      if (ins < CHUNK_INSTR) lds_dma16(src + ins * 1024, wbuf + buf * CHUNK_BYTES + ins * 1024);
    }
  };
  issue_w(0, 0);
  if (NCH > 1) issue_w(1, 1);

  f32x16v acc[NCB][RW];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int pr = 0; pr < RW; ++pr)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][pr][i] = 0.f;

  // per-lane B-fragment bases: pixel (row wave*RW, column c0 + dx), slot (k16*2 + h) ^ swz = (2 k16) ^ (h ^ swz)
  int bbase[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
    bbase[dx] = ((wave * RW) * TC + c0 + dx) * PB + ((h ^ lp_swz<CIN>(c0 + dx)) << 4);

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    // my pieces of chunk ch (and, first time, of the input tile) have landed - the DMAs of chunk ch+1 may still be in
    // flight (VMEM completes in order: all but the youngest WAVE_INSTR operations are done); after the barrier
    // everybody's pieces of chunk ch have, and everybody is done reading the buffer of chunk ch-1, which chunk ch+2 reuses
    if (ch + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAVE_INSTR) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ch + 2 < NCH) issue_w(ch + 2, (ch + 2) % NBUF);
    const int dy = ch / NKG, kg = ch % NKG;                      // the pack is [dy][k16][dx][cb]: a chunk is contiguous
#pragma unroll
    for (int kk = 0; kk < KC16; ++kk) {
      const int k16 = kg * KC16 + kk;
      const char* wb = wbuf + (ch % NBUF) * CHUNK_BYTES + kk * STEP_BYTES + lane * 16;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        u32x4 af[NCB], bf[RW];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          af[cb] = *reinterpret_cast<const u32x4*>(wb + (dx * NCB + cb) * 1024);
#pragma unroll
        for (int pr = 0; pr < RW; ++pr)
          bf[pr] = *reinterpret_cast<const u32x4*>(tile + (bbase[dx] ^ (k16 << 5)) + (pr + dy) * TC * PB);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int pr = 0; pr < RW; ++pr) acc[cb][pr] = LP<T>::mfma32(af[cb], bf[pr], acc[cb][pr]);
      }
    }
  }

  // ---- epilogue: affine (+ GLU | + residual) in registers, then through LDS so that HBM sees whole lines.
  // The accumulator layout gives a lane 4 consecutive channels (8 bytes) of one pixel: stored directly that is 16
  // partial writes per 128-byte line (measured: WRITE_SIZE 2x the tensor).  Instead every wave stages its RW x 32
  // pixels x OC channels in its own LDS region [pixel][OC] (16-byte chunks XOR-swizzled by the pixel index; the input
  // tile and weight buffers are dead by now), reads them back 16 bytes per lane with 8 (4) consecutive lanes covering
  // one pixel, and stores full pixel rows.  The residual tile comes in the same way in the opposite direction (LDS-DMA
  // with the swizzle on the source address) and the output overwrites it in place.
  constexpr int NOB = EPI == kEpiGlu ? NCB / 2 : NCB;
  constexpr int OC = NOB * 32, OB = OC * 2, NCHK = OB / 16;          // output channels, bytes and 16-byte chunks per pixel
  constexpr int STG_WAVE = RW * 32 * OB, STG_INSTR = STG_WAVE / 1024;
  static_assert(4 * STG_WAVE + COUT * 8 <= TILE_BYTES + NBUF * CHUNK_BYTES, "staging fits the dead tile + weight buffers");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                     // every wave is done reading the tile / weights
  if (tid < COUT) {
    aff[tid] = a.scale ? a.scale[tid] : 1.f;
    aff[COUT + tid] = a.shift ? a.shift[tid] : 0.f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  char* stg = smem + wave * STG_WAVE;
  const int64_t orow = (int64_t)(a.W + 2) * a.ocp * 2;
  char* ob = a.out + ((int64_t)b * (a.H + 2) + y0 + wave * RW + 1) * orow + (int64_t)(x0 + 1) * (a.ocp * 2) + a.oco * 2;
  if (EPI == kEpiRes) {
    const int64_t rrow = (int64_t)(a.W + 2) * a.rcp * 2;
    const char* rb = a.res + ((int64_t)b * (a.H + 2) + y0 + wave * RW + 1) * rrow + (int64_t)(x0 + 1) * (a.rcp * 2) + a.rco * 2;
#pragma unroll
    for (int j = 0; j < STG_INSTR; ++j) {
      const int S = j * 64 + lane, pw = S / NCHK, q = (S % NCHK) ^ (pw & (NCHK - 1));
      lds_dma16(rb + (pw >> 5) * rrow + (int64_t)(pw & 31) * (a.rcp * 2) + q * 16, stg + j * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int pr = 0; pr < RW; ++pr) {
    const int pw = pr * 32 + c0;
#pragma unroll
    for (int cb = 0; cb < NOB; ++cb) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int ch0 = cb * 32 + 8 * rg + 4 * h;
        const f32x4w s = *reinterpret_cast<const f32x4w*>(aff + ch0);
        const f32x4w sh = *reinterpret_cast<const f32x4w*>(aff + COUT + ch0);
        char* sp = stg + pw * OB + (((cb * 4 + rg) ^ (pw & (NCHK - 1))) << 4) + 8 * h;
        float o[4];
        if (EPI == kEpiGlu) {
          const f32x4w gs = *reinterpret_cast<const f32x4w*>(aff + COUT / 2 + ch0);
          const f32x4w gsh = *reinterpret_cast<const f32x4w*>(aff + COUT + COUT / 2 + ch0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v = acc[cb][pr][4 * rg + q] * s[q] + sh[q];
            const float g = acc[cb + NCB / 2][pr][4 * rg + q] * gs[q] + gsh[q];
            o[q] = v * sigmoidf_fast(g);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = acc[cb][pr][4 * rg + q] * s[q] + sh[q];
          if (EPI == kEpiRes) {
            const u32x2 rv = *reinterpret_cast<const u32x2*>(sp);
            o[0] += LP<T>::lo(rv[0]);
            o[1] += LP<T>::hi(rv[0]);
            o[2] += LP<T>::lo(rv[1]);
            o[3] += LP<T>::hi(rv[1]);
          }
        }
        u32x2 pk;
        pk[0] = LP<T>::pack2(o[0], o[1]);
        pk[1] = LP<T>::pack2(o[2], o[3]);
        *reinterpret_cast<u32x2*>(sp) = pk;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // this wave's own staging writes have landed
#pragma unroll
  for (int j = 0; j < STG_INSTR; ++j) {
    const int S = j * 64 + lane, pw = S / NCHK, q = (S % NCHK) ^ (pw & (NCHK - 1));
    const u32x4 v = *reinterpret_cast<const u32x4*>(stg + S * 16);
    *reinterpret_cast<u32x4*>(ob + (pw >> 5) * orow + (int64_t)(pw & 31) * (a.ocp * 2) + q * 16) = v;
  }
}

template <class T, int CIN, int COUT, int EPI, bool UP, int TR>
__global__ __launch_bounds__(256, (lp_conv_occ<CIN, COUT, TR>())) void lp_conv3x3_kernel(LpConvArgs a) {
  __shared__ __attribute__((aligned(1024))) char smem[lp_conv_smem_bytes<CIN, COUT, TR>()];
  lp_conv3x3_body<T, CIN, COUT, EPI, UP, TR>(a, xcd_remap(blockIdx.x, gridDim.x), smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// The two ResBlocks of a generator stage (util.py:110-130 behind INIT_STAGE_GImgup / NEXT_STAGE_G, :773, :818) - four dependent
// convolutions on 64 channels - in ONE launch.  At 32^2 and 64^2 a convolution's body is ~5 us of work behind ~7 us of
// kernel-to-kernel dependency latency in a replayed graph, and these eight layers sit on the step's dependent chain (DESIGN.md
// 3.8e, 3.17).  A workgroup owns one tile (image b, rows ty, columns tx) through ALL four layers:
//     L0  x   -> tmp   conv 64 -> 128 + affine + GLU          L2  a   -> tmp   the same for the second block
//     L1  tmp -> a     conv 64 -> 64 + affine + x             L3  tmp -> b     ... + a
// and what a layer reads from its neighbours - the one-pixel halo of the previous layer's output - is guarded by per-tile flags
// instead of a kernel boundary: before layer k a workgroup waits until the (up to 8) neighbouring tiles of the SAME image have
// published layer k - 1 (which also means they are done READING the buffer layer k is about to overwrite: tmp is written twice).
// A flag holds the number of launches that have published it; every launch adds exactly one to each, so "published in this
// launch" is "> the value my own flag had when I started" - nothing to reset between launches (hipGraph replays pass the same
// arguments every time).
// Progress: workgroups are dispatched in blockIdx order, the tile index IS blockIdx (no XCD remap here), and a workgroup only
// waits for tiles at most tiles_x + 1 indices ahead of it - so of any set of resident workgroups all but the last few rows can
// finish and free their slots, however many other kernels (other graph lanes running this same kernel included) compete for the
// CUs: no workgroup ever waits for one that cannot be scheduled.  Every wait is bounded (2^20 polls, a fraction of a second) and reports through
// the error word instead of hanging the device.
struct LpChainArgs {
  LpConvArgs l[4];
  unsigned* flags;          // [4][ntiles] + 1 error word
  int ntiles;
};

template <class T, int TR>
__global__ __launch_bounds__(256, 2) void lp_resblocks_kernel(LpChainArgs c) {
  constexpr int SM = lp_conv_smem_bytes<64, 128, TR>() > lp_conv_smem_bytes<64, 64, TR>() ? lp_conv_smem_bytes<64, 128, TR>()
                                                                                         : lp_conv_smem_bytes<64, 64, TR>();
  __shared__ __attribute__((aligned(1024))) char smem[SM];
  __shared__ unsigned s_target;
  const int t = blockIdx.x, tid = threadIdx.x;
  const int tiles_x = c.l[0].tiles_x, tiles_y = c.l[0].tiles_y;
  const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
  if (tid == 0) s_target = __hip_atomic_load(c.flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  __syncthreads();
  const unsigned target = s_target;
  // this thread's neighbour (threads 0 .. 8; 4 = the tile itself: nothing to wait for)
  int nb = -1;
  if (tid < 9 && tid != 4) {
    const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
    if ((unsigned)ny < (unsigned)tiles_y && (unsigned)nx < (unsigned)tiles_x) nb = t + (ny - ty) * tiles_x + (nx - tx);
  }
  // hand-off protocol of MI355X_MICROARCH.md (inter-workgroup visibility): producer - every storing wave drains its stores, the
  // workgroup's barrier, then ONE lane: agent-scope release (writes back this XCD's L2), drained, relaxed agent flag store;
  // consumer - relaxed polls, ONE agent-scope acquire (invalidates this CU's L1), drained, the workgroup's barrier, plain loads.
  auto publish = [&](int k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(c.flags + k * c.ntiles + t, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto wait_for = [&](int k) {                     // the neighbours have published layer k
    if (tid < 64) {                                // wave 0: lanes 0 .. 8 poll, then one acquire for the CU
      if (nb >= 0) {
        const unsigned* f = c.flags + k * c.ntiles + nb;
        int it = 0;
        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
          __builtin_amdgcn_s_sleep(4);
          if (++it > (1 << 20)) {                  // never spin forever: flag the launch as failed and go on
            __hip_atomic_store(c.flags + 4 * c.ntiles, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  };
  lp_conv3x3_body<T, 64, 128, kEpiGlu, false, TR>(c.l[0], t, smem);
  publish(0);
  wait_for(0);
  lp_conv3x3_body<T, 64, 64, kEpiRes, false, TR>(c.l[1], t, smem);
  publish(1);
  wait_for(1);
  lp_conv3x3_body<T, 64, 128, kEpiGlu, false, TR>(c.l[2], t, smem);
  publish(2);
  wait_for(2);
  lp_conv3x3_body<T, 64, 64, kEpiRes, false, TR>(c.l[3], t, smem);
  publish(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// upBlock (util.py:74-80) by sub-pixel decomposition: Upsample(x2, nearest) -> conv3x3 only ever combines a 2x2
// low-resolution neighbourhood per output pixel, so output phase (a, b) = (row & 1, column & 1) is a 2x2 convolution of
// the LOW-resolution image with pre-summed taps (summed in fp32, rounded once to T):
//     a = 0: rows {y-1: w[0], y: w[1]+w[2]}      a = 1: rows {y: w[0]+w[1], y+1: w[2]}       (same for columns / b)
// 16 (phase, tap) products per low-res pixel instead of 36: 2.25x fewer MFMAs than the direct form on the up-sampled grid.
// Workgroup = 4 low-res rows x 32 low-res columns (8 x 64 outputs) x 64 couts; wave = (row pair rp, row phase a): its 8
// accumulators are [2 low-res rows][column phase b][value | gate block].  Same staging as lp_conv3x3_kernel: the
// low-res halo tile once by swizzled LDS-DMA, the phase weights streamed in 16-KB chunks [k16][column group][8 combos]
// (column group 0 = the outer columns dx = -1 (b = 0) and +1 (b = 1), 1 = the centre column feeding both b), double
// buffered, one barrier per chunk; an A fragment feeds 2 MFMAs, a B fragment 2-4.
//
// HK = 3 | 5: the image head that reads this upBlock's output (GET_IMAGE_G_noAct util.py:909-919, HK = 3; conv_output
// model.py:224, HK = 5) is computed here too, from the output tile while it sits in LDS: V[(c, dx)][row][x'] =
// sum_{dy, ci} w[c][ci][dy][dx] h[ci][row + dy - P][x'] on MFMA 16x16x32 (the (output channel, kernel column) pairs are
// the rows of the A fragment, as in lp_to3_kernel), then the HK-term shift-sum over dx.  A tile only holds its own 8 x 64
// pixels, so it produces PARTIAL sums for the (8 + 2P) x (64 + 2P) outputs its pixels reach; lp_head_combine_kernel adds
// the <= 4 partials of every pixel in a fixed order (no float atomics) and applies tanh / + a SRb.  With a.out == null the
// 32-channel feature image - which only the head would read - never goes to HBM.
//
// ATT: NEXT_STAGE_G.forward opens with `c_code, att = self.att(h_code, word_embs)` on exactly the tensor this kernel writes
// (util.py:814-817; c_code[:, q] depends on h[:, q] and the words only), so each wave attends to the words for its own
// 2 x 64 pixels while they sit in its staging region (lp_attend_tile: B fragments = two ds_read_b128 of the staged pixel):
// c_code goes to channels [att.coff, att.coff + 32) of the same pixels of `out`, the attention map to att.attn.  The
// stand-alone attention launch of the next stage - and its re-read of h - disappear from G_SR_NET_low's dependent chain.
template <class T, int CIN, int HK, bool ATT = false>
__global__ __launch_bounds__(256, 2) void lp_upconv_glu_kernel(LpConvArgs a) {
  constexpr int TRL = 4, TC = 34, NPIX = (TRL + 2) * TC, PB = CIN * 2, NSL = CIN / 8, COUT = 64;
  constexpr int TILE_SLOTS = NPIX * NSL, TILE_INSTR = (TILE_SLOTS + 63) / 64, TILE_BYTES = TILE_INSTR * 1024;
  constexpr int CHUNK_INSTR = 16, CHUNK_BYTES = CHUNK_INSTR * 1024;
  constexpr int NK16 = CIN / 16, NCH = 2 * NK16, NBUF = 3, WAVE_INSTR = CHUNK_INSTR / 4;   // see lp_conv3x3_kernel
  __shared__ __attribute__((aligned(1024))) char smem[TILE_BYTES + NBUF * CHUNK_BYTES];
  char* tile = smem;
  char* wbuf = smem + TILE_BYTES;
  float* aff = reinterpret_cast<float*>(smem + TILE_BYTES + NBUF * CHUNK_BYTES - COUT * 8);
  const int tid = threadIdx.x, lane = tid & 63, c0 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave >> 1, ph = wave & 1;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TRL, x0 = tx * 32;                         // low-res origin (unpadded)
  u32x4 hf[HK > 0 ? HK : 1];                                     // the head's A fragments: fetched now, used after the epilogue
  if constexpr (HK > 0) {
#pragma unroll
    for (int k = 0; k < HK; ++k) hf[k] = *reinterpret_cast<const u32x4*>(a.hw + (k * 64 + lane) * 16);
  }
  u32x4 fa[4];                                                   // the attention's A fragments (this sample's projected words)
  if constexpr (ATT) {
#pragma unroll
    for (int f = 0; f < 4; ++f) fa[f] = *reinterpret_cast<const u32x4*>(a.att.frag + (int64_t)b * 4096 + (f * 64 + lane) * 16);
  }
  {
    const int64_t rowb = (int64_t)(a.Wi + 2) * a.xcp * 2;
    const char* xb = a.x + (int64_t)b * (a.Hi + 2) * rowb;
#pragma unroll
    for (int k = 0; k < (TILE_INSTR + 3) / 4; ++k) {
      const int ins = wave + 4 * k;
      if (ins < TILE_INSTR) {
        const int S = ins * 64 + lane;
        int pix = S / NSL;
        const int ps = S % NSL;
        if (pix >= NPIX) pix = 0;
        const int r = pix / TC, c = pix - r * TC;
        const int ls = ps ^ lp_swz<CIN>(c);
        lds_dma16(xb + (y0 + r) * rowb + (int64_t)(x0 + c) * (a.xcp * 2) + ls * 16, tile + ins * 1024);
      }
    }
  }
  auto issue_w = [&](int ch, int buf) {
    const char* src = a.wpack + (int64_t)ch * CHUNK_BYTES + lane * 16;
#pragma unroll
    for (int k = 0; k < CHUNK_INSTR / 4; ++k) {
      const int ins = wave + 4 * k;
      lds_dma16(src + ins * 1024, wbuf + buf * CHUNK_BYTES + ins * 1024);
    }
  };
  issue_w(0, 0);
  issue_w(1, 1);
  f32x16v acc[2][2][2];                                          // [low-res row][column phase b][value | gate]
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rr][bb][cb][i] = 0.f;
  // B-fragment bases: tile row (2 rp + ph) [+ rr + dys], tile column c0 + dxi (dxi = 0..2 <-> dx = -1..+1)
  int bbase[3];
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi)
    bbase[dxi] = ((2 * rp + ph) * TC + c0 + dxi) * PB + ((h ^ lp_swz<CIN>(c0 + dxi)) << 4);

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAVE_INSTR) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ch + 2 < NCH) issue_w(ch + 2, (ch + 2) % NBUF);
    const int k16 = ch >> 1;
    const char* wb = wbuf + (ch % NBUF) * CHUNK_BYTES + lane * 16 + ph * (8 * 1024);   // this row phase's 4 combos
    if ((ch & 1) == 0) {                                          // outer columns: dx = -1 feeds b = 0, dx = +1 feeds b = 1
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int dys = 0; dys < 2; ++dys) {
          u32x4 af[2], bf[2];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) af[cb] = *reinterpret_cast<const u32x4*>(wb + ((sb * 2 + dys) * 2 + cb) * 1024);
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
            bf[rr] = *reinterpret_cast<const u32x4*>(tile + (bbase[2 * sb] ^ (k16 << 5)) + (rr + dys) * TC * PB);
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[rr][sb][cb] = LP<T>::mfma32(af[cb], bf[rr], acc[rr][sb][cb]);
        }
    } else {                                                      // centre column: feeds both column phases
#pragma unroll
      for (int dys = 0; dys < 2; ++dys) {
        u32x4 bf[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
          bf[rr] = *reinterpret_cast<const u32x4*>(tile + (bbase[1] ^ (k16 << 5)) + (rr + dys) * TC * PB);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          u32x4 af[2];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) af[cb] = *reinterpret_cast<const u32x4*>(wb + ((sb * 2 + dys) * 2 + cb) * 1024);
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[rr][sb][cb] = LP<T>::mfma32(af[cb], bf[rr], acc[rr][sb][cb]);
        }
      }
    }
  }

  // epilogue through LDS (see lp_conv3x3_kernel): a wave owns two output rows (2 (y0 + 2 rp + rr) + ph) of 64 pixels x
  // 32 channels; staged [rr][64 pixels][64 bytes] with the 4 chunks of a pixel XOR-swizzled, stored as whole pixel rows
  constexpr int OB = 64, NCHK = 4, STG_WAVE = 2 * 64 * OB, STG_INSTR = STG_WAVE / 1024;
  static_assert(4 * STG_WAVE + COUT * 8 <= TILE_BYTES + NBUF * CHUNK_BYTES, "staging fits the dead tile + weight buffers");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (tid < COUT) {
    aff[tid] = a.scale ? a.scale[tid] : 1.f;
    aff[COUT + tid] = a.shift ? a.shift[tid] : 0.f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  char* stg = smem + wave * STG_WAVE;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int pw = rr * 64 + 2 * c0 + sb;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int ch0 = 8 * rg + 4 * h;
        const f32x4w s = *reinterpret_cast<const f32x4w*>(aff + ch0);
        const f32x4w sh = *reinterpret_cast<const f32x4w*>(aff + COUT + ch0);
        const f32x4w gs = *reinterpret_cast<const f32x4w*>(aff + 32 + ch0);
        const f32x4w gsh = *reinterpret_cast<const f32x4w*>(aff + COUT + 32 + ch0);
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = acc[rr][sb][0][4 * rg + q] * s[q] + sh[q];
          const float g = acc[rr][sb][1][4 * rg + q] * gs[q] + gsh[q];
          o[q] = v * sigmoidf_fast(g);
        }
        u32x2 pk;
        pk[0] = LP<T>::pack2(o[0], o[1]);
        pk[1] = LP<T>::pack2(o[2], o[3]);
        *reinterpret_cast<u32x2*>(stg + pw * OB + ((rg ^ ((pw >> 1) & (NCHK - 1))) << 4) + 8 * h) = pk;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (HK == 0 || a.out != nullptr) {
    const int64_t orow = (int64_t)(a.W + 2) * a.ocp * 2;
    char* ob = a.out + ((int64_t)b * (a.H + 2) + 2 * (y0 + 2 * rp) + ph + 1) * orow + (int64_t)(2 * x0 + 1) * (a.ocp * 2) +
               a.oco * 2;
#pragma unroll
    for (int j = 0; j < STG_INSTR; ++j) {
      const int S = j * 64 + lane, pw = S / NCHK, q = (S % NCHK) ^ ((pw >> 1) & (NCHK - 1));
      const u32x4 v = *reinterpret_cast<const u32x4*>(stg + S * 16);
      *reinterpret_cast<u32x4*>(ob + (pw >> 6) * (2 * orow) + (int64_t)(pw & 63) * (a.ocp * 2) + q * 16) = v;
    }
  }
  if constexpr (ATT) {
    // ---- the next stage's word attention on this wave's own 2 rows x 64 pixels (staged above; own lgkmcnt(0) passed)
    const int64_t Q = (int64_t)a.H * a.W;
    const int64_t orow = (int64_t)(a.W + 2) * a.ocp * 2;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * (y0 + 2 * rp + rr) + ph;
#pragma unroll
      for (int seg = 0; seg < 2; ++seg) {
        const int pw = rr * 64 + seg * 32 + c0, col = 2 * x0 + seg * 32 + c0, sw = (pw >> 1) & (NCHK - 1);
        const u32x4 b0 = *reinterpret_cast<const u32x4*>(stg + pw * OB + ((h ^ sw) << 4));          // channels 8 h ..
        const u32x4 b1 = *reinterpret_cast<const u32x4*>(stg + pw * OB + (((2 + h) ^ sw) << 4));    // channels 16 + 8 h ..
        const int64_t q = (int64_t)row * a.W + col;
        unsigned mb = 0;
        if (a.att.mbits) mb = a.att.mbits[a.att.mask_mode ? b : (int)(((int64_t)b * Q + q) % a.B)];   // GlobalAttention.py:111
        char* cp = a.out + ((int64_t)b * (a.H + 2) + row + 1) * orow + (int64_t)(col + 1) * (a.ocp * 2) + a.att.coff * 2;
        lp_attend_tile<T>(fa, b0, b1, mb, a.att.T, h, a.att.attn ? a.att.attn + (int64_t)b * a.att.T * Q + q : nullptr, Q, cp);
      }
    }
  }
  if constexpr (HK > 0) {
    // ---- fused image head on the tile in LDS: output row R (0..7) of the tile lives in the staging region of wave
    // (R >> 2) * 2 + (R & 1), half rr = (R >> 1) & 1, as [64 pixels][64 bytes] with the 16-byte chunk of channels 8 g ..
    // at physical chunk g ^ ((pixel >> 1) & 3): the B fragment of 16 neighbouring pixels x 4 channel groups is one
    // conflict-free ds_read_b128.  V image of a wave: [16 rows (c, dx)][VP], column x' + 2P, zero outside [0, 64): the
    // shift-sum out[c][x] = sum_dx V[(c, dx)][x + dx - P] then needs no bounds test.
    constexpr int P = HK / 2, NR = 8 + 2 * P, NX = 64 + 2 * P, VP = 76;   // VP >= 64 + 4P; rows 4 apart 16 banks apart
    static_assert(4 * STG_WAVE + 4 * 16 * VP * 4 + COUT * 8 <= TILE_BYTES + NBUF * CHUNK_BYTES, "V images fit behind the tile");
    const int p = lane & 15, g = lane >> 4;
    float* v = reinterpret_cast<float*>(smem + 4 * STG_WAVE) + wave * (16 * VP);
    for (int o = lane; o < 16 * 4 * P; o += 64) {                // the 2P zero columns on either side, once
      const int r = o / (4 * P), q = o - r * (4 * P);
      v[r * VP + (q < 2 * P ? q : 64 + q)] = 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // every wave's rows are staged (own lgkmcnt(0) above)
    float* pb = a.hpart + ((int64_t)(b * a.tiles_y + ty) * a.tiles_x + tx) * (3 * NR * NX);
#pragma unroll 1
    for (int yo = wave - P; yo < 8 + P; yo += 4) {               // this wave's output rows (tile-local, -P .. 7 + P)
      f32x4w acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
#pragma unroll
      for (int dy = 0; dy < HK; ++dy) {
        const int R = yo + dy - P;                               // source row: wave-uniform
        if ((unsigned)R < 8u) {
          const char* rb = smem + ((R >> 2) * 2 + (R & 1)) * STG_WAVE + ((R >> 1) & 1) * (64 * OB);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int xq = 16 * j + p;
            const u32x4 bf = *reinterpret_cast<const u32x4*>(rb + xq * OB + ((g ^ ((xq >> 1) & 3)) << 4));
            acc[j] = LP<T>::mfma16(hf[dy], bf, acc[j]);
          }
        }
      }
      // D[row = 4 g + i][col = p] -> V image, then the shift-sum over dx: lane = output column xi (and xi + 64)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[(4 * g + i) * VP + 2 * P + 16 * j + p] = acc[j][i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private image: no barrier
      float* prow = pb + (yo + P) * NX;
#pragma unroll
      for (int rnd = 0; rnd < 2; ++rnd) {
        const int xi = lane + 64 * rnd;                          // output column x = xi - P (tile-local)
        if (xi < NX) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float sum = 0.f;
#pragma unroll
            for (int dx = 0; dx < HK; ++dx) sum += v[(c * HK + dx) * VP + xi + dx];   // V column (x + dx - P) + 2P
            prow[c * (NR * NX) + xi] = sum;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the V image is rewritten for the next row
    }
  }
}

// Sum of the per-tile partial sums the head-fused upBlocks wrote, for up to 4 scales at once and both generators:
//   low[s]  = act_low(sum of G_SR_NET_low's 3x3 partials)                (GET_IMAGE_G_noAct: none; GET_IMAGE_G x16: tanh)
//   high[s] = tanh(sum of NetG_highweight's 5x5 partials) + alpha * low[s]   (model.py:280, 288, 297)
// A pixel receives at most 2 x 2 partials (tile rows ty, columns tx ascending: a fixed order).  thread = one pixel of
// one scale, all 3 channels of both images.
struct LpCombineScale {
  const float* pl;     // [B][ty][tx][3][10][66] or null
  const float* ph;     // [B][ty][tx][3][12][68] or null
  float* low;          // [B][3][H][W]
  float* high;         // [B][3][H][W] or null
  int H, W, tiles_x, tiles_y;
  int64_t first;       // index of this scale's first pixel in the launch's flat pixel range
};
struct LpCombineArgs {
  LpCombineScale s[4];
  int nscales, B, low_tanh;
  float alpha;
  int64_t total;
};

// tanh(v) = 1 - 2 / (exp(2v) + 1): absolute error ~1e-7, saturates cleanly (the fp32 heads' form, tgsr_conv_to3.hip)
__device__ __forceinline__ float lp_fast_tanh(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f); }

template <int P>
__device__ __forceinline__ void head_gather(const float* part, int b, int y, int x, int tiles_y, int tiles_x, float (&o)[3]) {
  constexpr int NR = 8 + 2 * P, NX = 64 + 2 * P;
  o[0] = o[1] = o[2] = 0.f;
  const int ty0 = (y - P) >= 0 ? (y - P) >> 3 : 0, ty1 = min((y + P) >> 3, tiles_y - 1);
  const int tx0 = (x - P) >= 0 ? (x - P) >> 6 : 0, tx1 = min((x + P) >> 6, tiles_x - 1);
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) {
      const float* pb = part + ((int64_t)(b * tiles_y + ty) * tiles_x + tx) * (3 * NR * NX) + (y - 8 * ty + P) * NX +
                        (x - 64 * tx + P);
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] += pb[c * NR * NX];
    }
}

__global__ __launch_bounds__(256) void lp_head_combine_kernel(LpCombineArgs a) {
  // thread = 4 consecutive pixels of a row (W % 64 == 0): 6 float4 stores instead of 24 scalar ones
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= a.total) return;
  int k = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (q < a.nscales && i >= a.s[q].first) k = q;
  const LpCombineScale& sc = a.s[k];
  int64_t t = i - sc.first;
  const int x = (int)(t % sc.W);
  t /= sc.W;
  const int y = (int)(t % sc.H);
  const int b = (int)(t / sc.H);
  const int64_t HW = (int64_t)sc.H * sc.W, oi = (int64_t)b * 3 * HW + (int64_t)y * sc.W + x;
  float lo[4][3];
  if (sc.pl) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      head_gather<1>(sc.pl, b, y, x + p, sc.tiles_y, sc.tiles_x, lo[p]);
      if (a.low_tanh) {
#pragma unroll
        for (int c = 0; c < 3; ++c) lo[p][c] = lp_fast_tanh(lo[p][c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *reinterpret_cast<float4*>(sc.low + oi + c * HW) = make_float4(lo[0][c], lo[1][c], lo[2][c], lo[3][c]);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(sc.low + oi + c * HW);
      lo[0][c] = v.x; lo[1][c] = v.y; lo[2][c] = v.z; lo[3][c] = v.w;
    }
  }
  if (sc.ph) {
    float hi[4][3];
#pragma unroll
    for (int p = 0; p < 4; ++p) head_gather<2>(sc.ph, b, y, x + p, sc.tiles_y, sc.tiles_x, hi[p]);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *reinterpret_cast<float4*>(sc.high + oi + c * HW) =
          make_float4(lp_fast_tanh(hi[0][c]) + a.alpha * lo[0][c], lp_fast_tanh(hi[1][c]) + a.alpha * lo[1][c],
                      lp_fast_tanh(hi[2][c]) + a.alpha * lo[2][c], lp_fast_tanh(hi[3][c]) + a.alpha * lo[3][c]);
  }
}

// wpack[k16][column group 2][combo 8][cb 2][lane 64][8] <- pre-summed sub-pixel taps of w[64][Cin][3][3].
// combo = a * 4 + sb * 2 + dys: row phase a, row offset dyi = a + dys (0..2 <-> dy = -1..+1), column phase b = sb and
// column offset dxi = (group 0: 2 sb | group 1: 1).
template <class T>
__global__ void lp_pack_upconv_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), l = (int)((i >> 3) & 63), cb = (int)((i >> 9) & 1), combo = (int)((i >> 10) & 7);
    const int grp = (int)((i >> 13) & 1), k16 = (int)(i >> 14);
    const int pa = combo >> 2, sb = (combo >> 1) & 1, dys = combo & 1;
    const int dyi = pa + dys, dxi = grp == 0 ? 2 * sb : 1;
    const int co = cb * 32 + (l & 31), ci = k16 * 16 + 8 * (l >> 5) + j;
    const float* g = w + ((int64_t)co * Cin + ci) * 9;
    // taps of the 3x3 kernel that land on low-res offset d (0..2) for phase p: p=0: {0 | 1,2 | -}, p=1: {- | 0,1 | 2}
    auto lo = [](int p, int d) { return p == 0 ? (d == 0 ? 0 : 1) : (d == 1 ? 0 : 2); };
    auto hi = [](int p, int d) { return p == 0 ? (d == 0 ? 0 : 2) : (d == 1 ? 1 : 2); };
    // fp32 sums in the order of the CPU model (oracle/tgsr_oracle_lp.py subpixel_upconv): rows first, then columns
    float v = 0.f;
    bool first = true;
    for (int kx = lo(sb, dxi); kx <= hi(sb, dxi); ++kx) {
      float c = g[lo(pa, dyi) * 3 + kx];
      if (hi(pa, dyi) != lo(pa, dyi)) c += g[hi(pa, dyi) * 3 + kx];
      v = first ? c : v + c;
      first = false;
    }
    wp[i] = LP<T>::one(v);
  }
}

// wpack[chunk = dy * (Cin/16) + k16][dx][cb][lane 64][8] <- w[Cout][Cin][3][3]:
// element j of lane l = w[cb*32 + (l & 31)][k16*16 + 8 (l >> 5) + j][dy][dx], rounded to T.
template <class T>
__global__ void lp_pack_conv3x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout, int Cin,
                                       int64_t total) {
  const int ncb = Cout / 32, nk16 = Cin / 16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
    int64_t t = i >> 9;
    const int cb = (int)(t % ncb);
    t /= ncb;
    const int dx = (int)(t % 3);
    t /= 3;
    const int k16 = (int)(t % nk16);
    const int dy = (int)(t / nk16);
    const int co = cb * 32 + (l & 31), ci = k16 * 16 + 8 * (l >> 5) + j;
    wp[i] = LP<T>::one(w[((int64_t)co * Cin + ci) * 9 + dy * 3 + dx]);
  }
}

// fp32 NCHW [B][C][H][W] -> zero-bordered channels-last T [B][H+2][W+2][cpitch] at channel offset coff (interior only)
template <class T>
__global__ void lp_from_nchw_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, int B, int C, int H,
                                    int W, int cpitch, int coff) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int xx = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    out[(((int64_t)b * (H + 2) + y + 1) * (W + 2) + xx + 1) * cpitch + coff + c] =
        LP<T>::one(x[(((int64_t)b * C + c) * H + y) * W + xx]);
  }
}

template <class T>
__global__ void lp_to_nchw_kernel(const unsigned short* __restrict__ x, float* __restrict__ out, int B, int C, int H,
                                  int W, int cpitch, int coff) {
  const int64_t total = (int64_t)B * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    int64_t t = i / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const unsigned v = x[(((int64_t)b * (H + 2) + y + 1) * (W + 2) + xx + 1) * cpitch + coff + c];
    out[i] = LP<T>::lo(v);
  }
}

// lp image of one 2-byte type -> the same image in the other (bf16 <-> f16): 8 elements per thread, the whole buffer incl.
// the zero border (0 -> 0).  One rounding (to nearest even) where the target has fewer bits.  The one place of the bf16
// configuration that needs it: NetG_highweight's 32^2 trunk runs with f16 operands, its output feeds bf16 layers.
template <class TS, class TD>
__global__ __launch_bounds__(256) void lp_convert_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const u32x4 v = src[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = LP<TD>::pack2(LP<TS>::lo(v[j]), LP<TS>::hi(v[j]));
    dst[i] = o;
  }
}

template <class T, int CIN, int COUT, int EPI, bool UP>
static int launch_lp_conv_tr(const LpConvArgs& a0, hipStream_t s) {
  LpConvArgs a = a0;
  // rows per workgroup: 8 when that still gives >= 2 workgroups per CU, else 4
  const bool big = (int64_t)a.B * (a.H / 8) * (a.W / 32) >= 512 && a.H % 8 == 0;
  a.tiles_x = a.W / 32;
  if (big) {
    a.tiles_y = a.H / 8;
    hipLaunchKernelGGL((lp_conv3x3_kernel<T, CIN, COUT, EPI, UP, 8>), dim3(a.B * a.tiles_x * a.tiles_y), dim3(256), 0, s, a);
  } else {
    a.tiles_y = a.H / 4;
    hipLaunchKernelGGL((lp_conv3x3_kernel<T, CIN, COUT, EPI, UP, 4>), dim3(a.B * a.tiles_x * a.tiles_y), dim3(256), 0, s, a);
  }
  return note_launch(hipGetLastError(), "lp_conv3x3_kernel");
}

template <class T, int CIN, int COUT>
static int launch_lp_conv_epi(const LpConvArgs& a, int epi, bool up, hipStream_t s) {
  if (epi == kEpiGlu) {
    if constexpr (COUT >= 64) {
      return up ? launch_lp_conv_tr<T, CIN, COUT, kEpiGlu, true>(a, s) : launch_lp_conv_tr<T, CIN, COUT, kEpiGlu, false>(a, s);
    } else {
      return TGSR_EUNSUPPORTED;
    }
  }
  if (up) return TGSR_EUNSUPPORTED;          // the reference only up-samples in front of a GLU block (util.py:74-80)
  if (epi == kEpiRes) return launch_lp_conv_tr<T, CIN, COUT, kEpiRes, false>(a, s);
  return launch_lp_conv_tr<T, CIN, COUT, kEpiAffine, false>(a, s);
}

template <class T>
static int launch_lp_conv(const LpConvArgs& a, int Cin, int Cout, int epi, bool up, hipStream_t s) {
  if (Cin == 64 && Cout == 128) return launch_lp_conv_epi<T, 64, 128>(a, epi, up, s);
  if (Cin == 64 && Cout == 64) return launch_lp_conv_epi<T, 64, 64>(a, epi, up, s);
  if (Cin == 32 && Cout == 64) return launch_lp_conv_epi<T, 32, 64>(a, epi, up, s);
  if (Cin == 32 && Cout == 32) return launch_lp_conv_epi<T, 32, 32>(a, epi, up, s);
  return TGSR_EUNSUPPORTED;
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_lp_packed_conv3x3_elems(int Cout, int Cin) { return (int64_t)Cout * Cin * 9; }

extern "C" int tgsr_lp_pack_conv3x3_weight(int dtype, const float* w, void* wpack, int Cout, int Cin, void* stream) {
  if (!w || !wpack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  if (Cout % 32 != 0 || Cin % 16 != 0) return TGSR_EUNSUPPORTED;
  const int64_t total = (int64_t)Cout * Cin * 9;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  unsigned short* o = static_cast<unsigned short*>(wpack);
  if (dtype == TGSR_DT_BF16)
    hipLaunchKernelGGL(lp_pack_conv3x3_kernel<BF16>, dim3(blocks), dim3(256), 0, as_stream(stream), w, o, Cout, Cin, total);
  else if (dtype == TGSR_DT_F16)
    hipLaunchKernelGGL(lp_pack_conv3x3_kernel<F16>, dim3(blocks), dim3(256), 0, as_stream(stream), w, o, Cout, Cin, total);
  else
    return TGSR_EINVAL;
  return note_launch(hipGetLastError(), "lp_pack_conv3x3_kernel");
}

extern "C" int tgsr_lp_conv3x3_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack,
                                   int Cout, const float* scale, const float* shift, const void* residual, int res_cpitch,
                                   int res_coff, void* out, int out_cpitch, int out_coff, int epilogue, int upsample,
                                   void* stream) {
  if (!x || !wpack || !out || B < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  if (epilogue != TGSR_EPI_AFFINE && epilogue != TGSR_EPI_AFFINE_GLU) return TGSR_EINVAL;
  if (epilogue == TGSR_EPI_AFFINE_GLU && residual) return TGSR_EINVAL;
  if (dtype != TGSR_DT_BF16 && dtype != TGSR_DT_F16) return TGSR_EINVAL;
  const int co = epilogue == TGSR_EPI_AFFINE_GLU ? Cout / 2 : Cout;
  if (W % 32 != 0 || H % 4 != 0 || (upsample && ((H | W) & 1))) return TGSR_EUNSUPPORTED;
  if (x_cpitch < Cin || x_cpitch % 8 != 0 || out_cpitch % 8 != 0 || out_coff % 8 != 0 || out_coff + co > out_cpitch)
    return TGSR_EUNSUPPORTED;                      // rows of pixels move as 16-byte pieces: offsets / pitches % 8 channels
  if (residual && (res_cpitch % 8 != 0 || res_coff % 8 != 0 || res_coff + co > res_cpitch)) return TGSR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wpack) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(residual) & 15))
    return TGSR_EUNSUPPORTED;
  if ((int64_t)(H + 2) * (W + 2) * (x_cpitch > out_cpitch ? x_cpitch : out_cpitch) * 2 >= (1ll << 31)) return TGSR_EUNSUPPORTED;
  LpConvArgs a;
  a.x = static_cast<const char*>(x); a.xcp = x_cpitch; a.B = B; a.H = H; a.W = W;
  a.Hi = upsample ? H / 2 : H; a.Wi = upsample ? W / 2 : W;
  a.wpack = static_cast<const char*>(wpack); a.scale = scale; a.shift = shift;
  a.res = static_cast<const char*>(residual); a.rcp = res_cpitch; a.rco = res_coff;
  a.out = static_cast<char*>(out); a.ocp = out_cpitch; a.oco = out_coff;
  a.tiles_x = a.tiles_y = 0;
  a.hw = nullptr; a.hpart = nullptr;
  const int epi = epilogue == TGSR_EPI_AFFINE_GLU ? kEpiGlu : (residual ? kEpiRes : kEpiAffine);
  if (dtype == TGSR_DT_BF16) return launch_lp_conv<BF16>(a, Cin, Cout, epi, upsample != 0, as_stream(stream));
  return launch_lp_conv<F16>(a, Cin, Cout, epi, upsample != 0, as_stream(stream));
}

extern "C" int64_t tgsr_lp_resblocks_flag_elems(int B, int H, int W) {
  if (B < 1 || H < 4 || W < 32 || H % 4 != 0 || W % 32 != 0) return 0;
  return 4ll * B * (H / 4) * (W / 32) + 1;
}

extern "C" int tgsr_lp_resblocks_fwd(int dtype, const void* x, int x_cpitch, int B, int H, int W, const void* const* wpack,
                                     const float* const* scale, const float* const* shift, void* tmp, int tmp_cpitch, void* a_out,
                                     int a_cpitch, void* b_out, int b_cpitch, unsigned* flags, void* stream) {
  if (!x || !wpack || !scale || !shift || !tmp || !a_out || !b_out || !flags || B < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (dtype != TGSR_DT_BF16 && dtype != TGSR_DT_F16) return TGSR_EINVAL;
  if (W % 32 != 0 || H % 4 != 0) return TGSR_EUNSUPPORTED;
  const int cps[4] = {x_cpitch, tmp_cpitch, a_cpitch, b_cpitch};
  const void* ptrs[4] = {x, tmp, a_out, b_out};
  int maxcp = 0;
  for (int i = 0; i < 4; ++i) {
    if (cps[i] < 64 || cps[i] % 8 != 0 || (reinterpret_cast<uintptr_t>(ptrs[i]) & 15)) return TGSR_EUNSUPPORTED;
    maxcp = cps[i] > maxcp ? cps[i] : maxcp;
    if (!wpack[i] || (reinterpret_cast<uintptr_t>(wpack[i]) & 15) || (scale[i] == nullptr) != (shift[i] == nullptr)) return TGSR_EINVAL;
  }
  if (tmp == x || tmp == a_out || tmp == b_out || a_out == x || a_out == b_out) return TGSR_EINVAL;   // four distinct roles
  if ((int64_t)(H + 2) * (W + 2) * maxcp * 2 >= (1ll << 31)) return TGSR_EUNSUPPORTED;
  LpChainArgs c;
  const char* in[4] = {static_cast<const char*>(x), static_cast<const char*>(tmp), static_cast<const char*>(a_out),
                       static_cast<const char*>(tmp)};
  const int incp[4] = {x_cpitch, tmp_cpitch, a_cpitch, tmp_cpitch};
  char* out[4] = {static_cast<char*>(tmp), static_cast<char*>(a_out), static_cast<char*>(tmp), static_cast<char*>(b_out)};
  const int outcp[4] = {tmp_cpitch, a_cpitch, tmp_cpitch, b_cpitch};
  const char* res[4] = {nullptr, static_cast<const char*>(x), nullptr, static_cast<const char*>(a_out)};
  const int rescp[4] = {0, x_cpitch, 0, a_cpitch};
  for (int i = 0; i < 4; ++i) {
    LpConvArgs& a = c.l[i];
    a.x = in[i]; a.xcp = incp[i]; a.B = B; a.H = H; a.W = W; a.Hi = H; a.Wi = W;
    a.wpack = static_cast<const char*>(wpack[i]); a.scale = scale[i]; a.shift = shift[i];
    a.res = res[i]; a.rcp = rescp[i]; a.rco = 0;
    a.out = out[i]; a.ocp = outcp[i]; a.oco = 0;
    a.tiles_x = W / 32; a.tiles_y = H / 4;
    a.hw = nullptr; a.hpart = nullptr;
  }
  c.flags = flags;
  c.ntiles = B * (H / 4) * (W / 32);
  const dim3 grid((unsigned)c.ntiles);
  if (dtype == TGSR_DT_BF16) hipLaunchKernelGGL((lp_resblocks_kernel<BF16, 4>), grid, dim3(256), 0, as_stream(stream), c);
  else hipLaunchKernelGGL((lp_resblocks_kernel<F16, 4>), grid, dim3(256), 0, as_stream(stream), c);
  return note_launch(hipGetLastError(), "lp_resblocks_kernel");
}

extern "C" int64_t tgsr_lp_packed_upconv_elems(int Cout, int Cin) { return (int64_t)Cout * Cin * 16; }

extern "C" int tgsr_lp_pack_upconv_weight(int dtype, const float* w, void* wpack, int Cout, int Cin, void* stream) {
  if (!w || !wpack) return TGSR_EINVAL;
  if (Cout != 64 || (Cin != 32 && Cin != 64)) return TGSR_EUNSUPPORTED;
  const int64_t total = (int64_t)Cout * Cin * 16;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  unsigned short* o = static_cast<unsigned short*>(wpack);
  if (dtype == TGSR_DT_BF16)
    hipLaunchKernelGGL(lp_pack_upconv_kernel<BF16>, dim3(blocks), dim3(256), 0, as_stream(stream), w, o, Cin, total);
  else if (dtype == TGSR_DT_F16)
    hipLaunchKernelGGL(lp_pack_upconv_kernel<F16>, dim3(blocks), dim3(256), 0, as_stream(stream), w, o, Cin, total);
  else
    return TGSR_EINVAL;
  return note_launch(hipGetLastError(), "lp_pack_upconv_kernel");
}

template <class T, int HK>
static void launch_lp_upconv(const LpConvArgs& a, int Cin, dim3 grid, hipStream_t s) {
  if (Cin == 64) hipLaunchKernelGGL((lp_upconv_glu_kernel<T, 64, HK>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((lp_upconv_glu_kernel<T, 32, HK>), grid, dim3(256), 0, s, a);
}

template <class T, int HK>
static void launch_lp_upconv_att(const LpConvArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((lp_upconv_glu_kernel<T, 64, HK, true>), grid, dim3(256), 0, s, a);
}

static int lp_upconv_launch(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack, int Cout,
                            const float* scale, const float* shift, void* out, int out_cpitch, int out_coff,
                            const void* head_wpack, int head_k, float* head_partial, void* stream,
                            const LpAttFuse* att = nullptr) {
  if (!x || !wpack || B < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (!out && !head_partial) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  if (dtype != TGSR_DT_BF16 && dtype != TGSR_DT_F16) return TGSR_EINVAL;
  if (head_partial && (!head_wpack || (head_k != 3 && head_k != 5))) return TGSR_EINVAL;
  if (Cout != 64 || (Cin != 32 && Cin != 64) || W % 32 != 0 || H % 4 != 0) return TGSR_EUNSUPPORTED;   // H, W: LOW-res size
  if (x_cpitch < Cin || x_cpitch % 8 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wpack) & 15) ||
      (reinterpret_cast<uintptr_t>(head_wpack) & 15))
    return TGSR_EUNSUPPORTED;
  if (out && (out_cpitch % 8 != 0 || out_coff % 8 != 0 || out_coff + 32 > out_cpitch || (reinterpret_cast<uintptr_t>(out) & 15) ||
              (int64_t)(2 * H + 2) * (2 * W + 2) * out_cpitch * 2 >= (1ll << 31)))
    return TGSR_EUNSUPPORTED;
  LpConvArgs a;
  a.x = static_cast<const char*>(x); a.xcp = x_cpitch; a.B = B; a.H = 2 * H; a.W = 2 * W; a.Hi = H; a.Wi = W;
  a.wpack = static_cast<const char*>(wpack); a.scale = scale; a.shift = shift; a.res = nullptr; a.rcp = a.rco = 0;
  a.out = static_cast<char*>(out); a.ocp = out_cpitch; a.oco = out_coff; a.tiles_x = W / 32; a.tiles_y = H / 4;
  a.hw = static_cast<const char*>(head_wpack); a.hpart = head_partial;
  const dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y));
  hipStream_t s = as_stream(stream);
  const int hk = head_partial ? head_k : 0;
  a.att = LpAttFuse{nullptr, nullptr, nullptr, 0, 0, 0};
  if (att) {
    // the attended tensor is G_SR_NET_low's 32-channel h_code: a 64 -> 64 (GLU: 32) upBlock whose image is written
    if (!out || Cin != 64 || (hk != 0 && hk != 3) || att->T < 1 || att->T > 32 || att->coff % 4 != 0 ||
        att->coff + 32 > out_cpitch || (att->coff < out_coff + 32 && out_coff < att->coff + 32))
      return TGSR_EUNSUPPORTED;
    a.att = *att;
    if (dtype == TGSR_DT_BF16) {
      if (hk == 0) launch_lp_upconv_att<BF16, 0>(a, grid, s);
      else launch_lp_upconv_att<BF16, 3>(a, grid, s);
    } else {
      if (hk == 0) launch_lp_upconv_att<F16, 0>(a, grid, s);
      else launch_lp_upconv_att<F16, 3>(a, grid, s);
    }
    return note_launch(hipGetLastError(), "lp_upconv_glu_kernel");
  }
  if (dtype == TGSR_DT_BF16) {
    if (hk == 0) launch_lp_upconv<BF16, 0>(a, Cin, grid, s);
    else if (hk == 3) launch_lp_upconv<BF16, 3>(a, Cin, grid, s);
    else launch_lp_upconv<BF16, 5>(a, Cin, grid, s);
  } else {
    if (hk == 0) launch_lp_upconv<F16, 0>(a, Cin, grid, s);
    else if (hk == 3) launch_lp_upconv<F16, 3>(a, Cin, grid, s);
    else launch_lp_upconv<F16, 5>(a, Cin, grid, s);
  }
  return note_launch(hipGetLastError(), "lp_upconv_glu_kernel");
}

extern "C" int tgsr_lp_upconv_glu_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack,
                                      int Cout, const float* scale, const float* shift, void* out, int out_cpitch,
                                      int out_coff, void* stream) {
  if (!out) return TGSR_EINVAL;
  return lp_upconv_launch(dtype, x, x_cpitch, B, Cin, H, W, wpack, Cout, scale, shift, out, out_cpitch, out_coff, nullptr, 0,
                          nullptr, stream);
}

extern "C" int64_t tgsr_lp_head_partial_elems(int B, int H, int W, int K) {
  const int P = K / 2;                                   // H, W: size of the head's image (the upBlock's OUTPUT)
  return (int64_t)B * (H / 8) * (W / 64) * 3 * (8 + 2 * P) * (64 + 2 * P);
}

extern "C" int tgsr_lp_upconv_glu_head_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W,
                                           const void* wpack, int Cout, const float* scale, const float* shift, void* out,
                                           int out_cpitch, int out_coff, const void* head_wpack, int head_k,
                                           float* head_partial, void* stream) {
  if (!head_partial) return TGSR_EINVAL;
  return lp_upconv_launch(dtype, x, x_cpitch, B, Cin, H, W, wpack, Cout, scale, shift, out, out_cpitch, out_coff, head_wpack,
                          head_k, head_partial, stream);
}

extern "C" int tgsr_lp_upconv_glu_att_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W,
                                          const void* wpack, int Cout, const float* scale, const float* shift, void* out,
                                          int out_cpitch, int out_coff, const void* head_wpack, int head_k,
                                          float* head_partial, const void* att_pack, int att_nsets, int att_set, int use_mask,
                                          int mask_mode, int T, int c_coff, float* attn, void* stream) {
  LpAttFuse f;
  const int rc = lp_att_fuse(att_pack, att_nsets, att_set, B, use_mask, mask_mode, T, c_coff, attn, &f);
  if (rc) return rc;
  return lp_upconv_launch(dtype, x, x_cpitch, B, Cin, H, W, wpack, Cout, scale, shift, out, out_cpitch, out_coff,
                          head_partial ? head_wpack : nullptr, head_k, head_partial, stream, &f);
}

extern "C" int tgsr_lp_head_combine(int nscales, int B, const int* H, const int* W, const float* const* partial_low,
                                    const float* const* partial_high, float* const* low, float* const* high, int low_tanh,
                                    float alpha, void* stream) {
  if (nscales < 1 || nscales > 4 || B < 1 || !H || !W || !partial_low || !partial_high || !low || !high) return TGSR_EINVAL;
  LpCombineArgs a;
  a.nscales = nscales; a.B = B; a.low_tanh = low_tanh; a.alpha = alpha;
  int64_t total = 0;
  for (int k = 0; k < nscales; ++k) {
    if (H[k] < 8 || W[k] < 64 || H[k] % 8 != 0 || W[k] % 64 != 0) return TGSR_EUNSUPPORTED;
    if (!low[k] || (partial_high[k] && !high[k])) return TGSR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(low[k]) & 15) || (reinterpret_cast<uintptr_t>(high[k]) & 15)) return TGSR_EUNSUPPORTED;
    a.s[k].pl = partial_low[k]; a.s[k].ph = partial_high[k]; a.s[k].low = low[k]; a.s[k].high = high[k];
    a.s[k].H = H[k]; a.s[k].W = W[k]; a.s[k].tiles_x = W[k] / 64; a.s[k].tiles_y = H[k] / 8; a.s[k].first = total;
    total += (int64_t)B * H[k] * W[k];
  }
  for (int k = nscales; k < 4; ++k) a.s[k] = a.s[0];
  a.total = total;
  if (total >= (1ll << 31) * 256) return TGSR_EUNSUPPORTED;
  hipLaunchKernelGGL(lp_head_combine_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "lp_head_combine_kernel");
}

extern "C" int tgsr_lp_from_nchw(int dtype, const float* x, void* out, int B, int C, int H, int W, int cpitch, int coff,
                                 void* stream) {
  if (!x || !out || B < 1 || C < 1 || H < 1 || W < 1 || coff < 0 || coff + C > cpitch) return TGSR_EINVAL;
  const int64_t total = (int64_t)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  unsigned short* o = static_cast<unsigned short*>(out);
  if (dtype == TGSR_DT_BF16)
    hipLaunchKernelGGL(lp_from_nchw_kernel<BF16>, dim3(blocks), dim3(256), 0, as_stream(stream), x, o, B, C, H, W, cpitch, coff);
  else if (dtype == TGSR_DT_F16)
    hipLaunchKernelGGL(lp_from_nchw_kernel<F16>, dim3(blocks), dim3(256), 0, as_stream(stream), x, o, B, C, H, W, cpitch, coff);
  else
    return TGSR_EINVAL;
  return note_launch(hipGetLastError(), "lp_from_nchw_kernel");
}

extern "C" int tgsr_lp_convert(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n_elems, void* stream) {
  if (!src || !dst || n_elems < 8 || (n_elems & 7) || src_dtype == dst_dtype) return TGSR_EINVAL;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0) return TGSR_EINVAL;
  const int64_t n8 = n_elems >> 3;
  const int blocks = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
  const u32x4* sp = static_cast<const u32x4*>(src);
  u32x4* dp = static_cast<u32x4*>(dst);
  if (src_dtype == TGSR_DT_F16 && dst_dtype == TGSR_DT_BF16)
    hipLaunchKernelGGL((lp_convert_kernel<F16, BF16>), dim3(blocks), dim3(256), 0, as_stream(stream), sp, dp, n8);
  else if (src_dtype == TGSR_DT_BF16 && dst_dtype == TGSR_DT_F16)
    hipLaunchKernelGGL((lp_convert_kernel<BF16, F16>), dim3(blocks), dim3(256), 0, as_stream(stream), sp, dp, n8);
  else
    return TGSR_EINVAL;
  return note_launch(hipGetLastError(), "lp_convert_kernel");
}

extern "C" int tgsr_lp_to_nchw(int dtype, const void* x, float* out, int B, int C, int H, int W, int cpitch, int coff,
                               void* stream) {
  if (!x || !out || B < 1 || C < 1 || H < 1 || W < 1 || coff < 0 || coff + C > cpitch) return TGSR_EINVAL;
  const int64_t total = (int64_t)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  const unsigned short* i = static_cast<const unsigned short*>(x);
  if (dtype == TGSR_DT_BF16)
    hipLaunchKernelGGL(lp_to_nchw_kernel<BF16>, dim3(blocks), dim3(256), 0, as_stream(stream), i, out, B, C, H, W, cpitch, coff);
  else if (dtype == TGSR_DT_F16)
    hipLaunchKernelGGL(lp_to_nchw_kernel<F16>, dim3(blocks), dim3(256), 0, as_stream(stream), i, out, B, C, H, W, cpitch, coff);
  else
    return TGSR_EINVAL;
  return note_launch(hipGetLastError(), "lp_to_nchw_kernel");
}
