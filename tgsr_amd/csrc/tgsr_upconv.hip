// upBlock (util.py:74-80: Upsample(x2, nearest) -> conv3x3 -> BatchNorm2d -> GLU) as FOUR 2x2 convolutions on the
// pre-upsample tensor (sub-pixel decomposition), fp32 on the MFMA units, inference (folded BN) form.
//
// With u[Y][X] = x[Y >> 1][X >> 1], output pixel (2y + a, 2x + b) of the 3x3 convolution only ever touches the 2x2
// source neighbourhood rows {y - 1 + a, y + a} x cols {x - 1 + b, x + b}: the taps that land on the same source pixel
// are pre-summed (tgsr_pack_upconv_weight):
//     a = 0: row y-1 <- w[0][.],          row y   <- w[1][.] + w[2][.]
//     a = 1: row y   <- w[0][.] + w[1][.], row y+1 <- w[2][.]          (same for columns / b)
// so each output needs 4 * Cin MACs instead of 9 * Cin - 2.25x fewer MFMAs than folding the up-sample into the tile
// read of the generic kernel (tgsr_conv3x3_fwd, upsample = 1), identical up to the rounding of the weight sums.
// The upBlocks are 41 % of the conv time of one SR forward (the 256^2 one alone 23 % of all FLOPs).
//
// Kernel: workgroup = 4 waves = 4 source rows x 32 source columns (8 x 64 outputs) x one GLU channel block (32 value +
// 32 gate channels); wave = 1 source row, 8 accumulators (4 phases x {value, gate}).  Per k-step (2 input channels)
// the 9 shifted input fragments are read once and feed the 16 (phase, tap) weight fragments.  Stages of 4 input
// channels are double buffered in LDS by LDS-DMA exactly like tgsr_conv3x3.hip; the two column phases of a lane are
// stored together as one 8-byte store (fully coalesced rows of the NCHW output plane).
#include "tgsr_common.h"

namespace tgsr {

typedef __attribute__((address_space(3))) void* lds_ptr2_t;
typedef const __attribute__((address_space(1))) void* glb_ptr2_t;
__device__ __attribute__((aligned(16))) float g_up_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct UpArgs {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;       // source (pre-upsample) dims
  const float* wpack;     // [chunk][phase 4][tap 4][ci 4][Cout]
  int Cout;
  const float* scale;
  const float* shift;
  float* out;             // [B][Cout/2][2H][2W]
  int64_t obs;
  int tiles_x, tiles_y, nchunks;
};

constexpr int kUpWV = 4;                         // waves per workgroup = source rows per tile (two workgroups per CU:
                                                 // one 8-wave workgroup stalls both waves of a SIMD at the same barrier;
                                                 // a persistent grid and 8-channel stages were measured too: no gain)
constexpr int kUpNCOL = 64;                      // 32 value + 32 gate columns
constexpr int kUpTR = kUpWV + 2, kUpTC = 34, kUpPLANE = kUpTR * kUpTC;
constexpr int kUpW = 16 * kConvCK * kUpNCOL;     // floats of weights per stage (4096)
constexpr int kUpWUnits = kUpW / 256;            // 16
constexpr int kUpIn = kConvCK * kUpPLANE;        // 1360
constexpr int kUpInUnits = (kUpIn + 63) / 64;    // 22
constexpr int kUpBuf = kUpW + kUpInUnits * 64;
constexpr int kUpWK = (kUpWUnits + kUpWV - 1) / kUpWV, kUpIK = (kUpInUnits + kUpWV - 1) / kUpWV;

__global__ __launch_bounds__(64 * kUpWV, 2) void upconv_glu_mfma_kernel(UpArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2 * kUpBuf + 2 * kUpNCOL];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int grp = blockIdx.y;
  const int ys = ty * kUpWV, xs = tx * 32;
  const float* xb = a.x + (int64_t)b * a.xbs;
  const uint32_t HW = (uint32_t)a.H * (uint32_t)a.W;

  // ---- DMA plan (per-lane source offsets, stage independent)
  int woff[kUpWK], ioff[kUpIK];
#pragma unroll
  for (int k = 0; k < kUpWK; ++k) {
    const int q = (wave + kUpWV * k) * 64 + lane;          // float4 index in the stage's weight block
    const int row = q / (kUpNCOL / 4);                     // (phase*4 + tap)*4 + ci
    const int c4 = q - row * (kUpNCOL / 4);
    const int seg = c4 >> 3, f4 = c4 & 7;
    const int col = (seg == 0 ? grp * 32 : (a.Cout >> 1) + grp * 32) + f4 * 4;
    woff[k] = row * a.Cout + col;
  }
#pragma unroll
  for (int k = 0; k < kUpIK; ++k) {
    const int idx = (wave + kUpWV * k) * 64 + lane;
    const int c = idx / kUpPLANE;
    const int rem = idx - c * kUpPLANE;
    const int r = rem / kUpTC, cc = rem - r * kUpTC;
    const int gy = ys - 1 + r, gx = xs - 1 + cc;
    const bool ok = c < kConvCK && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    ioff[k] = ok ? ((c << 28) | (gy * a.W + gx)) : -1;
  }
  auto issue = [&](int p, float* buf, int ch) {
    if (p < kUpWK) {
      const int u = wave + kUpWV * p;
      if (u < kUpWUnits) {
        const float* g = a.wpack + (int64_t)ch * 16 * kConvCK * a.Cout + woff[p];
        __builtin_amdgcn_global_load_lds((glb_ptr2_t)g, (lds_ptr2_t)(buf + u * 256), 16, 0, 0);
      }
    } else {
      const int k = p - kUpWK;
      const int u = wave + kUpWV * k;
      if (u < kUpInUnits) {
        const int v = ioff[k];
        const int c = ch * kConvCK + (v >> 28);
        const bool ok = v >= 0 && c < a.Cin;
        const float* g = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(v & 0x0fffffff) : g_up_zero;
        __builtin_amdgcn_global_load_lds((glb_ptr2_t)g, (lds_ptr2_t)(buf + kUpW + u * 64), 4, 0, 0);
      }
    }
  };
  constexpr int NP = kUpWK + kUpIK;
  static_assert(NP <= 8, "one DMA piece per (channel pair, phase) step must cover a stage");

#pragma unroll
  for (int p = 0; p < NP; ++p) issue(p, smem, 0);
  float* aff_s = smem + 2 * kUpBuf;
  if (tid < kUpNCOL) {
    const int col = (tid < 32 ? grp * 32 : (a.Cout >> 1) + grp * 32) + (tid & 31);
    aff_s[tid] = a.scale ? a.scale[col] : 1.f;
    aff_s[kUpNCOL + tid] = a.scale ? a.shift[col] : 0.f;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ph][cb][i] = 0.f;

  const int a_off = h * kUpNCOL + l31;
  const int b_off = kUpW + h * kUpPLANE + wave * kUpTC + l31;
  __syncthreads();

  for (int ch = 0; ch < a.nchunks; ++ch) {
    const float* cur = smem + (ch & 1) * kUpBuf;
    float* nxt = smem + ((ch + 1) & 1) * kUpBuf;
    const bool more = ch + 1 < a.nchunks;
#pragma unroll
    for (int kk = 0; kk < kConvCK / 2; ++kk) {
      float bf[3][3];
#pragma unroll
      for (int ro = 0; ro < 3; ++ro)
#pragma unroll
        for (int co = 0; co < 3; ++co) bf[ro][co] = cur[b_off + 2 * kk * kUpPLANE + ro * kUpTC + co];
      float av[4][2];
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) av[tp][cb] = cur[a_off + ((0 * 4 + tp) * kConvCK + 2 * kk) * kUpNCOL + cb * 32];
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int pa = ph >> 1, pb = ph & 1;
        float an[4][2];
        // first half of this phase's 8 MFMAs, then (pinned) the next phase's weight fragments + one DMA piece
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[ph][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tp][cb], bf[pa + (tp >> 1)][pb + (tp & 1)], acc[ph][cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more && kk * 4 + ph < NP) issue(kk * 4 + ph, nxt, ch + 1);
        if (ph + 1 < 4) {
#pragma unroll
          for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              an[tp][cb] = cur[a_off + (((ph + 1) * 4 + tp) * kConvCK + 2 * kk) * kUpNCOL + cb * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tp = 2; tp < 4; ++tp)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[ph][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tp][cb], bf[pa + (tp >> 1)][pb + (tp & 1)], acc[ph][cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ph + 1 < 4) {
#pragma unroll
          for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) av[tp][cb] = an[tp][cb];
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: affine + GLU; lane = source column x, the two column phases go out as one float2
  const int y = ys + wave, x = xs + l31;
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const int64_t HWo = (int64_t)Ho * Wo;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  if (y < a.H && x < a.W) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lc = acc_row(i, h);
      const float sv = aff_s[lc], tv = aff_s[kUpNCOL + lc], sg = aff_s[32 + lc], tg = aff_s[kUpNCOL + 32 + lc];
      const int c = grp * 32 + lc;
#pragma unroll
      for (int pa = 0; pa < 2; ++pa) {
        float o2[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const float v = acc[pa * 2 + pb][0][i] * sv + tv;
          const float g = acc[pa * 2 + pb][1][i] * sg + tg;
          o2[pb] = v * (1.f / (1.f + __expf(-g)));
        }
        *reinterpret_cast<float2*>(ob + (int64_t)c * HWo + (int64_t)(2 * y + pa) * Wo + 2 * x) = make_float2(o2[0], o2[1]);
      }
    }
  }
}

// wu[chunk][phase][tap][ci][Cout] <- w[Cout][Cin][3][3] with the row/column tap sums of the header comment
__global__ void pack_upconv_weight_kernel(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                          int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    int64_t t = i / Cout;
    const int ci = (int)(t % kConvCK);
    t /= kConvCK;
    const int tp = (int)(t % 4);
    t /= 4;
    const int ph = (int)(t % 4);
    const int chunk = (int)(t / 4);
    const int c = chunk * kConvCK + ci;
    float s = 0.f;
    if (c < Cin) {
      const int pa = ph >> 1, pb = ph & 1, dy = tp >> 1, dx = tp & 1;
      // rows: a=0: dy=0 -> {0}, dy=1 -> {1,2};  a=1: dy=0 -> {0,1}, dy=1 -> {2}
      const int ky0 = pa == 0 ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), ky1 = pa == 0 ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
      const int kx0 = pb == 0 ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), kx1 = pb == 0 ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
      const float* wc = w + ((int64_t)co * Cin + c) * 9;
      for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) s += wc[ky * 3 + kx];
    }
    wu[i] = s;
  }
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_packed_upconv_weight_elems(int Cout, int Cin) {
  return (int64_t)((Cin + kConvCK - 1) / kConvCK) * 16 * kConvCK * Cout;
}

extern "C" int tgsr_pack_upconv_weight(const float* w, float* wpack, int Cout, int Cin, void* stream) {
  if (!w || !wpack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  const int64_t total = tgsr_packed_upconv_weight_elems(Cout, Cin);
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(pack_upconv_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, wpack, Cout, Cin,
                     total);
  return note_launch(hipGetLastError(), "pack_upconv_weight_kernel");
}

extern "C" int tgsr_upconv3x3_glu_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W,
                                      const float* wpack, int Cout, const float* scale, const float* shift, float* out,
                                      int64_t out_bstride, void* stream) {
  if (!x || !wpack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  if (Cout % 64 != 0) return TGSR_EUNSUPPORTED;
  if ((int64_t)H * W >= (1 << 28) || (int64_t)Cin * H * W >= (1ll << 32)) return TGSR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(out) & 7) != 0 || (out_bstride & 1) != 0) return TGSR_EUNSUPPORTED;
  UpArgs a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.wpack = wpack; a.Cout = Cout;
  a.scale = scale; a.shift = shift; a.out = out; a.obs = out_bstride;
  a.tiles_x = (W + 31) / 32; a.tiles_y = (H + kUpWV - 1) / kUpWV; a.nchunks = (Cin + kConvCK - 1) / kConvCK;
  dim3 grid((unsigned)(B * a.tiles_x * a.tiles_y), (unsigned)(Cout / 64));
  hipLaunchKernelGGL(upconv_glu_mfma_kernel, grid, dim3(64 * kUpWV), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "upconv_glu_mfma_kernel");
}
