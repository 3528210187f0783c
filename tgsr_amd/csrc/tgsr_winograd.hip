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

namespace tgsr {

typedef __attribute__((address_space(3))) void* lds_ptrw_t;
typedef const __attribute__((address_space(1))) void* glb_ptrw_t;
__device__ __attribute__((aligned(16))) float g_wino_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct WinoArgs {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;
  const float* upack;     // [stage][pos 16][ci 8][Cout]
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
// 4 waves of a workgroup take 4 consecutive tile rows (8 x 32 outputs) and share the transformed weights:
//   stage = 8 input channels.  U [16 pos][8 ci][64 couts] (32 KB, LDS-DMA from the pre-transformed pack, double
//   buffered; rows rotated by 16 columns on odd ci so the two channel rows of a half-wave hit disjoint banks).
//   Raw input: each wave DMAs ITS OWN 4 halo rows [8 ci][4][40] (16-byte pieces, the tile starts 4 columns left of
//   the outputs) and transforms them into its private V image [16 pos][8 ci][16 tiles] (double buffered) - so the
//   only workgroup barrier per stage is the one that publishes U.
// With one wave per SIMD nothing hides a stall, so everything that is not an MFMA is placed INSIDE the MFMA stream
// of a stage (32 k-steps of 4 MFMAs = 128 cycles each): the operand fragments of step s+1, the DMA pieces of stage
// st+1 (steps 0-12), and - after a counted vmcnt wait for the raw pieces - the input transform of stage st+1 in 8
// slices (steps 16-23).
constexpr int kWCK = 8;                                  // input channels per stage
constexpr int kWTC = 40, kWPLANE = 4 * kWTC;             // per-wave raw rows: 4 x (32 + 8) columns per channel
constexpr int kWU = 16 * kWCK * 64;                      // floats of U per stage: 8192
constexpr int kWUK = kWU / 256 / 4;                      // U DMA pieces (1 KB) per wave per stage: 8
constexpr int kWRaw = kWCK * kWPLANE;                    // 1280 floats of raw input per wave per stage
constexpr int kWIK = kWRaw / 256;                        // raw DMA pieces (1 KB) per wave per stage: 5
constexpr int kWV = 16 * kWCK * 16;                      // V image: [pos 16][ci 8][16 tiles]
constexpr int kWSmem = 2 * kWU + 4 * kWRaw + 4 * 2 * kWV + 2 * 64;

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
  float* raw = smem + 2 * kWU + wave * kWRaw;            // this wave's raw rows
  float* vs = smem + 2 * kWU + 4 * kWRaw + wave * 2 * kWV;   // this wave's 2 V images
  float* aff_s = smem + 2 * kWU + 4 * kWRaw + 4 * 2 * kWV;

  auto gcol = [&](int lc) {                              // logical column (0..63) of this workgroup -> global cout
    if (GLU) return (lc < 32 ? grp * 32 : (a.Cout >> 1) + grp * 32 - 32) + lc;
    return grp * 64 + lc;
  };

  // ---- DMA plan.  U: LDS row r = pos*8 + ci holds logical column (cl - 16*(r&1)) mod 64 at position cl.
  int uoff[kWUK], ioff[kWIK];
#pragma unroll
  for (int k = 0; k < kWUK; ++k) {
    const int q = (wave + 4 * k) * 64 + lane;            // float4 index in the stage's weight block
    const int row = q >> 4, cl = (q & 15) * 4;
    uoff[k] = row * a.Cout + gcol((cl - 16 * (row & 1)) & 63);
  }
#pragma unroll
  for (int k = 0; k < kWIK; ++k) {
    const int e = (k * 64 + lane) * 4;                   // first float of this lane's 16-byte piece (wave private)
    const int c = e / kWPLANE;
    const int rem = e - c * kWPLANE;
    const int r = rem / kWTC, j = rem - r * kWTC;
    const int gy = y0 + 2 * wave - 1 + r, gx = x0 - 4 + j;
    const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   // W % 4 == 0: whole piece in/out
    ioff[k] = ok ? ((c << 28) | (gy * a.W + gx)) : -1;
  }
  auto issue_raw = [&](int k, int st) {
    const int v = ioff[k];
    const int c = st * kWCK + (v >> 28);
    const bool ok = v >= 0 && c < a.Cin;
    const float* g = ok ? xb + (uint64_t)(uint32_t)c * HW + (uint32_t)(v & 0x0fffffff) : g_wino_zero;
    __builtin_amdgcn_global_load_lds((glb_ptrw_t)g, (lds_ptrw_t)(raw + k * 256), 16, 0, 0);
  };
  auto issue_u = [&](int k, int st) {
    const float* g = a.upack + (int64_t)st * (kWU / 64) * a.Cout + uoff[k];
    __builtin_amdgcn_global_load_lds((glb_ptrw_t)g, (lds_ptrw_t)(us + (st & 1) * kWU + (wave + 4 * k) * 256), 16, 0, 0);
  };
  // one eighth of the input transform V = B^T d B of this lane's tile: channel 2*lg + (j >> 2), row i = j & 3 of V
  const int praw = 2 * l15 + 3;                          // left column of the 4x4 patch inside a raw row
  auto transform_slice = [&](int j, float* vdst) {
    const int c = 2 * lg + (j >> 2), i = j & 3;
    const float* rp = raw + c * kWPLANE + praw;
    const int ra = i == 0 ? 0 : (i == 3 ? 1 : (i == 1 ? 1 : 2)), rb2 = i == 0 ? 2 : (i == 3 ? 3 : (i == 1 ? 2 : 1));
    float tr[4];                                         // B^T row i: d0-d2 | d1+d2 | d2-d1 | d1-d3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float u = rp[ra * kWTC + q], w = rp[rb2 * kWTC + q];
      tr[q] = i == 1 ? u + w : u - w;
    }
    float* vp = vdst + c * 16 + l15 + (i * 4) * 128;     // V[pos = i*4 + jj][c][tile], pos stride 8 * 16
    vp[0 * 128] = tr[0] - tr[2];
    vp[1 * 128] = tr[1] + tr[2];
    vp[2 * 128] = tr[2] - tr[1];
    vp[3 * 128] = tr[1] - tr[3];
  };

  if (tid < 64) {
    const int col = gcol(tid);
    aff_s[tid] = a.scale ? a.scale[col] : 1.f;
    aff_s[64 + tid] = a.scale ? a.shift[col] : 0.f;
  }

  f32x4v M[16][4];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) M[p][cb][i] = 0.f;

  // ---- prologue: stage 0 raw + U, transform into V[0]
#pragma unroll
  for (int k = 0; k < kWIK; ++k) issue_raw(k, 0);
#pragma unroll
  for (int k = 0; k < kWUK; ++k) issue_u(k, 0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWUK) : "memory");     // the raw pieces (issued first) have landed
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) transform_slice(j, vs);
  __syncthreads();                                                 // vmcnt(0) + barrier: U(0) visible to all waves

  // lane-constant parts of every LDS address (the per-step parts are immediates)
  const int rotl = 16 * (lg & 1);
  int ucol[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) ucol[cb] = lg * 64 + ((cb * 16 + l15 + rotl) & 63);
  const int vlane = lg * 16 + l15;

  for (int st = 0; st < a.nstages; ++st) {
    const float* cur = us + (st & 1) * kWU;
    const float* vb = vs + (st & 1) * kWV + vlane;           // + (pos*8 + 4*ks) * 16
    const float* ub0 = cur + ucol[0];                        // + (pos*8 + 4*ks) * 64
    const float* ub1 = cur + ucol[1];
    const float* ub2 = cur + ucol[2];
    const float* ub3 = cur + ucol[3];
    float* vnxt = vs + ((st + 1) & 1) * kWV;
    const bool more = st + 1 < a.nstages;
    float bA, aA[4], bB, aB[4];
    bA = vb[0];
    aA[0] = ub0[0]; aA[1] = ub1[0]; aA[2] = ub2[0]; aA[3] = ub3[0];
    // 32 k-steps of 4 MFMAs.  One wave per SIMD: an instruction issued between two MFMAs runs under the first one's
    // 32 pipe cycles, so the non-MFMA work of a step is spread over its four gaps:
    //   gap 0: B fragment + 2 A fragments of step s+1     gap 1: the other 2 A fragments
    //   gap 2: one DMA piece of stage st+1 (steps 0-12) | raw wait (step 21) | transform slice (steps 22-29)
#pragma unroll
    for (int s2 = 0; s2 < 32; s2 += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int sidx = s2 + half, p = sidx >> 1;
        float& bv = half == 0 ? bA : bB;
        float(&av)[4] = half == 0 ? aA : aB;
        float& bn = half == 0 ? bB : bA;
        float(&an)[4] = half == 0 ? aB : aA;
        const int n1 = sidx + 1;
        const int o1 = ((n1 >> 1) * kWCK + 4 * (n1 & 1));     // row offset of step s+1 (without the lane part)
        M[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv, M[p][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (n1 < 32) { bn = vb[o1 * 16]; an[0] = ub0[o1 * 64]; an[1] = ub1[o1 * 64]; }
        __builtin_amdgcn_sched_barrier(0);
        M[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv, M[p][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (n1 < 32) { an[2] = ub2[o1 * 64]; an[3] = ub3[o1 * 64]; }
        __builtin_amdgcn_sched_barrier(0);
        M[p][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv, M[p][2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          if (sidx < kWIK) issue_raw(sidx, st + 1);                           // steps 0-4
          else if (sidx < kWIK + kWUK) issue_u(sidx - kWIK, st + 1);          // steps 5-12
          else if (sidx == 21) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWUK) : "memory");   // raw(st+1) landed
          else if (sidx >= 22 && sidx < 30) transform_slice(sidx - 22, vnxt); // steps 22-29
        }
        __builtin_amdgcn_sched_barrier(0);
        M[p][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv, M[p][3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // U(st+1) landed (vmcnt(0)) and visible; U(st) / V(st) may be overwritten
  }

  // ---- output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) + epilogue; lane = tile (l15), registers of
  // block cb = couts cb*16 + 4*lg + i; the two column phases of a tile leave as one float2
  const int oy = y0 + 2 * wave, ox = x0 + 2 * l15;
  const int64_t HWo = (int64_t)a.H * a.W;
  float* __restrict__ ob = a.out + (int64_t)b * a.obs;
  const float* __restrict__ rb = a.res ? a.res + (int64_t)b * a.rbs : nullptr;
  auto yout = [&](int cb, int i, int dy, int dx) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float cr = dy == 0 ? (r < 3 ? 1.f : 0.f) : (r == 0 ? 0.f : (r == 1 ? 1.f : -1.f));
      if (cr == 0.f) continue;
      const float m0 = M[r * 4 + 0][cb][i], m1 = M[r * 4 + 1][cb][i], m2 = M[r * 4 + 2][cb][i], m3 = M[r * 4 + 3][cb][i];
      const float red = dx == 0 ? (m0 + m1 + m2) : (m1 - m2 - m3);
      acc += cr * red;
    }
    return acc;
  };
  if (ox < a.W) {
#pragma unroll
    for (int cb = 0; cb < (GLU ? 2 : 4); ++cb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int lc = cb * 16 + 4 * lg + i;             // logical column of the value (or plain) channel
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          if (oy + dy >= a.H) continue;
          float o0, o1;
          int c;
          if (GLU) {
            const float sv = aff_s[lc], tv = aff_s[64 + lc], sg = aff_s[32 + lc], tg = aff_s[64 + 32 + lc];
            const float g0 = yout(cb + 2, i, dy, 0) * sg + tg, g1 = yout(cb + 2, i, dy, 1) * sg + tg;
            o0 = (yout(cb, i, dy, 0) * sv + tv) * (1.f / (1.f + __expf(-g0)));
            o1 = (yout(cb, i, dy, 1) * sv + tv) * (1.f / (1.f + __expf(-g1)));
            c = grp * 32 + lc;
          } else {
            const float sv = aff_s[lc], tv = aff_s[64 + lc];
            o0 = yout(cb, i, dy, 0) * sv + tv;
            o1 = yout(cb, i, dy, 1) * sv + tv;
            c = grp * 64 + lc;
          }
          const int64_t o = (int64_t)c * HWo + (int64_t)(oy + dy) * a.W + ox;
          if (!GLU && rb) {
            const float2 r = *reinterpret_cast<const float2*>(rb + o);
            o0 += r.x;
            o1 += r.y;
          }
          *reinterpret_cast<float2*>(ob + o) = make_float2(o0, o1);
        }
      }
    }
  }
}

// upack[stage][pos 16][ci 8][Cout] <- U = G g G^T, pos = i * 4 + j of the 4x4 transformed filter
__global__ void pack_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin,
                                        int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(idx % Cout);
    int64_t t = idx / Cout;
    const int ci = (int)(t % kWCK);
    t /= kWCK;
    const int pos = (int)(t % 16);
    const int st = (int)(t / 16);
    const int c = st * kWCK + ci;
    float u = 0.f;
    if (c < Cin) {
      const int i = pos >> 2, j = pos & 3;
      const float* g = w + ((int64_t)co * Cin + c) * 9;
      float gi[3];   // row i of G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] applied to the filter rows
      for (int q = 0; q < 3; ++q) {
        const float g0 = g[0 * 3 + q], g1 = g[1 * 3 + q], g2 = g[2 * 3 + q];
        gi[q] = i == 0 ? g0 : (i == 1 ? 0.5f * (g0 + g1 + g2) : (i == 2 ? 0.5f * (g0 - g1 + g2) : g2));
      }
      u = j == 0 ? gi[0] : (j == 1 ? 0.5f * (gi[0] + gi[1] + gi[2]) : (j == 2 ? 0.5f * (gi[0] - gi[1] + gi[2]) : gi[2]));
    }
    up[idx] = u;
  }
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_packed_wino_weight_elems(int Cout, int Cin) {
  return (int64_t)((Cin + kWCK - 1) / kWCK) * 16 * kWCK * Cout;
}

extern "C" int tgsr_pack_wino_weight(const float* w, float* upack, int Cout, int Cin, void* stream) {
  if (!w || !upack || Cout < 1 || Cin < 1) return TGSR_EINVAL;
  const int64_t total = tgsr_packed_wino_weight_elems(Cout, Cin);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_wino_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, upack, Cout, Cin,
                     total);
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
  if (Cout % 64 != 0) return TGSR_EUNSUPPORTED;
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
