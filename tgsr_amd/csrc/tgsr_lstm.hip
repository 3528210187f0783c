// Bidirectional LSTM text encoder (RNN_ENCODER.forward, util.py:233-260, eval mode) for gfx950.
//
// Two launches instead of nn.Embedding + pack_padded_sequence + cuDNN/MIOpen LSTM + pad_packed_sequence + transpose:
//   1. lstm_input_gates_kernel: the input projection of EVERY (sample, step, direction) at once as one fp32 MFMA
//      GEMM  G[b*T+t][d*4H+j] = emb[captions[b][t]] . w_ih[d][j] + b_ih[d][j] + b_hh[d][j]   (embedding gather
//      fused into the A-tile load; M = B*Tmax, N = 8H, K = ninput).
//   2. lstm_recurrent_kernel: one workgroup per (sample, direction), one thread per gate row with its w_hh row held
//      in registers for the whole sequence (H <= 128: 128 VGPRs), h broadcast from LDS; per step one mat-vec,
//      two barriers.  Packed-sequence semantics are explicit: each direction walks only the sample's own `len`
//      tokens, outputs past `len` are zero, the sentence code is [h_fwd(len-1), h_bwd(0)].
#include "tgsr_common.h"

namespace tgsr {

#ifdef TGSR_LSTM_STAMPS
// diagnostic build (tools/lstm_stamps.py): s_memtime of thread 0 of every workgroup at the phases of every step
__device__ unsigned long long g_lstamps[64 * 128];
#define TGSR_LSTAMP(k)                                                                                    \
  do {                                                                                                    \
    if (threadIdx.x == 0 && (k) < 126) {                                                                  \
      const int bid_ = blockIdx.y * gridDim.x + blockIdx.x;                                               \
      if (bid_ < 64) {                                                                                    \
        g_lstamps[bid_ * 128 + (k)] = __builtin_amdgcn_s_memtime();                                       \
        if ((k) == 0) g_lstamps[bid_ * 128 + 126] = __builtin_amdgcn_s_memrealtime();                     \
        g_lstamps[bid_ * 128 + 127] = __builtin_amdgcn_s_memrealtime();                                   \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
#else
#define TGSR_LSTAMP(k)
#endif

// C[m][n] = sum_k A[row(m)][k] * Bm[n][k] + bias0[n] + bias1[n];  A rows gathered through idx (token ids).
// Tile: 4 waves, wave w -> columns [n0 + 32w, +32), rows [m0, m0+32).  K chunks of 64 through LDS (pitch 65).
__global__ __launch_bounds__(256) void lstm_input_gates_kernel(const int64_t* __restrict__ captions, int width,
                                                               int Tmax, const float* __restrict__ emb, int ntoken,
                                                               const float* __restrict__ w_ih,
                                                               const float* __restrict__ b_ih,
                                                               const float* __restrict__ b_hh, int M, int N, int K,
                                                               float* __restrict__ gates) {
  constexpr int KC = 64, P = KC + 1;
  __shared__ float a_s[32 * P];
  __shared__ float b_s[128 * P];
  __shared__ int tok_s[32];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5, wave = tid >> 6;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 128;
  if (tid < 32) {
    const int m = m0 + tid;
    int tok = 0;
    if (m < M) {
      const int64_t v = captions ? captions[(int64_t)(m / Tmax) * width + (m % Tmax)] : (int64_t)m;
      tok = (v < 0 || v >= ntoken) ? 0 : (int)v;
    }
    tok_s[tid] = tok;
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();
    for (int idx = tid; idx < 32 * KC; idx += 256) {
      const int r = idx >> 6, k = idx & 63;
      a_s[r * P + k] = (k0 + k < K) ? emb[(int64_t)tok_s[r] * K + k0 + k] : 0.f;
    }
    for (int idx = tid; idx < 128 * KC; idx += 256) {
      const int r = idx >> 6, k = idx & 63;
      b_s[r * P + k] = (k0 + k < K && n0 + r < N) ? w_ih[(int64_t)(n0 + r) * K + k0 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < KC; k += 2) {
      const float av = a_s[l31 * P + k + hh];
      const float bv = b_s[(wave * 32 + l31) * P + k + hh];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  const int n = n0 + wave * 32 + l31;
  if (n < N) {
    const float bias = b_ih[n] + b_hh[n];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + acc_row(i, hh);
      if (m < M) gates[(int64_t)m * N + n] = acc[i] + bias;
    }
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// the register kernel's step is latency, not throughput: v_exp_f32 / v_rcp_f32 forms (absolute error ~1e-7, far inside the
// 1e-5 the encoder is pinned to) instead of the ~40-instruction library expf / tanhf
__device__ __forceinline__ float fsig_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// grid (B, 2); block 4H threads.  gates [B][Tmax][2][4H]; w_hh [2][4H][H].
// gates: [B][Tmax][2][4H] per position, or (captions != nullptr) a per-TOKEN table [ntoken][2][4H] indexed through
// the caption - the eval-mode form: the input projection of a frozen encoder is a function of the token only.
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_recurrent_kernel(const float* __restrict__ gates,
                                                               const int32_t* __restrict__ cap_lens, int Tmax,
                                                               const float* __restrict__ w_hh,
                                                               float* __restrict__ words_emb,
                                                               float* __restrict__ sent_emb,
                                                               const int64_t* __restrict__ captions, int width,
                                                               int ntoken, float* __restrict__ acts) {
  __shared__ __attribute__((aligned(16))) float h_s[H];
  __shared__ float g_s[4 * H];
  const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
  int len = cap_lens[b];
  len = len < 0 ? 0 : (len > Tmax ? Tmax : len);
  float w[H];
  {
    const float4* wr = reinterpret_cast<const float4*>(w_hh + ((int64_t)d * 4 * H + j) * H);
#pragma unroll
    for (int k = 0; k < H / 4; ++k) {
      const float4 v = wr[k];
      w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
  }
  float c = 0.f, hcur = 0.f;
  if (j < H) h_s[j] = 0.f;
  float* wout = words_emb + ((int64_t)b * 2 * H + d * H + j) * Tmax;  // valid for j < H
  if (j < H)
    for (int t = len; t < Tmax; ++t) wout[t] = 0.f;
  __syncthreads();
  for (int s = 0; s < len; ++s) {
    const int t = d == 0 ? s : len - 1 - s;
    int64_t row = (int64_t)b * Tmax + t;
    if (captions) {
      const int64_t v = captions[(int64_t)b * width + t];
      row = (v < 0 || v >= ntoken) ? 0 : v;
    }
    // four independent accumulation chains (one dependent chain of H FMAs was the step's latency: 1.7 us per step at
    // H = 128), combined pairwise
    float g0 = gates[(row * 2 + d) * 4 * H + j], g1 = 0.f, g2 = 0.f, g3 = 0.f;
#pragma unroll
    for (int k = 0; k < H / 4; ++k) {
      const float4 hv = *reinterpret_cast<const float4*>(h_s + 4 * k);
      g0 = fmaf(w[4 * k], hv.x, g0);
      g1 = fmaf(w[4 * k + 1], hv.y, g1);
      g2 = fmaf(w[4 * k + 2], hv.z, g2);
      g3 = fmaf(w[4 * k + 3], hv.w, g3);
    }
    g_s[j] = (g0 + g1) + (g2 + g3);
    __syncthreads();
    if (j < H) {
      const float ig = fsig_(g_s[j]), fg = fsig_(g_s[H + j]), gg = ftanh_(g_s[2 * H + j]), og = fsig_(g_s[3 * H + j]);
      c = fg * c + ig * gg;
      hcur = og * ftanh_(c);
      h_s[j] = hcur;
      wout[t] = hcur;
      if (acts) {   // training: gate activations and cell state of this step, [B][Tmax][2][5][H]
        float* ap = acts + (((int64_t)b * Tmax + t) * 2 + d) * 5 * H + j;
        ap[0] = ig; ap[H] = fg; ap[2 * H] = gg; ap[3 * H] = og; ap[4 * H] = c;
      }
    }
    __syncthreads();
  }
  if (j < H) sent_emb[(int64_t)b * 2 * H + d * H + j] = hcur;
}

// The recurrence as shipped for H = 128: KS * H threads, thread t = (unit j = t / KS, K slice q = t % KS) owns ALL FOUR gate
// rows of unit j over 1/KS of the K = H products.  What the stamps of the one-row kernel above say about a step
// (tools/lstm_stamps.py, 2.4 GHz, 4 400 cycles): its 128 FMAs are not the cost - (a) the step's gate pre-activations were a
// global load behind a scalar caption load at the head of the FMA chain (~1.2 us of exposed latency per step), (b) every wave
// re-read all of h as 32 broadcast ds_read_b128, 1 KB of returned data each: 2 048 cycles per step on the CU's one LDS pipe,
// (c) two barriers and the g_s round trip between the mat-vec and the gate math.  Here: (a) all steps' pre-activations are
// staged in LDS during the prologue (TP steps x 4H floats, 48 KB), (b) a wave reads 1/KS of h per lane - a quarter of the LDS
// traffic, (c) the KS partial sums of a unit's four gates meet by DPP quad_perm adds between neighbouring lanes, so the gate
// math needs no LDS exchange; h is double-buffered and a step has ONE barrier.
//
// NO PACKED FP32 INSTRUCTIONS (this TU is built with -fno-slp-vectorize and the code below keeps its weights in scalar
// registers).  A first version fed v_pk_fma_f32 with (i, f) / (g, o) weight pairs and was 1-2 us faster - and gave different
// results whenever an MFMA-heavy kernel (lp_upconv_glu_kernel, the 128^2 lp convolutions) shared its CU: a v_pk_*_f32 whose
// source or destination registers are the target of a load issued right behind it (ds_read_b128 v[0:3] after
// v_pk_fma_f32 ..., v[0:1] - the register allocator reuses registers like that all the time) can see the NEW contents when
// the matrix pipe is busy with another wave's MFMAs; the hardware does not interlock it and the compiler's hazard recognizer
// does not know it.  tools/lds_neighbour_check.py reproduces it in 40 lines (packed FMAs fed from ds_read_b128 against
// v_fma_f32 on the same data: 0 mismatches alone, 40 000 beside lp_upconv_glu_kernel; with every load in registers of its own:
// 0 again), tests/test_hip_concurrency.py keeps it out.
// Per launch (B = 16, T = 18, rocprofv3): one row per thread 29.4-31.0 us; this kernel 20.2 us; with the K slices interleaved so
// that the prologue's weight loads coalesce (below) 14.7 us.  (One activation per lane + a DPP broadcast round the quad instead
// of four per lane: 15.0 - the quarter-rate transcendentals are not what a step waits for.)
__device__ __forceinline__ float lane_xor1(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}

__device__ __forceinline__ float lane_xor2(float x) {   // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, false));
}

template <int H, int TP, int KS>
__global__ __launch_bounds__(KS * H) void lstm_recurrent2_kernel(const float* __restrict__ gates,
                                                                 const int32_t* __restrict__ cap_lens, int Tmax,
                                                                 const float* __restrict__ w_hh,
                                                                 float* __restrict__ words_emb, float* __restrict__ sent_emb,
                                                                 const int64_t* __restrict__ captions, int width, int ntoken,
                                                                 float* __restrict__ acts) {
  static_assert(KS == 2 || KS == 4, "K split over 2 or 4 neighbouring lanes");
  constexpr int HK = H / KS, NT = KS * H;
  __shared__ __attribute__((aligned(16))) float h_s[2][H];
  __shared__ float gpre_s[TP * 4 * H];                         // [step s][4H], in the order the recurrence consumes them
  __shared__ int row_s[TP];
  const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
  const int j = tid / KS, q = tid & (KS - 1);
  TGSR_LSTAMP(0);
  int len = cap_lens[b];
  len = len < 0 ? 0 : (len > Tmax ? Tmax : len);
  // Load order of the prologue: the caption tokens (one per thread, the head of a dependent chain), then the 32 float4 of weights
  // per thread, which stay in flight while the chain continues - caption -> gate-table rows staged in LDS.  The prologue is
  // 7.7 us of the kernel's 20 whatever the order (stamps): what it waited for was the weight loads' address pattern - a lane's 128 bytes
  // of a row are 128 bytes away from its neighbour's, so every one of the 32 load instructions touches 64 cache lines.  A
  // per-weight-version pack in lane order (like the gate table) would make them 1-KB runs; short of that the K slices are
  // interleaved (below): the four lanes of a unit read 64 contiguous bytes per instruction - 16 lines instead of 64.
  int64_t row_of_step = 0;
  if (tid < len) {                                             // the gate-table row of every step (caption token, or position)
    const int t = d == 0 ? tid : len - 1 - tid;
    row_of_step = (int64_t)b * Tmax + t;
    if (captions) {
      const int64_t v = captions[(int64_t)b * width + t];
      row_of_step = (v < 0 || v >= ntoken) ? 0 : v;
    }
  }
  // SCALAR registers and v_fma_f32 on purpose - see the note on packed fp32 instructions above
  float wi[HK], wf[HK], wg[HK], wo[HK];                        // rows i, f, g, o of unit j, this thread's K slice
  {
    // K slice of lane q = the float4 pieces q, q + KS, q + 2 KS ... of the row: the KS lanes of a unit read 16 KS contiguous
    // bytes per load instruction (a contiguous H / KS run per lane put every lane on a cache line of its own: 64 lines per
    // instruction, and the prologue was 7.7 of the kernel's 20 us; now 2.9)
    const float* base = w_hh + ((int64_t)d * 4 * H + j) * H;
    const float4* pi = reinterpret_cast<const float4*>(base) + q;
    const float4* pf = reinterpret_cast<const float4*>(base + (int64_t)H * H) + q;
    const float4* pg = reinterpret_cast<const float4*>(base + (int64_t)2 * H * H) + q;
    const float4* po = reinterpret_cast<const float4*>(base + (int64_t)3 * H * H) + q;
#pragma unroll
    for (int k = 0; k < HK / 4; ++k) {
      const float4 vi = pi[KS * k], vf = pf[KS * k], vg = pg[KS * k], vo = po[KS * k];
      wi[4 * k] = vi.x; wi[4 * k + 1] = vi.y; wi[4 * k + 2] = vi.z; wi[4 * k + 3] = vi.w;
      wf[4 * k] = vf.x; wf[4 * k + 1] = vf.y; wf[4 * k + 2] = vf.z; wf[4 * k + 3] = vf.w;
      wg[4 * k] = vg.x; wg[4 * k + 1] = vg.y; wg[4 * k + 2] = vg.z; wg[4 * k + 3] = vg.w;
      wo[4 * k] = vo.x; wo[4 * k + 1] = vo.y; wo[4 * k + 2] = vo.z; wo[4 * k + 3] = vo.w;
    }
  }
  if (tid < len) row_s[tid] = (int)row_of_step;
  __syncthreads();
#pragma unroll 6
  for (int s = 0; s < len; ++s) {                              // independent loads: in flight together, and with the weights below
    const float* gp = gates + ((int64_t)row_s[s] * 2 + d) * 4 * H;
#pragma unroll
    for (int e = tid; e < 4 * H; e += NT) gpre_s[s * 4 * H + e] = gp[e];
  }
  float c = 0.f, hcur = 0.f;
  if (tid < H) h_s[0][tid] = 0.f;
  float* wout = words_emb + ((int64_t)b * 2 * H + d * H + j) * Tmax;
  if (q == 0)
    for (int t = len; t < Tmax; ++t) wout[t] = 0.f;
  __syncthreads();
  TGSR_LSTAMP(1);
  for (int s = 0; s < len; ++s) {
    const int t = d == 0 ? s : len - 1 - s;
    const float* hb = h_s[s & 1] + 4 * q;                      // this lane's pieces of h: 4 (q + KS k), as its weights
    const float* gq = gpre_s + s * 4 * H + j;
    const float q0 = gq[0], q1 = gq[H], q2 = gq[2 * H], q3 = gq[3 * H];   // issued ahead of the h reads
    float ai0 = 0.f, af0 = 0.f, ag0 = 0.f, ao0 = 0.f, ai1 = 0.f, af1 = 0.f, ag1 = 0.f, ao1 = 0.f;   // two chains per row
#pragma unroll
    for (int k = 0; k < HK / 4; ++k) {
      const float4 hv = *reinterpret_cast<const float4*>(hb + 4 * KS * k);
      ai0 = fmaf(wi[4 * k], hv.x, ai0); af0 = fmaf(wf[4 * k], hv.x, af0); ag0 = fmaf(wg[4 * k], hv.x, ag0); ao0 = fmaf(wo[4 * k], hv.x, ao0);
      ai1 = fmaf(wi[4 * k + 1], hv.y, ai1); af1 = fmaf(wf[4 * k + 1], hv.y, af1); ag1 = fmaf(wg[4 * k + 1], hv.y, ag1); ao1 = fmaf(wo[4 * k + 1], hv.y, ao1);
      ai0 = fmaf(wi[4 * k + 2], hv.z, ai0); af0 = fmaf(wf[4 * k + 2], hv.z, af0); ag0 = fmaf(wg[4 * k + 2], hv.z, ag0); ao0 = fmaf(wo[4 * k + 2], hv.z, ao0);
      ai1 = fmaf(wi[4 * k + 3], hv.w, ai1); af1 = fmaf(wf[4 * k + 3], hv.w, af1); ag1 = fmaf(wg[4 * k + 3], hv.w, ag1); ao1 = fmaf(wo[4 * k + 3], hv.w, ao1);
    }
    TGSR_LSTAMP(2 + 3 * s);
    // the other K slices sit in the neighbouring lanes: DPP quad_perm adds (a __shfl_xor is a ds_bpermute: an LDS round trip
    // per value); the order of the adds is the same on every lane of a unit: (slice 0 + 1) + (2 + 3)
    float v4[4] = {ai0 + ai1, af0 + af1, ag0 + ag1, ao0 + ao1};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float o = lane_xor1(v4[e]);
      v4[e] = (q & 1) ? o + v4[e] : v4[e] + o;
      if (KS == 4) {
        const float o2 = lane_xor2(v4[e]);
        v4[e] = (q & 2) ? o2 + v4[e] : v4[e] + o2;
      }
    }
    const float si = v4[0] + q0, sf = v4[1] + q1, sg = v4[2] + q2, so = v4[3] + q3;
    const float ig = fsig_(si), fg = fsig_(sf), gg = ftanh_(sg), og = fsig_(so);
    c = fg * c + ig * gg;
    hcur = og * ftanh_(c);
    if (q == 0) {
      h_s[(s + 1) & 1][j] = hcur;
      wout[t] = hcur;
      if (acts) {   // training: gate activations and cell state of this step, [B][Tmax][2][5][H]
        float* ap = acts + (((int64_t)b * Tmax + t) * 2 + d) * 5 * H + j;
        ap[0] = ig; ap[H] = fg; ap[2 * H] = gg; ap[3 * H] = og; ap[4 * H] = c;
      }
    }
    TGSR_LSTAMP(3 + 3 * s);
    __syncthreads();
    TGSR_LSTAMP(4 + 3 * s);
  }
  if (q == 0) sent_emb[(int64_t)b * 2 * H + d * H + j] = hcur;
}

// Any H (<= 256): w_hh streamed from L2 every step; correctness path for configurations the register kernel
// does not cover.
__global__ void lstm_recurrent_generic_kernel(const float* __restrict__ gates, const int32_t* __restrict__ cap_lens,
                                              int Tmax, int H, const float* __restrict__ w_hh,
                                              float* __restrict__ words_emb, float* __restrict__ sent_emb,
                                              const int64_t* __restrict__ captions, int width, int ntoken,
                                              float* __restrict__ acts) {
  __shared__ float h_s[256];
  __shared__ float g_s[1024];
  const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
  int len = cap_lens[b];
  len = len < 0 ? 0 : (len > Tmax ? Tmax : len);
  const float* wr = w_hh + ((int64_t)d * 4 * H + j) * H;
  float c = 0.f, hcur = 0.f;
  if (j < H) h_s[j] = 0.f;
  float* wout = words_emb + ((int64_t)b * 2 * H + d * H + (j < H ? j : 0)) * Tmax;
  if (j < H)
    for (int t = len; t < Tmax; ++t) wout[t] = 0.f;
  __syncthreads();
  for (int s = 0; s < len; ++s) {
    const int t = d == 0 ? s : len - 1 - s;
    int64_t row = (int64_t)b * Tmax + t;
    if (captions) {
      const int64_t v = captions[(int64_t)b * width + t];
      row = (v < 0 || v >= ntoken) ? 0 : v;
    }
    float g = gates[(row * 2 + d) * 4 * H + j];
    for (int k = 0; k < H; ++k) g = fmaf(wr[k], h_s[k], g);
    g_s[j] = g;
    __syncthreads();
    if (j < H) {
      const float ig = sigmoidf_(g_s[j]), fg = sigmoidf_(g_s[H + j]), gg = tanhf(g_s[2 * H + j]),
                  og = sigmoidf_(g_s[3 * H + j]);
      c = fg * c + ig * gg;
      hcur = og * tanhf(c);
      h_s[j] = hcur;
      wout[t] = hcur;
      if (acts) {
        float* ap = acts + (((int64_t)b * Tmax + t) * 2 + d) * 5 * H + j;
        ap[0] = ig; ap[H] = fg; ap[2 * H] = gg; ap[3 * H] = og; ap[4 * H] = c;
      }
    }
    __syncthreads();
  }
  if (j < H) sent_emb[(int64_t)b * 2 * H + d * H + j] = hcur;
}

// Backward through time of one (sample, direction): walks the sample's steps in reverse processing order and emits
// the pre-activation gate gradients dgates[b][t][d][4H] (zero past len) and hprev[b][t][d][H] (the hidden state that
// entered step t) - the operands of the weight-gradient GEMMs  dW_ih = dgates^T x,  dW_hh = dgates^T hprev,
// dx = dgates W_ih.  grid (B, 2), 4H threads: thread (q = tid / H, k = tid % H) keeps W_hh[d][q*H .. q*H+H)[k] (a
// column segment) in registers for dh_prev[k] = sum_j W_hh[j][k] dgate[j]; the four partial sums meet in LDS.
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bwd_kernel(const int32_t* __restrict__ cap_lens, int Tmax,
                                                         const float* __restrict__ w_hh,
                                                         const float* __restrict__ acts,
                                                         const float* __restrict__ words_emb,
                                                         const float* __restrict__ d_words,
                                                         const float* __restrict__ d_sent,
                                                         float* __restrict__ dgates, float* __restrict__ hprev) {
  __shared__ float dg_s[4 * H];
  __shared__ float part_s[4 * H];
  const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x, q = tid / H, k = tid - q * H;
  int len = cap_lens[b];
  len = len < 0 ? 0 : (len > Tmax ? Tmax : len);
  float w[H];
#pragma unroll
  for (int jj = 0; jj < H; ++jj) w[jj] = w_hh[((int64_t)d * 4 * H + q * H + jj) * H + k];
  // steps past the caption: no gradient
  for (int t = len; t < Tmax; ++t) {
    dgates[(((int64_t)b * Tmax + t) * 2 + d) * 4 * H + tid] = 0.f;
    if (tid < H) hprev[(((int64_t)b * Tmax + t) * 2 + d) * H + tid] = 0.f;
  }
  const float* hrow = words_emb + ((int64_t)b * 2 * H + d * H + k) * Tmax;     // h[t] of unit k (tid < H)
  const float* dwrow = d_words + ((int64_t)b * 2 * H + d * H + k) * Tmax;
  float dh_carry = 0.f, dc_carry = 0.f;                                        // valid for tid < H
  if (tid < H && d_sent) dh_carry = d_sent[(int64_t)b * 2 * H + d * H + k];    // final hidden state = sentence code
  for (int s = len - 1; s >= 0; --s) {
    const int t = d == 0 ? s : len - 1 - s;
    const int tp = d == 0 ? t - 1 : t + 1;                                     // time of the previous step (s > 0)
    const int64_t row = ((int64_t)b * Tmax + t) * 2 + d;
    if (tid < H) {
      const float* ap = acts + row * 5 * H + k;
      const float ig = ap[0], fg = ap[H], gg = ap[2 * H], og = ap[3 * H], c = ap[4 * H];
      const float cprev = s > 0 ? acts[(((int64_t)b * Tmax + tp) * 2 + d) * 5 * H + 4 * H + k] : 0.f;
      const float hp = s > 0 ? hrow[tp] : 0.f;
      const float dh = dwrow[t] + dh_carry;
      const float tc = tanhf(c);
      const float dc = dc_carry + dh * og * (1.f - tc * tc);
      dg_s[k] = dc * gg * ig * (1.f - ig);                 // d a_i
      dg_s[H + k] = dc * cprev * fg * (1.f - fg);          // d a_f
      dg_s[2 * H + k] = dc * ig * (1.f - gg * gg);         // d a_g
      dg_s[3 * H + k] = dh * tc * og * (1.f - og);         // d a_o
      dc_carry = dc * fg;
      hprev[row * H + k] = hp;
    }
    __syncthreads();
    dgates[row * 4 * H + tid] = dg_s[tid];
    float p = 0.f;
#pragma unroll
    for (int jj = 0; jj < H; ++jj) p = fmaf(w[jj], dg_s[q * H + jj], p);
    part_s[tid] = p;
    __syncthreads();
    if (tid < H) dh_carry = part_s[k] + part_s[H + k] + part_s[2 * H + k] + part_s[3 * H + k];
    __syncthreads();
  }
}

// out[k] = sum_{p < n} parts[p][k], p ascending (bias gradients = column sums of dgates)
__global__ void lstm_colsum_kernel(const float* __restrict__ parts, int n, int m, float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) {
    float s = 0.f;
    for (int p = 0; p < n; ++p) s += parts[(int64_t)p * m + k];
    out[k] = s;
  }
}

}  // namespace tgsr

using namespace tgsr;

#ifdef TGSR_LSTM_STAMPS
extern "C" int tgsr_debug_read_lstamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(tgsr::g_lstamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -3;
}
#endif

static int launch_recurrent(const float* gates, const int32_t* cap_lens, int B, int Tmax, const float* w_hh, int H,
                            float* words_emb, float* sent_emb, const int64_t* captions, int width, int ntoken,
                            hipStream_t s, float* acts = nullptr) {
  dim3 grid(B, 2);
  static const bool one_row = getenv("TGSR_LSTM_ONE_ROW") != nullptr;       // A/B switch (tools)
  if (H == 128 && !one_row && Tmax <= 24)
    hipLaunchKernelGGL((lstm_recurrent2_kernel<128, 24, 4>), grid, dim3(512), 0, s, gates, cap_lens, Tmax, w_hh, words_emb,
                       sent_emb, captions, width, ntoken, acts);
  else if (H == 128)
    hipLaunchKernelGGL(lstm_recurrent_kernel<128>, grid, dim3(512), 0, s, gates, cap_lens, Tmax, w_hh, words_emb,
                       sent_emb, captions, width, ntoken, acts);
  else if (H == 64)
    hipLaunchKernelGGL(lstm_recurrent_kernel<64>, grid, dim3(256), 0, s, gates, cap_lens, Tmax, w_hh, words_emb,
                       sent_emb, captions, width, ntoken, acts);
  else if (H == 32)
    hipLaunchKernelGGL(lstm_recurrent_kernel<32>, grid, dim3(128), 0, s, gates, cap_lens, Tmax, w_hh, words_emb,
                       sent_emb, captions, width, ntoken, acts);
  else
    hipLaunchKernelGGL(lstm_recurrent_generic_kernel, grid, dim3(4 * H), 0, s, gates, cap_lens, Tmax, H, w_hh,
                       words_emb, sent_emb, captions, width, ntoken, acts);
  return note_launch(hipGetLastError(), "lstm_recurrent_kernel");
}

extern "C" int tgsr_bilstm_fwd(const int64_t* captions, int width, const int32_t* cap_lens, int B, int Tmax,
                               const float* emb, int ntoken, int ninput, const float* w_ih, const float* w_hh,
                               const float* b_ih, const float* b_hh, int H, float* gates_ws, float* words_emb,
                               float* sent_emb, void* stream) {
  if (!captions || !cap_lens || !emb || !w_ih || !w_hh || !b_ih || !b_hh || !gates_ws || !words_emb || !sent_emb)
    return TGSR_EINVAL;
  if (B < 1 || Tmax < 1 || Tmax > width || ntoken < 1 || ninput < 1 || H < 1) return TGSR_EINVAL;
  if (H > 256 || (4 * H) % 64 != 0) return TGSR_EUNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const int M = B * Tmax, N = 8 * H;
  hipLaunchKernelGGL(lstm_input_gates_kernel, dim3((N + 127) / 128, (M + 31) / 32), dim3(256), 0, s, captions, width,
                     Tmax, emb, ntoken, w_ih, b_ih, b_hh, M, N, ninput, gates_ws);
  int rc = note_launch(hipGetLastError(), "lstm_input_gates_kernel");
  if (rc) return rc;
  return launch_recurrent(gates_ws, cap_lens, B, Tmax, w_hh, H, words_emb, sent_emb, nullptr, 0, 0, s);
}

extern "C" int tgsr_lstm_gate_table(const float* emb, int ntoken, int ninput, const float* w_ih, const float* b_ih,
                                    const float* b_hh, int H, float* table, void* stream) {
  if (!emb || !w_ih || !b_ih || !b_hh || !table || ntoken < 1 || ninput < 1 || H < 1) return TGSR_EINVAL;
  const int N = 8 * H;
  hipLaunchKernelGGL(lstm_input_gates_kernel, dim3((N + 127) / 128, (ntoken + 31) / 32), dim3(256), 0,
                     as_stream(stream), (const int64_t*)nullptr, 1, 1, emb, ntoken, w_ih, b_ih, b_hh, ntoken, N, ninput,
                     table);
  return note_launch(hipGetLastError(), "lstm_input_gates_kernel(table)");
}

// ------------------------------------------------------------------------------------------------------------ GRU
// RNN_ENCODER with cfg.RNN_TYPE == 'GRU' (util.py:207-211): 1-layer bidirectional torch.nn.GRU, eval mode, over a per-token
// gate table [ntoken][2][3H] = emb[tok] . w_ih[d]^T + b_ih[d] + (b_hr, b_hz, 0)[d] (tgsr_gru_gate_table: the r / z halves of
// b_hh fold into the table; b_hn stays inside r * (W_hn h + b_hn)).  grid (B, 2), block 3H threads: thread j holds row j of
// W_hh[d] in registers (gate order r, z, n), one mat-vec + one gate pass per step, packed-sequence semantics as the LSTM.
template <int H>
__global__ __launch_bounds__(3 * H) void gru_recurrent_kernel(const float* __restrict__ table,
                                                              const int32_t* __restrict__ cap_lens, int Tmax,
                                                              const float* __restrict__ w_hh, const float* __restrict__ b_hn,
                                                              float* __restrict__ words_emb, float* __restrict__ sent_emb,
                                                              const int64_t* __restrict__ captions, int width, int ntoken,
                                                              float* __restrict__ acts) {
  // captions == nullptr (training): `table` holds one row of gate pre-activations per (b, t) - the embedded, dropped-out input
  // through W_ih - instead of one per token; acts [B][Tmax][2][4][H] then receives (r, z, n, W_hn h + b_hn) of every step for
  // gru_bwd_kernel
  __shared__ __attribute__((aligned(16))) float h_s[H];
  __shared__ float g_s[3 * H];                 // r, z pre-activations; W_hn h + b_hn
  __shared__ float xn_s[H];                    // W_in x + b_in of the step
  const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
  int len = cap_lens[b];
  len = len < 0 ? 0 : (len > Tmax ? Tmax : len);
  float w[H];
  {
    const float4* wr = reinterpret_cast<const float4*>(w_hh + ((int64_t)d * 3 * H + j) * H);
#pragma unroll
    for (int k = 0; k < H / 4; ++k) {
      const float4 v = wr[k];
      w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
  }
  const float bn = j >= 2 * H ? b_hn[d * H + j - 2 * H] : 0.f;
  float hcur = 0.f;
  if (j < H) h_s[j] = 0.f;
  float* wout = words_emb + ((int64_t)b * 2 * H + d * H + (j < H ? j : 0)) * Tmax;
  if (j < H)
    for (int t = len; t < Tmax; ++t) wout[t] = 0.f;
  __syncthreads();
  for (int s = 0; s < len; ++s) {
    const int t = d == 0 ? s : len - 1 - s;
    int64_t row;
    if (captions) {
      const int64_t v = captions[(int64_t)b * width + t];
      row = (v < 0 || v >= ntoken) ? 0 : v;
    } else {
      row = (int64_t)b * Tmax + t;
    }
    const float x = table[(row * 2 + d) * 3 * H + j];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;            // four independent chains, combined pairwise (as the LSTM kernel)
#pragma unroll
    for (int k = 0; k < H / 4; ++k) {
      const float4 hv = *reinterpret_cast<const float4*>(h_s + 4 * k);
      g0 = fmaf(w[4 * k], hv.x, g0);
      g1 = fmaf(w[4 * k + 1], hv.y, g1);
      g2 = fmaf(w[4 * k + 2], hv.z, g2);
      g3 = fmaf(w[4 * k + 3], hv.w, g3);
    }
    const float hd = (g0 + g1) + (g2 + g3);
    if (j < 2 * H) {
      g_s[j] = x + hd;
    } else {
      g_s[j] = hd + bn;
      xn_s[j - 2 * H] = x;
    }
    __syncthreads();
    if (j < H) {
      const float r = fsig_(g_s[j]), z = fsig_(g_s[H + j]);
      const float n = ftanh_(xn_s[j] + r * g_s[2 * H + j]);
      if (acts) {
        float* ap = acts + ((((int64_t)b * Tmax + t) * 2 + d) * 4) * H + j;
        ap[0] = r; ap[H] = z; ap[2 * H] = n; ap[3 * H] = g_s[2 * H + j];
      }
      hcur = (1.f - z) * n + z * hcur;
      h_s[j] = hcur;
      wout[t] = hcur;
    }
    __syncthreads();
  }
  if (j < H) sent_emb[(int64_t)b * 2 * H + d * H + j] = hcur;
}

// BPTT of one (sample, direction) of the GRU: thread q * H + k keeps column k of the rows [q H, (q + 1) H) of W_hh (gate q = r, z, n)
// for dh_prev[k] = sum_j W_hh[j][k] dgh[j]; per step the H unit threads turn dh into the gate gradients
//   dn = dh (1 - z), dz = dh (h_prev - n), da_n = dn (1 - n^2), da_z = dz z (1 - z), d(W_hn h + b_hn) = da_n r, da_r = da_n hn r (1 - r)
// dgx [B][Tmax][2][3H] = (da_r, da_z, da_n): the gradient at x W_ih^T + b_ih; dgh = (da_r, da_z, da_n r): at h W_hh^T + b_hh.
template <int H>
__global__ __launch_bounds__(3 * H) void gru_bwd_kernel(const int32_t* __restrict__ cap_lens, int Tmax, const float* __restrict__ w_hh,
                                                        const float* __restrict__ acts, const float* __restrict__ words_emb,
                                                        const float* __restrict__ d_words, const float* __restrict__ d_sent,
                                                        float* __restrict__ dgx, float* __restrict__ dgh, float* __restrict__ hprev) {
  __shared__ float dg_s[3 * H];
  __shared__ float part_s[3 * H];
  const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x, q = tid / H, k = tid - q * H;
  int len = cap_lens[b];
  len = len < 0 ? 0 : (len > Tmax ? Tmax : len);
  float w[H];
#pragma unroll
  for (int jj = 0; jj < H; ++jj) w[jj] = w_hh[((int64_t)d * 3 * H + q * H + jj) * H + k];
  for (int t = len; t < Tmax; ++t) {                       // steps past the caption: no gradient
    const int64_t row = ((int64_t)b * Tmax + t) * 2 + d;
    dgx[row * 3 * H + tid] = 0.f;
    dgh[row * 3 * H + tid] = 0.f;
    if (tid < H) hprev[row * H + tid] = 0.f;
  }
  const float* hrow = words_emb + ((int64_t)b * 2 * H + d * H + k) * Tmax;
  const float* dwrow = d_words + ((int64_t)b * 2 * H + d * H + k) * Tmax;
  float dh_carry = 0.f;
  if (tid < H && d_sent) dh_carry = d_sent[(int64_t)b * 2 * H + d * H + k];
  for (int s = len - 1; s >= 0; --s) {
    const int t = d == 0 ? s : len - 1 - s;
    const int tp = d == 0 ? t - 1 : t + 1;
    const int64_t row = ((int64_t)b * Tmax + t) * 2 + d;
    float dh_direct = 0.f;
    if (tid < H) {
      const float* ap = acts + row * 4 * H + k;
      const float r = ap[0], z = ap[H], n = ap[2 * H], hn = ap[3 * H];
      const float hp = s > 0 ? hrow[tp] : 0.f;
      const float dh = dwrow[t] + dh_carry;
      const float dan = dh * (1.f - z) * (1.f - n * n);
      const float daz = dh * (hp - n) * z * (1.f - z);
      const float dhn = dan * r;
      const float dar = dan * hn * r * (1.f - r);
      dh_direct = dh * z;
      dg_s[k] = dar; dg_s[H + k] = daz; dg_s[2 * H + k] = dhn;
      float* gx = dgx + row * 3 * H + k;
      gx[0] = dar; gx[H] = daz; gx[2 * H] = dan;
      hprev[row * H + k] = hp;
    }
    __syncthreads();
    dgh[row * 3 * H + tid] = dg_s[tid];
    float p = 0.f;
#pragma unroll
    for (int jj = 0; jj < H; ++jj) p = fmaf(w[jj], dg_s[q * H + jj], p);
    part_s[tid] = p;
    __syncthreads();
    if (tid < H) dh_carry = dh_direct + (part_s[k] + part_s[H + k] + part_s[2 * H + k]);
    __syncthreads();
  }
}

extern "C" int tgsr_gru_gate_table(const float* emb, int ntoken, int ninput, const float* w_ih, const float* b_ih,
                                   const float* b_hh_rz, int H, float* table, void* stream) {
  if (!emb || !w_ih || !b_ih || !b_hh_rz || !table || ntoken < 1 || ninput < 1 || H < 1) return TGSR_EINVAL;
  const int N = 6 * H;
  hipLaunchKernelGGL(lstm_input_gates_kernel, dim3((N + 127) / 128, (ntoken + 31) / 32), dim3(256), 0,
                     as_stream(stream), (const int64_t*)nullptr, 1, 1, emb, ntoken, w_ih, b_ih, b_hh_rz, ntoken, N, ninput,
                     table);
  return note_launch(hipGetLastError(), "lstm_input_gates_kernel(gru table)");
}

extern "C" int tgsr_bigru_table_fwd(const int64_t* captions, int width, const int32_t* cap_lens, int B, int Tmax,
                                    const float* table, int ntoken, const float* w_hh, const float* b_hn, int H,
                                    float* words_emb, float* sent_emb, void* stream) {
  if (!captions || !cap_lens || !table || !w_hh || !b_hn || !words_emb || !sent_emb) return TGSR_EINVAL;
  if (B < 1 || Tmax < 1 || Tmax > width || ntoken < 1 || H < 1) return TGSR_EINVAL;
  hipStream_t s = as_stream(stream);
  const dim3 grid(B, 2);
  if (H == 128)
    hipLaunchKernelGGL(gru_recurrent_kernel<128>, grid, dim3(384), 0, s, table, cap_lens, Tmax, w_hh, b_hn, words_emb, sent_emb,
                       captions, width, ntoken, (float*)nullptr);
  else if (H == 64)
    hipLaunchKernelGGL(gru_recurrent_kernel<64>, grid, dim3(192), 0, s, table, cap_lens, Tmax, w_hh, b_hn, words_emb, sent_emb,
                       captions, width, ntoken, (float*)nullptr);
  else if (H == 32)
    hipLaunchKernelGGL(gru_recurrent_kernel<32>, grid, dim3(96), 0, s, table, cap_lens, Tmax, w_hh, b_hn, words_emb, sent_emb,
                       captions, width, ntoken, (float*)nullptr);
  else
    return TGSR_EUNSUPPORTED;
  return note_launch(hipGetLastError(), "gru_recurrent_kernel");
}

// Training forward of the bidirectional GRU on dense (embedded, dropped-out) inputs x [B][Tmax][ninput]: the gate GEMM
// (b_hh_rz = b_hh with its n third zeroed: r / z biases fold, b_hn stays inside r * (W_hn h + b_hn)), then the recurrence with the
// per-step activations saved for tgsr_bigru_bwd.  gates_ws [B*Tmax][2][3H], acts [B][Tmax][2][4][H].
extern "C" int tgsr_bigru_train_fwd(const float* x, const int32_t* cap_lens, int B, int Tmax, int ninput, const float* w_ih,
                                    const float* w_hh, const float* b_ih, const float* b_hh_rz, const float* b_hn, int H,
                                    float* gates_ws, float* acts, float* words_emb, float* sent_emb, void* stream) {
  if (!x || !cap_lens || !w_ih || !w_hh || !b_ih || !b_hh_rz || !b_hn || !gates_ws || !acts || !words_emb || !sent_emb)
    return TGSR_EINVAL;
  if (B < 1 || Tmax < 1 || ninput < 1 || H < 1) return TGSR_EINVAL;
  if (H != 128 && H != 64 && H != 32) return TGSR_EUNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const int M = B * Tmax, N = 6 * H;
  hipLaunchKernelGGL(lstm_input_gates_kernel, dim3((N + 127) / 128, (M + 31) / 32), dim3(256), 0, s, (const int64_t*)nullptr, 1, 1,
                     x, M, w_ih, b_ih, b_hh_rz, M, N, ninput, gates_ws);
  int rc = note_launch(hipGetLastError(), "lstm_input_gates_kernel(gru train)");
  if (rc) return rc;
  const dim3 grid(B, 2);
  const int64_t* nocap = nullptr;
  if (H == 128)
    hipLaunchKernelGGL(gru_recurrent_kernel<128>, grid, dim3(384), 0, s, gates_ws, cap_lens, Tmax, w_hh, b_hn, words_emb, sent_emb,
                       nocap, 0, 0, acts);
  else if (H == 64)
    hipLaunchKernelGGL(gru_recurrent_kernel<64>, grid, dim3(192), 0, s, gates_ws, cap_lens, Tmax, w_hh, b_hn, words_emb, sent_emb,
                       nocap, 0, 0, acts);
  else
    hipLaunchKernelGGL(gru_recurrent_kernel<32>, grid, dim3(96), 0, s, gates_ws, cap_lens, Tmax, w_hh, b_hn, words_emb, sent_emb,
                       nocap, 0, 0, acts);
  return note_launch(hipGetLastError(), "gru_recurrent_kernel(train)");
}

// BPTT of both directions.  dgx / dgh [B][Tmax][2][3H] (gradients at the input-side / hidden-side gate pre-activations), hprev
// [B][Tmax][2][H] (h of the previous step), dbias [2][2][3H] = (d b_ih, d b_hh) = their column sums (NULL: skipped).
extern "C" int tgsr_bigru_bwd(const int32_t* cap_lens, int B, int Tmax, int H, const float* w_hh, const float* acts,
                              const float* words_emb, const float* d_words, const float* d_sent, float* dgx, float* dgh,
                              float* hprev, float* dbias, void* stream) {
  if (!cap_lens || !w_hh || !acts || !words_emb || !d_words || !dgx || !dgh || !hprev || B < 1 || Tmax < 1) return TGSR_EINVAL;
  hipStream_t s = as_stream(stream);
  const dim3 grid(B, 2);
  if (H == 128)
    hipLaunchKernelGGL(gru_bwd_kernel<128>, grid, dim3(384), 0, s, cap_lens, Tmax, w_hh, acts, words_emb, d_words, d_sent, dgx, dgh, hprev);
  else if (H == 64)
    hipLaunchKernelGGL(gru_bwd_kernel<64>, grid, dim3(192), 0, s, cap_lens, Tmax, w_hh, acts, words_emb, d_words, d_sent, dgx, dgh, hprev);
  else if (H == 32)
    hipLaunchKernelGGL(gru_bwd_kernel<32>, grid, dim3(96), 0, s, cap_lens, Tmax, w_hh, acts, words_emb, d_words, d_sent, dgx, dgh, hprev);
  else
    return TGSR_EUNSUPPORTED;
  int rc = note_launch(hipGetLastError(), "gru_bwd_kernel");
  if (rc || !dbias) return rc;
  const int m = 6 * H;
  hipLaunchKernelGGL(lstm_colsum_kernel, dim3((m + 255) / 256), dim3(256), 0, s, dgx, B * Tmax, m, dbias);
  hipLaunchKernelGGL(lstm_colsum_kernel, dim3((m + 255) / 256), dim3(256), 0, s, dgh, B * Tmax, m, dbias + m);
  return note_launch(hipGetLastError(), "lstm_colsum_kernel");
}

extern "C" int tgsr_bilstm_table_fwd(const int64_t* captions, int width, const int32_t* cap_lens, int B, int Tmax,
                                     const float* table, int ntoken, const float* w_hh, int H, float* words_emb,
                                     float* sent_emb, void* stream) {
  if (!captions || !cap_lens || !table || !w_hh || !words_emb || !sent_emb) return TGSR_EINVAL;
  if (B < 1 || Tmax < 1 || Tmax > width || ntoken < 1 || H < 1) return TGSR_EINVAL;
  if (H > 256 || (4 * H) % 64 != 0) return TGSR_EUNSUPPORTED;
  return launch_recurrent(table, cap_lens, B, Tmax, w_hh, H, words_emb, sent_emb, captions, width, ntoken,
                          as_stream(stream));
}

// Training forward on dense (already embedded, possibly dropped-out) inputs x [B][Tmax][ninput]; also saves the gate
// activations and cell states acts [B][Tmax][2][5][H] for tgsr_bilstm_bwd.
extern "C" int tgsr_bilstm_train_fwd(const float* x, const int32_t* cap_lens, int B, int Tmax, int ninput,
                                     const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int H,
                                     float* gates_ws, float* acts, float* words_emb, float* sent_emb, void* stream) {
  if (!x || !cap_lens || !w_ih || !w_hh || !b_ih || !b_hh || !gates_ws || !acts || !words_emb || !sent_emb)
    return TGSR_EINVAL;
  if (B < 1 || Tmax < 1 || ninput < 1 || H < 1) return TGSR_EINVAL;
  if (H > 256 || (4 * H) % 64 != 0) return TGSR_EUNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const int M = B * Tmax, N = 8 * H;
  hipLaunchKernelGGL(lstm_input_gates_kernel, dim3((N + 127) / 128, (M + 31) / 32), dim3(256), 0, s,
                     (const int64_t*)nullptr, 1, 1, x, M, w_ih, b_ih, b_hh, M, N, ninput, gates_ws);
  int rc = note_launch(hipGetLastError(), "lstm_input_gates_kernel(train)");
  if (rc) return rc;
  return launch_recurrent(gates_ws, cap_lens, B, Tmax, w_hh, H, words_emb, sent_emb, nullptr, 0, 0, s, acts);
}

extern "C" int tgsr_bilstm_bwd(const int32_t* cap_lens, int B, int Tmax, int H, const float* w_hh, const float* acts,
                               const float* words_emb, const float* d_words, const float* d_sent, float* dgates,
                               float* hprev, float* dbias, void* stream) {
  if (!cap_lens || !w_hh || !acts || !words_emb || !d_words || !dgates || !hprev || B < 1 || Tmax < 1)
    return TGSR_EINVAL;
  hipStream_t s = as_stream(stream);
  dim3 grid(B, 2);
  if (H == 128)
    hipLaunchKernelGGL(lstm_bwd_kernel<128>, grid, dim3(512), 0, s, cap_lens, Tmax, w_hh, acts, words_emb, d_words,
                       d_sent, dgates, hprev);
  else if (H == 64)
    hipLaunchKernelGGL(lstm_bwd_kernel<64>, grid, dim3(256), 0, s, cap_lens, Tmax, w_hh, acts, words_emb, d_words,
                       d_sent, dgates, hprev);
  else if (H == 32)
    hipLaunchKernelGGL(lstm_bwd_kernel<32>, grid, dim3(128), 0, s, cap_lens, Tmax, w_hh, acts, words_emb, d_words,
                       d_sent, dgates, hprev);
  else
    return TGSR_EUNSUPPORTED;
  int rc = note_launch(hipGetLastError(), "lstm_bwd_kernel");
  if (rc || !dbias) return rc;
  const int m = 8 * H;   // dbias[2][4H] = sum over (b, t) of dgates (= d b_ih = d b_hh)
  hipLaunchKernelGGL(lstm_colsum_kernel, dim3((m + 255) / 256), dim3(256), 0, s, dgates, B * Tmax, m, dbias);
  return note_launch(hipGetLastError(), "lstm_colsum_kernel");
}
