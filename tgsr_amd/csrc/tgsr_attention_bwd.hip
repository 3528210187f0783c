// Backward of the generator's word attention (GlobalAttentionGeneral.forward, GlobalAttention.py:87-130) for gfx950.
//
// Per pixel q of sample b, with s[t] = sum_i src[i][t] h[i][q], P = softmax_t(masked s), C[i] = sum_t src[i][t] P[t]:
//     dP[t]  = sum_i src[i][t] dC[i][q]                      MFMA, same operand form as the forward's first GEMM
//     dS[t]  = P[t] (dP[t] - sum_t' P[t'] dP[t'])            in-lane + one lane^32 exchange
//     dh[i]  = sum_t src[i][t] dS[t]                         MFMA, dS consumed from the accumulator registers
//     dsrc[i][t] += sum_q (dC[i][q] P[t][q] + h[i][q] dS[t][q])   contraction over PIXELS: the four operand tiles go
//                                                            through LDS once ([32][33] images) to put the channel /
//                                                            word index on the lane, then 2 x 16 MFMA k-steps
// P and S are recomputed from h and src (cheaper than storing [B][T][Q] twice).  The attention map output itself
// carries no gradient (the reference only visualises it).  dsrc is accumulated in registers over all pixel blocks
// of a workgroup, reduced across its 4 waves in LDS, and written as ONE slab per workgroup (summed by the caller).
#include "tgsr_common.h"

namespace tgsr {

struct AttnBwdArgs {
  const float* h;
  int64_t hbs;
  const float* src;      // [B][idf][32]
  const uint8_t* mask;
  int mask_mode, B, T, Q, blocks_per_wave;
  const float* dc;       // [B][idf][Q] dense
  float* dh;             // [B][idf][Q] dense
  float* dsrc_part;      // [B][nchunks][idf][32]
};

template <int NI>
__global__ __launch_bounds__(256) void word_attention_bwd_kernel(AttnBwdArgs a) {
  constexpr int IDF = 32 * NI, P = 33;
  __shared__ float src_s[IDF * 32];
  __shared__ float srcT_s[32 * IDF];
  __shared__ unsigned mbits_s[256];
  __shared__ float tile_s[4][4][32 * P];   // per wave: dC, h (one 32-channel block at a time), P, dS images
  __shared__ float red_s[NI][32 * 32];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const float* sb = a.src + (int64_t)b * IDF * 32;
  for (int o = tid; o < IDF * 32; o += 256) {
    const float v = sb[o];
    src_s[o] = v;
    srcT_s[(o & 31) * IDF + (o >> 5)] = v;
  }
  for (int o = tid; o < NI * 1024; o += 256) (&red_s[0][0])[o] = 0.f;
  const int nrows = a.mask ? (a.B < 256 ? a.B : 256) : 0;
  for (int r = tid; r < nrows; r += 256) {
    unsigned m = 0;
    for (int t = 0; t < a.T; ++t) m |= (a.mask[r * a.T + t] ? 1u : 0u) << t;
    mbits_s[r] = m;
  }
  __syncthreads();

  const float* hb = a.h + (int64_t)b * a.hbs;
  const float* dcb = a.dc + (int64_t)b * IDF * a.Q;
  float* dhb = a.dh + (int64_t)b * IDF * a.Q;
  float* tdc = tile_s[wave][0];
  float* th = tile_s[wave][1];
  float* tp = tile_s[wave][2];
  float* tds = tile_s[wave][3];

  f32x16 dsrc[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) dsrc[k][i] = 0.f;

  const int blk0 = (blockIdx.x * 4 + wave) * a.blocks_per_wave;
  for (int bi = 0; bi < a.blocks_per_wave; ++bi) {
    const int q0 = (blk0 + bi) * 32;
    if (q0 >= a.Q) break;
    const int q = q0 + l31;
    const bool qok = q < a.Q;
    // S and dP: k = channel pairs
    f32x16 s, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
    float hv[IDF / 2], dv[IDF / 2];
#pragma unroll
    for (int k = 0; k < IDF / 2; ++k) {
      hv[k] = qok ? hb[(int64_t)(2 * k + hh) * a.Q + q] : 0.f;
      dv[k] = qok ? dcb[(int64_t)(2 * k + hh) * a.Q + q] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < IDF / 2; ++k) {
      const float av = src_s[(2 * k + hh) * 32 + l31];
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(av, hv[k], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dv[k], dp, 0, 0, 0);
    }
    unsigned mb = 0;
    if (a.mask) {
      const int mrow = a.mask_mode ? b : (int)(((int64_t)b * a.Q + q) % a.B);
      if (mrow < 256) mb = mbits_s[mrow];
      else for (int t = 0; t < a.T; ++t) mb |= (a.mask[mrow * a.T + t] ? 1u : 0u) << t;
    }
    const unsigned valid = (a.T >= 32 ? 0xffffffffu : ((1u << a.T) - 1u)) & ~mb;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (!((valid >> acc_row(i, hh)) & 1u)) s[i] = -INFINITY;
      mx = fmaxf(mx, s[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = __expf(s[i] - mx); sum += s[i]; }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    float rd = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] *= inv; rd = fmaf(s[i], dp[i], rd); }
    rd += __shfl_xor(rd, 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) dp[i] = qok ? s[i] * (dp[i] - rd) : 0.f;   // dp now holds dS
    if (!qok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = 0.f;
    }
    // P and dS images [t][q] for the pixel contraction
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      tp[acc_row(i, hh) * P + l31] = s[i];
      tds[acc_row(i, hh) * P + l31] = dp[i];
    }
#pragma unroll
    for (int blk = 0; blk < NI; ++blk) {
      // dh block: A = srcT[t][i], B = dS registers
      f32x16 c;
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (acc_row(r, 0) < a.T)
          c = __builtin_amdgcn_mfma_f32_32x32x2f32(srcT_s[acc_row(r, hh) * IDF + blk * 32 + l31], dp[r], c, 0, 0, 0);
      if (qok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) dhb[(int64_t)(blk * 32 + acc_row(i, hh)) * a.Q + q] = c[i];
      }
      // channel images [i][q] of this block (hv/dv registers hold channel 2k+hh at pixel q)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        tdc[(2 * k + hh) * P + l31] = dv[blk * 16 + k];
        th[(2 * k + hh) * P + l31] = hv[blk * 16 + k];
      }
      __builtin_amdgcn_wave_barrier();
      // dsrc[i][t] += sum_q dC[i][q] P[t][q] + h[i][q] dS[t][q]; lane = (A: channel i | B: word t), k = pixel pairs
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        dsrc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(tdc[l31 * P + 2 * k + hh], tp[l31 * P + 2 * k + hh], dsrc[blk], 0, 0, 0);
        dsrc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(th[l31 * P + 2 * k + hh], tds[l31 * P + 2 * k + hh], dsrc[blk], 0, 0, 0);
      }
    }
  }
  // reduce the 4 waves, one slab per workgroup: [idf][32]; lane = word t, register rows = channel i
  // (the waves take turns, in order: an LDS atomicAdd here made d(conv_context.weight) differ by 1e-11 from run to run)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int blk = 0; blk < NI; ++blk)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* d = &red_s[blk][acc_row(i, hh) * 32 + l31];
          *d = (w == 0 ? 0.f : *d) + dsrc[blk][i];
        }
    }
    __syncthreads();
  }
  float* o = a.dsrc_part + ((int64_t)b * gridDim.x + blockIdx.x) * IDF * 32;
  for (int e = tid; e < IDF * 32; e += 256) o[e] = (&red_s[0][0])[e];
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_word_attention_bwd_chunks(int Q) {
  // one pixel block (32 pixels) per wave and chunk where that is what fills the chip: at B = 16 the 32^2 / 64^2 / 128^2
  // stages get 8 / 32 / 64 chunks = 128 / 512 / 1024 workgroups (with 8 blocks per wave they were 16 / 64 / 256
  // workgroups of four latency-bound waves each: 0.41 ms per step for a 17 us matrix-pipe job)
  int n = (Q + 127) / 128;
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}

extern "C" int tgsr_word_attention_bwd(const float* h, int64_t h_bstride, const float* src, const uint8_t* mask,
                                       int mask_mode, int B, int idf, int T, int Q, const float* dc, float* dh,
                                       float* dsrc_part, void* stream) {
  if (!h || !src || !dc || !dh || !dsrc_part || B < 1 || T < 1 || Q < 1) return TGSR_EINVAL;
  if (T > 32 || (idf != 32 && idf != 64)) return TGSR_EUNSUPPORTED;
  AttnBwdArgs a;
  a.h = h; a.hbs = h_bstride; a.src = src; a.mask = mask; a.mask_mode = mask_mode; a.B = B; a.T = T; a.Q = Q;
  a.dc = dc; a.dh = dh; a.dsrc_part = dsrc_part;
  const int nch = tgsr_word_attention_bwd_chunks(Q);
  const int nblk = (Q + 31) / 32;
  a.blocks_per_wave = (nblk + nch * 4 - 1) / (nch * 4);
  dim3 grid(nch, B);
  if (idf == 32) hipLaunchKernelGGL(word_attention_bwd_kernel<1>, grid, dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(word_attention_bwd_kernel<2>, grid, dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "word_attention_bwd_kernel");
}
