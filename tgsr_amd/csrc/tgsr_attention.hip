// Word-level attention of the generator (GlobalAttentionGeneral.forward, GlobalAttention.py:87-130) for gfx950.
//
//   word_project_kernel : src[b][i][t] = sum_c w_ctx[i][c] * words[b][c][t]      (conv1x1, :100-102), T padded to 32
//   word_attention_kernel<NI>: per wave, 32 pixels:
//       S[t][q]  = sum_i src[i][t] * h[i][q]          MFMA 32x32x2 f32, A = src (lane = word), B = h (lane = pixel,
//                                                       read straight from HBM: 32 consecutive pixels of a channel plane)
//       mask (reference quirk or per-sample), softmax over the 32 word rows: 16 registers in-lane + one
//                                                       cross-half exchange (lane ^ 32) - no LDS, no transposes
//       C[i][q]  = sum_t src[i][t] * P[t][q]          MFMA again; P is consumed as the B operand directly from the
//                                                       accumulator registers (k-step r pairs words t0(r) and t0(r)+4)
//   Every store is 32 consecutive pixels of one plane (attn[b][t][:], c_code[b][i][:]) = 128-B runs.
// The kernel is HBM-bound ((2*idf + T) * 4 bytes per pixel against 4*idf*T flops): one pass over h, one write of
// c_code and attn, versus bmm + masked_fill + softmax + 2 transposes + bmm in the reference.
#include "tgsr_common.h"
#include "tgsr_text_blocks.h"

namespace tgsr {

// grid (B, idf / 8), 256 threads: thread (i = 8 * blockIdx.y + tid / 32, t = tid % 32) owns one output; the
// sample's words are staged once in LDS as [c][32] (zero padded), the weight row is a half-wave-uniform float4 stream.
__global__ __launch_bounds__(256) void word_project_kernel(const float* __restrict__ words,
                                                           const float* __restrict__ w_ctx, float* __restrict__ src,
                                                           int idf, int cdf, int T) {
  extern __shared__ __attribute__((aligned(16))) float ws[];   // [cdf][32]
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* wb = words + (int64_t)b * cdf * T;
  for (int o = tid; o < cdf * 32; o += 256) {
    const int c = o >> 5, t = o & 31;
    ws[o] = t < T ? wb[c * T + t] : 0.f;
  }
  __syncthreads();
  const int i = blockIdx.y * 8 + (tid >> 5), t = tid & 31;
  const float* wr = w_ctx + (int64_t)i * cdf;
  float acc = 0.f;
  int c = 0;
  if ((cdf & 3) == 0) {
#pragma unroll 4
    for (; c < cdf; c += 4) {
      const float4 wv = *reinterpret_cast<const float4*>(wr + c);
      acc = fmaf(wv.x, ws[(c + 0) * 32 + t], acc);
      acc = fmaf(wv.y, ws[(c + 1) * 32 + t], acc);
      acc = fmaf(wv.z, ws[(c + 2) * 32 + t], acc);
      acc = fmaf(wv.w, ws[(c + 3) * 32 + t], acc);
    }
  }
  for (; c < cdf; ++c) acc = fmaf(wr[c], ws[c * 32 + t], acc);
  src[((int64_t)b * idf + i) * 32 + t] = acc;
}

// word_project_block (tgsr_text_blocks.h): grid (B, nsets, idf / 32).
__global__ __launch_bounds__(256) void word_project_mfma_kernel(ProjArgs a) {
  __shared__ float red[kProjRedFloats];
  word_project_block(a, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

struct AttnArgs {
  const float* h;
  int64_t hbs;
  const float* src;      // [B][idf][32]
  const uint8_t* mask;   // [B][T] or null
  int mask_mode, B, T, Q;
  float* c_code;
  int64_t cbs;
  float* attn;           // [B][T][Q] or null
};

template <int NI>  // idf = 32 * NI
__global__ __launch_bounds__(256) void word_attention_kernel(AttnArgs a) {
  constexpr int IDF = 32 * NI;
  __shared__ float src_s[IDF * 32];   // [i][t]  (GEMM1 A operand: lanes = t, conflict-free)
  constexpr int TP = IDF + 1;         // row pitch of srcT: the transposing store below walks a column (bank = t + i: conflict free)
  __shared__ float srcT_s[32 * TP];   // [t][i]  (GEMM2 A operand: lanes = i, conflict-free)
  __shared__ unsigned mbits_s[256];   // packed mask rows (bit t = masked), up to 256 rows cached

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = tid >> 6;
  const int b = blockIdx.y;

  const float* sb = a.src + (int64_t)b * IDF * 32;
  for (int o = tid; o < IDF * 32; o += 256) {
    const float v = sb[o];
    src_s[o] = v;
    srcT_s[(o & 31) * TP + (o >> 5)] = v;
  }
  const int nrows = a.mask ? (a.B < 256 ? a.B : 256) : 0;
  for (int r = tid; r < nrows; r += 256) {
    unsigned m = 0;
    for (int t = 0; t < a.T; ++t) m |= (a.mask[r * a.T + t] ? 1u : 0u) << t;
    mbits_s[r] = m;
  }
  __syncthreads();

  // A wave walks over 32-pixel tiles (stride = waves of the sample's workgroups): the staging above is paid once per
  // workgroup, and the h rows of the next tile are in flight while this one is computed.
  const int ntiles = (a.Q + 31) >> 5, tstride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;
  const float* hb = a.h + (int64_t)b * a.hbs;
  float* cb = a.c_code + (int64_t)b * a.cbs;
  float hv[IDF / 2];
  {
    const int q = tile * 32 + l31;
#pragma unroll
    for (int k = 0; k < IDF / 2; ++k) hv[k] = q < a.Q ? hb[(int64_t)(2 * k + hh) * a.Q + q] : 0.f;
  }
  while (tile < ntiles) {
    const int q = tile * 32 + l31;
    const bool qok = q < a.Q;
    const int tnext = tile + tstride;
    float hn[IDF / 2];
    {
      const int qn = tnext * 32 + l31;
      const bool nok = tnext < ntiles && qn < a.Q;
#pragma unroll
      for (int k = 0; k < IDF / 2; ++k) hn[k] = nok ? hb[(int64_t)(2 * k + hh) * a.Q + qn] : 0.f;
    }

    // ---- GEMM1: S[t][q], k = channel pairs
    f32x16 s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
    for (int k = 0; k < IDF / 2; ++k) {
      const float av = src_s[(2 * k + hh) * 32 + l31];
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(av, hv[k], s, 0, 0, 0);
    }

    // ---- mask + softmax over words (rows of S: 16 in this lane, 16 in lane ^ 32)
    unsigned mb = 0;
    if (a.mask) {
      int mrow = a.mask_mode ? b : (int)(((int64_t)b * a.Q + q) % a.B);   // GlobalAttention.py:111 mask.repeat(queryL,1)
      if (mrow < 256) {
        mb = mbits_s[mrow];
      } else {
        for (int t = 0; t < a.T; ++t) mb |= (a.mask[mrow * a.T + t] ? 1u : 0u) << t;
      }
    }
    const unsigned valid = (a.T >= 32 ? 0xffffffffu : ((1u << a.T) - 1u)) & ~mb;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = acc_row(i, hh);
      if (!((valid >> t) & 1u)) s[i] = -INFINITY;
      mx = fmaxf(mx, s[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i] = __expf(s[i] - mx);   // exp(-inf) = 0 for masked / padded words
      sum += s[i];
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] *= inv;

    if (a.attn && qok) {
      float* __restrict__ ab = a.attn + (int64_t)b * a.T * a.Q + q;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t = acc_row(i, hh);
        if (t < a.T) ab[(int64_t)t * a.Q] = s[i];
      }
    }

    // ---- GEMM2: C[i][q] = sum_t src[i][t] P[t][q]; k-step r pairs words t0 = acc_row(r,0) and t0 + 4 (= this lane
    // half's own register r), so P never leaves the accumulator registers.
#pragma unroll
    for (int blk = 0; blk < NI; ++blk) {
      f32x16 c;
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (acc_row(r, 0) < a.T) {   // wave-uniform; words >= T have P = 0 anyway
          const float av = srcT_s[acc_row(r, hh) * TP + blk * 32 + l31];
          c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s[r], c, 0, 0, 0);
        }
      }
      if (qok) {
        float* __restrict__ cq = cb + q;
#pragma unroll
        for (int i = 0; i < 16; ++i) cq[(int64_t)(blk * 32 + acc_row(i, hh)) * a.Q] = c[i];
      }
    }
#pragma unroll
    for (int k = 0; k < IDF / 2; ++k) hv[k] = hn[k];
    tile = tnext;
  }
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_word_attention_fwd(const float* h, int64_t h_bstride, const float* words, const float* w_ctx,
                                       const uint8_t* mask, int mask_mode, int B, int idf, int cdf, int T, int Q,
                                       float* src_ws, float* c_code, int64_t c_bstride, float* attn, void* stream) {
  if (!h || !src_ws || !c_code || B < 1 || cdf < 1 || Q < 1 || T < 1) return TGSR_EINVAL;
  if ((words == nullptr) != (w_ctx == nullptr)) return TGSR_EINVAL;
  if (T > 32 || (idf != 32 && idf != 64 && idf != 128)) return TGSR_EUNSUPPORTED;
  hipStream_t s = as_stream(stream);
  if (words) {   // else: src_ws already holds the projection (tgsr_word_project_fwd)
    if (cdf > 1024) return TGSR_EUNSUPPORTED;   // words of one sample are staged in LDS (cdf * 128 bytes)
    hipLaunchKernelGGL(word_project_kernel, dim3(B, idf / 8), dim3(256), (size_t)cdf * 32 * sizeof(float), s, words,
                       w_ctx, src_ws, idf, cdf, T);
    int rc = note_launch(hipGetLastError(), "word_project_kernel");
    if (rc) return rc;
  }
  AttnArgs a;
  a.h = h; a.hbs = h_bstride; a.src = src_ws; a.mask = mask; a.mask_mode = mask_mode;
  a.B = B; a.T = T; a.Q = Q; a.c_code = c_code; a.cbs = c_bstride; a.attn = attn;
  // enough workgroups to fill the chip (~4 per CU over the batch), each walking over several 128-pixel tiles
  int gx = (Q + 127) / 128;
  const int cap = (1024 + B - 1) / B;
  if (gx > cap) gx = cap;
  dim3 grid(gx, B);
  if (idf == 32) hipLaunchKernelGGL(word_attention_kernel<1>, grid, dim3(256), 0, s, a);
  else if (idf == 64) hipLaunchKernelGGL(word_attention_kernel<2>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(word_attention_kernel<4>, grid, dim3(256), 0, s, a);
  return note_launch(hipGetLastError(), "word_attention_kernel");
}

extern "C" int tgsr_word_project_fwd(const float* words, const float* const* w_ctx, int nsets, int B, int idf, int cdf,
                                     int T, float* src_out, void* stream) {
  if (!words || !w_ctx || !src_out || nsets < 1 || B < 1 || cdf < 1 || T < 1) return TGSR_EINVAL;
  if (nsets > 4 || T > 32 || idf < 32 || (idf & 31)) return TGSR_EUNSUPPORTED;
  ProjArgs a;
  a.words = words; a.out = src_out; a.B = B; a.idf = idf; a.cdf = cdf; a.T = T;
  for (int i = 0; i < 4; ++i) a.w[i] = i < nsets ? w_ctx[i] : nullptr;
  for (int i = 0; i < nsets; ++i)
    if (!a.w[i]) return TGSR_EINVAL;
  hipLaunchKernelGGL(word_project_mfma_kernel, dim3(B, nsets, idf / 32), dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "word_project_mfma_kernel");
}
