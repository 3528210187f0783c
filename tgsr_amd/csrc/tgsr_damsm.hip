// DAMSM word/region attention (func_attention, GlobalAttention.py:33-74) for gfx950, batched over every
// (image, caption) pair in ONE launch, fused with the cosine / log-sum-exp tail of words_loss (losses.py:73-113).
//
// The reference loops over captions in Python: for caption i it repeats the caption B times, runs
// bmm -> softmax over words -> transpose -> x gamma1 -> softmax over regions -> bmm against all B images, then
// cosine_similarity, exp, sum, log: ~12 small launches x B iterations.  Here one workgroup owns one (image j,
// caption i) pair and never leaves the chip:
//   A. S[l][s] = sum_d word_i[d][l] * ctx_j[d][s]        MFMA 32x32x2 f32: A = words from LDS (lane = word),
//                                                         B = ctx straight from HBM/L2 (lane = region, coalesced)
//      softmax over the words of each region              16 accumulator registers in-lane + one lane^32 exchange
//   B. x gamma1, softmax over the S regions of each word  rows of a [32][321] LDS image, wave shuffles
//   C. wc[d][l] = sum_s ctx_j[d][s] * attn[l][s]          MFMA: A = ctx re-staged per wave as [32 d][64 s] (pitch 65),
//                                                         B = attn rows from the LDS image (pitch 321: conflict-free)
//      cos(word_l, wc_l), log sum_l exp(gamma2 * cos)     in-lane over the accumulator rows + LDS atomics
// Pair enumeration: GRID (blockIdx -> (j, i), words of caption i, regions of image j; output sim[j][i] and the
// diagonal attention maps) or PAIRED (pair p uses query p and context p: the stand-alone func_attention API,
// outputs weightedContext and attn).
#include "tgsr_common.h"

namespace tgsr {

struct DamsmArgs {
  const float* words;      // [B][ndf][Tw]
  const int32_t* lens;     // [B] or null (= Tw)
  const float* ctx;        // [B][ndf][S]
  int B, ndf, Tw, S, paired;
  float gamma1, gamma2;
  float* sim;              // [B][B] (GRID) or null
  float* att_out;          // GRID: [B][Tw][S] written for j == i;  PAIRED: [B][Tw][S] for every pair;  or null
  float* wc_out;           // PAIRED: [B][ndf][Tw] or null
};

constexpr int kSP = 321;   // pitch of the attention image rows (S <= 320)

__global__ __launch_bounds__(256) void damsm_pair_kernel(DamsmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* word_s = lds;                        // [ndf][32]   words of the caption, zero for l >= L
  float* p_s = word_s + a.ndf * 32;           // [32][kSP]   attention image
  float* cs = p_s + 32 * kSP;                 // [4 waves][32][65] ctx staging for phase C
  float* red_s = cs + 4 * 32 * 65;            // [3][32]: dot, |wc|^2, |word|^2 per word

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int j, i;
  if (a.paired) {
    j = i = blockIdx.x;
  } else {
    j = blockIdx.x / a.B;
    i = blockIdx.x - j * a.B;
  }
  int L = a.lens ? a.lens[i] : a.Tw;
  L = L < 1 ? 1 : (L > a.Tw ? a.Tw : L);
  const int S = a.S, ndf = a.ndf;
  const float* wb = a.words + (int64_t)i * ndf * a.Tw;
  const float* cb = a.ctx + (int64_t)j * ndf * S;

  for (int o = tid; o < ndf * 32; o += 256) {
    const int d = o >> 5, l = o & 31;
    word_s[o] = l < L ? wb[d * a.Tw + l] : 0.f;
  }
  if (tid < 96) red_s[tid] = 0.f;
  __syncthreads();

  // ---- A: scores + softmax over words, 32 regions per wave pass
  const int nsb = (S + 31) / 32;
  for (int sb = wave; sb < nsb; sb += 4) {
    const int s = sb * 32 + l31;
    const bool sok = s < S;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < ndf / 2; k0 += 16) {
      float bv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) bv[k] = sok ? cb[(int64_t)(2 * (k0 + k) + hh) * S + s] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(word_s[(2 * (k0 + k) + hh) * 32 + l31], bv[k], acc, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (acc_row(r, hh) >= L) acc[r] = -INFINITY;
      mx = fmaxf(mx, acc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = __expf(acc[r] - mx);
      sum += acc[r];
    }
    sum += __shfl_xor(sum, 32);
    const float inv = a.gamma1 / sum;                       // softmax over words, then x gamma1 (GlobalAttention.py:56,64)
#pragma unroll
    for (int r = 0; r < 16; ++r) p_s[acc_row(r, hh) * kSP + s] = sok ? acc[r] * inv : 0.f;
  }
  // zero the padding columns [nsb*32, 320) that phase C multiplies
  for (int o = tid; o < 32 * (320 - nsb * 32); o += 256) {
    const int l = o / (320 - nsb * 32), c = o - l * (320 - nsb * 32);
    p_s[l * kSP + nsb * 32 + c] = 0.f;
  }
  __syncthreads();

  // ---- B: softmax over regions for each word row (8 rows per wave)
  for (int l = wave * 8; l < wave * 8 + 8; ++l) {
    float* row = p_s + l * kSP;
    if (l < L) {
      float v[5];
      float mx = -INFINITY;
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const int s = lane + 64 * m;
        v[m] = s < S ? row[s] : -INFINITY;
        mx = fmaxf(mx, v[m]);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        v[m] = __expf(v[m] - mx);
        sum += v[m];
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
      const float inv = 1.f / sum;
      const bool wr = a.att_out && (a.paired || i == j);
      float* ao = a.att_out ? a.att_out + ((int64_t)i * a.Tw + l) * S : nullptr;
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const int s = lane + 64 * m;
        if (s < S) {
          const float p = v[m] * inv;
          row[s] = p;
          if (wr) ao[s] = p;
        }
      }
    } else {
      for (int s = lane; s < 320; s += 64) row[s] = 0.f;
      if (a.att_out && (a.paired || i == j) && l < a.Tw) {
        float* ao = a.att_out + ((int64_t)i * a.Tw + l) * S;
        for (int s = lane; s < S; s += 64) ao[s] = 0.f;
      }
    }
  }
  __syncthreads();

  // ---- C: weighted context on MFMA, 32 feature rows per wave pass, then the cosine partial sums
  float* my_cs = cs + wave * 32 * 65;
  float dot = 0.f, nwc = 0.f, nwd = 0.f;
  for (int db = wave; db < ndf / 32; db += 4) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int sc = 0; sc < 5; ++sc) {
      __builtin_amdgcn_wave_barrier();
      for (int o = lane; o < 32 * 64; o += 64) {       // [32 d][64 s] <- ctx rows, coalesced along s
        const int r = o >> 6, c = o & 63;
        const int s = sc * 64 + c;
        my_cs[r * 65 + c] = s < S ? cb[(int64_t)(db * 32 + r) * S + s] : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll 8
      for (int k = 0; k < 32; ++k) {
        const float av = my_cs[l31 * 65 + 2 * k + hh];                    // A[d = l31][s]
        const float bv = p_s[l31 * kSP + sc * 64 + 2 * k + hh];           // B[s][l = l31]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
    }
    // acc: column = word l31, rows = features db*32 + acc_row(r, hh)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = db * 32 + acc_row(r, hh);
      const float w = word_s[d * 32 + l31];
      dot = fmaf(acc[r], w, dot);
      nwc = fmaf(acc[r], acc[r], nwc);
      nwd = fmaf(w, w, nwd);
      if (a.wc_out && l31 < L) a.wc_out[((int64_t)i * ndf + d) * a.Tw + l31] = acc[r];
    }
  }
  dot += __shfl_xor(dot, 32);
  nwc += __shfl_xor(nwc, 32);
  nwd += __shfl_xor(nwd, 32);
  // the four waves' partial sums meet in a fixed order (an LDS atomicAdd here made sim differ in the last bit run to run)
  float* wpart_s = red_s + 96;                 // [4 waves][96] (dynamic LDS: the launch reserves it)
  if (hh == 0) {
    wpart_s[wave * 96 + l31] = dot;
    wpart_s[wave * 96 + 32 + l31] = nwc;
    wpart_s[wave * 96 + 64 + l31] = nwd;
  }
  __syncthreads();
  if (tid < 96) red_s[tid] = ((wpart_s[tid] + wpart_s[96 + tid]) + wpart_s[192 + tid]) + wpart_s[288 + tid];
  __syncthreads();
  if (a.sim && wave == 0) {
    float e = 0.f;
    if (lane < L) {
      const float den = fmaxf(sqrtf(red_s[64 + lane]) * sqrtf(red_s[32 + lane]), 1e-8f);   // losses.py:12-18
      e = __expf(a.gamma2 * (red_s[lane] / den));                                            // losses.py:106
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) e += __shfl_xor(e, o);
    if (lane == 0) a.sim[(int64_t)j * a.B + i] = __logf(e);                                  // losses.py:107-108
  }
}

}  // namespace tgsr

using namespace tgsr;

static int damsm_launch(DamsmArgs a, void* stream) {
  if (!a.words || !a.ctx || a.B < 1 || a.ndf < 32 || a.Tw < 1 || a.S < 1) return TGSR_EINVAL;
  if (a.ndf % 32 != 0 || a.Tw > 32 || a.S > 320 || a.ndf > 512) return TGSR_EUNSUPPORTED;
  const size_t lds = sizeof(float) * ((size_t)a.ndf * 32 + 32 * kSP + 4 * 32 * 65 + 96 + 4 * 96);
  const int pairs = a.paired ? a.B : a.B * a.B;
  static bool attr_set[64] = {false};   // > 64 KB of dynamic LDS needs the opt-in once per DEVICE (a function attribute
  int dev = 0;                          // belongs to the device's code object, not to the process)
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(damsm_pair_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return note_launch(hipGetLastError(), "hipFuncSetAttribute(damsm_pair_kernel)");
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(damsm_pair_kernel, dim3(pairs), dim3(256), lds, as_stream(stream), a);
  return note_launch(hipGetLastError(), "damsm_pair_kernel");
}

extern "C" int tgsr_damsm_words_fwd(const float* words, const int32_t* cap_lens, const float* ctx, int B, int ndf,
                                    int Tw, int S, float gamma1, float gamma2, float* sim, float* att_diag,
                                    void* stream) {
  if (!sim) return TGSR_EINVAL;
  DamsmArgs a;
  a.words = words; a.lens = cap_lens; a.ctx = ctx; a.B = B; a.ndf = ndf; a.Tw = Tw; a.S = S; a.paired = 0;
  a.gamma1 = gamma1; a.gamma2 = gamma2; a.sim = sim; a.att_out = att_diag; a.wc_out = nullptr;
  return damsm_launch(a, stream);
}

extern "C" int tgsr_func_attention_fwd(const float* query, const float* context, int B, int ndf, int L, int S,
                                       float gamma1, float* weighted_context, float* attn, void* stream) {
  if (!weighted_context || !attn) return TGSR_EINVAL;
  DamsmArgs a;
  a.words = query; a.lens = nullptr; a.ctx = context; a.B = B; a.ndf = ndf; a.Tw = L; a.S = S; a.paired = 1;
  a.gamma1 = gamma1; a.gamma2 = 0.f; a.sim = nullptr; a.att_out = attn; a.wc_out = weighted_context;
  return damsm_launch(a, stream);
}
