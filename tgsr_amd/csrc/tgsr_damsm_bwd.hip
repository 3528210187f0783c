// Backward of the DAMSM word/region similarity grid (tgsr_damsm_words_fwd; words_loss, losses.py:73-113 through
// func_attention, GlobalAttention.py:33-74) for gfx950.
//
// One workgroup (256 threads) owns one (image j, caption i) pair, recomputes the pair's forward quantities and
// pushes G = dLoss/dsim[j][i] back to the caption's words and the image's region features:
//   s[l][r]   = sum_d word[d][l] ctx[d][r]              a1 = softmax_l(s)   a2 = softmax_r(gamma1 * a1)
//   wc[d][l]  = sum_r ctx[d][r] a2[l][r]                cos_l = <word_l, wc_l> / max(|word_l| |wc_l|, 1e-8)
//   sim       = log sum_l exp(gamma2 cos_l)
// Two thread mappings alternate: thread = region r (scores, both softmax backward passes; ctx arrives in 32-row slabs
// through LDS, the [ndf][32] word / dwc matrices are LDS broadcasts) and thread = (feature d, half of the words)
// (weighted context and every gradient row; each thread keeps its 16 words of word, wc, dwc, gword in registers, ctx and the
// chunk's columns of the [32][S] attention images are staged through LDS in chunks of 32 regions).
// Per-pair gradients go to gw_part[j][i][ndf][32] and gc_part[i][j][ndf][S]; tgsr_reduce_dim0 sums them in a fixed
// order (deterministic, no float atomics).  A loss kernel (28 MFLOP per pair) on the vector units - which ran 0.96 ms ALONE on the
// device in the G/D + DAMSM step and is a quarter of a DAMSM pre-training step: round 6 took its dependent global loads out
// (0.69 ms with its reductions) and gave a pair 512 threads instead of 256 (two waves per SIMD).
#include "tgsr_common.h"

namespace tgsr {

struct DamsmBwdArgs {
  const float* words;      // [B][ndf][Tw]
  const int32_t* lens;     // [B] or null
  const float* ctx;        // [B][ndf][S]
  const float* gsim;       // [B img][B cap]
  int B, ndf, Tw, S;
  float gamma1, gamma2;
  float* ws;               // [B*B][3][32][S] workspace: a1, a2, da2 -> ds
  float* gw_part;          // [B img][B cap][ndf][32]
  float* gc_part;          // [B cap][B img][ndf][S]
};

constexpr int kBwdChunk = 32;   // regions per LDS chunk in the thread = d phases
constexpr int kBwdNT = 512;     // threads per pair: 8 waves = 2 per SIMD (one wave per SIMD issues a VALU instruction every 4 cycles, two
                                // fill the 2-cycle slots: the kernel is VALU / LDS issue bound)

// Thread mappings (round 6: 512 threads per pair instead of 256):
//   thread = region r (tid < S <= 320: five waves, one pass): scores + softmax over words, the da2 products, softmax-r backward;
//   thread = (feature d = tid & 255, word half h = tid >> 8): weighted context, cosine / log-sum-exp backward, every gradient row -
//     a thread keeps its 16 words of word / wc / dwc / gword in registers; the two halves of a gctx value meet in LDS (h = 0 first).
// Reductions over d per word: the four waves that hold a word's 256 features (waves 4 h .. 4 h + 3), combined in a fixed order.
__global__ __launch_bounds__(kBwdNT) void damsm_pair_bwd_kernel(DamsmBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ndf = a.ndf, S = a.S;
  float* word_s = lds;                       // [ndf][32]
  float* dwc_s = word_s + ndf * 32;          // [ndf][32]
  float* cs = dwc_s + ndf * 32;              // [ndf][33] ctx chunk
  float* gcs = cs + ndf * 33;                // [ndf][33] gctx chunk
  // (cs .. gcs also hold the region-mapped phases' [32][S] ctx slab: the region is as large as the larger of the two uses)
  const int csz = 2 * ndf * 33 > 32 * S ? 2 * ndf * 33 : 32 * S;
  float* red = cs + csz;                     // [4][32] reductions
  float* wpart = red + 128;                  // [8 waves][128]: the waves' partial sums of those, combined in a fixed order
  // the chunk's columns of the attention images, [region][32 words]: the thread = d phases read every (word, region) value of the
  // chunk in EVERY thread - as LDS broadcasts instead of dependent global loads from the per-pair workspace
  float* a2s = wpart + 8 * 128;              // [32 regions][32]
  float* a3s = a2s + kBwdChunk * 32;         // [32 regions][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x / a.B, i = blockIdx.x - j * a.B;
  int L = a.lens ? a.lens[i] : a.Tw;
  L = L < 1 ? 1 : (L > a.Tw ? a.Tw : L);
  const float* wb = a.words + (int64_t)i * ndf * a.Tw;
  const float* cb = a.ctx + (int64_t)j * ndf * S;
  const float G = a.gsim[(int64_t)j * a.B + i];
  float* a1 = a.ws + (int64_t)blockIdx.x * 3 * 32 * S;
  float* a2 = a1 + 32 * S;
  float* a3 = a2 + 32 * S;

  for (int o = tid; o < ndf * 32; o += kBwdNT) {
    const int d = o >> 5, l = o & 31;
    word_s[o] = l < L ? wb[d * a.Tw + l] : 0.f;
  }
  if (tid < 128) red[tid] = 0.f;
  __syncthreads();

  // ---- thread = r: scores s[l][r] = sum_d word[d][l] ctx[d][r] (ctx in slabs of 32 feature rows through LDS), softmax over words
  float* slab = cs;
  {
    float s[32];
#pragma unroll
    for (int l = 0; l < 32; ++l) s[l] = 0.f;
    for (int d0 = 0; d0 < ndf; d0 += 32) {
      for (int o = tid; o < 32 * S; o += kBwdNT) slab[o] = cb[(int64_t)d0 * S + o];
      __syncthreads();
      if (tid < S) {
        for (int dd = 0; dd < 32; ++dd) {
          const float c = slab[dd * S + tid];
          const float4* wp = reinterpret_cast<const float4*>(word_s + (d0 + dd) * 32);
#pragma unroll
          for (int l4 = 0; l4 < 8; ++l4) {
            const float4 w4 = wp[l4];
            s[4 * l4] = fmaf(w4.x, c, s[4 * l4]); s[4 * l4 + 1] = fmaf(w4.y, c, s[4 * l4 + 1]);
            s[4 * l4 + 2] = fmaf(w4.z, c, s[4 * l4 + 2]); s[4 * l4 + 3] = fmaf(w4.w, c, s[4 * l4 + 3]);
          }
        }
      }
      __syncthreads();
    }
    if (tid < S) {
      const int r = tid;
      float mx = -INFINITY;
#pragma unroll
      for (int l = 0; l < 32; ++l) {
        if (l >= L) s[l] = -INFINITY;
        mx = fmaxf(mx, s[l]);
      }
      float sum = 0.f;
#pragma unroll
      for (int l = 0; l < 32; ++l) {
        s[l] = expf(s[l] - mx);
        sum += s[l];
      }
      const float inv = 1.f / sum;
#pragma unroll
      for (int l = 0; l < 32; ++l) {
        const float p = s[l] * inv;
        a1[l * S + r] = p;
        a2[l * S + r] = a.gamma1 * p;        // softmax over regions comes next
      }
    }
  }
  __syncthreads();

  // ---- wave per word row: softmax over regions -> a2 (a lane's <= 5 values of the row stay in registers between the passes; S <= 320)
  for (int l = wave; l < 32; l += kBwdNT / 64) {
    float* row = a2 + l * S;
    if (l < L) {
      float v[5];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int r = lane + 64 * k;
        v[k] = r < S ? row[r] : -INFINITY;
        mx = fmaxf(mx, v[k]);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        v[k] = lane + 64 * k < S ? expf(v[k] - mx) : 0.f;
        sum += v[k];
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
      const float inv = 1.f / sum;
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if (lane + 64 * k < S) row[lane + 64 * k] = v[k] * inv;
    } else {
      for (int r = lane; r < S; r += 64) row[r] = 0.f;
    }
  }
  __syncthreads();

  // ---- thread = (d, h): weighted context wc[d][l] for the 16 words of half h (ctx and a2 staged through LDS in chunks of 32 regions)
  const int d = tid & 255, h = tid >> 8, l0 = 16 * h;
  const bool dok = d < ndf;
  float wrow[16], wc[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) {
    wrow[l] = dok ? word_s[d * 32 + l0 + l] : 0.f;
    wc[l] = 0.f;
  }
  for (int r0 = 0; r0 < S; r0 += kBwdChunk) {
    for (int o = tid; o < ndf * kBwdChunk; o += kBwdNT) {
      const int dd = o / kBwdChunk, rr = o - dd * kBwdChunk;
      cs[dd * 33 + rr] = r0 + rr < S ? cb[(int64_t)dd * S + r0 + rr] : 0.f;
    }
    for (int o = tid; o < 32 * kBwdChunk; o += kBwdNT) {    // a2[l][r0 + rr] -> a2s[rr][l] (coalesced over rr)
      const int l = o / kBwdChunk, rr = o - l * kBwdChunk;
      a2s[rr * 32 + l] = r0 + rr < S ? a2[l * S + r0 + rr] : 0.f;
    }
    __syncthreads();
    if (dok) {
      const int nr = S - r0 < kBwdChunk ? S - r0 : kBwdChunk;
      for (int rr = 0; rr < nr; ++rr) {
        const float c = cs[d * 33 + rr];
        const float4* ap = reinterpret_cast<const float4*>(a2s + rr * 32 + l0);
#pragma unroll
        for (int l4 = 0; l4 < 4; ++l4) {
          const float4 av = ap[l4];
          wc[4 * l4] = fmaf(c, av.x, wc[4 * l4]);
          wc[4 * l4 + 1] = fmaf(c, av.y, wc[4 * l4 + 1]);
          wc[4 * l4 + 2] = fmaf(c, av.z, wc[4 * l4 + 2]);
          wc[4 * l4 + 3] = fmaf(c, av.w, wc[4 * l4 + 3]);
        }
      }
    }
    __syncthreads();
  }

  // ---- cosine + log-sum-exp backward: per word dot / norms over d (a word's features sit in the four waves of its half)
  {
#pragma unroll
    for (int l = 0; l < 16; ++l) {
      float dt = wrow[l] * wc[l], nc = wc[l] * wc[l], nw = wrow[l] * wrow[l];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        dt += __shfl_xor(dt, o);
        nc += __shfl_xor(nc, o);
        nw += __shfl_xor(nw, o);
      }
      if (lane == 0) {
        wpart[wave * 128 + l0 + l] = dt;
        wpart[wave * 128 + 32 + l0 + l] = nc;
        wpart[wave * 128 + 64 + l0 + l] = nw;
      }
    }
  }
  __syncthreads();
  if (tid < 96) {                            // word tid & 31 lives in half (tid & 31) >> 4: waves 4 half .. 4 half + 3, in that order
    const int w0 = 4 * ((tid & 31) >> 4);
    red[tid] = ((wpart[w0 * 128 + tid] + wpart[(w0 + 1) * 128 + tid]) + wpart[(w0 + 2) * 128 + tid]) + wpart[(w0 + 3) * 128 + tid];
  }
  __syncthreads();
  float gw[16], dwc[16];
  {
    // p_l = softmax_l(gamma2 cos_l) over the caption's words; dcos_l = G gamma2 p_l
    float mx = -INFINITY;
    for (int l = 0; l < 32; ++l) {
      if (l < L) {
        const float den = fmaxf(sqrtf(red[64 + l]) * sqrtf(red[32 + l]), 1e-8f);
        mx = fmaxf(mx, a.gamma2 * (red[l] / den));
      }
    }
    float sum = 0.f;
    for (int l = 0; l < 32; ++l) {
      if (l < L) {
        const float den = fmaxf(sqrtf(red[64 + l]) * sqrtf(red[32 + l]), 1e-8f);
        sum += expf(a.gamma2 * (red[l] / den) - mx);
      }
    }
#pragma unroll
    for (int l = 0; l < 16; ++l) {
      const int lw = l0 + l;
      float gwl = 0.f, dw = 0.f;
      if (lw < L) {
        const float nw = sqrtf(red[64 + lw]), nc = sqrtf(red[32 + lw]);
        const float cosv = red[lw] / fmaxf(nw * nc, 1e-8f);
        const float dcos = G * a.gamma2 * expf(a.gamma2 * cosv - mx) / sum;
        if (nw * nc > 1e-8f) {
          const float inv = 1.f / (nw * nc);
          dw = dcos * (wrow[l] * inv - cosv * wc[l] / (nc * nc));
          gwl = dcos * (wc[l] * inv - cosv * wrow[l] / (nw * nw));
        } else {                                  // clamped denominator: cos = dot / eps
          dw = dcos * wrow[l] * 1e8f;
          gwl = dcos * wc[l] * 1e8f;
        }
      }
      dwc[l] = dw;
      gw[l] = gwl;
      if (dok) dwc_s[d * 32 + lw] = dw;
    }
  }
  __syncthreads();

  // ---- thread = r, pass A: da2[l][r] = sum_d dwc[d][l] ctx[d][r]; row dots sum_r da2 a2 for the softmax-r backward
  {
    float g[32];
#pragma unroll
    for (int l = 0; l < 32; ++l) g[l] = 0.f;
    for (int d0 = 0; d0 < ndf; d0 += 32) {                 // ctx in slabs of 32 rows through LDS, as for the scores
      for (int o = tid; o < 32 * S; o += kBwdNT) slab[o] = cb[(int64_t)d0 * S + o];
      __syncthreads();
      if (tid < S) {
        for (int dd = 0; dd < 32; ++dd) {
          const float c = slab[dd * S + tid];
          const float4* wp = reinterpret_cast<const float4*>(dwc_s + (d0 + dd) * 32);
#pragma unroll
          for (int l4 = 0; l4 < 8; ++l4) {
            const float4 w4 = wp[l4];
            g[4 * l4] = fmaf(w4.x, c, g[4 * l4]); g[4 * l4 + 1] = fmaf(w4.y, c, g[4 * l4 + 1]);
            g[4 * l4 + 2] = fmaf(w4.z, c, g[4 * l4 + 2]); g[4 * l4 + 3] = fmaf(w4.w, c, g[4 * l4 + 3]);
          }
        }
      }
      __syncthreads();
    }
    float rd[32];
#pragma unroll
    for (int l = 0; l < 32; ++l) {
      rd[l] = 0.f;
      if (tid < S) {
        a3[l * S + tid] = g[l];
        rd[l] = g[l] * a2[l * S + tid];
      }
    }
#pragma unroll
    for (int l = 0; l < 32; ++l) {
      float v = rd[l];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) wpart[wave * 128 + 96 + l] = v;
    }
  }
  __syncthreads();
  if (tid < 32)                              // the regions sit in waves 0 .. 4 (S <= 320): combined in that order
    red[96 + tid] = (((wpart[96 + tid] + wpart[128 + 96 + tid]) + wpart[256 + 96 + tid]) + wpart[384 + 96 + tid]) + wpart[512 + 96 + tid];
  __syncthreads();
  // pass B: dx = a2 (da2 - rowdot); da1 = gamma1 dx; ds = a1 (da1 - sum_l da1 a1)  -> a3
  if (tid < S) {
    const int r = tid;
    float da1[32], p1[32], acc = 0.f;
#pragma unroll
    for (int l = 0; l < 32; ++l) {
      p1[l] = a1[l * S + r];
      da1[l] = l < L ? a.gamma1 * a2[l * S + r] * (a3[l * S + r] - red[96 + l]) : 0.f;
      acc = fmaf(da1[l], p1[l], acc);
    }
#pragma unroll
    for (int l = 0; l < 32; ++l) a3[l * S + r] = p1[l] * (da1[l] - acc);
  }
  __syncthreads();

  // ---- thread = (d, h): gword[d][l] += sum_r ds[l][r] ctx[d][r];  gctx[d][r] = sum_l (word[d][l] ds[l][r] + dwc[d][l] a2[l][r])
  float* gc = a.gc_part + ((int64_t)i * a.B + j) * ndf * S;
  for (int r0 = 0; r0 < S; r0 += kBwdChunk) {
    for (int o = tid; o < ndf * kBwdChunk; o += kBwdNT) {
      const int dd = o / kBwdChunk, rr = o - dd * kBwdChunk;
      cs[dd * 33 + rr] = r0 + rr < S ? cb[(int64_t)dd * S + r0 + rr] : 0.f;
    }
    for (int o = tid; o < 32 * kBwdChunk; o += kBwdNT) {
      const int l = o / kBwdChunk, rr = o - l * kBwdChunk;
      const bool in = r0 + rr < S;
      a2s[rr * 32 + l] = in ? a2[l * S + r0 + rr] : 0.f;
      a3s[rr * 32 + l] = in ? a3[l * S + r0 + rr] : 0.f;
    }
    __syncthreads();
    const int nr = S - r0 < kBwdChunk ? S - r0 : kBwdChunk;
    float gpart[kBwdChunk];
    if (dok) {
#pragma unroll
      for (int rr = 0; rr < kBwdChunk; ++rr) {
        float g = 0.f;
        if (rr < nr) {
          const float c = cs[d * 33 + rr];
          const float4* p2 = reinterpret_cast<const float4*>(a2s + rr * 32 + l0);
          const float4* p3 = reinterpret_cast<const float4*>(a3s + rr * 32 + l0);
#pragma unroll
          for (int l4 = 0; l4 < 4; ++l4) {
            const float4 d4 = p3[l4], q4 = p2[l4];
            const float dsv[4] = {d4.x, d4.y, d4.z, d4.w}, a2v[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int l = 4 * l4 + k;
              gw[l] = fmaf(dsv[k], c, gw[l]);
              g = fmaf(wrow[l], dsv[k], g);
              g = fmaf(dwc[l], a2v[k], g);
            }
          }
        }
        gpart[rr] = g;
      }
      if (h == 0) {
#pragma unroll
        for (int rr = 0; rr < kBwdChunk; ++rr) gcs[d * 33 + rr] = gpart[rr];
      }
    }
    __syncthreads();
    if (dok && h == 1) {                     // the second half of the words on top of the first: a fixed order
#pragma unroll
      for (int rr = 0; rr < kBwdChunk; ++rr) gcs[d * 33 + rr] += gpart[rr];
    }
    __syncthreads();
    for (int o = tid; o < ndf * kBwdChunk; o += kBwdNT) {
      const int dd = o / kBwdChunk, rr = o - dd * kBwdChunk;
      if (rr < nr) gc[(int64_t)dd * S + r0 + rr] = gcs[dd * 33 + rr];
    }
    __syncthreads();
  }
  if (dok) {
    float* gwp = a.gw_part + (((int64_t)j * a.B + i) * ndf + d) * 32 + l0;
#pragma unroll
    for (int l = 0; l < 16; ++l) gwp[l] = gw[l];
  }
}

// out[k] = sum_{p < n} parts[p][k], p ascending (deterministic)
__global__ void reduce_dim0_kernel(const float* __restrict__ parts, int n, int64_t m, float* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < n; ++p) s += parts[(int64_t)p * m + k];
    out[k] = s;
  }
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int64_t tgsr_damsm_words_bwd_ws_elems(int B, int ndf, int S) {
  return (int64_t)B * B * (3 * 32 * (int64_t)S + (int64_t)ndf * 32 + (int64_t)ndf * S);
}

extern "C" int tgsr_damsm_words_bwd(const float* words, const int32_t* cap_lens, const float* ctx, const float* grad_sim,
                                    int B, int ndf, int Tw, int S, float gamma1, float gamma2, float* ws,
                                    float* grad_words32, float* grad_ctx, void* stream) {
  if (!words || !ctx || !grad_sim || !ws || !grad_words32 || !grad_ctx || B < 1 || Tw < 1 || S < 1) return TGSR_EINVAL;
  if (ndf < 32 || ndf > 256 || ndf % 32 != 0 || Tw > 32 || S > 320) return TGSR_EUNSUPPORTED;
  DamsmBwdArgs a;
  a.words = words; a.lens = cap_lens; a.ctx = ctx; a.gsim = grad_sim; a.B = B; a.ndf = ndf; a.Tw = Tw; a.S = S;
  a.gamma1 = gamma1; a.gamma2 = gamma2;
  a.ws = ws;
  a.gw_part = ws + (int64_t)B * B * 3 * 32 * S;
  a.gc_part = a.gw_part + (int64_t)B * B * ndf * 32;
  const size_t csz = (size_t)ndf * 33 * 2 > (size_t)32 * S ? (size_t)ndf * 33 * 2 : (size_t)32 * S;
  const size_t lds = sizeof(float) * ((size_t)ndf * 32 * 2 + csz + 128 + 8 * 128 + 2 * kBwdChunk * 32);
  static bool attr_set[64] = {false};   // the > 64 KB opt-in belongs to the DEVICE's code object: once per device, not per
  int dev = 0;                          // process (one process driving two GPUs would otherwise fail on the second)
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(damsm_pair_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return note_launch(hipGetLastError(), "hipFuncSetAttribute(damsm_pair_bwd_kernel)");
    attr_set[dev] = true;
  }
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(damsm_pair_bwd_kernel, dim3(B * B), dim3(kBwdNT), lds, s, a);
  int rc = note_launch(hipGetLastError(), "damsm_pair_bwd_kernel");
  if (rc) return rc;
  // grad_words32[i][ndf][32] = sum_j gw_part[j][i];  grad_ctx[j][ndf][S] = sum_i gc_part[i][j]
  const int64_t mw = (int64_t)B * ndf * 32, mc = (int64_t)B * ndf * S;
  hipLaunchKernelGGL(reduce_dim0_kernel, dim3((unsigned)((mw + 255) / 256 < 1024 ? (mw + 255) / 256 : 1024)), dim3(256),
                     0, s, a.gw_part, B, mw, grad_words32);
  rc = note_launch(hipGetLastError(), "reduce_dim0_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_dim0_kernel, dim3((unsigned)((mc + 255) / 256 < 1024 ? (mc + 255) / 256 : 1024)), dim3(256),
                     0, s, a.gc_part, B, mc, grad_ctx);
  return note_launch(hipGetLastError(), "reduce_dim0_kernel");
}
