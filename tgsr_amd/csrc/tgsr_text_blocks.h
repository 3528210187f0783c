// Device-side building blocks of the text tail (word projection, CA_NET), shared by the stand-alone kernels
// (tgsr_attention.hip, tgsr_gemm.hip) and by the one-launch text_tail_kernel (tgsr_text_tail.hip): the same code, hence
// the same bits, whichever launch a caller takes.
#pragma once
#include "tgsr_common.h"

namespace tgsr {

// The conv1x1 of GlobalAttentionGeneral (GlobalAttention.py:100-102) on the MFMA units for up to 4 weight sets in ONE
// launch (the generator stages attend to the same words through different conv_context weights): one workgroup per
// (sample b, set, 32-channel block ib), 4 waves, each a quarter of the channel pairs of
// src[i][t] = sum_c W[i][c] words[c][t] (A = W, lane = i; B = words, lane = t), summed through LDS in a fixed order.
struct ProjArgs {
  const float* words;
  const float* w[4];
  float* out;            // [nsets][B][idf][32]
  int B, idf, cdf, T;
  // optional (idf == 32 only): the same projections rounded to a 2-byte type and laid out as the four MFMA 32x32x16 A
  // fragments the reduced-precision attention consumes ([nsets][B][4 fragments][64 lanes][8], tgsr_lp_misc.hip): fragments
  // 0, 1 = scores GEMM (row = word t, k = channel 16 f + 8 (lane >> 5) + j), 2, 3 = context GEMM (row = channel, k = word
  // 16 (f - 2) + 8 (j >> 2) + 4 (lane >> 5) + (j & 3): the order the softmax leaves in the accumulator registers).  A
  // kernel that attends in its epilogue (lp_stem_kernel, lp_upconv_glu_kernel) loads them with one 16-byte load per lane.
  unsigned short* frag = nullptr;
  int frag_dt = 0;       // TGSR_DT_BF16 | TGSR_DT_F16
};

__device__ __forceinline__ unsigned short lp_round_one(float v, int dt) {
  return dt == TGSR_DT_BF16 ? __builtin_bit_cast(unsigned short, (__bf16)v) : __builtin_bit_cast(unsigned short, (_Float16)v);
}

constexpr int kProjRedFloats = 4 * 16 * 64;

__device__ __forceinline__ void word_project_block(const ProjArgs& a, int b, int set, int ib, float* red_raw) {
  float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(red_raw);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5, wave = tid >> 6;
  const float* wr = a.w[set] + (int64_t)(ib * 32 + l31) * a.cdf;
  const float* wb = a.words + (int64_t)b * a.cdf * a.T;
  const bool tok = l31 < a.T;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nk = (a.cdf + 1) >> 1;
  for (int ks = wave; ks < nk; ks += 32) {               // 8 k-steps of this wave per trip: their loads go out together
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 2 * (ks + 4 * j) + hh;
      const bool ok = ks + 4 * j < nk && c < a.cdf;
      av[j] = ok ? wr[c] : 0.f;
      bv[j] = ok && tok ? wb[c * a.T + l31] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  float* ob = a.out + ((int64_t)(set * a.B + b) * a.idf + ib * 32) * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = tid + 256 * j, r = o >> 6, ln = o & 63;
    const float v = red[0][r][ln] + red[1][r][ln] + red[2][r][ln] + red[3][r][ln];
    ob[acc_row(r, ln >> 5) * 32 + (ln & 31)] = v;
    if (a.frag) {                                             // idf == 32 (checked by the launcher): ib == 0
      const int i = acc_row(r, ln >> 5), t = ln & 31;
      unsigned short* fb = a.frag + (int64_t)(set * a.B + b) * (4 * 64 * 8);
      const unsigned short q = lp_round_one(v, a.frag_dt);
      fb[(((i >> 4) * 64 + ((i >> 3) & 1) * 32 + t) << 3) + (i & 7)] = q;
      fb[(((2 + (t >> 4)) * 64 + ((t >> 2) & 1) * 32 + i) << 3) + 4 * ((t >> 3) & 1) + (t & 3)] = q;
    }
  }
}

// CA_NET (util.py:372-400) for 16 samples x 4 condition channels per workgroup (256 threads):
//   y = fc(sent) [4 ncf];  (mu | logvar) = y[:2ncf] * sigmoid(y[2ncf:]);  c_code = eps * exp(logvar / 2) + mu.
// Y[16 rows][16 samples] on v_mfma_f32_16x16x4_f32: A = 16 rows of the Linear's weight (slot r = 4 * (i - i0) + kind:
// the four rows one output channel i needs - i, 2ncf + i (mu and its gate), ncf + i, 3ncf + i (logvar and its gate) -
// land in the four accumulator registers of ONE lane), B = the sentence codes (lane = sample).  Wave w sums the
// quarter [w K/4, (w+1) K/4) of the K = tdim products, lane group g = lane >> 4 a contiguous K/16 run of it (float4
// loads of both operands); the four partial tiles meet in LDS in a fixed order.  The weight rows are read once per
// 16 samples (the row-per-thread form read them once per sample, one 1-KB-strided row per lane: 12 us at B = 16).
struct CaArgs {
  const float* sent;     // [B][tdim]
  const float* w;        // [4 ncf][tdim]
  const float* bias;     // [4 ncf]
  const float* eps;      // [B][ncf] or null
  float* c_code;         // [B][ncf] or null
  float* mu;             // [B][ncf]
  float* logvar;         // [B][ncf]
  int B, tdim, ncf;
};

constexpr int kCaRedFloats = 4 * 4 * 64;

// requires tdim % 16 == 0; 16-byte aligned rows when tdim % 64 == 0 (checked by the launchers)
__device__ __forceinline__ void ca_net_block(const CaArgs& a, int iblk, int sblk, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
  const int kc = a.tdim >> 4;                               // products per (wave, lane group)
  const int k0 = wave * (a.tdim >> 2) + g * kc;
  const int i0 = iblk * 4, smp = sblk * 16 + n;
  // A: slot r = lane & 15 -> channel i0 + (r >> 2), kind r & 3
  const int ia = i0 + (n >> 2), kind = n & 3;
  const bool a_ok = ia < a.ncf, b_ok = smp < a.B;
  const int row = (kind & 2 ? a.ncf : 0) + (kind & 1 ? 2 * a.ncf : 0) + (a_ok ? ia : 0);
  const float* wr = a.w + (int64_t)row * a.tdim + k0;
  const float* xr = a.sent + (int64_t)(b_ok ? smp : 0) * a.tdim + k0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if ((kc & 3) == 0) {
    for (int s = 0; s < kc; s += 16) {                      // up to 4 float4 of each operand in flight
      float4 av[4], bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = s + 4 * j < kc;
        av[j] = in && a_ok ? *reinterpret_cast<const float4*>(wr + s + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[j] = in && b_ok ? *reinterpret_cast<const float4*>(xr + s + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc, 0, 0, 0);
      }
    }
  } else {
    for (int s = 0; s < kc; ++s)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? wr[s] : 0.f, b_ok ? xr[s] : 0.f, acc, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[(wave * 4 + j) * 64 + lane] = acc[j];
  __syncthreads();
  if (tid < 64) {
    // D: lane (sample n, channel i0 + g) holds kinds 0..3 in its four registers
    const int i = i0 + g;
    if (i < a.ncf && b_ok) {
      float y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rw = (j & 2 ? a.ncf : 0) + (j & 1 ? 2 * a.ncf : 0) + i;
        y[j] = ((red[j * 64 + lane] + red[(4 + j) * 64 + lane]) + (red[(8 + j) * 64 + lane] + red[(12 + j) * 64 + lane])) +
               a.bias[rw];
      }
      const float m = y[0] * (1.f / (1.f + __expf(-y[1])));
      const float lv = y[2] * (1.f / (1.f + __expf(-y[3])));
      const int64_t o = (int64_t)smp * a.ncf + i;
      a.mu[o] = m;
      a.logvar[o] = lv;
      if (a.c_code) a.c_code[o] = a.eps[o] * __expf(0.5f * lv) + m;
    }
  }
}

}  // namespace tgsr
