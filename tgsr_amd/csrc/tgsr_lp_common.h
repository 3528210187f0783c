// Shared pieces of the reduced-precision (bf16 / f16 storage, fp32 accumulate) inference kernels - BASELINE.json
// configs[4].  Activations live in HBM as zero-bordered channels-last images
//     [B][H + 2][W + 2][cpitch]   (2-byte elements; border pixels are zero and are never written by any kernel),
// so a 3x3 / pad-1 convolution needs no bounds handling, a pixel's channels are one contiguous run (an MFMA k-slice of
// 8 channels = one 16-byte load) and `torch.cat((h_code, c_code), 1)` (util.py:771, 817) is a channel offset.
#pragma once
#include "tgsr_common.h"

namespace tgsr {

struct BF16 {};
struct F16 {};

typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(3))) void* lds_vptr_t;

template <class T>
struct LP;

template <>
struct LP<BF16> {
  static __device__ __forceinline__ f32x16v mfma32(u32x4 a, u32x4 b, f32x16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4w mfma16(u32x4 a, u32x4 b, f32x4w c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  // round-to-nearest-even pair (v_cvt_pk_bf16_f32); lo in bits 0..15
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
  }
  static __device__ __forceinline__ float lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
  static __device__ __forceinline__ float hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
  static __device__ __forceinline__ unsigned short one(float v) {
    return __builtin_bit_cast(unsigned short, (__bf16)v);
  }
};

template <>
struct LP<F16> {
  static __device__ __forceinline__ f32x16v mfma32(u32x4 a, u32x4 b, f32x16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4w mfma16(u32x4 a, u32x4 b, f32x4w c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) {   // v_cvt_f16_f32 x2: round-to-nearest-even
    f16x2 v;
    v[0] = (_Float16)lo;
    v[1] = (_Float16)hi;
    return __builtin_bit_cast(unsigned, v);
  }
  static __device__ __forceinline__ float lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
  static __device__ __forceinline__ float hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }
  static __device__ __forceinline__ unsigned short one(float v) {
    return __builtin_bit_cast(unsigned short, (_Float16)v);
  }
};

// One 1-KiB LDS-DMA piece: lane l copies 16 bytes from its own global address to lds_wave_base + 16 l.  Issued from
// inline assembly (hipcc would otherwise put s_waitcnt vmcnt(0) in front of every later ds_read); all ordering is the
// caller's explicit s_waitcnt vmcnt + barrier.
__device__ __forceinline__ void lds_dma16(const void* g, void* lds_wave_base) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory");
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): __frcp_rn expands to the 10-instruction IEEE division sequence, which made the
// GLU epilogue (64 gates per lane) half of the kernel's VALU work
__device__ __forceinline__ float sigmoidf_fast(float g) { return __builtin_amdgcn_rcpf(1.f + __expf(-g)); }

// What a kernel needs to attend to the words for the pixels it has just produced (GlobalAttentionGeneral.forward,
// GlobalAttention.py:87-130, fused into the epilogue of the kernel that computes h: util.py:768-771, 814-817).
struct LpAttFuse {
  const char* frag;        // [B][4 fragments][64 lanes][8] of T: the projected words as MFMA A fragments (tgsr_text_tail_lp_fwd)
  const uint32_t* mbits;   // [B] packed mask rows (bit t = masked) or null
  float* attn;             // [B][T][H*W] fp32 or null
  int T, mask_mode, coff;  // words (<= 32); 0 = the reference's mask quirk, 1 = per-sample; first channel of c_code in `out`
};

// One 32-pixel tile of the word attention on MFMA 32x32x16 - THE arithmetic of the reduced-precision attention, shared by the
// stand-alone kernel (lp_word_attention_kernel) and the producers that attend in their epilogue (lp_stem_kernel,
// lp_upconv_glu_kernel), so every route gives the same bits.  Lane = (pixel l31, half hh):
//   S[t][q] = sum_i src[i][t] h[i][q]   (fa[0], fa[1]; b0 / b1 = channels 8 hh .. and 16 + 8 hh .. of the lane's pixel)
//   mask (bit t of mb), softmax over the 32 word rows: 16 in this lane, 16 in lane ^ 32
//   C[i][q] = sum_t src[i][t] P[t][q]   (fa[2], fa[3]; P rounded to T and consumed from the accumulator registers)
// attn_q = &attn[b][0][q] or null (row stride Q); cp = address of channel `coff` of the lane's pixel in the output image.
template <class T>
__device__ __forceinline__ void lp_attend_tile(const u32x4 (&fa)[4], u32x4 b0, u32x4 b1, unsigned mb, int Tw, int hh,
                                               float* __restrict__ attn_q, int64_t Q, char* __restrict__ cp) {
  f32x16v s;
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = 0.f;
  s = LP<T>::mfma32(fa[0], b0, s);
  s = LP<T>::mfma32(fa[1], b1, s);
  const unsigned valid = (Tw >= 32 ? 0xffffffffu : ((1u << Tw) - 1u)) & ~mb;
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
    s[i] = __expf(s[i] - mx);
    sum += s[i];
  }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] *= inv;
  if (attn_q) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = acc_row(i, hh);
      if (t < Tw) attn_q[(int64_t)t * Q] = s[i];
    }
  }
  // weighted context: B operand of k-step ks = registers 8 ks .. 8 ks + 7 of P, rounded to T (their word order is the one
  // the context GEMM's A fragments were packed in)
  u32x4 p0, p1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p0[j] = LP<T>::pack2(s[2 * j], s[2 * j + 1]);
    p1[j] = LP<T>::pack2(s[8 + 2 * j], s[8 + 2 * j + 1]);
  }
  f32x16v c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = LP<T>::mfma32(fa[2], p0, c);
  c = LP<T>::mfma32(fa[3], p1, c);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    u32x2 pk;
    pk[0] = LP<T>::pack2(c[4 * rg], c[4 * rg + 1]);
    pk[1] = LP<T>::pack2(c[4 * rg + 2], c[4 * rg + 3]);
    *reinterpret_cast<u32x2*>(cp + (8 * rg + 4 * hh) * 2) = pk;
  }
}

// `att_pack` (tgsr_text_tail_lp_fwd) -> the fused-attention arguments of one stage
static inline int lp_att_fuse(const void* att_pack, int att_nsets, int att_set, int B, int use_mask, int mask_mode, int T, int c_coff,
                       float* attn, LpAttFuse* f) {
  if (!att_pack || att_nsets < 1 || att_set < 0 || att_set >= att_nsets || (reinterpret_cast<uintptr_t>(att_pack) & 15))
    return TGSR_EINVAL;
  const char* p = static_cast<const char*>(att_pack);
  f->frag = p + (size_t)att_set * B * 4096;
  f->mbits = use_mask ? reinterpret_cast<const uint32_t*>(p + (size_t)att_nsets * B * 4096) : nullptr;
  f->attn = attn; f->T = T; f->mask_mode = mask_mode; f->coff = c_coff;
  return TGSR_OK;
}


}  // namespace tgsr
