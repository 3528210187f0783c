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

}  // namespace tgsr
