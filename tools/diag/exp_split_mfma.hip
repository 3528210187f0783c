// Experiment (not product code): can fp32 matrix products run on the bf16 matrix pipe?
//
// On gfx950 `v_mfma_f32_16x16x4_f32` runs at 64 FLOP/clk/SIMD - the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate - and every
// VALU instruction beside it costs its full issue time (profiles/HISTORY.md 3.1c / 3.1e measured ~7.7 cycles each): the fp32 convolution
// kernels are bound by that.  An fp32 number is exactly the sum of three bf16 numbers (round-to-nearest split: 9 + 9 + 9 >= 24
// significand bits), a bf16 x bf16 product is exact in fp32, and the bf16 MFMA accumulates in fp32 - so
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0) + [a1 b2 + a2 b1 + a2 b2]
// with the bracket <= 2^-26 |a b|: six bf16 MFMAs ("x6") instead of one fp32 MFMA of the same K, at 16x the rate each.
// This program measures (1) the error of x3 / x6 / x9 dot products against fp64, beside the fp32 MFMA's own, and (2) the
// sustained rate of each form with every SIMD busy, with and without VALU work beside the MFMAs.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/diag/bin/exp_split_mfma tools/diag/exp_split_mfma.hip && tools/diag/bin/exp_split_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ uint32_t rne_bf16_bits(float x) {   // upper 16 bits of x rounded to nearest even (no NaN handling)
  const uint32_t u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
// x = p[0] + p[1] + p[2] (+ residual), each a bf16 value held as fp32 bits; RNE: round to nearest, else truncate
template <bool RNE>
__device__ __forceinline__ void split3(float x, uint32_t (&p)[3]) {
  p[0] = RNE ? rne_bf16_bits(x) : (__float_as_uint(x) & 0xffff0000u);
  const float r1 = x - __uint_as_float(p[0]);
  p[1] = RNE ? rne_bf16_bits(r1) : (__float_as_uint(r1) & 0xffff0000u);
  const float r2 = r1 - __uint_as_float(p[1]);
  p[2] = RNE ? rne_bf16_bits(r2) : (__float_as_uint(r2) & 0xffff0000u);
}
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------- numerics
// One wave per 16 x 16 output tile: C = A[16][K] B[K][16].  mode 0: fp32 MFMA; 1: x6, truncating split; 2: x6, RNE split, one
// accumulator; 3: x6 RNE, small terms in their own accumulator; 4: x9 RNE (two accumulators); 5: x3 RNE (a0 b0 + a0 b1 + a1 b0)
__global__ void numerics_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K, int mode) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const float* a = A + (size_t)blockIdx.x * 16 * K;
  const float* b = B + (size_t)blockIdx.x * 16 * K;      // stored [col][K]
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
  if (mode == 0) {
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r * K + k + g], b[r * K + k + g], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 32) {
      u32x4 ap[3], bp[3];
      for (int j = 0; j < 4; ++j) {
        uint32_t x0[3], x1[3], y0[3], y1[3];
        const float a0 = a[r * K + k + 8 * g + 2 * j], a1 = a[r * K + k + 8 * g + 2 * j + 1];
        const float b0 = b[r * K + k + 8 * g + 2 * j], b1 = b[r * K + k + 8 * g + 2 * j + 1];
        if (mode == 1) { split3<false>(a0, x0); split3<false>(a1, x1); split3<false>(b0, y0); split3<false>(b1, y1); }
        else { split3<true>(a0, x0); split3<true>(a1, x1); split3<true>(b0, y0); split3<true>(b1, y1); }
        for (int p = 0; p < 3; ++p) {
          ap[p][j] = (x0[p] >> 16) | x1[p];
          bp[p][j] = (y0[p] >> 16) | y1[p];
        }
      }
      if (mode == 5) {
        acc = mfma_bf16(ap[1], bp[0], acc);
        acc = mfma_bf16(ap[0], bp[1], acc);
        acc = mfma_bf16(ap[0], bp[0], acc);
      } else if (mode == 1 || mode == 2) {
        acc = mfma_bf16(ap[2], bp[0], acc);
        acc = mfma_bf16(ap[0], bp[2], acc);
        acc = mfma_bf16(ap[1], bp[1], acc);
        acc = mfma_bf16(ap[1], bp[0], acc);
        acc = mfma_bf16(ap[0], bp[1], acc);
        acc = mfma_bf16(ap[0], bp[0], acc);
      } else {
        if (mode == 4) {
          lo = mfma_bf16(ap[2], bp[2], lo);
          lo = mfma_bf16(ap[2], bp[1], lo);
          lo = mfma_bf16(ap[1], bp[2], lo);
        }
        lo = mfma_bf16(ap[2], bp[0], lo);
        lo = mfma_bf16(ap[0], bp[2], lo);
        lo = mfma_bf16(ap[1], bp[1], lo);
        lo = mfma_bf16(ap[1], bp[0], lo);
        lo = mfma_bf16(ap[0], bp[1], lo);
        acc = mfma_bf16(ap[0], bp[0], acc);
      }
    }
  }
  for (int i = 0; i < 4; ++i) C[(size_t)blockIdx.x * 256 + (4 * g + i) * 16 + r] = acc[i] + lo[i];
}

// -------------------------------------------------------------------------------------------------------------- throughput
// Operands in registers, NACC accumulators per wave, `iters` trips; one trip = K 64 of a 16 x 16 (x NACC) product:
// fp32: 16 MFMAs 16x16x4 per accumulator; x6: 2 k-blocks x 6 MFMAs 16x16x32 per accumulator; FILL = independent v_fma_f32 per MFMA.
template <int MODE, int NACC, int FILL>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float fa = seed + lane, fb = 1.0f - 1e-7f * lane;
  u32x4 ap[3], bp[3];
  for (int p = 0; p < 3; ++p)
    for (int j = 0; j < 4; ++j) {
      ap[p][j] = 0x3f803f80u + ((lane * 7 + p * 3 + j) & 0x7f);
      bp[p][j] = 0x3f803f80u + ((lane * 5 + p + j * 11) & 0x7f);
    }
  float fill[4] = {seed, seed + 1.f, seed + 2.f, seed + 3.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[n], 0, 0, 0);
#pragma unroll
          for (int f = 0; f < FILL; ++f) fill[f & 3] = fmaf(fill[f & 3], 1.0000001f, 1e-9f);
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
#pragma unroll
          for (int m = 0; m < 6; ++m) {
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
            acc[n] = mfma_bf16(ap[pa[m]], bp[pb[m]], acc[n]);
#pragma unroll
            for (int f = 0; f < FILL; ++f) fill[f & 3] = fmaf(fill[f & 3], 1.0000001f, 1e-9f);
          }
        }
    }
  }
  float s = fill[0] + fill[1] + fill[2] + fill[3];
  for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  if (s == 12345.678f) out[0] = s;       // keeps the work alive
}

template <int MODE, int NACC, int FILL>
static void rate(const char* name, float* dout, int waves_per_simd) {
  const int iters = MODE == 0 ? 4000 : 12000;
  const int blocks = 256 * waves_per_simd;                 // 256-thread workgroups: one wave per SIMD each
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  rate_kernel<MODE, NACC, FILL><<<blocks, 256>>>(dout, 200, 1.f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  rate_kernel<MODE, NACC, FILL><<<blocks, 256>>>(dout, iters, 1.f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = (double)blocks * 4 * iters * NACC * (16.0 * 16 * 64 * 2);      // fp32-equivalent (useful) FLOP
  const double mfmas = (double)iters * NACC * (MODE == 0 ? 16 : 12);
  // cycles per MFMA per SIMD at 2.4 GHz nominal (the chip may hold a lower clock: the TFLOP/s column is what counts)
  printf("%-44s %2d waves/SIMD  %8.3f ms  %8.1f fp32-equivalent TFLOP/s   %.1f ns per MFMA per SIMD\n", name, waves_per_simd, ms,
         flop / (ms * 1e-3) * 1e-12, ms * 1e6 / (mfmas * waves_per_simd));
}

int main() {
  // ------------------------------------------------------------------------------------------------------------ numerics
  const int T = 256;
  for (int K : {64, 576, 2304}) {
    std::vector<float> A((size_t)T * 16 * K), B((size_t)T * 16 * K);
    uint64_t s = 0x9E3779B97F4A7C15ull + K;
    auto rnd = [&]() {                                     // sum of 4 uniforms, ~ normal, scale ~ 1
      float v = 0.f;
      for (int i = 0; i < 4; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        v += (float)((s >> 40) & 0xffffff) / 16777216.f - 0.5f;
      }
      return v * 1.7320508f;
    };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    std::vector<double> ref((size_t)T * 256);
    double rms = 0;
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double acc = 0;
          for (int k = 0; k < K; ++k) acc += (double)A[((size_t)t * 16 + i) * K + k] * (double)B[((size_t)t * 16 + j) * K + k];
          ref[(size_t)t * 256 + i * 16 + j] = acc;
          rms += acc * acc;
        }
    rms = std::sqrt(rms / ref.size());
    float *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, A.size() * 4));
    CHECK(hipMalloc(&dB, B.size() * 4));
    CHECK(hipMalloc(&dC, ref.size() * 4));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    const char* names[6] = {"fp32 MFMA 16x16x4", "x6 truncating split", "x6 RNE split, one accumulator", "x6 RNE split, small terms apart",
                            "x9 RNE split, small terms apart", "x3 RNE split"};
    // sequential fp32 fma on the host, the CPU reference's kind of sum, for scale
    double emax = 0, e2 = 0;
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          float acc = 0.f;
          for (int k = 0; k < K; ++k) acc = fmaf(A[((size_t)t * 16 + i) * K + k], B[((size_t)t * 16 + j) * K + k], acc);
          const double e = std::fabs((double)acc - ref[(size_t)t * 256 + i * 16 + j]);
          emax = std::fmax(emax, e);
          e2 += e * e;
        }
    printf("K %5d  rms |C| %.3f   %-36s max err %.3e   rms err %.3e  (relative to rms |C|: %.2e / %.2e)\n", K, rms, "host: sequential fmaf", emax,
           std::sqrt(e2 / ref.size()), emax / rms, std::sqrt(e2 / ref.size()) / rms);
    for (int mode = 0; mode < 6; ++mode) {
      numerics_kernel<<<T, 64>>>(dA, dB, dC, K, mode);
      CHECK(hipDeviceSynchronize());
      std::vector<float> C(ref.size());
      CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
      emax = 0;
      e2 = 0;
      for (size_t i = 0; i < C.size(); ++i) {
        const double e = std::fabs((double)C[i] - ref[i]);
        emax = std::fmax(emax, e);
        e2 += e * e;
      }
      printf("K %5d  rms |C| %.3f   %-36s max err %.3e   rms err %.3e  (relative to rms |C|: %.2e / %.2e)\n", K, rms, names[mode], emax,
             std::sqrt(e2 / C.size()), emax / rms, std::sqrt(e2 / C.size()) / rms);
    }
    CHECK(hipFree(dA));
    CHECK(hipFree(dB));
    CHECK(hipFree(dC));
  }
  // ---------------------------------------------------------------------------------------------------------- throughput
  float* dout;
  CHECK(hipMalloc(&dout, 64));
  for (int w : {1, 2}) {
    rate<0, 4, 0>("fp32 MFMA 16x16x4, 4 accumulators", dout, w);
    rate<0, 4, 1>("fp32 MFMA 16x16x4 + 1 v_fma per MFMA", dout, w);
    rate<0, 4, 2>("fp32 MFMA 16x16x4 + 2 v_fma per MFMA", dout, w);
    rate<0, 4, 4>("fp32 MFMA 16x16x4 + 4 v_fma per MFMA", dout, w);
    rate<1, 4, 0>("x6 on bf16 MFMA 16x16x32, 4 accumulators", dout, w);
    rate<1, 4, 1>("x6 on bf16 MFMA 16x16x32 + 1 v_fma per MFMA", dout, w);
    rate<1, 4, 2>("x6 on bf16 MFMA 16x16x32 + 2 v_fma per MFMA", dout, w);
    rate<1, 4, 4>("x6 on bf16 MFMA 16x16x32 + 4 v_fma per MFMA", dout, w);
  }
  CHECK(hipFree(dout));
  return 0;
}
