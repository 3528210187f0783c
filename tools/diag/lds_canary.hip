// Diagnostic: a workgroup fills its LDS allocation with a pattern, keeps re-writing it for `ticks` (100 MHz) and then checks
// it.  Workgroups of the SAME kernel are each other's neighbours on a CU: a word that comes back with another workgroup's
// pattern was written by that workgroup - i.e. the two LDS allocations overlap.  (tools/lds_granule_check.py)
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int WORDS>
__global__ __launch_bounds__(256) void lds_canary_kernel(unsigned* report, unsigned long long ticks, int rewrite) {
  __shared__ unsigned buf[WORDS];
  for (int i = threadIdx.x; i < WORDS; i += 256) buf[i] = (unsigned)i * 2654435761u ^ (blockIdx.x * 977u);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    if (rewrite && __builtin_amdgcn_s_memrealtime() - t0 < ticks / 2)       // first half of the wait: keep writing
      for (int i = threadIdx.x; i < WORDS; i += 256) buf[i] = (unsigned)i * 2654435761u ^ (blockIdx.x * 977u);
    __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < WORDS; i += 256) {
    const unsigned want = (unsigned)i * 2654435761u ^ (blockIdx.x * 977u);
    const unsigned got = buf[i];
    if (got != want) {
      const unsigned n = atomicAdd(report, 1u);
      atomicMin(report + 1, (unsigned)i);
      atomicMax(report + 2, (unsigned)i);
      if (n < 64) {
        report[4 + 4 * n] = blockIdx.x;
        report[5 + 4 * n] = (unsigned)i;
        report[6 + 4 * n] = got;
        report[7 + 4 * n] = want;
      }
    }
  }
}

#define CASE(W) if (words == W) { hipLaunchKernelGGL(lds_canary_kernel<W>, dim3(blocks), dim3(256), 0, s, report, ticks, rewrite); return hipGetLastError() == hipSuccess ? 0 : -2; }
extern "C" int lds_canary_words(int blocks, int words, unsigned long long ticks, int rewrite, unsigned* report, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  CASE(4096) CASE(4096 + 32) CASE(4096 + 64) CASE(4096 + 128) CASE(4096 + 192) CASE(4096 + 256) CASE(4096 + 320) CASE(4096 + 384)
  CASE(4096 + 512) CASE(4096 + 768) CASE(4096 + 1024) CASE(12288) CASE(12544) CASE(15360) CASE(15616) CASE(4146) CASE(12544 + 100) CASE(13568) CASE(10240) CASE(8192) CASE(20480) CASE(20224) CASE(40960) CASE(13312) CASE(13824) CASE(12568) CASE(4224) CASE(640)
  return -1;
}
extern "C" int lds_canary(int blocks, int kbytes, unsigned long long ticks, unsigned* report, void* stream) {
  return lds_canary_words(blocks, kbytes * 256, ticks, 0, report, stream);
}

// Register canary: NR VGPRs per lane hold a pattern through the wait (empty asm statements pin them in registers).
template <int NR, int NT>
__global__ __launch_bounds__(NT) void vgpr_canary_kernel(unsigned* report, unsigned long long ticks) {
  unsigned r[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) r[i] = (unsigned)(i + 1) * 2654435761u ^ (threadIdx.x * 40503u) ^ (blockIdx.x * 977u);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(r[i]));
    __builtin_amdgcn_s_sleep(4);
  }
  unsigned bad = 0;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    asm volatile("" : "+v"(r[i]));
    bad += r[i] != ((unsigned)(i + 1) * 2654435761u ^ (threadIdx.x * 40503u) ^ (blockIdx.x * 977u));
  }
  if (bad) atomicAdd(report, bad);
}

extern "C" int vgpr_canary(int blocks, int threads, unsigned long long ticks, unsigned* report, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (threads == 512) hipLaunchKernelGGL((vgpr_canary_kernel<150, 512>), dim3(blocks), dim3(512), 0, s, report, ticks);
  else hipLaunchKernelGGL((vgpr_canary_kernel<200, 256>), dim3(blocks), dim3(256), 0, s, report, ticks);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Packed-FMA canary: the same chain of fused multiply-adds once with v_pk_fma_f32 (two lanes of a register pair, the
// multiplicand broadcast by op_sel) and once with v_fma_f32; the two must agree bit for bit.
typedef float f32x2c __attribute__((ext_vector_type(2)));
template <int NT>
__global__ __launch_bounds__(NT) void pkfma_canary_kernel(unsigned* report, int iters) {
  f32x2c w[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    w[i].x = 0.001f * (float)((threadIdx.x * 7 + i * 13) % 97) - 0.04f;
    w[i].y = 0.001f * (float)((threadIdx.x * 11 + i * 5) % 89) - 0.03f;
  }
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2c a = {0.f, 0.f};
    float sx = 0.f, sy = 0.f;
    const float h0 = 0.5f + 0.001f * (float)(it & 63);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float h = h0 + 0.01f * (float)i;
      a = __builtin_elementwise_fma(w[i], f32x2c{h, h}, a);
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sx) : "v"(w[i].x), "v"(h));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sy) : "v"(w[i].y), "v"(h));
    }
    bad += (a.x != sx) + (a.y != sy);
  }
  if (bad) atomicAdd(report, bad);
}
extern "C" int pkfma_canary(int blocks, int threads, int iters, unsigned* report, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (threads == 512) hipLaunchKernelGGL(pkfma_canary_kernel<512>, dim3(blocks), dim3(512), 0, s, report, iters);
  else hipLaunchKernelGGL(pkfma_canary_kernel<256>, dim3(blocks), dim3(256), 0, s, report, iters);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Packed-FMA-from-LDS canary: the multiplicands come from LDS by ds_read_b128 every iteration (as in the LSTM recurrence);
// the reference chain uses the analytically known values and v_fma_f32.
template <int NT, int LDSPAD, int MODE = 0>
__global__ __launch_bounds__(NT) void pkfma_lds_canary_kernel(unsigned* report, int iters) {
  __shared__ __attribute__((aligned(16))) float h_s[2][128];
  __shared__ float pad_s[LDSPAD];
  if (iters < 0) pad_s[threadIdx.x] = 1.f;
  f32x2c w[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    w[i].x = 0.001f * (float)((threadIdx.x * 7 + i * 13) % 97) - 0.04f;
    w[i].y = 0.001f * (float)((threadIdx.x * 11 + i * 5) % 89) - 0.03f;
  }
  const int q = threadIdx.x & 3;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float h0 = 0.5f + 0.001f * (float)(it & 63);
    if (threadIdx.x < 128) h_s[it & 1][threadIdx.x] = h0 + 0.01f * (float)(threadIdx.x & 31);
    __syncthreads();
    const float* hb = h_s[it & 1] + q * 32;
    f32x2c a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    float sx0 = 0.f, sy0 = 0.f, sx1 = 0.f, sy1 = 0.f;
    float4 hall[8];
    if (MODE == 1) {                  // every load has registers of its own: nothing a packed FMA reads is reloaded in this iteration
#pragma unroll
      for (int k = 0; k < 8; ++k) hall[k] = *reinterpret_cast<const float4*>(hb + 4 * k);
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(hall[k].x), "+v"(hall[k].y), "+v"(hall[k].z), "+v"(hall[k].w));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 hv = MODE == 1 ? hall[k] : *reinterpret_cast<const float4*>(hb + 4 * k);
      if (MODE == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // distance between the packed FMAs and the next load
      a0 = __builtin_elementwise_fma(w[4 * k], f32x2c{hv.x, hv.x}, a0);
      a1 = __builtin_elementwise_fma(w[4 * k + 1], f32x2c{hv.y, hv.y}, a1);
      a0 = __builtin_elementwise_fma(w[4 * k + 2], f32x2c{hv.z, hv.z}, a0);
      a1 = __builtin_elementwise_fma(w[4 * k + 3], f32x2c{hv.w, hv.w}, a1);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float e0 = h0 + 0.01f * (float)(4 * k), e1 = h0 + 0.01f * (float)(4 * k + 1), e2 = h0 + 0.01f * (float)(4 * k + 2),
                  e3 = h0 + 0.01f * (float)(4 * k + 3);
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sx0) : "v"(w[4 * k].x), "v"(e0));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sy0) : "v"(w[4 * k].y), "v"(e0));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sx1) : "v"(w[4 * k + 1].x), "v"(e1));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sy1) : "v"(w[4 * k + 1].y), "v"(e1));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sx0) : "v"(w[4 * k + 2].x), "v"(e2));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sy0) : "v"(w[4 * k + 2].y), "v"(e2));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sx1) : "v"(w[4 * k + 3].x), "v"(e3));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sy1) : "v"(w[4 * k + 3].y), "v"(e3));
    }
    bad += (a0.x != sx0) + (a0.y != sy0) + (a1.x != sx1) + (a1.y != sy1);
  }
  if (bad) atomicAdd(report, bad);
}
extern "C" int pkfma_lds_canary(int blocks, int big_lds, int iters, unsigned* report, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (big_lds == 2) hipLaunchKernelGGL((pkfma_lds_canary_kernel<512, 512, 1>), dim3(blocks), dim3(512), 0, s, report, iters);
  else if (big_lds == 3) hipLaunchKernelGGL((pkfma_lds_canary_kernel<512, 512, 2>), dim3(blocks), dim3(512), 0, s, report, iters);
  else if (big_lds) hipLaunchKernelGGL((pkfma_lds_canary_kernel<512, 12288>), dim3(blocks), dim3(512), 0, s, report, iters);
  else hipLaunchKernelGGL((pkfma_lds_canary_kernel<512, 512>), dim3(blocks), dim3(512), 0, s, report, iters);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// LDS read canary: values written to LDS (double-buffered, one barrier per iteration - the LSTM's scheme) are read back by
// ds_read_b128 (mode 0) or ds_read_b32 (mode 1) and compared with what was written.
template <int MODE>
__global__ __launch_bounds__(512) void lds_read_canary_kernel(unsigned* report, int iters) {
  __shared__ __attribute__((aligned(16))) float h_s[2][128];
  const int q = threadIdx.x & 3;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float h0 = 0.5f + 0.001f * (float)(it & 63);
    if (threadIdx.x < 128) h_s[it & 1][threadIdx.x] = h0 + 0.01f * (float)(threadIdx.x & 31);
    __syncthreads();
    const float* hb = h_s[it & 1] + q * 32;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float4 hv;
      if (MODE == 0) hv = *reinterpret_cast<const float4*>(hb + 4 * k);
      else {
        const volatile float* hp = hb + 4 * k;
        hv.x = hp[0]; hv.y = hp[1]; hv.z = hp[2]; hv.w = hp[3];
      }
      bad += (hv.x != h0 + 0.01f * (float)(4 * k)) + (hv.y != h0 + 0.01f * (float)(4 * k + 1)) +
             (hv.z != h0 + 0.01f * (float)(4 * k + 2)) + (hv.w != h0 + 0.01f * (float)(4 * k + 3));
    }
  }
  if (bad) atomicAdd(report, bad);
}
extern "C" int lds_read_canary(int blocks, int mode, int iters, unsigned* report, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == 0) hipLaunchKernelGGL(lds_read_canary_kernel<0>, dim3(blocks), dim3(512), 0, s, report, iters);
  else hipLaunchKernelGGL(lds_read_canary_kernel<1>, dim3(blocks), dim3(512), 0, s, report, iters);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
