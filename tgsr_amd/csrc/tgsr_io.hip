// The data edge of the SR path on the GPU: the reference builds its image pyramid on the CPU with
// Pillow / torchvision (datasets.py:151-197 get_imgs_blur: transforms.Resize -> PIL bilinear resample,
// ImageFilter.GaussianBlur(radius=2) -> PIL extended box blur x3 per axis, ToTensor + Normalize(0.5, 0.5)).
// These kernels restate Pillow's INTEGER arithmetic on planar uint8 images so the result is byte-identical:
//   resample_kernel<AXIS>: out = clip8((2^21 + sum_t in[first + t] * k[t]) >> 22), taps k in 22-bit fixed point
//                          (computed on the host in double exactly like Pillow's precompute_coeffs), horizontal pass
//                          then vertical pass, each rounded to uint8 like ImagingResample.
//   box_blur_kernel<AXIS>: out = (ww * sum_{|d|<=r} in[clamp(x+d)] + fw * (in[clamp(x-r-1)] + in[clamp(x+r+1)]) + 2^23)
//                          >> 24 in uint32 (ImagingLineBoxBlur's sliding window written in closed form).
//   u8_normalize_kernel  : (u8 / 255 - 0.5) / 0.5 with float32 division, subtraction, division (no FMA, no reciprocal).
// All HBM-bound streaming passes over images of at most a few MB; layout [N planes][H][W].
#include "tgsr_common.h"

namespace tgsr {

template <int AXIS>   // 0: resample along W (rows stay), 1: along H
__global__ void resample_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int N, int Hin, int Win,
                                int Hout, int Wout, const int32_t* __restrict__ bounds, const int32_t* __restrict__ coef,
                                int ksize) {
  const int64_t total = (int64_t)N * Hout * Wout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wout);
    const int64_t t = i / Wout;
    const int y = (int)(t % Hout);
    const int n = (int)(t / Hout);
    const int o = AXIS == 0 ? x : y;
    const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int32_t* k = coef + (int64_t)o * ksize;
    const uint8_t* p = in + (int64_t)n * Hin * Win + (AXIS == 0 ? (int64_t)y * Win + first : (int64_t)first * Win + x);
    const int stride = AXIS == 0 ? 1 : Win;
    int ss = 1 << 21;
    for (int j = 0; j < cnt; ++j) ss += (int)p[(int64_t)j * stride] * k[j];
    ss >>= 22;
    out[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

template <int AXIS>
__global__ void box_blur_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int N, int H, int W, int r,
                                uint32_t ww, uint32_t fw) {
  const int64_t total = (int64_t)N * H * W;
  const int L = AXIS == 0 ? W : H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int64_t t = i / W;
    const int y = (int)(t % H);
    const int64_t n = t / H;
    const uint8_t* base = in + n * H * W + (AXIS == 0 ? (int64_t)y * W : x);
    const int stride = AXIS == 0 ? 1 : W;
    const int c = AXIS == 0 ? x : y;
    auto at = [&](int q) { return (uint32_t)base[(int64_t)(q < 0 ? 0 : (q >= L ? L - 1 : q)) * stride]; };
    uint32_t acc = 0;
    for (int d = -r; d <= r; ++d) acc += at(c + d);
    const uint32_t bulk = acc * ww + (at(c - r - 1) + at(c + r + 1)) * fw;
    out[i] = (uint8_t)((bulk + (1u << 23)) >> 24);
  }
}

__global__ void u8_normalize_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)in[i], 255.0f), 0.5f), 0.5f);
}

static inline int io_grid(int64_t n) {
  const int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_resize_bilinear_u8(const uint8_t* in, int N, int Hin, int Win, int Hout, int Wout,
                                       const int32_t* hbounds, const int32_t* hcoef, int hk, const int32_t* vbounds,
                                       const int32_t* vcoef, int vk, uint8_t* tmp, uint8_t* out, void* stream) {
  if (!in || !out || N < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return TGSR_EINVAL;
  const bool doh = Wout != Win, dov = Hout != Hin;
  if ((doh && (!hbounds || !hcoef || hk < 1)) || (dov && (!vbounds || !vcoef || vk < 1))) return TGSR_EINVAL;
  if (doh && dov && !tmp) return TGSR_EINVAL;
  hipStream_t s = as_stream(stream);
  const uint8_t* cur = in;
  if (doh) {                                      // Pillow: horizontal pass first, into an image of [Hin][Wout]
    uint8_t* dst = dov ? tmp : out;
    hipLaunchKernelGGL(resample_kernel<0>, dim3(io_grid((int64_t)N * Hin * Wout)), dim3(256), 0, s, cur, dst, N, Hin, Win,
                       Hin, Wout, hbounds, hcoef, hk);
    cur = dst;
  }
  if (dov)
    hipLaunchKernelGGL(resample_kernel<1>, dim3(io_grid((int64_t)N * Hout * Wout)), dim3(256), 0, s, cur, out, N, Hin, Wout,
                       Hout, Wout, vbounds, vcoef, vk);
  if (!doh && !dov) {
    const hipError_t e = hipMemcpyAsync(out, in, (size_t)N * Hin * Win, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return note_launch(e, "tgsr_resize_bilinear_u8 copy");
  }
  return note_launch(hipGetLastError(), "resample_kernel");
}

extern "C" int tgsr_gaussian_blur_u8(const uint8_t* in, int N, int H, int W, int radius, uint32_t ww, uint32_t fw,
                                     int passes, uint8_t* tmp, uint8_t* out, void* stream) {
  if (!in || !tmp || !out || N < 1 || H < 1 || W < 1 || radius < 0 || passes < 1) return TGSR_EINVAL;
  hipStream_t s = as_stream(stream);
  const int g = io_grid((int64_t)N * H * W);
  // 2 * passes launches ping-ponging between tmp and out, arranged so that the last one writes `out`
  const uint8_t* cur = in;
  const int launches = 2 * passes;
  for (int k = 0; k < launches; ++k) {
    uint8_t* dst = ((launches - 1 - k) & 1) ? tmp : out;
    if (k < passes) hipLaunchKernelGGL(box_blur_kernel<0>, dim3(g), dim3(256), 0, s, cur, dst, N, H, W, radius, ww, fw);
    else hipLaunchKernelGGL(box_blur_kernel<1>, dim3(g), dim3(256), 0, s, cur, dst, N, H, W, radius, ww, fw);
    cur = dst;
  }
  return note_launch(hipGetLastError(), "box_blur_kernel");
}

extern "C" int tgsr_u8_normalize(const uint8_t* in, float* out, int64_t n, void* stream) {
  if (!in || !out || n < 1) return TGSR_EINVAL;
  hipLaunchKernelGGL(u8_normalize_kernel, dim3(io_grid(n)), dim3(256), 0, as_stream(stream), in, out, n);
  return note_launch(hipGetLastError(), "u8_normalize_kernel");
}
