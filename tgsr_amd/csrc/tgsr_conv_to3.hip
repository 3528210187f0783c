// KxK convolution to 3 output channels (the image heads) for gfx950.
//
// Replaces GET_IMAGE_G_noAct.img (conv3x3 ngf->3, util.py:913-915) and NetG_highweight.conv_output
// (conv5x5 ngf->3 + Tanh, model.py:224) fused with `one * . + a * SRb` (model.py:280/288/297).
// Cout = 3 does not fill an MFMA tile (a 32-row tile would waste 90 % of it), and at 256x256 the op reads
// 8.4 MB and writes 0.8 MB per image against 0.1-0.3 GFLOP: it is a streaming kernel, so it runs on the VALU:
//   * workgroup = 16 x 64 output pixels, thread = 4 consecutive pixels x 3 channels (12 accumulators);
//   * the input is staged in LDS 8 channels at a time with its halo; a thread reads its 4+K-1 input floats of a
//     row as two aligned ds_read_b128 and reuses them for all K taps x 4 pixels x 3 channels;
//   * weights are wave-uniform -> scalar loads, used as the SGPR operand of v_fma.
#include "tgsr_common.h"

namespace tgsr {

struct To3Args {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;
  const float* w;  // [3][Cin][K][K]
  const float* addend;
  float alpha;
  float* out;
  int tiles_x, tiles_y;
};

template <int K, int ACT>
__global__ __launch_bounds__(256) void conv_to3_kernel(To3Args a) {
  constexpr int P = K / 2, CK = 8, TH = 16, TW = 64;
  constexpr int TR = TH + K - 1;
  constexpr int PITCH = 72;  // >= TW + 8 so that 8 floats from column 4*tx always stay inside the row
  __shared__ __attribute__((aligned(16))) float in_s[CK * TR * PITCH];

  const int tid = threadIdx.x;
  const int txi = tid & 15, tyi = tid >> 4;  // 16 x 16 threads: 4 pixels wide each
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int64_t HW = (int64_t)a.H * a.W;
  const float* xb = a.x + (int64_t)b * a.xbs;

  float acc[3][4];
#pragma unroll
  for (int co = 0; co < 3; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;

  for (int c0 = 0; c0 < a.Cin; c0 += CK) {
    __syncthreads();
    // LDS column j holds input column x0 - 4 + j (4-float left margin keeps the 16-B reads aligned)
    for (int idx = tid; idx < CK * TR * PITCH; idx += 256) {
      const int c = idx / (TR * PITCH);
      const int rem = idx - c * (TR * PITCH);
      const int r = rem / PITCH, j = rem - r * PITCH;
      const int gy = y0 - P + r, gx = x0 - 4 + j;
      float v = 0.f;
      if (c0 + c < a.Cin && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
        v = xb[(int64_t)(c0 + c) * HW + (int64_t)gy * a.W + gx];
      in_s[idx] = v;
    }
    __syncthreads();
    const int cmax = (a.Cin - c0) < CK ? (a.Cin - c0) : CK;
    for (int c = 0; c < cmax; ++c) {
      const float* wc = a.w + (int64_t)(c0 + c) * K * K;  // + co * Cin*K*K
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const float* row = in_s + (c * TR + tyi + ky) * PITCH + 4 * txi;
        const float4 v0 = *reinterpret_cast<const float4*>(row);
        const float4 v1 = *reinterpret_cast<const float4*>(row + 4);
        const float4 v2 = *reinterpret_cast<const float4*>(row + 8);
        const float in[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
        // pixel p (column x0 + 4*txi + p) tap kx reads LDS column 4*txi + 4 + p + kx - P
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            const float wv = wc[(int64_t)co * a.Cin * K * K + ky * K + kx];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(wv, in[4 + p + kx - P], acc[co][p]);
          }
        }
      }
    }
  }

  const int y = y0 + tyi, xx = x0 + 4 * txi;
  if (y < a.H) {
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const int64_t o = ((int64_t)b * 3 + co) * HW + (int64_t)y * a.W + xx;
      float r[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float v = acc[co][p];
        if (ACT == TGSR_ACT_TANH_AXPY) {
          v = tanhf(v);
          if (a.addend && xx + p < a.W) v += a.alpha * a.addend[o + p];
        }
        r[p] = v;
      }
      if (xx + 3 < a.W && (a.W & 3) == 0) {
        *reinterpret_cast<float4*>(a.out + o) = make_float4(r[0], r[1], r[2], r[3]);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (xx + p < a.W) a.out[o + p] = r[p];
      }
    }
  }
}

template <int K, int ACT>
static int launch_to3(To3Args a, hipStream_t s) {
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + 15) / 16;
  hipLaunchKernelGGL((conv_to3_kernel<K, ACT>), dim3((unsigned)(a.B * a.tiles_x * a.tiles_y)), dim3(256), 0, s, a);
  return note_launch(hipGetLastError(), "conv_to3_kernel");
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_conv_to3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* w,
                                 int K, int act, const float* addend, float alpha, float* out, void* stream) {
  if (!x || !w || !out || B < 1 || Cin < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (K != 3 && K != 5) return TGSR_EUNSUPPORTED;
  if (act != TGSR_ACT_NONE && act != TGSR_ACT_TANH_AXPY) return TGSR_EINVAL;
  if (act == TGSR_ACT_NONE && addend) return TGSR_EINVAL;
  To3Args a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.w = w;
  a.addend = addend; a.alpha = alpha; a.out = out; a.tiles_x = a.tiles_y = 0;
  hipStream_t s = as_stream(stream);
  if (K == 3) return act == TGSR_ACT_NONE ? launch_to3<3, TGSR_ACT_NONE>(a, s) : launch_to3<3, TGSR_ACT_TANH_AXPY>(a, s);
  return act == TGSR_ACT_NONE ? launch_to3<5, TGSR_ACT_NONE>(a, s) : launch_to3<5, TGSR_ACT_TANH_AXPY>(a, s);
}
