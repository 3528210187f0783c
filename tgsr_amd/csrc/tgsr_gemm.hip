// Small fp32 MFMA GEMM with bias for the two trainable heads of CNN_ENCODER (util.py:300-301, 364-367):
//   emb_features : conv1x1 768 -> nef on the 17x17 region map  out[b][o][s] = sum_k W[o][k] x[b][k][s]        ("NN")
//   emb_cnn_code : Linear 2048 -> nef                           out[b][o]    = sum_k W[o][k] x[b][k] + bias[o] ("NT")
// One kernel: C[m][n] = sum_k A[m][k] * B(k, n) + bias[m], MFMA "A" = weight rows (lane = m, staged in LDS,
// pitch 65), MFMA "B" = activations (lane = n): NN reads them straight from HBM (n contiguous), NT stages a
// [32 n][64 k] tile per wave.  Output strides are free, so both NCHW planes and [B][nef] rows are written directly.
#include "tgsr_common.h"
#include "tgsr_text_blocks.h"

namespace tgsr {

struct GemmArgs {
  const float* A;      // [M][lda]
  const float* B;      // NN: [K][ldb] (n contiguous);  NT: [N][ldb] (k contiguous)
  const float* bias;   // [M] or null
  float* C;
  int M, N, K, lda, ldb;
  int64_t csm, csn;    // C[m*csm + n*csn]
  int64_t bsB, bsC;    // batch strides (blockIdx.z)
};

template <bool NT>
__global__ __launch_bounds__(256) void gemm_bias_kernel(GemmArgs a) {
  constexpr int KC = 64, P = KC + 1;
  __shared__ float a_s[32 * P];
  __shared__ float b_s[NT ? 4 * 32 * P : 1];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5, wave = tid >> 6;
  const int m0 = blockIdx.y * 32, n0 = (blockIdx.x * 4 + wave) * 32;
  const float* Bb = a.B + (int64_t)blockIdx.z * a.bsB;
  float* Cb = a.C + (int64_t)blockIdx.z * a.bsC;
  const int n = n0 + l31;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < a.K; k0 += KC) {
    __syncthreads();
    for (int idx = tid; idx < 32 * KC; idx += 256) {
      const int r = idx >> 6, k = idx & 63;
      a_s[r * P + k] = (m0 + r < a.M && k0 + k < a.K) ? a.A[(int64_t)(m0 + r) * a.lda + k0 + k] : 0.f;
    }
    if (NT) {
      float* mine = b_s + wave * 32 * P;
      for (int idx = lane; idx < 32 * KC; idx += 64) {
        const int r = idx >> 6, k = idx & 63;
        mine[r * P + k] = (n0 + r < a.N && k0 + k < a.K) ? Bb[(int64_t)(n0 + r) * a.ldb + k0 + k] : 0.f;
      }
    }
    __syncthreads();
    if (NT) {
      const float* mine = b_s + wave * 32 * P;
#pragma unroll 8
      for (int k = 0; k < KC; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s[l31 * P + k + hh], mine[l31 * P + k + hh], acc, 0, 0, 0);
    } else {
      float bv[KC / 2];
#pragma unroll
      for (int k = 0; k < KC / 2; ++k) {
        const int kk = k0 + 2 * k + hh;
        bv[k] = (n < a.N && kk < a.K) ? Bb[(int64_t)kk * a.ldb + n] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < KC / 2; ++k)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s[l31 * P + 2 * k + hh], bv[k], acc, 0, 0, 0);
    }
  }
  if (n < a.N) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + acc_row(i, hh);
      if (m < a.M) Cb[(int64_t)m * a.csm + (int64_t)n * a.csn] = acc[i] + (a.bias ? a.bias[m] : 0.f);
    }
  }
}

// The discriminators' logit heads: a 4x4 / stride-4 convolution of a 4x4 map to ONE channel = one dot product of
// K = 16 * 8 ndf values per sample.  As a GEMM that is a single 32 x 32 tile walking K = 8192 alone (0.9 ms measured);
// here one workgroup per sample reduces its row in a fixed order (float4 loads, tree in LDS: reproducible).
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int K) {
  __shared__ float red[256];
  const float* xr = x + (int64_t)blockIdx.x * K;
  float s = 0.f;
  if ((K & 3) == 0) {
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
      const float4 a = *reinterpret_cast<const float4*>(xr + k), b = *reinterpret_cast<const float4*>(w + k);
      s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  } else {
    for (int k = threadIdx.x; k < K; k += 256) s += xr[k] * w[k];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + (bias ? bias[0] : 0.f);
}

// dx[b][k] = dy[b] w[k]  and  dw[k] = sum_b dy[b] x[b][k] (b in order); thread = one k
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ w, float* __restrict__ dx,
                                                         float* __restrict__ dw, int B, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const float wk = dx ? w[k] : 0.f;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = dy[b];
    if (dx) dx[(int64_t)b * K + k] = g * wk;
    if (dw) s += g * x[(int64_t)b * K + k];
  }
  if (dw) dw[k] = s;
}

// CA_NET (util.py:372-400) in one launch: x = fc(sent_emb) [4 ncf]; h = x[:2ncf] * sigmoid(x[2ncf:]) (GLU);
// mu = h[:ncf], logvar = h[ncf:]; c_code = eps * exp(0.5 logvar) + mu.  Grid (sample, half): a workgroup owns the
// outputs i of its half of [0, ncf) and computes the four Linear rows each of them needs (i, i + 2ncf for mu; ncf + i,
// 3ncf + i for logvar) - a thread per row, float4 along its weight row (independent loads: a wave-per-row form with a
// shuffle reduction per row serialised 100 L2 latencies and took 212 us), the GLU pairs meet in LDS.  The eager form
// was a library GEMM plus eight pointwise launches.  eps comes from the caller (torch's generator: the reference
// consumes exactly ncf normals per sample here, util.py:388-396).
__global__ __launch_bounds__(256) void ca_net_kernel(const float* __restrict__ sent, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ eps,
                                                     int tdim, int ncf, float* __restrict__ c_code,
                                                     float* __restrict__ mu, float* __restrict__ logvar) {
  extern __shared__ float sm[];            // [tdim] sentence code, then [4][per] Linear outputs
  float* xs = sm;
  float* ys = sm + tdim;
  const int b = blockIdx.x, per = (ncf + 1) / 2, i0 = blockIdx.y * per, ni = min(per, ncf - i0);
  for (int k = threadIdx.x; k < tdim; k += 256) xs[k] = sent[(int64_t)b * tdim + k];
  __syncthreads();
  for (int t = threadIdx.x; t < 4 * ni; t += 256) {
    const int kind = t / ni, i = i0 + t - kind * ni;
    const int row = (kind & 2 ? ncf : 0) + (kind & 1 ? 2 * ncf : 0) + i;      // 0: i, 1: i + 2ncf, 2: ncf + i, 3: 3ncf + i
    const float* wr = w + (int64_t)row * tdim;
    float s0 = 0.f, s1 = 0.f;
    if ((tdim & 7) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
      for (int k = 0; k < tdim; k += 8) {
        const float4 a = *reinterpret_cast<const float4*>(wr + k), c = *reinterpret_cast<const float4*>(wr + k + 4);
        s0 += a.x * xs[k] + a.y * xs[k + 1] + a.z * xs[k + 2] + a.w * xs[k + 3];
        s1 += c.x * xs[k + 4] + c.y * xs[k + 5] + c.z * xs[k + 6] + c.w * xs[k + 7];
      }
    } else {
      for (int k = 0; k < tdim; ++k) s0 += wr[k] * xs[k];
    }
    ys[kind * per + (i - i0)] = s0 + s1 + bias[row];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < ni; t += 256) {
    const int i = i0 + t;
    const float m = ys[t] * (1.f / (1.f + __expf(-ys[per + t])));
    const float lv = ys[2 * per + t] * (1.f / (1.f + __expf(-ys[3 * per + t])));
    mu[(int64_t)b * ncf + i] = m;
    logvar[(int64_t)b * ncf + i] = lv;
    if (c_code) c_code[(int64_t)b * ncf + i] = eps[(int64_t)b * ncf + i] * __expf(0.5f * lv) + m;
  }
}

// ca_net_block (tgsr_text_blocks.h): grid (ceil(ncf / 4), ceil(B / 16)).
__global__ __launch_bounds__(256) void ca_net_mfma_kernel(CaArgs a) {
  __shared__ float red[kCaRedFloats];
  ca_net_block(a, blockIdx.x, blockIdx.y, red);
}

static int gemm_launch(const GemmArgs& a, bool nt, int batch, hipStream_t s) {
  dim3 grid((a.N + 127) / 128, (a.M + 31) / 32, batch);
  if (nt) hipLaunchKernelGGL(gemm_bias_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gemm_bias_kernel<false>, grid, dim3(256), 0, s, a);
  return note_launch(hipGetLastError(), "gemm_bias_kernel");
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_conv1x1_fwd(const float* x, int B, int Cin, int S, const float* w, const float* bias, int Cout,
                                float* out, void* stream) {
  if (!x || !w || !out || B < 1 || Cin < 1 || S < 1 || Cout < 1) return TGSR_EINVAL;
  GemmArgs a;
  a.A = w; a.B = x; a.bias = bias; a.C = out; a.M = Cout; a.N = S; a.K = Cin; a.lda = Cin; a.ldb = S;
  a.csm = S; a.csn = 1; a.bsB = (int64_t)Cin * S; a.bsC = (int64_t)Cout * S;
  return gemm_launch(a, false, B, as_stream(stream));
}

extern "C" int tgsr_ca_net_fwd(const float* sent_emb, const float* w, const float* bias, const float* eps, int B, int tdim,
                               int ncf, float* c_code, float* mu, float* logvar, void* stream) {
  if (!sent_emb || !w || !bias || !mu || !logvar || B < 1 || tdim < 1 || ncf < 1 || (c_code && !eps)) return TGSR_EINVAL;
  if ((tdim & 15) == 0 && ((tdim & 63) != 0 || ((reinterpret_cast<uintptr_t>(sent_emb) | reinterpret_cast<uintptr_t>(w)) & 15) == 0)) {
    CaArgs a;
    a.sent = sent_emb; a.w = w; a.bias = bias; a.eps = eps; a.c_code = c_code; a.mu = mu; a.logvar = logvar;
    a.B = B; a.tdim = tdim; a.ncf = ncf;
    hipLaunchKernelGGL(ca_net_mfma_kernel, dim3((ncf + 3) / 4, (B + 15) / 16), dim3(256), 0, as_stream(stream), a);
    return note_launch(hipGetLastError(), "ca_net_mfma_kernel");
  }
  if ((size_t)(tdim + 4 * ncf) * sizeof(float) > 60 * 1024) return TGSR_EUNSUPPORTED;
  hipLaunchKernelGGL(ca_net_kernel, dim3(B, 2), dim3(256), (size_t)(tdim + 4 * ncf + 8) * sizeof(float), as_stream(stream),
                     sent_emb, w, bias, eps, tdim, ncf, c_code, mu, logvar);
  return note_launch(hipGetLastError(), "ca_net_kernel");
}

extern "C" int tgsr_rowdot_fwd(const float* x, const float* w, const float* bias, float* out, int B, int K, void* stream) {
  if (!x || !w || !out || B < 1 || K < 1) return TGSR_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) return TGSR_EUNSUPPORTED;
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, w, bias, out, K);
  return note_launch(hipGetLastError(), "rowdot_fwd_kernel");
}

extern "C" int tgsr_rowdot_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, int B, int K,
                               void* stream) {
  if (!dy || B < 1 || K < 1 || (dx && !w) || (dw && !x) || (!dx && !dw)) return TGSR_EINVAL;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3((K + 255) / 256), dim3(256), 0, as_stream(stream), dy, x, w, dx, dw, B, K);
  return note_launch(hipGetLastError(), "rowdot_bwd_kernel");
}

extern "C" int tgsr_linear_fwd(const float* x, int B, int K, const float* w, const float* bias, int Cout, float* out,
                               void* stream) {
  if (!x || !w || !out || B < 1 || K < 1 || Cout < 1) return TGSR_EINVAL;
  GemmArgs a;
  a.A = w; a.B = x; a.bias = bias; a.C = out; a.M = Cout; a.N = B; a.K = K; a.lda = K; a.ldb = K;
  a.csm = 1; a.csn = Cout; a.bsB = 0; a.bsC = 0;
  return gemm_launch(a, true, 1, as_stream(stream));
}
