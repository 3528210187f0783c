// Fused 3x3 convolution for gfx950: fp32 implicit GEMM on v_mfma_f32_32x32x2_f32 with the reference's
// epilogues (BatchNorm-eval affine, GLU, residual add) and the nearest-x2 up-sample folded into the LDS read.
//
// Replaces conv3x3 / upBlock / ResBlock.block of the reference (util.py:62-65, 74-80, 110-130) - see
// include/tgsr_hip.h.  Written for CDNA4 directly:
//   * GEMM orientation D[cout][pixel] = W[cout][k] * X[k][pixel]: the MFMA "A" operand is the weight
//     (lane = cout), "B" is the input (lane = pixel).  One accumulator register then holds 32 consecutive
//     pixels of one output channel across lanes 0..31 -> every store instruction writes two 128-B runs of
//     an NCHW plane, and the GLU pair (c, c + C/2) lives in the same lane/register of two accumulators.
//   * a workgroup = 4 waves = TH x 32 output pixels x (1..4) blocks of 32 output channels; each wave owns
//     R rows and ALL the channel blocks, so a weight fragment is reused R times and an input fragment NCB times
//     from registers, and both come from LDS as conflict-free ds_read_b32 (32 consecutive dwords per half-wave).
//   * K loop = input-channel chunks of 8 staged in LDS: [8][TR][PITCH] halo tile + [9 taps][8][NCB*32]
//     weights (pre-packed on the device so the weight stage is straight float4 copies).
//   * 48 KB LDS and ~128 accumulator VGPRs per workgroup -> 3 workgroups per CU overlap one another's
//     staging with MFMA issue (fp32 MFMA is 64 cycles/instruction: LDS and L2 have large slack).
#include "tgsr_common.h"

namespace tgsr {

struct ConvArgs {
  const float* x;
  int64_t xbs;
  int B, Cin, H, W;
  const float* wpack;
  int Cout;
  const float* scale;
  const float* shift;
  const float* res;
  int64_t rbs;
  float* out;
  int64_t obs;
  int Ho, Wo, tiles_x, tiles_y, nchunks;
};

template <int NOB, bool GLU, bool UP, int R>
struct ConvCfg {
  static constexpr int NCB = NOB * (GLU ? 2 : 1);  // accumulator blocks of 32 couts per row
  static constexpr int NCOL = NCB * 32;            // weight columns held in LDS
  static constexpr int TH = 4 * R;                 // output rows per workgroup (4 waves)
  static constexpr int TR = UP ? TH / 2 + 2 : TH + 2;  // staged input rows (with halo)
  static constexpr int TC = UP ? 18 : 34;              // staged input cols (with halo)
  static constexpr int PITCH = TC;
  static constexpr int PLANE = TR * PITCH;
  static constexpr int IN_ELEMS = kConvCK * PLANE;
  static constexpr int W_ELEMS = 9 * kConvCK * NCOL;
};

template <int NOB, bool GLU, bool UP, int R>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(ConvArgs a) {
  using C = ConvCfg<NOB, GLU, UP, R>;
  constexpr int NCB = C::NCB, NCOL = C::NCOL, TH = C::TH, TR = C::TR, TC = C::TC, PITCH = C::PITCH, PLANE = C::PLANE;
  __shared__ __attribute__((aligned(16))) float smem[C::IN_ELEMS + C::W_ELEMS];
  float* in_s = smem + C::W_ELEMS;
  float* w_s = smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int grp = blockIdx.y;
  const int ty0 = ty * TH, tx0 = tx * 32;
  const int sy0 = (UP ? (ty0 >> 1) : ty0) - 1, sx0 = (UP ? (tx0 >> 1) : tx0) - 1;

  f32x16 acc[NCB][R];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.f;

  // lane-dependent LDS offsets (in floats)
  const int a_off = h * NCOL + l31;
  int b_col[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) b_col[kx] = UP ? (((l31 + kx - 1) >> 1) + 1) : (l31 + kx);
  const int b_base = h * PLANE;
  const int wrow = wave * R;

  const float* xb = a.x + (int64_t)b * a.xbs;
  const int64_t HW = (int64_t)a.H * a.W;

  for (int ch = 0; ch < a.nchunks; ++ch) {
    __syncthreads();  // previous chunk's fragment reads are done
    // ---- stage weights: [9*8 rows][NCOL] <- wpack[ch][tap][ci][Cout], NCB segments of 32 floats per row
    {
      const float* wsrc = a.wpack + (int64_t)ch * 9 * kConvCK * a.Cout;
      constexpr int NF4 = 9 * kConvCK * NCB * 8;
      for (int idx = tid; idx < NF4; idx += 256) {
        const int row = idx / (NCB * 8);
        const int rem = idx - row * (NCB * 8);
        const int seg = rem >> 3, f4 = rem & 7;
        int col;
        if (GLU)
          col = (seg < NOB ? (grp * NOB + seg) * 32 : (a.Cout >> 1) + (grp * NOB + seg - NOB) * 32);
        else
          col = (grp * NOB + seg) * 32;
        const float4 v = *reinterpret_cast<const float4*>(wsrc + (int64_t)row * a.Cout + col + f4 * 4);
        *reinterpret_cast<float4*>(w_s + row * NCOL + seg * 32 + f4 * 4) = v;
      }
    }
    // ---- stage the input halo tile (zero padding at the image border and past Cin)
    {
      const int c0 = ch * kConvCK;
      for (int idx = tid; idx < kConvCK * TR * TC; idx += 256) {
        const int c = idx / (TR * TC);
        const int rem = idx - c * (TR * TC);
        const int r = rem / TC, cc = rem - r * TC;
        const int gy = sy0 + r, gx = sx0 + cc;
        float v = 0.f;
        if (c0 + c < a.Cin && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
          v = xb[(int64_t)(c0 + c) * HW + (int64_t)gy * a.W + gx];
        in_s[c * PLANE + r * PITCH + cc] = v;
      }
    }
    __syncthreads();

    // ---- 9 taps x 4 channel pairs: NCB + R ds_read_b32, NCB*R MFMA each
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      int b_row[R];
#pragma unroll
      for (int r = 0; r < R; ++r) b_row[r] = (UP ? (((wrow + r + ky - 1) >> 1) + 1) : (wrow + r + ky)) * PITCH;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int kk = 0; kk < kConvCK / 2; ++kk) {
          float av[NCB], bv[R];
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) av[cb] = w_s[a_off + ((ky * 3 + kx) * kConvCK + 2 * kk) * NCOL + cb * 32];
#pragma unroll
          for (int r = 0; r < R; ++r) bv[r] = in_s[b_base + 2 * kk * PLANE + b_row[r] + b_col[kx]];
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < R; ++r)
              acc[cb][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb], bv[r], acc[cb][r], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: affine (BN eval), GLU, residual; each register = 32 consecutive pixels of one channel
  const int x = tx0 + l31;
  const int64_t HWo = (int64_t)a.Ho * a.Wo;
  float* ob = a.out + (int64_t)b * a.obs;
  const float* rb = a.res ? a.res + (int64_t)b * a.rbs : nullptr;
#pragma unroll
  for (int j = 0; j < NOB; ++j) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int cv = (grp * NOB + j) * 32 + acc_row(i, h);
      float sv = 1.f, tv = 0.f, sg = 1.f, tg = 0.f;
      if (a.scale) {
        sv = a.scale[cv];
        tv = a.shift[cv];
        if (GLU) {
          sg = a.scale[cv + (a.Cout >> 1)];
          tg = a.shift[cv + (a.Cout >> 1)];
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int y = ty0 + wrow + r;
        float v = acc[j][r][i] * sv + tv;
        if (GLU) {
          const float g = acc[NOB + j][r][i] * sg + tg;
          v = v * (1.f / (1.f + __expf(-g)));
        }
        if (y < a.Ho && x < a.Wo) {
          const int64_t o = (int64_t)cv * HWo + (int64_t)y * a.Wo + x;
          if (!GLU && rb) v += rb[o];
          ob[o] = v;
        }
      }
    }
  }
}

template <int NOB, bool GLU, bool UP, int R>
static int launch_conv(const ConvArgs& a, int groups, hipStream_t s) {
  using C = ConvCfg<NOB, GLU, UP, R>;
  ConvArgs k = a;
  k.tiles_x = (a.Wo + 31) / 32;
  k.tiles_y = (a.Ho + C::TH - 1) / C::TH;
  dim3 grid((unsigned)(a.B * k.tiles_x * k.tiles_y), (unsigned)groups);
  hipLaunchKernelGGL((conv3x3_mfma_kernel<NOB, GLU, UP, R>), grid, dim3(256), 0, s, k);
  return note_launch(hipGetLastError(), "conv3x3_mfma_kernel");
}

template <int NOB, bool GLU>
static int dispatch_up_r(const ConvArgs& a, int groups, bool up, int R, hipStream_t s) {
  if (up) return R == 2 ? launch_conv<NOB, GLU, true, 2>(a, groups, s) : launch_conv<NOB, GLU, true, 1>(a, groups, s);
  return R == 2 ? launch_conv<NOB, GLU, false, 2>(a, groups, s) : launch_conv<NOB, GLU, false, 1>(a, groups, s);
}

}  // namespace tgsr

using namespace tgsr;

extern "C" int tgsr_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* wpack,
                                int Cout, const float* scale, const float* shift, const float* residual,
                                int64_t res_bstride, float* out, int64_t out_bstride, int epilogue, int upsample,
                                void* stream) {
  if (!x || !wpack || !out || B < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return TGSR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TGSR_EINVAL;
  const bool glu = epilogue == TGSR_EPI_AFFINE_GLU;
  if (!glu && epilogue != TGSR_EPI_AFFINE) return TGSR_EINVAL;
  if (glu && residual) return TGSR_EINVAL;
  if (Cout % (glu ? 64 : 32) != 0) return TGSR_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.xbs = x_bstride; a.B = B; a.Cin = Cin; a.H = H; a.W = W;
  a.wpack = wpack; a.Cout = Cout; a.scale = scale; a.shift = shift;
  a.res = residual; a.rbs = res_bstride; a.out = out; a.obs = out_bstride;
  a.Ho = upsample ? 2 * H : H; a.Wo = upsample ? 2 * W : W;
  a.nchunks = (Cin + kConvCK - 1) / kConvCK;
  a.tiles_x = a.tiles_y = 0;
  // channel blocks per workgroup: 2 output blocks when the channel count allows it (more operand reuse)
  const int unit = glu ? 64 : 32;          // couts consumed per output block
  const int nob = (Cout % (2 * unit) == 0) ? 2 : 1;
  const int groups = Cout / (unit * nob);
  // rows per wave: 2 when that still gives every CU >= 2 workgroups, else 1 (small feature maps)
  const int64_t tiles2 = (int64_t)B * ((a.Ho + 7) / 8) * ((a.Wo + 31) / 32) * groups;
  const int R = tiles2 >= 512 ? 2 : 1;
  hipStream_t s = as_stream(stream);
  if (glu) return nob == 2 ? dispatch_up_r<2, true>(a, groups, upsample != 0, R, s)
                           : dispatch_up_r<1, true>(a, groups, upsample != 0, R, s);
  return nob == 2 ? dispatch_up_r<2, false>(a, groups, upsample != 0, R, s)
                  : dispatch_up_r<1, false>(a, groups, upsample != 0, R, s);
}
