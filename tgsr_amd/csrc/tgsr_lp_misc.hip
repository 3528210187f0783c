// Reduced-precision inference path: the 3-channel stems, the 3-channel image heads and the word attention on lp
// images (tgsr_lp_common.h).  BASELINE.json configs[4].
//
//   lp_stem_kernel      conv3x3 3 -> 2C + BN affine + GLU from the fp32 LR image (im2f util.py:741-744, convin
//                       model.py:228): 27 MACs per output, VALU, fp32 weights; writes C channels of an lp image.
//   lp_to3_kernel<K>    KxK conv C=32 -> 3 (+ tanh + alpha * addend) reading an lp image, writing the fp32 NCHW image
//                       (GET_IMAGE_G_noAct util.py:913-915, conv_output + a*SRb model.py:224, 280): MFMA 16x16x32 with
//                       (output channel, kernel column) pairs as the 16 rows of the A fragment - K MFMAs per 16 columns.
//   lp_word_attention_kernel   GlobalAttentionGeneral.forward (GlobalAttention.py:87-130) on MFMA 32x32x16: scores,
//                       masked softmax over words (in-lane + one lane^32 exchange) and the weighted context, with the
//                       softmax consumed as the second MFMA's B operand straight from the accumulator registers.
#include "tgsr_lp_common.h"

namespace tgsr {

// ------------------------------------------------------------------------------------------------------------ stem
struct LpStemArgs {
  const float* x;        // [B][3][H][W] fp32
  const float* w;        // [2C][3][3][3] fp32 (torch layout)
  const float* scale;    // [2C]
  const float* shift;
  unsigned short* out;   // lp image
  int B, H, W, C, ocp, oco;
  LpAttFuse att;         // ATT: attend to the words in the epilogue (INIT_STAGE_GImgup: im2f -> att, util.py:768-771)
};

// Workgroup = one row segment of 32 pixels x ALL output channels: thread = (pixel, group of 8 output channels: 8 value +
// 8 gate channels x 27 MACs, one 16-byte store).  The 3 x 3 x 34 input halo, the whole filter (2C x 27 floats, 6.9 KB for
// C = 32) and the BatchNorm affine go through LDS once per workgroup with coalesced loads; before, every thread fetched
// its 27 inputs itself, eight times over (once per channel group), and the 16 x 27 weights of its group through ~50
// dependent scalar loads: 11 us for 2 MFLOP per image at the head of the step's critical chain (profiles/r02i_bf16_
// kernel_stats.csv), now bound by one load round trip.
// ATT (C = 32): the 32 pixels x 32 channels the workgroup has just produced are also staged in LDS and wave 0 attends to the
// words for them (lp_attend_tile): c_code goes to channels [att.coff, att.coff + 32) of the same pixels of `out`, the
// attention map to att.attn - the stand-alone attention launch of the 32 x 32 stage and its re-read of h are gone.
template <class T, bool ATT>
__global__ __launch_bounds__(256) void lp_stem_kernel(LpStemArgs a) {
  // [2C][28] weights | [2C] scale | [2C] shift | [3][3][34] input, sized for C <= 64
  __shared__ __attribute__((aligned(16))) float stem_s[2 * 64 * 28 + 4 * 64 + 3 * 3 * 34];
  constexpr int HP = 80;                                               // bytes per staged pixel (64 + 16 pad)
  __shared__ __attribute__((aligned(16))) char h_s[ATT ? 32 * HP : 16];
  const int C2 = 2 * a.C, NG = a.C / 8;                                // NG channel groups, 32 * NG threads do the arithmetic
  float* ws = stem_s;
  float* sc = ws + C2 * 28;
  float* sh = sc + C2;
  float* in_s = sh + C2;
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tilesx = (a.W + 31) / 32;
  const int tx = t % tilesx;
  t /= tilesx;
  const int y = t % a.H, b = t / a.H, x0 = tx * 32;
  u32x4 fa[4];
  if constexpr (ATT) {
    if (tid < 64) {                                                    // wave 0's A fragments: in flight under the convolution
#pragma unroll
      for (int f = 0; f < 4; ++f)
        fa[f] = *reinterpret_cast<const u32x4*>(a.att.frag + (int64_t)b * 4096 + (f * 64 + tid) * 16);
    }
  }
  for (int i = tid; i < C2 * 27; i += 256) ws[(i / 27) * 28 + i % 27] = a.w[i];
  for (int i = tid; i < C2; i += 256) {
    sc[i] = a.scale[i];
    sh[i] = a.shift[i];
  }
  for (int i = tid; i < 3 * 3 * 34; i += 256) {
    const int c = i / 102, r = (i / 34) % 3, j = i % 34;
    const int yy = y + r - 1, xx = x0 + j - 1;
    in_s[i] = ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? a.x[(((int64_t)b * 3 + c) * a.H + yy) * a.W + xx] : 0.f;
  }
  __syncthreads();
  const int px = tid & 31, g = tid >> 5;
  const bool active = g < NG && x0 + px < a.W;
  if (!ATT && !active) return;
  if (active) {
  float in[27];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) in[c * 9 + dy * 3 + dx] = in_s[(c * 3 + dy) * 34 + px + dx];
  // (Round 3 history: built with the SLP vectorizer this loop became v_pk_fma_f32 fed by late re-reads of in_s into the same
  // registers, and the kernel intermittently computed other values - lanes 48-63, the later channels - whenever a hipGraph ran
  // MFMA-bound convolutions beside it.  That was the packed-fp32 hazard of profiles/HISTORY.md 3.13, not a property of this kernel: the
  // library's inference kernels are built without packed fp32 instructions now, tests/test_hip_concurrency.py watches it.)
  float o[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int cv = g * 8 + q, cg = a.C + cv;
    const float* wv = ws + cv * 28;                                    // rows of 28 floats: 16-byte aligned, read as float4
    const float* wg = ws + cg * 28;
    float v = 0.f, gt = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 7; ++k4) {
      const float4 a4 = *reinterpret_cast<const float4*>(wv + 4 * k4);
      const float4 g4 = *reinterpret_cast<const float4*>(wg + 4 * k4);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (4 * k4 + j < 27) {
          v = fmaf(av[j], in[4 * k4 + j], v);
          gt = fmaf(gv[j], in[4 * k4 + j], gt);
        }
      }
    }
    v = v * sc[cv] + sh[cv];
    gt = gt * sc[cg] + sh[cg];
    o[q] = v * sigmoidf_fast(gt);
  }
  u32x4 pk;
#pragma unroll
  for (int q = 0; q < 4; ++q) pk[q] = LP<T>::pack2(o[2 * q], o[2 * q + 1]);
  unsigned short* op = a.out + (((int64_t)b * (a.H + 2) + y + 1) * (a.W + 2) + x0 + px + 1) * a.ocp + a.oco + g * 8;
  *reinterpret_cast<u32x4*>(op) = pk;
  if constexpr (ATT) *reinterpret_cast<u32x4*>(h_s + px * HP + g * 16) = pk;
  }
  if constexpr (ATT) {
    __syncthreads();
    if (tid >= 64) return;
    const int l31 = tid & 31, hh = tid >> 5;
    const int64_t Q = (int64_t)a.H * a.W;
    const int64_t q = (int64_t)y * a.W + x0 + l31;
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(h_s + l31 * HP + hh * 16);
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(h_s + l31 * HP + 32 + hh * 16);
    unsigned mb = 0;
    if (a.att.mbits) mb = a.att.mbits[a.att.mask_mode ? b : (int)(((int64_t)b * Q + q) % a.B)];   // GlobalAttention.py:111
    char* cp = reinterpret_cast<char*>(a.out + (((int64_t)b * (a.H + 2) + y + 1) * (a.W + 2) + x0 + l31 + 1) * a.ocp + a.att.coff);
    lp_attend_tile<T>(fa, b0, b1, mb, a.att.T, hh, a.att.attn ? a.att.attn + (int64_t)b * a.att.T * Q + q : nullptr, Q, cp);
  }
}

// ------------------------------------------------------------------------------------------------------------ heads
struct LpTo3Args {
  const char* x;         // lp image, channels [0, 32)
  int xcp;
  const char* wpack;     // [K kernel rows][lane 64][8]
  const float* addend;   // [B][3][H][W] fp32 or null
  float alpha;
  float* out;            // [B][3][H][W] fp32
  int B, H, W, tiles_x, tiles_y;
};

// Workgroup = 4 waves = 8 rows x 32 columns of outputs; wave w owns rows 2w, 2w+1.
// Three output channels cannot fill an MFMA tile, so the kernel column dx is moved into the M dimension: the A
// fragment of kernel row dy holds the 3K "virtual channels" (c, dx) (15 of 16 rows for K = 5, 9 for K = 3),
//     V[(c, dx)][x'] = sum_{dy, ci} w[c][ci][dy][dx] * in[ci][y + dy][x'],       out[c][x] = sum_dx V[(c, dx)][x + dx],
// i.e. K MFMAs (16x16x32, one k-step = the 32 input channels) per 16 input columns instead of K*K, and the K-term shift-
// sum over dx goes through a small per-wave LDS image of V.  Per output row: 3 column tiles (32 + K - 1 <= 48 columns)
// x K MFMAs; measured against the tap-per-MFMA form this is 3.3x (K = 5) / 2x (K = 3) fewer MFMAs and B-fragment reads.
// Halo tile [(8 + K - 1) x 48 pixels][32 ch] by LDS-DMA, 16-byte slots swizzled by (column >> 1) & 3 (ds_read_b128 of 16
// neighbouring pixels x 4 channel groups conflict free; checked by simulation).  Pixels further than one outside the
// image (K = 5, or the tile's spare columns) are fetched from the image's top-left border pixel, zero by the layout rule.
template <class T, int K, int ACT>
__global__ __launch_bounds__(256) void lp_to3_kernel(LpTo3Args a) {
  constexpr int P = K / 2, TR = 8 + 2 * P, TC = 48, NPIX = TR * TC, VP = 52;
  constexpr int TILE_SLOTS = NPIX * 4, TILE_INSTR = (TILE_SLOTS + 63) / 64;
  // the per-wave V images go on top of the input tile once every wave is done reading it: 37 KB (5x5) / 31 KB (3x3) of
  // LDS per workgroup = 4-5 workgroups per CU in flight to cover each other's tile-load latency
  static_assert(4 * 16 * VP * 4 <= TILE_INSTR * 1024, "V images fit the tile");
  __shared__ __attribute__((aligned(1024))) char tile[TILE_INSTR * 1024];
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * 8, x0 = tx * 32;
  const int64_t rowb = (int64_t)(a.W + 2) * a.xcp * 2;
  const char* xb = a.x + (int64_t)b * (a.H + 2) * rowb;
#pragma unroll
  for (int k = 0; k < (TILE_INSTR + 3) / 4; ++k) {
    const int ins = wave + 4 * k;
    if (ins < TILE_INSTR) {
      const int S = ins * 64 + lane;
      int pix = S >> 2;
      const int ps = S & 3;
      if (pix >= NPIX) pix = 0;
      const int r = pix / TC, c = pix - r * TC;
      const int ls = ps ^ ((c >> 1) & 3);
      int sy = y0 + r + 1 - P, sx = x0 + c + 1 - P;            // padded source coordinates
      if ((unsigned)sy > (unsigned)(a.H + 1) || (unsigned)sx > (unsigned)(a.W + 1)) sy = sx = 0;
      lds_dma16(xb + sy * rowb + (int64_t)sx * (a.xcp * 2) + ls * 16, tile + ins * 1024);
    }
  }
  u32x4 af[K];
#pragma unroll
  for (int k = 0; k < K; ++k) af[k] = *reinterpret_cast<const u32x4*>(a.wpack + (k * 64 + lane) * 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x4w acc[2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[r][j][i] = 0.f;
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = 16 * j + p;
        const u32x4 bf = *reinterpret_cast<const u32x4*>(tile + ((2 * wave + r + dy) * TC + c) * 64 +
                                                         ((g ^ ((c >> 1) & 3)) << 4));
        acc[r][j] = LP<T>::mfma16(af[dy], bf, acc[r][j]);
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                  // every wave is done reading the tile
  float* v = reinterpret_cast<float*>(tile) + wave * (16 * VP);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    // D[row = 4 g + reg][col = p] -> V image [16 rows][48 columns] of this wave (pitch 52: rows 4 apart on distinct banks)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[(4 * g + i) * VP + 16 * j + p] = acc[r][j][i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // wave-private image: no barrier
    const int y = y0 + 2 * wave + r;
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {                         // lanes 0-31: channel 0 then 2; lanes 32-63: channel 1
      const int c = rnd == 0 ? (lane >> 5) : 2;
      if (rnd == 1 && lane >= 32) break;
      const int x = lane & 31;
      float o = 0.f;
#pragma unroll
      for (int dx = 0; dx < K; ++dx) o += v[(c * K + dx) * VP + x + dx];
      const int64_t oi = (((int64_t)b * 3 + c) * a.H + y) * a.W + x0 + x;
      if (ACT == TGSR_ACT_TANH_AXPY) o = tanhf(o) + (a.addend ? a.alpha * a.addend[oi] : 0.f);
      a.out[oi] = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the V image is rewritten for the next row
  }
}

// wpack[dy][lane][8] <- w[3][32][K][K]: element j of lane l = w[c][8 (l >> 4) + j][dy][dx] for row (l & 15) = c * K + dx
// < 3 K, else 0
template <class T>
__global__ void lp_pack_to3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int K, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = i & 7, l = (i >> 3) & 63, dy = i >> 9;
  const int row = l & 15, ci = 8 * (l >> 4) + j;
  const int c = row / K, dx = row - c * K;
  wp[i] = LP<T>::one(row < 3 * K ? w[((c * 32 + ci) * K + dy) * K + dx] : 0.f);
}

// ------------------------------------------------------------------------------------------------------------ attention
struct LpAttnArgs {
  const char* h;         // lp image, channels [0, 32) = h_code
  int hcp;
  const float* src;      // [B][32][32] fp32 projected words (tgsr_word_project_fwd), zero padded past T
  const uint8_t* mask;   // [B][T] or null
  int mask_mode, B, T, H, W;
  char* c;               // lp image receiving c_code at channels [cco, cco + 32)
  int ccp, cco;
  float* attn;           // [B][T][H*W] fp32 or null
};

template <class T>
__global__ __launch_bounds__(256) void lp_word_attention_kernel(LpAttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short frag_s[4][64][8];   // A fragments: GEMM1 k-steps 0,1; GEMM2 0,1
  __shared__ unsigned mbits_s[256];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = tid >> 6;
  const int b = blockIdx.y;
  const int Q = a.H * a.W;
  const float* sb = a.src + (int64_t)b * 32 * 32;          // [i][t]
  for (int o = tid; o < 4 * 64 * 8; o += 256) {
    const int j = o & 7, l = (o >> 3) & 63, f = o >> 9;
    const int lr = l & 31, lh = l >> 5;
    float v;
    if (f < 2) v = sb[(16 * f + 8 * lh + j) * 32 + lr];                                // A[row t = lr][k = i]
    else v = sb[lr * 32 + 16 * (f - 2) + 8 * (j >> 2) + 4 * lh + (j & 3)];             // A[row i = lr][k = t (permuted)]
    frag_s[f][l][j] = LP<T>::one(v);
  }
  const int nrows = a.mask ? (a.B < 256 ? a.B : 256) : 0;
  for (int r = tid; r < nrows; r += 256) {
    unsigned m = 0;
    for (int t = 0; t < a.T; ++t) m |= (a.mask[r * a.T + t] ? 1u : 0u) << t;
    mbits_s[r] = m;
  }
  __syncthreads();
  u32x4 fa[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) fa[f] = *reinterpret_cast<const u32x4*>(&frag_s[f][lane][0]);

  const int ntiles = Q >> 5, tstride = gridDim.x * 4;
  const int64_t hrow = (int64_t)(a.W + 2) * a.hcp * 2, crow = (int64_t)(a.W + 2) * a.ccp * 2;
  const char* hb = a.h + (int64_t)b * (a.H + 2) * hrow;
  char* cb = a.c + (int64_t)b * (a.H + 2) * crow;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += tstride) {
    const int q = tile * 32 + l31;
    const int y = q / a.W, x = q - y * a.W;
    const char* hp = hb + (y + 1) * hrow + (int64_t)(x + 1) * (a.hcp * 2) + hh * 16;
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(hp);          // channels 8 hh .. (k-step 0)
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(hp + 32);     // channels 16 + 8 hh ..
    unsigned mb = 0;
    if (a.mask) {
      const int mrow = a.mask_mode ? b : (int)(((int64_t)b * Q + q) % a.B);   // GlobalAttention.py:111 mask.repeat(queryL,1)
      if (mrow < 256) {
        mb = mbits_s[mrow];
      } else {
        for (int t = 0; t < a.T; ++t) mb |= (a.mask[mrow * a.T + t] ? 1u : 0u) << t;
      }
    }
    char* cp = cb + (y + 1) * crow + (int64_t)(x + 1) * (a.ccp * 2) + a.cco * 2;
    lp_attend_tile<T>(fa, b0, b1, mb, a.T, hh, a.attn ? a.attn + (int64_t)b * a.T * Q + q : nullptr, Q, cp);
  }
}

}  // namespace tgsr

using namespace tgsr;

static int lp_stem_launch(int dtype, const float* x, int B, int H, int W, const float* w, int C, const float* scale,
                          const float* shift, void* out, int out_cpitch, int out_coff, const LpAttFuse* att, void* stream) {
  if (!x || !w || !scale || !shift || !out || B < 1 || H < 1 || W < 1 || C < 8) return TGSR_EINVAL;
  if (C % 8 != 0 || out_cpitch % 8 != 0 || out_coff % 8 != 0 || out_coff + C > out_cpitch ||
      (reinterpret_cast<uintptr_t>(out) & 15))
    return TGSR_EUNSUPPORTED;
  LpStemArgs a;
  a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.out = static_cast<unsigned short*>(out);
  a.B = B; a.H = H; a.W = W; a.C = C; a.ocp = out_cpitch; a.oco = out_coff;
  if (C > 64) return TGSR_EUNSUPPORTED;                      // 8 channel groups of 8 per 256-thread workgroup
  const dim3 grid((unsigned)((int64_t)B * H * ((W + 31) / 32)));
  if (att) {
    if (C != 32 || W % 32 != 0 || att->T < 1 || att->T > 32 || att->coff % 4 != 0 || att->coff + 32 > out_cpitch ||
        (att->coff < out_coff + C && out_coff < att->coff + 32))
      return TGSR_EUNSUPPORTED;
    a.att = *att;
    if (dtype == TGSR_DT_BF16) hipLaunchKernelGGL((lp_stem_kernel<BF16, true>), grid, dim3(256), 0, as_stream(stream), a);
    else if (dtype == TGSR_DT_F16) hipLaunchKernelGGL((lp_stem_kernel<F16, true>), grid, dim3(256), 0, as_stream(stream), a);
    else return TGSR_EINVAL;
    return note_launch(hipGetLastError(), "lp_stem_kernel");
  }
  a.att = LpAttFuse{nullptr, nullptr, nullptr, 0, 0, 0};
  if (dtype == TGSR_DT_BF16) hipLaunchKernelGGL((lp_stem_kernel<BF16, false>), grid, dim3(256), 0, as_stream(stream), a);
  else if (dtype == TGSR_DT_F16) hipLaunchKernelGGL((lp_stem_kernel<F16, false>), grid, dim3(256), 0, as_stream(stream), a);
  else return TGSR_EINVAL;
  return note_launch(hipGetLastError(), "lp_stem_kernel");
}

extern "C" int tgsr_lp_stem_fwd(int dtype, const float* x, int B, int H, int W, const float* w, int C, const float* scale,
                                const float* shift, void* out, int out_cpitch, int out_coff, void* stream) {
  return lp_stem_launch(dtype, x, B, H, W, w, C, scale, shift, out, out_cpitch, out_coff, nullptr, stream);
}

extern "C" int tgsr_lp_stem_att_fwd(int dtype, const float* x, int B, int H, int W, const float* w, int C, const float* scale,
                                    const float* shift, void* out, int out_cpitch, int out_coff, const void* att_pack,
                                    int att_nsets, int att_set, int use_mask, int mask_mode, int T, int c_coff, float* attn,
                                    void* stream) {
  LpAttFuse f;
  const int rc = lp_att_fuse(att_pack, att_nsets, att_set, B, use_mask, mask_mode, T, c_coff, attn, &f);
  if (rc) return rc;
  return lp_stem_launch(dtype, x, B, H, W, w, C, scale, shift, out, out_cpitch, out_coff, &f, stream);
}

extern "C" int tgsr_lp_pack_to3_weight(int dtype, const float* w, void* wpack, int Cin, int K, void* stream) {
  if (!w || !wpack) return TGSR_EINVAL;
  if (Cin != 32 || (K != 3 && K != 5)) return TGSR_EUNSUPPORTED;
  const int total = K * 512;
  unsigned short* o = static_cast<unsigned short*>(wpack);
  if (dtype == TGSR_DT_BF16)
    hipLaunchKernelGGL(lp_pack_to3_kernel<BF16>, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), w, o, K, total);
  else if (dtype == TGSR_DT_F16)
    hipLaunchKernelGGL(lp_pack_to3_kernel<F16>, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), w, o, K, total);
  else
    return TGSR_EINVAL;
  return note_launch(hipGetLastError(), "lp_pack_to3_kernel");
}

template <class T>
static int launch_to3(const LpTo3Args& a, int K, int act, hipStream_t s) {
  const dim3 grid((unsigned)(a.B * a.tiles_x * a.tiles_y));
  if (K == 3 && act == TGSR_ACT_NONE) hipLaunchKernelGGL((lp_to3_kernel<T, 3, TGSR_ACT_NONE>), grid, dim3(256), 0, s, a);
  else if (K == 3) hipLaunchKernelGGL((lp_to3_kernel<T, 3, TGSR_ACT_TANH_AXPY>), grid, dim3(256), 0, s, a);
  else if (act == TGSR_ACT_NONE) hipLaunchKernelGGL((lp_to3_kernel<T, 5, TGSR_ACT_NONE>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((lp_to3_kernel<T, 5, TGSR_ACT_TANH_AXPY>), grid, dim3(256), 0, s, a);
  return note_launch(hipGetLastError(), "lp_to3_kernel");
}

extern "C" int tgsr_lp_conv_to3_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack,
                                    int K, int act, const float* addend, float alpha, float* out, void* stream) {
  if (!x || !wpack || !out || B < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (act != TGSR_ACT_NONE && act != TGSR_ACT_TANH_AXPY) return TGSR_EINVAL;
  if (Cin != 32 || (K != 3 && K != 5) || W % 32 != 0 || H % 8 != 0 || x_cpitch < 32 || x_cpitch % 8 != 0 ||
      (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wpack) & 15))
    return TGSR_EUNSUPPORTED;
  LpTo3Args a;
  a.x = static_cast<const char*>(x); a.xcp = x_cpitch; a.wpack = static_cast<const char*>(wpack);
  a.addend = addend; a.alpha = alpha; a.out = out; a.B = B; a.H = H; a.W = W;
  a.tiles_x = W / 32; a.tiles_y = H / 8;
  if (dtype == TGSR_DT_BF16) return launch_to3<BF16>(a, K, act, as_stream(stream));
  if (dtype == TGSR_DT_F16) return launch_to3<F16>(a, K, act, as_stream(stream));
  return TGSR_EINVAL;
}

extern "C" int tgsr_lp_word_attention_fwd(int dtype, const void* h, int h_cpitch, const float* src, const uint8_t* mask,
                                          int mask_mode, int B, int idf, int T, int H, int W, void* c_img, int c_cpitch,
                                          int c_coff, float* attn, void* stream) {
  if (!h || !src || !c_img || B < 1 || T < 1 || H < 1 || W < 1) return TGSR_EINVAL;
  if (idf != 32 || T > 32 || W % 32 != 0 || h_cpitch < 32 || h_cpitch % 8 != 0 || c_cpitch % 4 != 0 || c_coff % 4 != 0 ||
      c_coff + 32 > c_cpitch || (reinterpret_cast<uintptr_t>(h) & 15) || (reinterpret_cast<uintptr_t>(c_img) & 7))
    return TGSR_EUNSUPPORTED;
  LpAttnArgs a;
  a.h = static_cast<const char*>(h); a.hcp = h_cpitch; a.src = src; a.mask = mask; a.mask_mode = mask_mode;
  a.B = B; a.T = T; a.H = H; a.W = W; a.c = static_cast<char*>(c_img); a.ccp = c_cpitch; a.cco = c_coff; a.attn = attn;
  const int Q = H * W;
  int gx = (Q + 127) / 128;
  const int cap = (2048 + B - 1) / B;
  if (gx > cap) gx = cap;
  const dim3 grid(gx, B);
  if (dtype == TGSR_DT_BF16)
    hipLaunchKernelGGL(lp_word_attention_kernel<BF16>, grid, dim3(256), 0, as_stream(stream), a);
  else if (dtype == TGSR_DT_F16)
    hipLaunchKernelGGL(lp_word_attention_kernel<F16>, grid, dim3(256), 0, as_stream(stream), a);
  else
    return TGSR_EINVAL;
  return note_launch(hipGetLastError(), "lp_word_attention_kernel");
}
