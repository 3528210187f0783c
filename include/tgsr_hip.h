/*
 * tgsr_hip.h - C ABI of libtgsr_hip.so: the MI355X (gfx950 / CDNA4) kernels of the TGSR
 * text-conditioned super-resolution hot path.
 *
 * The reference (cxm12/TGSR) has no FFI layer: its boundary is Python module/class names backed by stock
 * torch ops.  Each entry point below replaces the torch call sequence cited next to it (paths relative to
 * the reference root).  Conventions for every function:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless named host_*;
 *   - fp32, NCHW, dense inner (C,H,W) layout; the batch stride is explicit (in elements) so a kernel can read
 *     or write a channel slice of a wider buffer (this is how the reference's torch.cat, util.py:771/817,
 *     disappears);
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - never allocates, never synchronises, no global mutable state (the attention mask is an argument,
 *     not module state as in GlobalAttention.py:84-85), safe to capture into a hipGraph;
 *   - returns TGSR_OK (0) or a negative TGSR_E* code; nothing throws.
 */
#ifndef TGSR_HIP_H
#define TGSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGSR_OK 0
#define TGSR_EINVAL (-1)       /* bad pointer / size / flag combination */
#define TGSR_EUNSUPPORTED (-2) /* shape outside what the kernels tile for (see each function) */
#define TGSR_ELAUNCH (-3)      /* hipLaunch reported an error; see tgsr_last_error() */

/* Epilogue selectors for tgsr_conv3x3_fwd. */
#define TGSR_EPI_AFFINE 0     /* y = conv*scale + bias                        (conv3x3 + BatchNorm2d eval) */
#define TGSR_EPI_AFFINE_GLU 1 /* y = a[:C/2] * sigmoid(a[C/2:]), a = affine   (+ GLU, util.py:45-53)       */

/* Activation selectors for tgsr_conv_to3_fwd. */
#define TGSR_ACT_NONE 0 /* y = conv                                  (GET_IMAGE_G_noAct, util.py:909-919) */
#define TGSR_ACT_TANH_AXPY 1 /* y = tanh(conv) + alpha * addend      (conv_output + a*SRb, model.py:224,280) */

/* ABI version of this header; tgsr_abi_version() must return the same number. */
#define TGSR_ABI_VERSION 2
int tgsr_abi_version(void);
/* Static string describing the last launch error seen by this process (debug aid). */
const char* tgsr_last_error(void);

/*
 * Re-lay a conv weight [Cout][Cin][K][K] (torch layout) as [ceil(Cin/4)][K*K][4][Cout] with zero-filled
 * channel padding: the order tgsr_conv3x3_fwd streams it into LDS.  Replaces nothing in the reference (layout
 * only); done once per weight version.  wpack must hold tgsr_packed_weight_elems(Cout,Cin,K) floats.
 */
int64_t tgsr_packed_weight_elems(int Cout, int Cin, int K);
int tgsr_pack_conv_weight(const float* w, float* wpack, int Cout, int Cin, int K, void* stream);
/* The pack of the DATA-GRADIENT convolution straight from the forward conv's weight: w is [Cin][Cout][K][K] (the forward
 * layer maps Cout -> Cin channels) and the packed filter is w'[co][ci][ky][kx] = w[ci][co][K-1-ky][K-1-kx] - what
 * autograd's conv_transpose of a stride-1 "same" convolution computes; no flip / transpose / copy kernels. */
int tgsr_pack_conv_weight_dgrad(const float* w, float* wpack, int Cout, int Cin, int K, void* stream);

/*
 * BatchNorm2d (eval) as a per-channel affine: scale = weight / sqrt(running_var + eps),
 * bias' = bias - running_mean * scale.  Replaces nn.BatchNorm2d.eval() at util.py:77,116,119.
 */
int tgsr_bn_fold(const float* weight, const float* bias, const float* running_mean, const float* running_var,
                 float eps, float* scale, float* shift, int C, void* stream);

/*
 * Fused 3x3 convolution (stride 1, zero pad 1, no bias), fp32 on the MFMA units.
 * Replaces, in one launch:
 *   conv3x3 -> BatchNorm2d(eval) -> GLU                 ResBlock.block[0..2] util.py:114-117, im2f util.py:741-744,
 *                                                       convin/residual24/48 model.py:228-232
 *   conv3x3 -> BatchNorm2d(eval) [+ residual]           ResBlock.block[3..4] + `out += residual` util.py:118-129
 *   Upsample(x2, nearest) -> conv3x3 -> BN -> GLU       upBlock util.py:74-80 (upsample=1: the x2 gather is folded
 *                                                       into the LDS tile read, the 4x larger tensor never exists)
 * x        [B][Cin][H][W], batch stride x_bstride elements
 * wpack    from tgsr_pack_conv_weight(K=3)
 * scale/shift [Cout] affine applied to the conv result (NULL,NULL = identity)
 * residual NULL or [B][Cout_out][Ho][Wo] (batch stride res_bstride) added after the affine; only with
 *          TGSR_EPI_AFFINE
 * out      [B][Cout_out][Ho][Wo], Ho = H*(upsample?2:1); Cout_out = Cout/2 with GLU else Cout
 * Supported: Cout % 32 == 0 (Cout % 64 == 0 with GLU); any B, Cin, H, W >= 1.
 */
int tgsr_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* wpack, int Cout,
                     const float* scale, const float* shift, const float* residual, int64_t res_bstride, float* out,
                     int64_t out_bstride, int epilogue, int upsample, void* stream);

/*
 * upBlock (util.py:74-80) by sub-pixel decomposition: Upsample(x2, nearest) -> conv3x3 -> affine (BN eval) -> GLU
 * computed as four 2x2 convolutions on the PRE-upsample tensor with pre-summed taps: 4*Cin MACs per output instead
 * of 9*Cin, same result up to the rounding of the weight sums (tgsr_conv3x3_fwd(upsample=1) is the 9-tap form, kept
 * for the training path).  wpack from tgsr_pack_upconv_weight (tgsr_packed_upconv_weight_elems floats).
 * x [B][Cin][H][W] (batch stride), out [B][Cout/2][2H][2W] (batch stride, 8-byte aligned); Cout % 64 == 0.
 */
int64_t tgsr_packed_upconv_weight_elems(int Cout, int Cin);
int tgsr_pack_upconv_weight(const float* w, float* wpack, int Cout, int Cin, void* stream);
int tgsr_upconv3x3_glu_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* wpack, int Cout,
                           const float* scale, const float* shift, float* out, int64_t out_bstride, void* stream);

/*
 * The same upBlock by Winograd F(2x2, 3x3) on the up-sampled image with the up-sampling folded into the input
 * transform: the doubled rows / columns zero 7 of the 16 Winograd positions, leaving 9 products per 2x2 outputs =
 * 2.25 multiplies per output (sub-pixel form above: 4, direct: 9).  fp32; the transforms only add / subtract.
 * upack from tgsr_pack_upwino_weight (tgsr_packed_upwino_weight_elems floats; glu = 1 groups value channels with
 * their gates and must match the entry point used).  Cout % 64 == 0, Cin % 4 == 0,
 * W % 4 == 0, x 16-byte aligned (batch stride % 4 == 0), out 8-byte aligned with an even batch stride.
 */
int64_t tgsr_packed_upwino_weight_elems(int Cout, int Cin);
int tgsr_pack_upwino_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream);
int tgsr_upwino_glu_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                        const float* scale, const float* shift, float* out, int64_t out_bstride, void* stream);
/* The same without GLU: out [B][Cout][2H][2W] = affine(conv3x3(upsample(x))) (scale/shift may be NULL: the raw
 * convolution BatchNorm's batch statistics are taken of in training).  Pack with glu = 0. */
int tgsr_upwino_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                    const float* scale, const float* shift, float* out, int64_t out_bstride, void* stream);

/*
 * upBlock by Winograd F(4x4, 3x3) on the up-sampled grid with the x2 folded into the input transform: 25 multiplies per
 * 16 outputs (tgsr_upwino_glu_fwd: 36) - the third transformed row / column vanishes and the fifth is -1/3 of the fourth,
 * so 5 x 5 positions carry products and their B operands come from 16 values per tile (tgsr_upwino4.hip).  `glu` != 0:
 * out [B][Cout/2][2H][2W] = GLU(affine(conv3x3(upsample(x)))), else [B][Cout][2H][2W] without the gate (scale/shift may be
 * NULL).  upack from tgsr_pack_upwino4_weight(glu) (tgsr_packed_upwino4_weight_elems floats, per-wave fragment order: the A
 * operands go from L2 straight into registers).  Cout % 64 == 0, Cin % 8 == 0, W % 4 == 0, x / out / upack 16-byte aligned
 * with batch strides % 4 == 0.  Numerics: F(4x4)'s (see
 * tgsr_wino4_conv3x3_fwd); callers route by output size (tgsr_amd.ops.upwino4_wanted).
 */
int64_t tgsr_packed_upwino4_weight_elems(int Cout, int Cin);
int tgsr_pack_upwino4_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream);
int tgsr_upwino4_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                     const float* scale, const float* shift, float* out, int64_t out_bstride, int glu, void* stream);

/*
 * The same fused 3x3 convolution as tgsr_conv3x3_fwd (upsample = 0) by Winograd F(2x2, 3x3): 16 multiplies per 4
 * outputs instead of 36.  fp32 throughout; the transforms only use {0, +-1, +-1/2} (end-to-end error on the shipped
 * checkpoint indistinguishable from the direct fp32 form).  upack from tgsr_pack_wino_weight
 * (tgsr_packed_wino_weight_elems floats; `glu` != 0 groups each value channel block with its gate block and must
 * match the epilogue the pack is used with).  Cout % 32 == 0 (64-channel groups when Cout % 64 == 0, else 32), Cin % 4 == 0, W % 4 == 0, x 16-byte aligned, out / residual 8-byte
 * aligned with even batch strides; same epilogue selectors and residual rule as tgsr_conv3x3_fwd.
 */
int64_t tgsr_packed_wino_weight_elems(int Cout, int Cin);
int tgsr_pack_wino_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream);
/* Winograd pack of the data-gradient convolution from the forward weight [Cin][Cout][3][3] (see
 * tgsr_pack_conv_weight_dgrad); plain channel order (no GLU grouping). */
int tgsr_pack_wino_weight_dgrad(const float* w, float* upack, int Cout, int Cin, void* stream);
int tgsr_wino_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                          const float* scale, const float* shift, const float* residual, int64_t res_bstride,
                          float* out, int64_t out_bstride, int epilogue, void* stream);

/*
 * The same convolution by Winograd F(4x4, 3x3): 36 multiplies per 16 outputs (1.78x fewer than F(2x2), 4x fewer than the
 * direct form), for the large layers of the inference path - ResBlock.block util.py:110-130 at 128^2 and above.  fp32
 * throughout; the transforms hold 4, 5, 8 and 1/24, so its rounding error is ~15x that of the other two forms per layer
 * (3e-5 on unit-scale data): on the shipped checkpoint the finest image stays at 3.0e-5 (max, against fp64) with the 128^2
 * and 64^2 layers on this kernel, 2.1e-4 with the 32^2 layers on it too (profiles/HISTORY.md 3.1e) - callers route by layer size.  upack from
 * tgsr_pack_wino4_weight (tgsr_packed_wino4_weight_elems floats; U = G g G^T computed in double, rounded once; `glu` must
 * match the epilogue).  Cout % 64 == 0, Cin % 4 == 0, W % 4 == 0; x, out, residual 16-byte aligned with batch strides
 * % 4 == 0; same epilogue selectors and residual rule as tgsr_conv3x3_fwd.
 */
int64_t tgsr_packed_wino4_weight_elems(int Cout, int Cin);
int tgsr_pack_wino4_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream);
/* F(4x4) pack of the data-gradient convolution from the forward weight [Cin][Cout][3][3] (see tgsr_pack_conv_weight_dgrad). */
int tgsr_pack_wino4_weight_dgrad(const float* w, float* upack, int Cout, int Cin, void* stream);
int tgsr_wino4_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                           const float* scale, const float* shift, const float* residual, int64_t res_bstride,
                           float* out, int64_t out_bstride, int epilogue, void* stream);

/* The register-fed form of the same kernel: the A operands straight from L2 in per-wave fragment order.  Cout % 128 == 0: a
 * workgroup = one tile row x 128 accumulator rows (the input transform amortised over twice the channels); else 64-row groups
 * in 4-wave workgroups, two independent per CU.  Same contract with Cin % 8 == 0; its own pack layout
 * (tgsr_packed_wino4_weight_elems floats, 16-byte aligned). */
int tgsr_pack_wino4_wide_weight(const float* w, float* upack, int Cout, int Cin, int glu, void* stream);
/* ... of the data-gradient convolution, from the forward weight [Cin][Cout][3][3] (see tgsr_pack_conv_weight_dgrad) */
int tgsr_pack_wino4_wide_weight_dgrad(const float* w, float* upack, int Cout, int Cin, void* stream);
/* ... with the statistics epilogue of tgsr_wino4_conv3x3_stats_fwd (training forward): one pair per channel and 4 x 64 wave tile */
int tgsr_wino4_wide_stats_nslots(int B, int H, int W, int Cout);
int tgsr_wino4_wide_conv3x3_stats_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack,
                                      int Cout, float* out, int64_t out_bstride, float* stat_partial, void* stream);
int tgsr_wino4_wide_conv3x3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack, int Cout,
                                const float* scale, const float* shift, const float* residual, int64_t res_bstride,
                                float* out, int64_t out_bstride, int epilogue, void* stream);

/*
 * tgsr_conv_to3_set_pipe(on): the streaming form of tgsr_conv_to3_fwd with its LDS copies two stages ahead in a ring of three buffers
 * (counted waits) and the filter in LDS - on by default where W % 4 == 0 and the filter takes <= 16 KB (TGSR_TO3_PIPE=0 in the
 * environment turns it off); the images are bit-identical either way.  Returns the previous setting.
 */
int tgsr_conv_to3_set_pipe(int on);

/*
 * KxK convolution (K = 3 or 5, stride 1, zero pad K/2, no bias) to 3 output channels + optional epilogue.
 * Replaces GET_IMAGE_G_noAct.img (util.py:913-915; K=3, TGSR_ACT_NONE) and
 * conv_output = conv5x5 + Tanh followed by `one*. + a*SRb` (model.py:224, 280/288/297; K=5, TGSR_ACT_TANH_AXPY).
 * x [B][Cin][H][W] (batch stride x_bstride), w [3][Cin][K][K] (torch layout), addend NULL or [B][3][H][W] dense,
 * out [B][3][H][W] dense.
 */
int tgsr_conv_to3_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* w, int K,
                      int act, const float* addend, float alpha, float* out, void* stream);

/*
 * Word-level attention of the generator (GlobalAttentionGeneral.forward, GlobalAttention.py:87-130) in two
 * launches: the 1x1 projection of the word embeddings (:100-102) and a fused QK^T -> mask -> softmax over words
 * -> PV kernel on MFMA (:107-128).
 * h        [B][idf][Q] (batch stride h_bstride), idf in {32,64,128}
 * words    [B][cdf][T] dense, T <= 32;  w_ctx [idf][cdf]
 * mask     NULL or uint8 [B][T], non-zero = padded word (captions == 0, trainer_objective.py:136-140)
 * mask_mode 0 = reference behaviour: score row b*Q+q is masked with mask[(b*Q+q) % B]
 *               (`mask.repeat(queryL,1)`, GlobalAttention.py:111); 1 = per-sample masking (mask[b])
 * src_ws   workspace, B*idf*32 floats (the projected words, zero padded to 32).  With words == w_ctx == NULL the
 *          projection is skipped and src_ws is read as the output of tgsr_word_project_fwd.
 * c_code   [B][idf][Q] (batch stride c_bstride);  attn [B][T][Q] dense (may be NULL: not written)
 */
int tgsr_word_attention_fwd(const float* h, int64_t h_bstride, const float* words, const float* w_ctx,
                            const uint8_t* mask, int mask_mode, int B, int idf, int cdf, int T, int Q, float* src_ws,
                            float* c_code, int64_t c_bstride, float* attn, void* stream);

/*
 * The projection alone, for up to 4 conv_context weight sets over the same words in one launch (the generator's
 * stages share the word embeddings): src_out[set][B][idf][32] = w_ctx[set] [idf][cdf] x words [B][cdf][T], zero padded
 * to 32 words.  w_ctx: HOST array of nsets device pointers.  idf % 32 == 0, T <= 32.
 */
int tgsr_word_project_fwd(const float* words, const float* const* w_ctx, int nsets, int B, int idf, int cdf, int T,
                          float* src_out, void* stream);

/*
 * Bidirectional 1-layer LSTM text encoder, eval mode (RNN_ENCODER.forward util.py:233-260: Embedding ->
 * pack_padded_sequence -> LSTM -> pad_packed_sequence -> transpose; final hidden = sentence code).
 * captions int64 [B][width]; cap_lens int32 [B] (each 1..Tmax <= width); emb [ntoken][ninput];
 * w_ih [2][4H][ninput], w_hh [2][4H][H], b_ih/b_hh [2][4H]  (direction 0 = forward, 1 = reverse; gate order i,f,g,o)
 * gates_ws workspace B*Tmax*2*4H floats.
 * words_emb [B][2H][Tmax] (zero beyond each length), sent_emb [B][2H].   H <= 256, 4H % 64 == 0.
 */
int tgsr_bilstm_fwd(const int64_t* captions, int width, const int32_t* cap_lens, int B, int Tmax, const float* emb,
                    int ntoken, int ninput, const float* w_ih, const float* w_hh, const float* b_ih,
                    const float* b_hh, int H, float* gates_ws, float* words_emb, float* sent_emb, void* stream);

/*
 * Eval-mode form of the same encoder: with frozen weights the input projection depends on the token only, so
 * tgsr_lstm_gate_table builds table[ntoken][2][4H] = emb[tok] . w_ih[d]^T + b_ih[d] + b_hh[d] once per weight version
 * (the same GEMM kernel, same arithmetic per row) and tgsr_bilstm_table_fwd runs the recurrence reading its gate
 * pre-activations through the caption: one launch per batch instead of two, bit-identical to tgsr_bilstm_fwd.
 */
int tgsr_lstm_gate_table(const float* emb, int ntoken, int ninput, const float* w_ih, const float* b_ih,
                         const float* b_hh, int H, float* table, void* stream);
int tgsr_bilstm_table_fwd(const int64_t* captions, int width, const int32_t* cap_lens, int B, int Tmax,
                          const float* table, int ntoken, const float* w_hh, int H, float* words_emb, float* sent_emb,
                          void* stream);

/*
 * The GRU branch of RNN_ENCODER (util.py:207-211: `nn.GRU(ninput, nhidden, 1, batch_first=True, bidirectional=True)` when
 * cfg.RNN_TYPE == 'GRU'; forward util.py:233-260, sent_emb = the final hidden states, :257), eval mode, same packed-sequence
 * semantics.  Gate order r, z, n.  tgsr_gru_gate_table: table[ntoken][2][3H] = emb[tok] . w_ih[d]^T + b_ih[d] + b_hh_rz[d], where
 * b_hh_rz [2][3H] = b_hh with its n third zeroed (b_hn enters inside r * (W_hn h + b_hn)); tgsr_bigru_table_fwd: the recurrence
 * through the caption, b_hn [2][H].  words_emb [B][2H][Tmax] (zero behind a caption), sent_emb [B][2H].  H in {32, 64, 128}.
 */
int tgsr_gru_gate_table(const float* emb, int ntoken, int ninput, const float* w_ih, const float* b_ih, const float* b_hh_rz,
                        int H, float* table, void* stream);
int tgsr_bigru_table_fwd(const int64_t* captions, int width, const int32_t* cap_lens, int B, int Tmax, const float* table,
                         int ntoken, const float* w_hh, const float* b_hn, int H, float* words_emb, float* sent_emb,
                         void* stream);

/*
 * Training path of the GRU branch (the reference trains whichever cell cfg.RNN_TYPE names, pretrain_DAMSM.py:49-98 through
 * util.py:233-260).  tgsr_bigru_train_fwd: x [B][Tmax][ninput] = the embedded, dropped-out captions; gates_ws [B*Tmax][2][3H]
 * workspace; acts [B][Tmax][2][4][H] = (r, z, n, W_hn h + b_hn) of every step.  tgsr_bigru_bwd walks every (sample, direction) back
 * through time from d_words [B][2H][Tmax] and d_sent [B][2H] (NULL = zero): dgx / dgh [B][Tmax][2][3H] = the gradients at the
 * input-side (x W_ih^T + b_ih) and hidden-side (h W_hh^T + b_hh) gate pre-activations, hprev [B][Tmax][2][H] = h of the previous
 * step, dbias [2][2][3H] = (d b_ih, d b_hh) (NULL: skipped).  The caller's GEMMs give dW_ih = dgx^T x, dW_hh = dgh^T hprev,
 * dx = dgx W_ih.  Fixed summation orders.  H in {32, 64, 128}.
 */
int tgsr_bigru_train_fwd(const float* x, const int32_t* cap_lens, int B, int Tmax, int ninput, const float* w_ih,
                         const float* w_hh, const float* b_ih, const float* b_hh_rz, const float* b_hn, int H, float* gates_ws,
                         float* acts, float* words_emb, float* sent_emb, void* stream);
int tgsr_bigru_bwd(const int32_t* cap_lens, int B, int Tmax, int H, const float* w_hh, const float* acts, const float* words_emb,
                   const float* d_words, const float* d_sent, float* dgx, float* dgh, float* hprev, float* dbias, void* stream);

/*
 * Training path of the same encoder.  tgsr_bilstm_train_fwd takes the already embedded (and dropped-out) inputs
 * x [B][Tmax][ninput] and additionally saves acts [B][Tmax][2][5][H] (i, f, g, o activations and the cell state of
 * every step).  tgsr_bilstm_bwd walks every (sample, direction) back through time from d_words [B][2H][Tmax] and
 * d_sent [B][2H] (may be NULL) and emits the pre-activation gate gradients dgates [B][Tmax][2][4H] (zero past the
 * caption), hprev [B][Tmax][2][H] (the hidden state entering each step) and dbias [2][4H] (= d b_ih = d b_hh; may be
 * NULL).  The weight / input gradients are plain GEMMs on those (tgsr_linear_fwd):
 *   dW_ih[d] = dgates[:, :, d]^T x,   dW_hh[d] = dgates[:, :, d]^T hprev[:, :, d],   dx = dgates W_ih.
 * H in {32, 64, 128} for the backward.
 */
int tgsr_bilstm_train_fwd(const float* x, const int32_t* cap_lens, int B, int Tmax, int ninput, const float* w_ih,
                          const float* w_hh, const float* b_ih, const float* b_hh, int H, float* gates_ws, float* acts,
                          float* words_emb, float* sent_emb, void* stream);
int tgsr_bilstm_bwd(const int32_t* cap_lens, int B, int Tmax, int H, const float* w_hh, const float* acts,
                    const float* words_emb, const float* d_words, const float* d_sent, float* dgates, float* hprev,
                    float* dbias, void* stream);

/*
 * DAMSM word/region attention for the whole (image, caption) grid in one launch: func_attention
 * (GlobalAttention.py:33-74) as driven by words_loss (losses.py:73-113) - the B-iteration Python loop, word.repeat,
 * both softmaxes, the two bmm and the cosine / exp / sum / log tail.
 * words [B][ndf][Tw] dense (Tw <= 32), cap_lens int32 [B] (NULL = Tw for all), ctx [B][ndf][S] dense
 * (S = 17*17 <= 320), ndf % 32 == 0, ndf <= 512.
 * sim      [B_img][B_cap] = log sum_{w < len} exp(gamma2 * cos(word_w, region-context_w))   (losses.py:102-109;
 *          the caller multiplies by gamma3 and applies the class mask / cross entropy)
 * att_diag NULL or [B][Tw][S]: attention of image i on caption i, rows >= cap_lens[i] zero       (losses.py:93)
 */
int tgsr_damsm_words_fwd(const float* words, const int32_t* cap_lens, const float* ctx, int B, int ndf, int Tw, int S,
                         float gamma1, float gamma2, float* sim, float* att_diag, void* stream);

/*
 * Backward of tgsr_damsm_words_fwd (the attention maps carry no gradient, like in the reference): grad_sim
 * [B_img][B_cap] -> grad_words32 [B][ndf][32] (word axis padded to 32; the caller keeps [:Tw]) and grad_ctx
 * [B][ndf][S].  Each (image, caption) pair is recomputed by one workgroup; the per-pair gradients are summed in a
 * fixed order (deterministic).  ws: tgsr_damsm_words_bwd_ws_elems(B, ndf, S) floats.  ndf % 32 == 0, ndf <= 256,
 * Tw <= 32, S <= 320.
 */
int64_t tgsr_damsm_words_bwd_ws_elems(int B, int ndf, int S);
int tgsr_damsm_words_bwd(const float* words, const int32_t* cap_lens, const float* ctx, const float* grad_sim, int B,
                         int ndf, int Tw, int S, float gamma1, float gamma2, float* ws, float* grad_words32,
                         float* grad_ctx, void* stream);

/*
 * Stand-alone func_attention(query, context, gamma1) (GlobalAttention.py:33-74): pair p attends query p to context p.
 * query [B][ndf][L] (L <= 32), context [B][ndf][S] (S <= 320) -> weighted_context [B][ndf][L], attn [B][L][S].
 */
int tgsr_func_attention_fwd(const float* query, const float* context, int B, int ndf, int L, int S, float gamma1,
                            float* weighted_context, float* attn, void* stream);

/*
 * The two trainable heads of CNN_ENCODER (util.py:300-301, 364-367) as fp32 MFMA GEMMs with bias:
 * tgsr_conv1x1_fwd : emb_features, conv1x1 Cin -> Cout on [B][Cin][S] (S = ih*iw, 17*17) -> out [B][Cout][S]
 * tgsr_linear_fwd  : emb_cnn_code, out[b][o] = sum_k w[o][k] x[b][k] + bias[o]; x [B][K], w [Cout][K]
 * bias may be NULL.  (The Inception-v3 trunk that produces their inputs is third-party torchvision arithmetic.)
 */
int tgsr_conv1x1_fwd(const float* x, int B, int Cin, int S, const float* w, const float* bias, int Cout, float* out,
                     void* stream);
int tgsr_linear_fwd(const float* x, int B, int K, const float* w, const float* bias, int Cout, float* out,
                    void* stream);

/*
 * The discriminators' logit heads (nn.Conv2d(8 ndf, 1, kernel_size=4, stride=4) on a 4x4 map, called by
 * losses.py:292-316 / 359-366 as netD.COND_DNET / UNCOND_DNET): out[b] = sum_k x[b][k] w[k] + bias[0], one workgroup
 * per sample, fixed-order reduction.  Backward: dx[b][k] = dy[b] w[k] (dx may be NULL), dw[k] = sum_b dy[b] x[b][k]
 * (dw may be NULL).  x, w 16-byte aligned.
 */
int tgsr_rowdot_fwd(const float* x, const float* w, const float* bias, float* out, int B, int K, void* stream);

/*
 * CA_NET.forward (util.py:372-400) in one launch: x = fc(sent_emb) (w [4 ncf][tdim], bias [4 ncf]); GLU halves;
 * mu = h[:ncf], logvar = h[ncf:] ([B][ncf] each); c_code = eps * exp(0.5 * logvar) + mu with eps [B][ncf] standard
 * normals drawn by the caller (the reference draws them from torch's generator, util.py:388-396).  c_code and eps may
 * both be NULL (the x8 generators discard c_code, model.py:51-52).
 */
int tgsr_ca_net_fwd(const float* sent_emb, const float* w, const float* bias, const float* eps, int B, int tdim, int ncf,
                    float* c_code, float* mu, float* logvar, void* stream);
int tgsr_rowdot_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, int B, int K, void* stream);

/*
 * The text tail of an inference step in ONE launch (trainer_objective.py:134-146 between the text encoder and the
 * generator): the conv_context projections of tgsr_word_project_fwd (src_out [nsets][B][idf][32]), CA_NET's mu / logvar
 * [B][ncf] of tgsr_ca_net_fwd (same arithmetic; c_code is not produced - the x8 / x16 generators discard it,
 * model.py:51-52) and mask[B][T] = (captions[b][t] == 0) as bytes 0 / 1 (trainer_objective.py:136-140; torch.bool storage).
 * captions int64 [B][width], width >= T.  tdim % 16 == 0 (sent_emb / ca_w 16-byte aligned when tdim % 64 == 0); other
 * limits as tgsr_word_project_fwd.
 */
int tgsr_text_tail_fwd(const float* words, const float* const* w_ctx, int nsets, int B, int idf, int cdf, int T,
                       float* src_out, const float* sent_emb, const float* ca_w, const float* ca_b, int tdim, int ncf,
                       float* mu, float* logvar, const int64_t* captions, int width, uint8_t* mask, void* stream);

/*
 * tgsr_text_tail_fwd that ALSO leaves what the reduced-precision generators need to attend to the words inside the kernels
 * that produce h (tgsr_lp_stem_att_fwd, tgsr_lp_upconv_glu_att_fwd): `att_pack`, tgsr_lp_att_pack_bytes(nsets, B) bytes,
 * 16-byte aligned =
 *     [nsets][B][4 fragments][64 lanes][8] elements of `lp_dtype`: the projections src_out rounded once and laid out as
 *         the MFMA 32x32x16 A fragments of the scores GEMM (0, 1) and the context GEMM (2, 3);
 *     [B] uint32: bit t = (captions[b][t] == 0), t < T.
 * idf must be 32 (the generators' ngf).  src_out / mu / logvar / mask as tgsr_text_tail_fwd, bit-identical.
 */
int64_t tgsr_lp_att_pack_bytes(int nsets, int B);
int tgsr_text_tail_lp_fwd(const float* words, const float* const* w_ctx, int nsets, int B, int idf, int cdf, int T,
                          float* src_out, const float* sent_emb, const float* ca_w, const float* ca_b, int tdim, int ncf,
                          float* mu, float* logvar, const int64_t* captions, int width, uint8_t* mask, int lp_dtype,
                          void* att_pack, void* stream);

/*
 * n <= 16 dense device-to-device copies in one launch: dst / src / nbytes are HOST arrays; sizes and addresses must be
 * multiples of 4 bytes.  (GraphedStep.replay: the new inputs of every lane go into the captured step's static buffers
 * with one launch instead of three hipMemcpyAsync per lane.)
 */
int tgsr_multi_copy(int n, void* const* dst, const void* const* src, const int64_t* nbytes, void* stream);

/*
 * The gradient collective of the data-parallel path behind the C ABI (`allreduce_flat`, an RCCL wrapper; the
 * reference has no collective - trainer_objective.py:31 is single-process).  tgsr_allreduce_flat: buf[i] = scale * sum over ranks
 * of buf[i], in place, n floats, asynchronous on `stream` (ncclAllReduce, then one scaling launch unless scale == 1) - what
 * tgsr_amd.parallel.FlatGradBucket.all_reduce_mean does through torch.distributed by default.  Communicators: rank 0 calls
 * tgsr_comm_unique_id(id) (128 bytes), the caller carries those bytes to every rank, every rank calls tgsr_comm_init(&comm, id,
 * rank, world) with its device current; tgsr_comm_destroy(comm) at the end.  librccl is opened lazily (dlopen): where it is absent
 * tgsr_comm_available() returns 0 and the other four return TGSR_EUNSUPPORTED - the rest of the library is unaffected.
 */
int tgsr_comm_available(void);
int tgsr_comm_unique_id(void* id128);
int tgsr_comm_init(void** comm, const void* id128, int rank, int world);
int tgsr_allreduce_flat(void* comm, float* buf, int64_t n, float scale, void* stream);
int tgsr_comm_count(void* comm, int* world, int* rank); /* ncclCommCount / ncclCommUserRank of a communicator: what RCCL itself says */
int tgsr_comm_destroy(void* comm);

/*
 * out[i] = t[i] + alpha * s[i] for n <= 4 dense fp32 images in one launch (host arrays of device pointers; numel[i] a
 * multiple of 4, pointers 16-byte aligned).  The second half of NetG_highweight's heads (model.py:280, 288, 297:
 * `ims = one * conv_output(out) + a * SRb`): the caller computes t = tanh(conv5x5(out)) with tgsr_conv_to3_fwd(addend = NULL)
 * on the high-frequency branch's own stream - it needs nothing of G_SR_NET_low - and only this axpy waits for the
 * low-frequency images.  fma(alpha, s, t), exactly what tgsr_conv_to3_fwd's epilogue evaluates when given the addend.
 */
int tgsr_axpy_images(int n, float* const* out, const float* const* t, const float* const* s, const int64_t* numel, float alpha,
                     void* stream);

/*
 * NetG_highweight(weightmap=True) (model.py:235-245, 276-297; models16.py:119-125, 149-178): `ims_k = one_k * conv_output(out) +
 * a_k * SRb_k` with a_k a trainable [H, W] map (initial value 1) broadcast over batch and channels.
 *   fwd: out[bc][p] = t[bc][p] + amap[p] * s[bc][p]     (t = conv_output(out): tgsr_conv_to3_fwd without addend)
 *   bwd: ds = amap * dy (ds may be NULL), damap[p] = sum_bc dy[bc][p] * s[bc][p] (damap may be NULL), planes added in index order.
 * BC = B * 3 planes of HW pixels, dense; HW % 4 == 0, pointers 16-byte aligned (else TGSR_EUNSUPPORTED).
 */
/* ------------------------------------------------------------------------------------------------------------------
 * CNN_ENCODER's frozen Inception-v3 trunk (util.py:263-368, every parameter `requires_grad = False`, util.py:274-275; run by
 * generator_loss for the DAMSM ranking term, losses.py:375-389): forward and the gradient with respect to the image.
 *
 * tgsr_gconv: a convolution of any KH x KW (KH KW <= 64), stride 1 | 2, zero padding, as one implicit GEMM on the fp32 MFMA.
 *   dgrad = 0  forward.  A = w' [M = Cout][K = Cin KH KW] (tgsr_gconv_pack(dgrad = 0): the filter times the folded BatchNorm scale),
 *              S = x based at its first channel, [B] samples `s_bstride` floats apart, Hs x Ws pixels; pixel grid PH x PW = the
 *              OUTPUT size; out[b][m][PH][PW] (+)= relu?(sum + bias[m]) based at the output's channel slice, samples
 *              `o_bstride` apart (a block's branches write straight into their slices of its concatenation).
 *   dgrad = 1  data gradient.  A = w'T [M = Cin][K = Cout KH KW] (tgsr_gconv_pack(dgrad = 1)), S = g = d(loss)/d(pre-activation) with
 *              the forward's OUTPUT size Hs x Ws, pixel grid PH x PW = the forward's INPUT size; out = dx (accumulate = 1: +=,
 *              a block input collects its branches' gradients one after the other).  bias / relu must be 0.
 *   mask (nullable; laid out like `out`, same batch stride): the contribution is kept where mask > 0 and dropped elsewhere - the
 *              ReLU of the tensor whose gradient is being written (a 0 / 1 factor distributes over the sum of a tensor's consumers,
 *              so every consumer applies it to its own contribution and no separate mask pass runs).  Also on the pool backwards.
 *   ws: tgsr_gconv_ws_elems(B, M, PH, PW, K) floats (0 when the shape is not split over K).  Deterministic (slabs summed in order).
 * tgsr_maxpool3s2_*: F.max_pool2d(x, 3, stride 2); the backward routes dy to the FIRST maximum of a window (torch's rule).
 * tgsr_avgpool3: F.avg_pool2d(x, 3, 1, 1) (count_include_pad); symmetric, so it is also its own backward (accumulate = 1).
 * tgsr_plane_mean(_bwd): the 8 x 8 global average (F.avg_pool2d(x, 8) on an 8 x 8 map).  tgsr_relu_mask: out = dy * (y > 0).
 * tgsr_bilinear_*: nn.Upsample(size = (OH, OW), mode = 'bilinear') (align_corners = False), util.py:310, on dense planes.
 */
int tgsr_gconv_set_form(int split); /*
 * out[k] = ((parts[0][k] + parts[1][k]) + ...) over n dense tensors of m floats stacked back to back (m % 4 == 0, 16-byte aligned):
 * the branches' contributions to a Mixed block's input gradient (util.py:281-298's blocks), each written by its own branch's stream,
 * summed in the order a one-stream walk accumulates them.
 */
int tgsr_sum_stack(const float* parts, int n, int64_t m, float* out, void* stream);

/*
 * dx[b][c][2Y + py][2X + px] (+)= t_{py px}[b][c][Y][X] over `planes` = B * C dense planes of H x W: weaves the four parity classes of
 * a stride-2 convolution's data gradient - each a stride-1 data gradient over its own taps (tgsr_gconv), dense at half resolution,
 * ceil((H - py) / 2) x ceil((W - px) / 2) pixels - into the gradient of the input (util.py:281-298's stride-2 layers: a quarter of the
 * products of the direct form).  mask (dense, like dx) keeps the value where mask > 0; accumulate != 0 adds.
 */
int tgsr_interleave2x2(const float* t00, const float* t01, const float* t10, const float* t11, float* dx, int64_t planes, int H, int W,
                       int accumulate, const float* mask, void* stream);

/* 1 (default): three-piece bf16 form where K % 16 == 0, <= 25 taps, stride-1 data gradient; 0: fp32 MFMA.  Returns the old value */
int tgsr_gconv_nsplit(int M, int N, int K);
int64_t tgsr_gconv_ws_elems(int B, int M, int PH, int PW, int K);
int tgsr_gconv(int dgrad, const float* A, const float* S, int64_t s_bstride, int B, int Hs, int Ws, int M, int K, int PH, int PW,
               int KH, int KW, int stride, int padh, int padw, const float* bias, int relu, int accumulate, const float* mask,
               float* out, int64_t o_bstride, float* ws, void* stream);
int tgsr_gconv_pack(const float* w, const float* scale, float* out, int Cout, int Cin, int KK, int dgrad, void* stream);
int tgsr_maxpool3s2_fwd(const float* x, int64_t x_bstride, int B, int C, int H, int W, float* out, int64_t o_bstride, void* stream);
int tgsr_maxpool3s2_bwd(const float* x, int64_t x_bstride, const float* dy, int64_t dy_bstride, int B, int C, int H, int W, float* dx,
                        int64_t dx_bstride, int accumulate, const float* mask, void* stream);
int tgsr_avgpool3(const float* x, int64_t x_bstride, int B, int C, int H, int W, float* out, int64_t o_bstride, int accumulate,
                  const float* mask, void* stream);
int tgsr_plane_mean(const float* x, float* out, int64_t planes, int HW, void* stream);
int tgsr_plane_mean_bwd(const float* dy, float* dx, int64_t planes, int HW, void* stream);
int tgsr_relu_mask(const float* dy, int64_t dy_bstride, const float* y, int64_t y_bstride, float* out, int64_t o_bstride, int B,
                   int64_t per_sample, void* stream);
int tgsr_bilinear_fwd(const float* x, int64_t planes, int H, int W, int OH, int OW, float* out, void* stream);
int tgsr_bilinear_bwd(const float* dy, int64_t planes, int H, int W, int OH, int OW, float* dx, void* stream);

/*
 * The adversarial terms of discriminator_loss / generator_loss (losses.py:290-316, 358-371) in one launch: out[0] = sum_i
 * weight[i] * BCEWithLogits(l[i], target[i]) over l = [a (na logits); b (nb logits, may be NULL / 0)] - the reference's up to five
 * mean-reduced nn.BCEWithLogitsLoss calls with their 1/n means and /2, /3 combination folded into `weight`.  One workgroup, fixed
 * summation order.  bwd: da / db = dy[0] * weight * (sigmoid(l) - target) (either may be NULL).
 */
int tgsr_weighted_bce_fwd(const float* a, int na, const float* b, int nb, const float* target, const float* weight, float* out,
                          void* stream);
int tgsr_weighted_bce_bwd(const float* dy, const float* a, int na, const float* b, int nb, const float* target, const float* weight,
                          float* da, float* db, void* stream);

/*
 * Adam over flat fp32 buffers (torch.optim.Adam's update, trainer_objective.py's optimizers: no amsgrad; weight_decay = L2 added to
 * the gradient), one pass: param, grad, exp_avg, exp_avg_sq are dense buffers of n floats, 16-byte aligned.  state = 3 device
 * floats [step, 1 - beta1^step, sqrt(1 - beta2^step)]: advance != 0 first bumps the step and recomputes the two corrections on the
 * device (a one-thread launch: the count survives hipGraph replays); a caller with several parameter ranges advances once and
 * passes advance = 0 for the others.  The hyper-parameters arrive as the doubles the caller holds: 1 - beta is rounded to fp32 from
 * the double difference, as torch does (1.f - 0.999f is 4.7e-5 away from 0.001f).
 */
int tgsr_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* state, int64_t n, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int advance, void* stream);

int tgsr_axpy_map_fwd(const float* t, const float* s, const float* amap, float* out, int BC, int HW, void* stream);
int tgsr_axpy_map_bwd(const float* dy, const float* s, const float* amap, float* ds, float* damap, int BC, int HW, void* stream);

/*
 * Eval-mode BatchNorm2d + activation on a raw convolution output, dense NCHW: out = act(raw * scale[c] + shift[c]) with
 * scale / shift from tgsr_bn_fold (running statistics) - downBlock / Block3x3_leakRelu under .eval() (util.py:92-98; the
 * train-mode form is tgsr_bn_train_fwd).  act: 0 none, 2 LeakyReLU(0.2) (the selector values of tgsr_bn_train_fwd's `glu`).
 * bwd: draw = dy * act'(out) * scale[c] (out = the forward output; may be NULL for act 0).  HW % 4 == 0, 16-byte aligned.
 */
int tgsr_affine_act_fwd(const float* raw, const float* scale, const float* shift, float* out, int B, int C, int HW, int act,
                        void* stream);
int tgsr_affine_act_bwd(const float* dy, const float* out, const float* scale, float* draw, int B, int C, int HW, int act,
                        void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training path (BatchNorm2d batch statistics + backward).  The reference trains through torch autograd over
 * nn.Conv2d / nn.BatchNorm2d(train) / GLU / nn.Upsample (util.py:74-80, 110-130); these entry points are the
 * hand-written forward/backward of the same blocks.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Number of partial-sum splits tgsr_bn_train_* use for (B, C, HW); partial_ws must hold C * nsplit * 4 floats. */
/*
 * Small layers (one workgroup per channel covers the batch: B * HW < 8192, no GLU - the discriminators' 16^2 .. 4^2 maps) run the
 * statistics / reduce pass INSIDE the normalise / apply kernel: one launch instead of two, bit-identical results.
 * tgsr_bn_set_fuse_small(on): default on (TGSR_BN_FUSE_SMALL=0 in the environment turns it off); returns the previous setting.
 */
int tgsr_bn_set_fuse_small(int on);
int tgsr_bn_train_nsplit(int B, int C, int HW);

/*
 * nn.BatchNorm2d in training mode (+ GLU | + residual) on the raw conv output: batch mean / biased variance per
 * channel over (B, H, W), running statistics updated in place with `momentum` and the unbiased variance (either
 * both NULL or both given), y = GLU(bn(raw)) (glu=1, C even, out has C/2 channels) or bn(raw) (+ residual).
 * `glu` is the activation selector: 0 = none (+ residual), 1 = GLU, 2 = LeakyReLU(0.2) (downBlock util.py:92-98 and
 * the discriminators' conv3x3 -> BN -> LeakyReLU blocks; no residual).
 * raw [B][C][HW] dense, HW % 4 == 0.  Saves mean/invstd/scale/shift [C] for the backward.  num_batches_tracked (device
 * int64 scalar, may be NULL) is incremented by one, as nn.BatchNorm2d.forward does in training mode.
 */
int tgsr_bn_train_fwd(const float* raw, int B, int C, int HW, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, int glu, const float* residual,
                      int64_t res_bstride, float* partial_ws, float* mean, float* invstd, float* scale, float* shift,
                      float* out, int64_t out_bstride, int64_t* num_batches_tracked, void* stream);

/*
 * The statistics pass folded into the producing convolution.  tgsr_wino_conv3x3_stats_fwd = tgsr_wino_conv3x3_fwd with the
 * plain epilogue (no affine, no residual: the raw output BatchNorm's batch statistics are taken of) that also writes
 * stat_partial [Cout][nslots][2]: one (sum, sum of squares) pair per channel and wave tile, nslots =
 * tgsr_wino_stats_nslots(B, H, W, Cout).  tgsr_bn_train_fwd_from_stats = tgsr_bn_train_fwd without its pass over the raw
 * tensor: it combines those pairs (double precision, fixed order) instead.
 */
int tgsr_wino_stats_nslots(int B, int H, int W, int Cout);
int tgsr_wino_conv3x3_stats_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack,
                                int Cout, float* out, int64_t out_bstride, float* stat_partial, void* stream);
/* The same for the F(4x4, 3x3) kernel (training forward of the 128 x 128 layers): one pair per channel and 4 x 64 wave tile. */
int tgsr_wino4_stats_nslots(int B, int H, int W, int Cout);
int tgsr_wino4_conv3x3_stats_fwd(const float* x, int64_t x_bstride, int B, int Cin, int H, int W, const float* upack,
                                 int Cout, float* out, int64_t out_bstride, float* stat_partial, void* stream);
int tgsr_bn_train_fwd_from_stats(const float* raw, int B, int C, int HW, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, int glu, const float* residual,
                                 int64_t res_bstride, const float* stat_partial, int nslots, float* mean, float* invstd,
                                 float* scale, float* shift, float* out, int64_t out_bstride, int64_t* num_batches_tracked,
                                 void* stream);

/*
 * Backward of the above: dout [B][C or C/2][HW] dense -> draw [B][C][HW] (gradient wrt the raw conv output),
 * dgamma, dbeta [C].  The gradient wrt a residual input is dout itself.  sums_ws: no longer used (the apply pass combines
 * the partials itself); any non-NULL pointer, e.g. partial_ws.
 */
int tgsr_bn_train_bwd(const float* dout, const float* raw, int B, int C, int HW, const float* scale,
                      const float* shift, const float* mean, const float* invstd, int glu, float* partial_ws,
                      float* sums_ws, float* draw, float* dgamma, float* dbeta, void* stream);

/*
 * The discriminators' convolutions: downBlock's nn.Conv2d(Cin, Cout, 4, 2, 1, bias=False) (util.py:92-98) and the 3x3
 * stride-1 pad-1 convolutions of their 4x4-pixel, 512...2048-channel blocks, as one fp32 MFMA implicit-GEMM kernel
 * (128 x 128 x 16 LDS tiles, operands gathered from the NCHW tensors by index arithmetic - no im2col buffer; the
 * reduction is split into slabs where M x N alone cannot fill the chip, summed in a fixed order: reproducible, no
 * float atomics).  H, W = INPUT size (even for the 4x4 form); out / dy [B][Cout][H/2][W/2] (4x4) or [B][Cout][H][W]
 * (3x3), dense; w in torch layout.  BatchNorm + LeakyReLU behind it = tgsr_bn_train_fwd(glu = 2); `act` = 1 applies
 * LeakyReLU(0.2) in the epilogue (the discriminators' first layer has no BatchNorm).
 *   *_fwd   out = conv(x, w)              *_dgrad  dx = conv_transpose(dy, w)          *_wgrad  dw = sum_pixels dy x_gather
 * ws: *_ws_elems(op, ...) floats with op = 0 forward, 1 data gradient, 2 weight gradient (slabs; for the 4x4 data
 * gradient also the per-parity-class weight regrouping, rebuilt by every call: weights change per step).
 * The generator's 3x3 convolutions (32 ... 128 channels on 32^2 ... 256^2 pixels) are tgsr_wino_conv3x3_fwd /
 * tgsr_conv3x3_fwd, not these.
 * tgsr_leaky_relu: out = x > 0 ? x : 0.2 x, or with y_for_bwd != NULL the backward out = x * (y > 0 ? 1 : 0.2).
 * tgsr_dconv_set_split(on): on != 0 (the default; TGSR_DCONV_SPLIT=0 in the environment turns it off) lets the GEMMs of both
 * forms run on the bf16 matrix pipe with every fp32 operand split exactly into three bf16 pieces and six of the nine piece
 * products accumulated in fp32 (what is left out is <= 2^-26 of a product; measured error against fp64: below the fp32 MFMA's) -
 * fp32 in, fp32 out, 2.4x the fp32 MFMA rate; shapes it does not take (K % 16 != 0, ...) and on == 0 use the fp32 MFMA.
 * Returns the previous setting.  Process-wide; not meant to be flipped while launches are being issued from other threads.
 */
int tgsr_dconv_set_split(int on);
/* 1 when tgsr_conv4x4s2_{fwd (op 0), dgrad (1), wgrad (2)} takes the split form for this shape under the current setting (operand
 * alignment aside: the calls themselves also require 16-byte aligned w / dy / ws): what a caller's bookkeeping needs to price a launch. */
int tgsr_conv4x4s2_split_form(int op, int B, int Cin, int H, int W, int Cout);
int tgsr_conv3x3_gemm_split_form(int op, int B, int Cin, int H, int W, int Cout);
int64_t tgsr_conv4x4s2_ws_elems(int op, int B, int Cin, int H, int W, int Cout);
int tgsr_conv4x4s2_fwd(const float* x, int B, int Cin, int H, int W, const float* w, int Cout, int act, float* ws,
                       float* out, void* stream);
int tgsr_conv4x4s2_dgrad(const float* dy, int B, int Cin, int H, int W, const float* w, int Cout, float* ws, float* dx,
                         void* stream);
int tgsr_conv4x4s2_wgrad(const float* dy, const float* x, int B, int Cin, int H, int W, int Cout, float* ws, float* dw,
                         void* stream);
int64_t tgsr_conv3x3_gemm_ws_elems(int op, int B, int Cin, int H, int W, int Cout);
int tgsr_conv3x3_gemm_fwd(const float* x, int B, int Cin, int H, int W, const float* w, int Cout, float* ws, float* out,
                          void* stream);
int tgsr_conv3x3_gemm_dgrad(const float* dy, int B, int Cin, int H, int W, const float* w, int Cout, float* ws, float* dx,
                            void* stream);
int tgsr_conv3x3_gemm_wgrad(const float* dy, const float* x, int B, int Cin, int H, int W, int Cout, float* ws, float* dw,
                            void* stream);
int tgsr_leaky_relu(const float* x, const float* y_for_bwd, float* out, int64_t n, void* stream);

/*
 * GLU.forward (util.py:45-53) as a stand-alone op - inside the conv blocks it is the producing kernel's epilogue; the
 * module itself is callable upstream (CA_NET, util.py:381).  x viewed as [outer][2][half] (the two channel halves of
 * [B, 2C, ...]: outer = B, half = C * inner), dense fp32.  dy == NULL: out[outer][half] = x[.][0][.] * sigmoid(x[.][1][.]);
 * dy [outer][half] != NULL: the backward, out = dx [outer][2][half].
 */
int tgsr_glu(const float* x, const float* dy, float* out, int64_t outer, int64_t half, void* stream);

/* Backward of nn.Upsample(scale_factor=2, 'nearest'): out[bc][y][x] = sum of in[bc][2y..2y+1][2x..2x+1]. */
int tgsr_sumpool2x2(const float* x, int64_t BC, int H, int W, float* out, void* stream);

/*
 * uint8 image epilogue of the reference's caller (trainer_objective.py:153-155):
 * out[i] = round(clip((x[i] + 1) * 127.5, 0, 255)), float32 arithmetic and round-half-to-even exactly as the numpy
 * expression, so the bytes equal the host-side result.  x, out dense, n elements.
 */
int tgsr_to_uint8(const float* x, uint8_t* out, int64_t n, void* stream);

/*
 * The image pyramid of the reference's data loader on the GPU (datasets.py:151-197 get_imgs_blur, :236-278), byte-
 * identical to the Pillow arithmetic it delegates to.  Images are planar uint8 [N][H][W] (N = batch x 3 planes).
 *   tgsr_resize_bilinear_u8  transforms.Resize on a PIL image = PIL resize(BILINEAR): horizontal then vertical pass of
 *       a triangle filter in 22-bit fixed point.  hbounds / vbounds: int32 [out][2] = (first input index, tap count);
 *       hcoef / vcoef: int32 [out][hk | vk] taps (the host computes both in double exactly like Pillow's
 *       precompute_coeffs; a pass whose size does not change is skipped and its tables may be NULL).  tmp: N*Hin*Wout
 *       bytes when both passes run.
 *   tgsr_gaussian_blur_u8    ImageFilter.GaussianBlur(radius): `passes` extended-box-blur passes per axis with integer
 *       radius, centre weight ww and far-pixel weight fw in 24-bit fixed point (radius 2, 3 passes: 1, 4473924, 1677722);
 *       clamped borders.  tmp: N*H*W bytes.  `in` may not alias `out` or `tmp`.
 *   tgsr_u8_normalize        ToTensor + Normalize(0.5, 0.5): float32 (u8 / 255 - 0.5) / 0.5 (datasets.py:286-288).
 */
int tgsr_resize_bilinear_u8(const uint8_t* in, int N, int Hin, int Win, int Hout, int Wout, const int32_t* hbounds,
                            const int32_t* hcoef, int hk, const int32_t* vbounds, const int32_t* vcoef, int vk,
                            uint8_t* tmp, uint8_t* out, void* stream);
int tgsr_gaussian_blur_u8(const uint8_t* in, int N, int H, int W, int radius, uint32_t ww, uint32_t fw, int passes,
                          uint8_t* tmp, uint8_t* out, void* stream);
int tgsr_u8_normalize(const uint8_t* in, float* out, int64_t n, void* stream);

/*
 * Weight gradient of tgsr_conv3x3_fwd: dw[Cout][Cin][3][3] = sum over (b, y, x) of grad_out * shifted input
 * (upsample=1: the input is read through the folded nearest-x2, H/W are the PRE-upsample sizes).
 * grad_out [B][Cout][Ho][Wo] dense; x [B][Cin][H][W] with batch stride; Cout % 32 == 0.
 * ws must hold tgsr_conv3x3_wgrad_ws_elems(...) floats (per-workgroup partial slabs, summed in a fixed order).
 * The data gradient is tgsr_conv3x3_fwd on the flipped / transposed weights (+ tgsr_sumpool2x2 for upsample=1).
 */
int64_t tgsr_conv3x3_wgrad_ws_elems(int B, int Cin, int Cout, int H, int W, int upsample);
int tgsr_conv3x3_wgrad(const float* grad_out, const float* x, int64_t x_bstride, int B, int Cin, int H, int W, int Cout,
                       int upsample, float* ws, float* dw, void* stream);

/*
 * The same weight gradient for the upBlock convolution (upsample = 1) in the domain of the up-sample-aware Winograd
 * form: dU'[p] = sum over LOW-resolution pixels of dM[p] (x) V[p] for the 9 positions, then dW = G'^T dU' G' - 4x fewer
 * multiplies than the direct form.  grad_out [B][Cout][2H][2W] dense (8-byte aligned), x [B][Cin][H][W] with batch
 * stride; Cout % 64 == 0, Cin % 32 == 0.  ws: tgsr_upwino_wgrad_ws_elems floats (per-workgroup partial slabs, summed
 * in a fixed order).  dw [Cout][Cin][3][3].
 */
int64_t tgsr_upwino_wgrad_ws_elems(int B, int Cin, int Cout, int H, int W);
int tgsr_upwino_wgrad(const float* grad_out, const float* x, int64_t x_bstride, int B, int Cin, int H, int W, int Cout,
                      float* ws, float* dw, void* stream);

/*
 * Weight gradient of the plain conv3x3 (upsample = 0) in the Winograd F(2x2,3x3) domain: dU[p] = sum over 2x2 output
 * tiles of dM[p] (x) V[p] for the 16 positions, then dW = G^T dU G - 2.25x fewer multiplies than tgsr_conv3x3_wgrad.
 * grad_out [B][Cout][H][W] dense, x [B][Cin][H][W] with batch stride; Cout % 64 == 0, Cin % 32 == 0.
 * ws: tgsr_wino_wgrad_ws_elems floats (per-workgroup partial slabs, summed in a fixed order).  dw [Cout][Cin][3][3].
 */
int64_t tgsr_wino_wgrad_ws_elems(int B, int Cin, int Cout, int H, int W);
int tgsr_wino_wgrad(const float* grad_out, const float* x, int64_t x_bstride, int B, int Cin, int H, int W, int Cout,
                    float* ws, float* dw, void* stream);

/*
 * Backward of tgsr_word_attention_fwd (the attention map output carries no gradient).  P is recomputed from h and
 * src.  dc [B][idf][Q] dense -> dh [B][idf][Q] dense and dsrc_part [B][nchunks][idf][32] (nchunks =
 * tgsr_word_attention_bwd_chunks(Q)); the caller sums the chunks and maps dsrc to conv_context.weight / words:
 * dW_ctx[i][c] = sum_{b,t} dsrc[b][i][t] words[b][c][t].   idf in {32, 64}, T <= 32.
 */
int tgsr_word_attention_bwd_chunks(int Q);
int tgsr_word_attention_bwd(const float* h, int64_t h_bstride, const float* src, const uint8_t* mask, int mask_mode,
                            int B, int idf, int T, int Q, const float* dc, float* dh, float* dsrc_part, void* stream);

/*
 * Backward of tgsr_conv_to3_fwd.  g = dy * (1 - t^2), t = out - alpha*addend for act = TGSR_ACT_TANH_AXPY (out = the
 * forward output), g = dy otherwise.  dx (may be NULL) [B][Cin][H][W] dense needs w; dw (may be NULL) [3][Cin][K][K]
 * needs x and ws = tgsr_conv_to3_bwd_ws_elems(...) floats.  d(addend) = alpha * dy is left to the caller.
 */
int64_t tgsr_conv_to3_bwd_ws_elems(int B, int Cin, int H, int W, int K);
int tgsr_conv_to3_bwd(const float* dy, const float* out, const float* addend, float alpha, const float* x,
                      int64_t x_bstride, const float* w, int B, int Cin, int H, int W, int K, int act, float* dx,
                      float* ws, float* dw, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Reduced-precision inference path (BASELINE.json configs[4]: "bf16 ... MFMA bf16 attention + fused conv").
 * Storage type `dtype` = TGSR_DT_BF16 or TGSR_DT_F16 (2-byte elements), MFMA operands in that type, fp32 accumulation,
 * fp32 epilogue arithmetic, one round-to-nearest-even per stored element.  Activation layout ("lp image"):
 *     [B][H + 2][W + 2][cpitch]  channels-last with a one-pixel ZERO border that no kernel ever writes
 * (allocate the buffer zeroed once).  `cpitch` is the channel pitch of the buffer, `coff` the first channel a call
 * reads / writes: two producers writing channel ranges [0,32) and [32,64) of one cpitch-64 image replace the
 * reference's torch.cat((h_code, c_code), 1) (util.py:771, 817).  All pointers are to element [0][0][0][0] of the
 * padded image.  Shapes: W % 32 == 0, H % 4 == 0 (every layer of the x8 / x16 generators at LR >= 32).
 */
#define TGSR_DT_BF16 1
#define TGSR_DT_F16 2

/* fp32 NCHW [B][C][H][W] <-> channels [coff, coff + C) of an lp image (interior pixels only). */
int tgsr_lp_from_nchw(int dtype, const float* x, void* out, int B, int C, int H, int W, int cpitch, int coff,
                      void* stream);
int tgsr_lp_to_nchw(int dtype, const void* x, float* out, int B, int C, int H, int W, int cpitch, int coff,
                    void* stream);

/* An lp image (any shape; the whole buffer incl. its zero border, n_elems % 8 == 0, 16-byte aligned) from one 2-byte type
 * to the other: TGSR_DT_F16 -> TGSR_DT_BF16 (one round-to-nearest-even) or back (exact while |x| < 65504).  Used where a
 * section of a generator runs with f16 operands inside the bf16 configuration (NetG_highweight's 32x32 trunk: the six
 * chained ResBlocks of model.py:258-262 are where 8-bit mantissas cost the finest image 7 dB; profiles/HISTORY.md 3.8d). */
int tgsr_lp_convert(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n_elems, void* stream);

/* conv weight [Cout][Cin][3][3] (fp32, torch layout) -> MFMA fragment order
 * [kernel row 3][Cin/16][kernel column 3][Cout/32][lane 64][8], rounded to `dtype`
 * (tgsr_lp_packed_conv3x3_elems 2-byte elements).  Cout % 32 == 0, Cin % 16 == 0. */
int64_t tgsr_lp_packed_conv3x3_elems(int Cout, int Cin);
int tgsr_lp_pack_conv3x3_weight(int dtype, const float* w, void* wpack, int Cout, int Cin, void* stream);

/*
 * The fused blocks of tgsr_conv3x3_fwd on lp images (same reference sites: ResBlock.block util.py:110-130, upBlock
 * util.py:74-80, residual24/48 model.py:229-232):
 *   out = epilogue(scale * conv3x3(upsample ? nearest_x2(x) : x) + shift), epilogue = TGSR_EPI_AFFINE (+ residual when
 *   `residual` != NULL) or TGSR_EPI_AFFINE_GLU (Cout/2 output channels).
 * H, W are the OUTPUT size (x is H/2 x W/2 when upsample != 0; only with TGSR_EPI_AFFINE_GLU).  scale / shift fp32
 * [Cout] (NULL, NULL = identity).  (Cin, Cout) in {(64,128), (64,64), (32,64), (32,32)}: the generator's layers.
 */
int tgsr_lp_conv3x3_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack,
                        int Cout, const float* scale, const float* shift, const void* residual, int res_cpitch,
                        int res_coff, void* out, int out_cpitch, int out_coff, int epilogue, int upsample,
                        void* stream);

/*
 * The two ResBlocks of a generator stage (util.py:110-130 as called by INIT_STAGE_GImgup / NEXT_STAGE_G, util.py:773, :818) on
 * 64-channel lp images in ONE launch: four dependent convolutions
 *     tmp = GLU(affine0(conv(x, w0)));  a = affine1(conv(tmp, w1)) + x;  tmp = GLU(affine2(conv(a, w2)));  b = affine3(conv(tmp, w3)) + a
 * each computed per tile by the very code of tgsr_lp_conv3x3_fwd (bit-identical results); a tile's dependence on its neighbours'
 * previous layer is guarded by per-tile flags in device memory instead of a kernel boundary (what a dependent launch costs in a
 * replayed graph is more than these layers' work at 32^2 and 64^2).  wpack / scale / shift: HOST arrays of 4 device pointers
 * (packs of [128,64,3,3], [64,64,3,3], [128,64,3,3], [64,64,3,3]; scale / shift NULL together = identity).  x, tmp, a_out, b_out:
 * four distinct images [B][H+2][W+2][cpitch >= 64], channels [0, 64).  flags: tgsr_lp_resblocks_flag_elems(B, H, W) uint32,
 * zeroed ONCE by the caller and then owned by this function across launches (one buffer per set of images: concurrent launches
 * on different streams need different flag buffers); its last word is set to 1 if a wait ever timed out (results invalid).
 * W % 32 == 0, H % 4 == 0.
 */
int64_t tgsr_lp_resblocks_flag_elems(int B, int H, int W);
int tgsr_lp_resblocks_fwd(int dtype, const void* x, int x_cpitch, int B, int H, int W, const void* const* wpack,
                          const float* const* scale, const float* const* shift, void* tmp, int tmp_cpitch, void* a_out,
                          int a_cpitch, void* b_out, int b_cpitch, unsigned* flags, void* stream);

/*
 * upBlock (util.py:74-80: Upsample(x2, nearest) -> conv3x3 -> BN -> GLU) on lp images by sub-pixel decomposition: each
 * output phase (row & 1, column & 1) is a 2x2 convolution of the LOW-resolution image with taps pre-summed in fp32 and
 * rounded once to `dtype` - 16 products per low-res pixel instead of 36 (tgsr_lp_conv3x3_fwd(upsample = 1) is the
 * direct 9-tap form on the same layout).  H, W = size of the LOW-resolution input x; out [B][2H+2][2W+2][out_cpitch]
 * receives Cout/2 = 32 channels at out_coff.  Cout == 64, Cin in {32, 64}, W % 32 == 0, H % 4 == 0.
 */
int64_t tgsr_lp_packed_upconv_elems(int Cout, int Cin);
int tgsr_lp_pack_upconv_weight(int dtype, const float* w, void* wpack, int Cout, int Cin, void* stream);
int tgsr_lp_upconv_glu_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack,
                           int Cout, const float* scale, const float* shift, void* out, int out_cpitch, int out_coff,
                           void* stream);

/*
 * upBlock WITH the image head that reads its output, in one launch (the feature image of the last stage - consumed by
 * nothing but its head: GET_IMAGE_G_noAct util.py:909-919 behind G_SR_NET_low.h_net3, conv_output model.py:224 behind
 * NetG_highweight.upscale8x - then never goes to HBM).  Arguments of tgsr_lp_upconv_glu_fwd, plus:
 *   out           may be NULL: the 32-channel feature image is not written;
 *   head_wpack    tgsr_lp_pack_to3_weight of the head's [3][32][K][K] filter, head_k = K in {3, 5};
 *   head_partial  tgsr_lp_head_partial_elems(B, 2H, 2W, K) floats: a workgroup tile (8 x 64 outputs) holds only its own
 *                 pixels, so it writes the PARTIAL head sums of the (8 + 2P) x (64 + 2P) outputs they reach, layout
 *                 [B][2H/8][2W/64][3][8 + 2P][64 + 2P], P = K/2.
 * tgsr_lp_head_combine adds the <= 4 partials of every pixel in a fixed order (tile rows, then columns: reproducible, no
 * float atomics) for up to 4 scales in ONE launch and finishes both generators' heads:
 *   low[s]  = sum of partial_low[s] (3x3 heads)  [tanh when low_tanh: GET_IMAGE_G of models16, util.py:894-905]
 *   high[s] = tanh(sum of partial_high[s] (5x5 heads)) + alpha * low[s]        (model.py:280, 288, 297)
 * H[s], W[s] = image size of scale s (multiples of 8 / 64); partial_low[s] == NULL: low[s] is an input (already
 * computed); partial_high[s] == NULL: high[s] is not produced.  low / high fp32 [B][3][H][W] dense.
 */
int64_t tgsr_lp_head_partial_elems(int B, int H, int W, int K);
int tgsr_lp_upconv_glu_head_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack,
                                int Cout, const float* scale, const float* shift, void* out, int out_cpitch, int out_coff,
                                const void* head_wpack, int head_k, float* head_partial, void* stream);
int tgsr_lp_head_combine(int nscales, int B, const int* H, const int* W, const float* const* partial_low,
                         const float* const* partial_high, float* const* low, float* const* high, int low_tanh,
                         float alpha, void* stream);

/*
 * The two 3-channel stems on the fp32 LR image: conv3x3 3 -> 2C + BatchNorm(eval) affine + GLU, written as C channels
 * of an lp image (im2f util.py:741-744, convin model.py:228).  x [B][3][H][W] fp32 dense, w [2C][3][3][3] fp32 (torch
 * layout, NOT rounded: 27 MACs per output run on the VALU), scale / shift [2C].  C % 8 == 0.
 */
int tgsr_lp_stem_fwd(int dtype, const float* x, int B, int H, int W, const float* w, int C, const float* scale,
                     const float* shift, void* out, int out_cpitch, int out_coff, void* stream);

/*
 * Word attention fused into the kernel that PRODUCES h (GlobalAttention.py:87-130 called at util.py:768-771 on im2f's output
 * and at util.py:814-817 on the previous stage's upBlock output; c_code[:, q] depends only on h[:, q] and the words):
 *   tgsr_lp_stem_att_fwd        = tgsr_lp_stem_fwd (C = 32, W % 32 == 0) + the first stage's attention;
 *   tgsr_lp_upconv_glu_att_fwd  = tgsr_lp_upconv_glu_fwd (head_partial NULL) / tgsr_lp_upconv_glu_head_fwd (head_k = 3)
 *                                 of a 64 -> 64 upBlock + the NEXT stage's attention on its output.
 * Same arithmetic as tgsr_lp_word_attention_fwd on the stored h (one shared device function: bit-identical results):
 * c_code -> channels [c_coff, c_coff + 32) of the same pixels of `out` (must not overlap [out_coff, out_coff + 32)),
 * attn [B][T][H_out * W_out] fp32 or NULL.  att_pack / att_nsets: what tgsr_text_tail_lp_fwd wrote for this batch; att_set:
 * which projection this stage attends through; use_mask 0: no mask; mask_mode as tgsr_word_attention_fwd.
 * The stand-alone attention launches - and their re-read of h - leave G_SR_NET_low's dependent chain.
 */
int tgsr_lp_stem_att_fwd(int dtype, const float* x, int B, int H, int W, const float* w, int C, const float* scale,
                         const float* shift, void* out, int out_cpitch, int out_coff, const void* att_pack, int att_nsets,
                         int att_set, int use_mask, int mask_mode, int T, int c_coff, float* attn, void* stream);
int tgsr_lp_upconv_glu_att_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack,
                               int Cout, const float* scale, const float* shift, void* out, int out_cpitch, int out_coff,
                               const void* head_wpack, int head_k, float* head_partial, const void* att_pack, int att_nsets,
                               int att_set, int use_mask, int mask_mode, int T, int c_coff, float* attn, void* stream);

/*
 * The image heads on an lp image (tgsr_conv_to3_fwd's reference sites: GET_IMAGE_G_noAct util.py:913-915; conv_output
 * + `one*. + a*SRb` model.py:224, 280/288/297).  w [3][32][K][K] fp32 -> wpack (K*512 2-byte elements: one 16-row MFMA
 * fragment per kernel row whose rows are the (output channel, kernel column) pairs); x = channels [0, 32) of an lp image; addend / out fp32 [B][3][H][W] dense.
 * Cin == 32, K in {3, 5}, W % 32 == 0, H % 8 == 0.
 */
int tgsr_lp_pack_to3_weight(int dtype, const float* w, void* wpack, int Cin, int K, void* stream);
int tgsr_lp_conv_to3_fwd(int dtype, const void* x, int x_cpitch, int B, int Cin, int H, int W, const void* wpack, int K,
                         int act, const float* addend, float alpha, float* out, void* stream);

/*
 * GlobalAttentionGeneral.forward (GlobalAttention.py:87-130) on lp images: h = channels [0, idf) of an lp image,
 * src = the fp32 word projection [B][idf][32] of tgsr_word_project_fwd (rounded to `dtype` inside), mask / mask_mode as
 * in tgsr_word_attention_fwd (0 = the reference's mask.repeat row order, GlobalAttention.py:109-116; 1 = per sample).
 * Writes c_code (rounded) to channels [c_coff, c_coff + idf) of c_img - normally the same image h lives in, which is
 * the reference's torch.cat((h_code, c_code), 1) - and the fp32 softmax to attn [B][T][H*W] (NULL = not needed).
 * idf == 32, T <= 32, W % 32 == 0.
 */
int tgsr_lp_word_attention_fwd(int dtype, const void* h, int h_cpitch, const float* src, const uint8_t* mask,
                               int mask_mode, int B, int idf, int T, int H, int W, void* c_img, int c_cpitch, int c_coff,
                               float* attn, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TGSR_HIP_H */
