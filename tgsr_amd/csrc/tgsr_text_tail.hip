// The text tail of one inference step in ONE launch: everything the generator needs from the text encoder's outputs
// besides the recurrence itself.
//   * the conv_context projections of the attention stages (GlobalAttention.py:100-102)  - word_project_block
//   * CA_NET's mu / logvar (util.py:372-400; c_code is discarded by the x8 / x16 generators, model.py:51-52) - ca_net_block
//   * mask = (captions[:, :T] == 0) (trainer_objective.py:136-140) as bytes 0 / 1 (torch.bool storage)
// Before: word_project + ca_net (on a second stream, behind an event) + a comparison kernel + a bool -> uint8 cast = four
// launches and a cross-stream dependency on the critical chain of a step whose 32^2 kernels last 5 us each.
// 1-D grid: [0, nproj) projection workgroups, [nproj, nproj + nca) CA_NET workgroups, then one mask workgroup.
#include "tgsr_common.h"
#include "tgsr_text_blocks.h"

namespace tgsr {

struct TextTailArgs {
  ProjArgs p;
  CaArgs c;
  const int64_t* captions;   // [B][width]
  uint8_t* mask;             // [B][T]
  uint32_t* mbits;           // [B] or null: bit t = (captions[b][t] == 0), t < T  (the packed form the fused attention reads)
  int width, nsets, nib, nproj, nca_i, nca;
};

__global__ __launch_bounds__(256) void text_tail_kernel(TextTailArgs a) {
  __shared__ float red[kProjRedFloats];
  static_assert(kCaRedFloats <= kProjRedFloats, "the CA_NET partial tiles share the projection's LDS");
  int blk = blockIdx.x;
  if (blk < a.nproj) {
    const int ib = blk % a.nib, set = (blk / a.nib) % a.nsets, b = blk / (a.nib * a.nsets);
    word_project_block(a.p, b, set, ib, red);
    return;
  }
  blk -= a.nproj;
  if (blk < a.nca) {
    ca_net_block(a.c, blk % a.nca_i, blk / a.nca_i, red);
    return;
  }
  const int n = a.p.B * a.p.T;
  for (int o = threadIdx.x; o < n; o += 256) {
    const int b = o / a.p.T, t = o - b * a.p.T;
    a.mask[o] = a.captions[(int64_t)b * a.width + t] == 0 ? 1 : 0;
  }
  if (a.mbits)
    for (int b = threadIdx.x; b < a.p.B; b += 256) {
      uint32_t m = 0;
      for (int t = 0; t < a.p.T; ++t) m |= (a.captions[(int64_t)b * a.width + t] == 0 ? 1u : 0u) << t;
      a.mbits[b] = m;
    }
}

}  // namespace tgsr

using namespace tgsr;

static int text_tail_launch(const float* words, const float* const* w_ctx, int nsets, int B, int idf, int cdf, int T,
                            float* src_out, const float* sent_emb, const float* ca_w, const float* ca_b, int tdim,
                            int ncf, float* mu, float* logvar, const int64_t* captions, int width, uint8_t* mask,
                            int lp_dtype, void* att_pack, void* stream) {
  if (!words || !w_ctx || !src_out || !sent_emb || !ca_w || !ca_b || !mu || !logvar || !captions || !mask || nsets < 1 ||
      B < 1 || cdf < 1 || T < 1 || tdim < 1 || ncf < 1 || width < T)
    return TGSR_EINVAL;
  if (nsets > 4 || T > 32 || idf < 32 || (idf & 31)) return TGSR_EUNSUPPORTED;
  if ((tdim & 15) != 0 ||
      ((tdim & 63) == 0 && ((reinterpret_cast<uintptr_t>(sent_emb) | reinterpret_cast<uintptr_t>(ca_w)) & 15) != 0))
    return TGSR_EUNSUPPORTED;
  TextTailArgs a;
  a.p.words = words; a.p.out = src_out; a.p.B = B; a.p.idf = idf; a.p.cdf = cdf; a.p.T = T;
  for (int i = 0; i < 4; ++i) a.p.w[i] = i < nsets ? w_ctx[i] : nullptr;
  for (int i = 0; i < nsets; ++i)
    if (!a.p.w[i]) return TGSR_EINVAL;
  a.c.sent = sent_emb; a.c.w = ca_w; a.c.bias = ca_b; a.c.eps = nullptr; a.c.c_code = nullptr; a.c.mu = mu;
  a.c.logvar = logvar; a.c.B = B; a.c.tdim = tdim; a.c.ncf = ncf;
  a.captions = captions; a.mask = mask; a.width = width;
  a.mbits = nullptr;
  if (att_pack) {
    if (idf != 32 || (lp_dtype != TGSR_DT_BF16 && lp_dtype != TGSR_DT_F16) || (reinterpret_cast<uintptr_t>(att_pack) & 15))
      return TGSR_EUNSUPPORTED;
    a.p.frag = static_cast<unsigned short*>(att_pack);
    a.p.frag_dt = lp_dtype;
    a.mbits = reinterpret_cast<uint32_t*>(static_cast<char*>(att_pack) + (size_t)nsets * B * 4096);
  }
  a.nsets = nsets; a.nib = idf / 32; a.nproj = B * nsets * a.nib;
  a.nca_i = (ncf + 3) / 4; a.nca = a.nca_i * ((B + 15) / 16);
  hipLaunchKernelGGL(text_tail_kernel, dim3(a.nproj + a.nca + 1), dim3(256), 0, as_stream(stream), a);
  return note_launch(hipGetLastError(), "text_tail_kernel");
}

extern "C" int tgsr_text_tail_fwd(const float* words, const float* const* w_ctx, int nsets, int B, int idf, int cdf, int T,
                                  float* src_out, const float* sent_emb, const float* ca_w, const float* ca_b, int tdim,
                                  int ncf, float* mu, float* logvar, const int64_t* captions, int width, uint8_t* mask,
                                  void* stream) {
  return text_tail_launch(words, w_ctx, nsets, B, idf, cdf, T, src_out, sent_emb, ca_w, ca_b, tdim, ncf, mu, logvar, captions,
                          width, mask, 0, nullptr, stream);
}

extern "C" int64_t tgsr_lp_att_pack_bytes(int nsets, int B) { return (int64_t)nsets * B * 4096 + 4 * (int64_t)B; }

extern "C" int tgsr_text_tail_lp_fwd(const float* words, const float* const* w_ctx, int nsets, int B, int idf, int cdf, int T,
                                     float* src_out, const float* sent_emb, const float* ca_w, const float* ca_b, int tdim,
                                     int ncf, float* mu, float* logvar, const int64_t* captions, int width, uint8_t* mask,
                                     int lp_dtype, void* att_pack, void* stream) {
  if (!att_pack) return TGSR_EINVAL;
  return text_tail_launch(words, w_ctx, nsets, B, idf, cdf, T, src_out, sent_emb, ca_w, ca_b, tdim, ncf, mu, logvar, captions,
                          width, mask, lp_dtype, att_pack, stream);
}
