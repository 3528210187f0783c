"""Drop-in for the reference's GlobalAttention.py: `GlobalAttentionGeneral` and `func_attention`, on HIP kernels.

Same class name, constructor, `applyMask` / `forward` protocol, return tuple and state_dict key
(`conv_context.weight [idf, cdf, 1, 1]`) as GlobalAttention.py:77-130.
"""
import torch
import torch.nn as nn

from . import ops


def conv1x1(in_planes, out_planes):
    "1x1 convolution, no bias (parameter holder; the arithmetic runs in tgsr_word_attention_fwd)"
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=False)


class GlobalAttentionGeneral(nn.Module):
    """GlobalAttention.py:77-130.  `correct_mask=True` is an opt-in that masks sample b with mask[b] instead of
    the reference's `mask.repeat(queryL, 1)` row order (GlobalAttention.py:109-116)."""

    def __init__(self, idf, cdf, correct_mask=False):
        super(GlobalAttentionGeneral, self).__init__()
        self.conv_context = conv1x1(cdf, idf)
        self.mask = None
        self.correct_mask = correct_mask

    def applyMask(self, mask):
        self.mask = mask  # batch x sourceL

    def forward(self, input, context, out=None, src=None):
        """input [B, idf, ih, iw], context [B, cdf, sourceL] -> (weightedContext [B, idf, ih, iw],
        attn [B, sourceL, ih, iw]).  `src`: this layer's word projection when the caller already has it
        (G_SR_NET_low projects the words for its three stages in one launch, ops.word_project); an argument, not
        module state, so the module stays re-entrant across streams."""
        if self.training:
            from .autograd import WordAttention
            if out is not None:
                raise RuntimeError("training path does not write into channel-slice views")
            return WordAttention.apply(input, context, self.conv_context.weight, self.mask, self.correct_mask)
        from . import custom_ops as C
        if out is None:
            return C.word_attention(input, context, self.conv_context.weight, self.mask, self.correct_mask, src)
        return out, C.word_attention_out(input, context, self.conv_context.weight, self.mask, self.correct_mask, src, out)


def func_attention(query, context, gamma1):
    """GlobalAttention.py:33-74 - served by the batched DAMSM kernel (miscc.losses.words_loss)."""
    from .miscc import losses
    return losses._func_attention(query, context, gamma1)
