"""Drop-in for the hot-path symbols of the reference's util.py, on the gfx950 kernels of libtgsr_hip.so.

Same class / function names, constructor arguments, forward signatures, return tuples and **state_dict keys**
as the reference (util.py:45-130, 175-260, 372-400, 726-823, 894-919), so `Checkpoint/face_S8/*.pth` load
unchanged.  The nn.Conv2d / nn.BatchNorm2d / nn.LSTM children are parameter holders that keep the key names;
they are never called - every forward goes through `tgsr_amd.ops` (fused conv + BN + GLU / residual, up-sample
folded into the conv, fused word attention, BiLSTM kernels).  `.eval()` modules take the fused inference kernels
(BatchNorm folded to an affine, no autograd graph); `.train()` modules take the training path of
`tgsr_amd.autograd` (batch-statistics BatchNorm, running-stat updates, HIP backward kernels).
"""
import os

import torch
import torch.nn as nn

from . import custom_ops as C            # registers torch.ops.tgsr.* (the PyTorch-ROCm custom operators over the C ABI)
from . import ops
from .GlobalAttention import GlobalAttentionGeneral as ATT_NET
from .miscc.config import cfg


# ------------------------------------------------------------------------------------------ fused-parameter cache
def _ver(*ts):
    return tuple((t.data_ptr(), t._version, t.device) for t in ts if t is not None)


# Inference convolutions without upsampling go through the Winograd F(2x2,3x3) kernel where it applies
# (ops.wino_supported: Cout % 64 == 0, Cin % 4 == 0, W % 4 == 0), the large layers (>= 64 x 64 pixels and a full round of
# workgroups: ops.wino4_wanted) through F(4x4,3x3) (TGSR_WINO4=0: F(2x2) there too); TGSR_WINOGRAD=0 keeps the direct kernel.
WINOGRAD = os.environ.get("TGSR_WINOGRAD", "1") != "0"


class _FusedParams:
    """Packed conv weight + folded BN affine for one conv(+bn) pair, rebuilt when any source tensor changes
    (load_state_dict / optimizer step / .cuda()).  The direct and the Winograd packs are built on first use."""

    def __init__(self):
        self.key = None
        self.wpack = self.upack = self.u4pack = self.scale = self.shift = None

    def _refresh(self, conv: nn.Conv2d, bn):
        src = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
        key = _ver(*src)
        if key != self.key:
            self.wpack = self.upack = self.u4pack = None
            if bn is not None:
                self.scale, self.shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
            else:
                self.scale = self.shift = None
            self.key = key

    def get(self, conv: nn.Conv2d, bn):
        self._refresh(conv, bn)
        if self.wpack is None:
            self.wpack = ops.pack_conv3x3_weight(conv.weight)
        return self.wpack, self.scale, self.shift

    def get_wino(self, conv: nn.Conv2d, bn, glu: bool):
        self._refresh(conv, bn)
        if self.upack is None:
            self.upack = ops.pack_wino_weight(conv.weight, glu=glu)
        return self.upack, self.scale, self.shift


    def get_wino4(self, conv: nn.Conv2d, bn, glu: bool):
        """(pack, scale, shift, wide): the register-fed form of the F(4x4) kernel (tgsr_wino4_wide_conv3x3_fwd: 128-row groups
        where Cout % 128 == 0, else 64-row groups in 4-wave workgroups) where the layer has an even number of 4-channel stages."""
        self._refresh(conv, bn)
        wide = conv.in_channels % 8 == 0
        if self.u4pack is None:
            self.u4pack = (C.pack_wino4w_weight(conv.weight.detach(), glu, False) if wide else
                           C.pack_wino4_weight(conv.weight.detach(), glu, False))
        return self.u4pack, self.scale, self.shift, wide


def _wino4_takes(x, cout, out, residual):
    """The F(4x4, 3x3) kernel: the layers ops.wino4_wanted names (>= 64 x 64 pixels, whole 8 x 64 tiles, 64-channel groups, a
    full round of workgroups) when every tensor is 16-byte aligned with batch strides % 4 == 0."""
    if x.dim() != 4 or not ops.wino4_wanted(x.shape[1], cout, x.shape[2], x.shape[3], x.shape[0]):
        return False
    for t in (x, out, residual):
        if t is not None and (t.data_ptr() % 16 != 0 or (t.shape[0] > 1 and t.stride(0) % 4 != 0) or t.stride(3) != 1 or
                              t.stride(2) != t.shape[3] or t.stride(1) != t.shape[2] * t.shape[3]):
            return False
    return True


def _wino_pays(x, cout, out, residual):
    """Winograd where the kernel takes the shape and is the faster one: always for 64-channel groups; 32-channel
    groups (Cout % 64 != 0: 8-row workgroup tiles) only when the image gives >= 256 workgroups (measured at B=16:
    32->32 @128^2 40 vs 71 us, @64^2 16 vs 20 us, @32^2 16 vs 13 us)."""
    if not ops.wino_supported(x, cout, out=out, residual=residual):
        return False
    if cout % 64 == 0:
        return True
    B, _, H, W = x.shape
    return B * ((W + 31) // 32) * ((H + 7) // 8) >= 256


def _conv_bn(x, fp: _FusedParams, conv, bn, glu=False, upsample=False, residual=None, out=None, training=False):
    """One fused block.  eval: conv + folded-BN affine + GLU/residual in one launch (optionally into `out`, a
    channel-slice view).  training: batch-statistics BN through tgsr_amd.autograd (differentiable)."""
    if training:
        from .autograd import conv_bn_act_train
        y = conv_bn_act_train(x, conv, bn, glu=glu, upsample=upsample, residual=residual)
        if out is not None:
            raise RuntimeError("training path does not write into channel-slice views")
        return y
    if WINOGRAD and not upsample and _wino4_takes(x, conv.out_channels, out, residual):
        upack, scale, shift, wide = fp.get_wino4(conv, bn, glu)
        if out is None:
            return (C.conv3x3_wino4w if wide else C.conv3x3_wino4)(x, upack, conv.out_channels, scale, shift, glu, residual)
        (C.conv3x3_wino4w_out if wide else C.conv3x3_wino4_out)(x, upack, conv.out_channels, scale, shift, glu, residual, out)
        return out
    if WINOGRAD and not upsample and _wino_pays(x, conv.out_channels, out, residual):
        upack, scale, shift = fp.get_wino(conv, bn, glu)
        if out is None:
            return C.conv3x3_wino(x, upack, conv.out_channels, scale, shift, glu, residual)
        C.conv3x3_wino_out(x, upack, conv.out_channels, scale, shift, glu, residual, out)
        return out
    wpack, scale, shift = fp.get(conv, bn)
    if out is None:
        return C.conv3x3_fused(x, wpack, conv.out_channels, scale, shift, glu, upsample, residual)
    C.conv3x3_fused_out(x, wpack, conv.out_channels, scale, shift, glu, upsample, residual, out)
    return out


def invalidate_caches(module: nn.Module):
    """Drop every packed-weight / folded-BatchNorm / gate-table cache under `module`, so the next eval forward rebuilds
    them.  The caches key on (data_ptr, version counter) of their source tensors, which every torch in-place op bumps;
    this is the explicit hook for writers that bypass the counters (raw-pointer kernels, `.data` tricks)."""
    for m in module.modules():
        for name in ("_fp", "_fp0", "_fp1"):
            fp = getattr(m, name, None)
            if isinstance(fp, _FusedParams):
                fp.key = None
        if hasattr(m, "_up_key"):
            m._up_key = None
        if hasattr(m, "_table_key"):
            m._key = m._table_key = None
        if hasattr(m, "_a_host"):
            m._a_host = (None, 0.5)


# ------------------------------------------------------------------------------------------ blocks
class GLU(nn.Module):
    """util.py:45-53: x[:, :C/2] * sigmoid(x[:, C/2:]).  Inside the fused Sequentials below it is a marker (its arithmetic
    is the producing convolution's epilogue); called on its own - as CA_NET does upstream, util.py:381 - it is one HIP
    launch (torch.ops.tgsr.glu, differentiable)."""

    def __init__(self):
        super(GLU, self).__init__()

    def forward(self, x):
        nc = x.size(1)
        assert nc % 2 == 0, 'channels dont divide 2!'
        return C.glu(x)


def conv1x1(in_planes, out_planes, bias=False):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=bias)


def conv3x3(in_planes, out_planes):
    "3x3 convolution with padding, no bias (util.py:62-65) - parameter holder"
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=False)


def conv5x5(in_planes, out_planes):
    "util.py:68-69 - parameter holder"
    return nn.Conv2d(in_planes, out_planes, kernel_size=5, stride=1, padding=2, bias=False)


class _UpBlock(nn.Sequential):
    """upBlock (util.py:74-80): [Upsample(x2 nearest), conv3x3, BatchNorm2d, GLU] as ONE launch
    (keys `1.weight`, `2.*`)."""

    def __init__(self, in_planes, out_planes):
        super().__init__(nn.Upsample(scale_factor=2, mode='nearest'), conv3x3(in_planes, out_planes * 2),
                         nn.BatchNorm2d(out_planes * 2), GLU())
        self._fp = _FusedParams()
        self._up_key = self._up_pack = self._up_aff = None

    def forward(self, x, out=None):
        conv, bn = self[1], self[2]
        if not self.training and conv.out_channels % 64 == 0:
            # inference: Winograd on the up-sampled grid with the up-sampling folded into the input transform (9 of the
            # 16 positions survive: 2.25 multiplies per output); shapes it does not take use the sub-pixel form (four
            # 2x2 convs on the pre-upsample tensor, 4 multiplies per output)
            wino = WINOGRAD and ops.upwino_supported(x, conv.out_channels, out=out)
            # ... and the F(4x4) form of it (25 of 36 positions: 1.56 multiplies per output) where ops.upwino4_wanted routes
            # the layer (output >= 256 x 256 pixels: the generator's last upBlock, whole tiles, a full round of workgroups) and the
            # tensors are 16-byte aligned
            w4 = (wino and ops.upwino4_wanted(x.shape[1], conv.out_channels, x.shape[2], x.shape[3], x.shape[0]) and
                  x.data_ptr() % 16 == 0 and (out is None or (out.data_ptr() % 16 == 0 and (out.shape[0] == 1 or out.stride(0) % 4 == 0))))
            src = [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
            key = (_ver(*src), wino, w4)
            if key != self._up_key:
                self._up_pack = (C.pack_upwino4_weight(conv.weight.detach(), True) if w4 else
                                 (ops.pack_upwino_weight if wino else ops.pack_upconv_weight)(conv.weight))
                self._up_aff = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
                self._up_key = key
            fn, fn_out = ((C.upwino4_glu, C.upwino4_glu_out) if w4 else
                          ((C.upwino_glu, C.upwino_glu_out) if wino else (C.upconv3x3_glu, C.upconv3x3_glu_out)))
            if out is None:
                return fn(x, self._up_pack, conv.out_channels, self._up_aff[0], self._up_aff[1])
            fn_out(x, self._up_pack, conv.out_channels, self._up_aff[0], self._up_aff[1], out)
            return out
        return _conv_bn(x, self._fp, conv, bn, glu=True, upsample=True, out=out, training=self.training)


def upBlock(in_planes, out_planes):
    return _UpBlock(in_planes, out_planes)


class _ConvBnGlu(nn.Sequential):
    """conv3x3 -> BatchNorm2d -> GLU (Block3x3_relu util.py:101-106, im2f util.py:741-744, convin model.py:228);
    keys `0.weight`, `1.*`."""

    def __init__(self, in_planes, out_planes):
        super().__init__(conv3x3(in_planes, out_planes * 2), nn.BatchNorm2d(out_planes * 2), GLU())
        self._fp = _FusedParams()

    def forward(self, x, out=None):
        return _conv_bn(x, self._fp, self[0], self[1], glu=True, out=out, training=self.training)


def Block3x3_relu(in_planes, out_planes):
    return _ConvBnGlu(in_planes, out_planes)


# ------------------------------------------------------------------------------------------ discriminator blocks
class _ConvBnLeaky(nn.Sequential):
    """[conv, BatchNorm2d, LeakyReLU(0.2)] as one fused training block (keys `0.weight`, `1.*`).  `kind` = "down":
    downBlock, nn.Conv2d(in, out, 4, 2, 1, bias=False) (util.py:92-98); "3x3": conv3x3 (the discriminators'
    Block3x3_leakRelu).  Training mode (what losses.py:290-374 run): batch-statistics BatchNorm, autograd.ConvBnLeaky.  Under
    .eval(): the running statistics, autograd.ConvBnLeakyEval (`groups` then only partitions the batch - every slice is
    normalised by the same running statistics)."""

    def __init__(self, conv, out_planes, kind):
        super().__init__(conv, nn.BatchNorm2d(out_planes), nn.LeakyReLU(0.2, inplace=True))
        self._kind = kind

    def forward(self, x, groups=None):
        """groups: sizes of consecutive batch slices normalised with their own batch statistics (autograd.ConvBnLeaky)."""
        if not self.training:
            from .autograd import ConvBnLeakyEval
            bn = self[1]
            if not bn.track_running_stats or bn.running_mean is None:
                raise NotImplementedError("eval-mode BatchNorm2d without running statistics")
            return ConvBnLeakyEval.apply(x, self[0].weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, self._kind, bn.eps)
        from .autograd import conv_bn_leaky_train
        return conv_bn_leaky_train(x, self[0], self[1], self._kind, groups)


def downBlock(in_planes, out_planes):
    "util.py:92-98: Conv2d(4, stride 2, pad 1, no bias) -> BatchNorm2d -> LeakyReLU(0.2); halves the spatial size"
    return _ConvBnLeaky(nn.Conv2d(in_planes, out_planes, 4, 2, 1, bias=False), out_planes, "down")


def Block3x3_leakRelu(in_planes, out_planes):
    "conv3x3 -> BatchNorm2d -> LeakyReLU(0.2); keeps the spatial size"
    return _ConvBnLeaky(conv3x3(in_planes, out_planes), out_planes, "3x3")


class _EncodeBy16(nn.Module):
    """Image -> ndf*8 channels at 1/16 of the resolution: Conv2d(3, ndf, 4, 2, 1) + LeakyReLU, then three downBlocks
    (the AttnGAN-style `encode_image_by_16times`; build-declared, see model.D_NET64)."""

    def __init__(self, ndf):
        super().__init__()
        self.conv0 = nn.Conv2d(3, ndf, 4, 2, 1, bias=False)
        self.down1 = downBlock(ndf, ndf * 2)
        self.down2 = downBlock(ndf * 2, ndf * 4)
        self.down3 = downBlock(ndf * 4, ndf * 8)

    def forward(self, x, groups=None):
        x = C.conv4x4s2(x, self.conv0.weight, True)
        return self.down3(self.down2(self.down1(x, groups), groups), groups)


def encode_image_by_16times(ndf):
    return _EncodeBy16(ndf)


class D_GET_LOGITS(nn.Module):
    """The conditional / unconditional logit heads `netD.COND_DNET(features, sent_emb)` / `netD.UNCOND_DNET(features)`
    that losses.py:292-316, 359-366 call.  features [B, 8 ndf, 4, 4]; conditional: the sentence code is tiled over the
    4x4 grid, concatenated, passed through conv3x3 -> BN -> LeakyReLU (jointConv); then a 4x4 / stride-4 convolution
    to one LOGIT per sample (no sigmoid: the reference's losses are BCEWithLogits)."""

    def __init__(self, ndf, nef, bcondition=False):
        super().__init__()
        self.df_dim, self.ef_dim, self.bcondition = ndf, nef, bcondition
        if bcondition:
            self.jointConv = Block3x3_leakRelu(ndf * 8 + nef, ndf * 8)
        self.outlogits = nn.Sequential(nn.Conv2d(ndf * 8, 1, kernel_size=4, stride=4))

    def forward(self, h_code, c_code=None, groups=None):
        from .autograd import RowDot
        if self.bcondition and c_code is not None:
            c = c_code.view(-1, self.ef_dim, 1, 1).repeat(1, 1, 4, 4)
            h_code = self.jointConv(torch.cat((h_code, c), 1), groups)
        conv = self.outlogits[0]                                  # a 4x4 / stride 4 conv on a 4x4 map = one dot product
        return RowDot.apply(h_code.reshape(h_code.size(0), -1), conv.weight.reshape(-1), conv.bias)


class _ResidualNoSum(nn.Sequential):
    """model.py:229-232 `residual24/48`: conv-BN-GLU-conv-BN, NO skip add (keys 0,1,3,4)."""

    def __init__(self, ngf):
        super().__init__(conv3x3(ngf, ngf * 2), nn.BatchNorm2d(ngf * 2), GLU(), conv3x3(ngf, ngf),
                         nn.BatchNorm2d(ngf))
        self._fp0, self._fp1 = _FusedParams(), _FusedParams()

    def forward(self, x):
        y = _conv_bn(x, self._fp0, self[0], self[1], glu=True, training=self.training)
        return _conv_bn(y, self._fp1, self[3], self[4], training=self.training)


class ResBlock(nn.Module):
    """util.py:110-130: x + BN(conv(GLU(BN(conv(x))))) in two launches (keys `block.{0,1,3,4}.*`)."""

    def __init__(self, channel_num, batchnorm=True):
        super(ResBlock, self).__init__()
        if not batchnorm:
            raise NotImplementedError("ResBlock(batchnorm=False) is only used by G_SR_NET_low_stage1 (dead: "
                                      "stage1=False, trainer_objective.py:56)")
        self.block = nn.Sequential(conv3x3(channel_num, channel_num * 2), nn.BatchNorm2d(channel_num * 2), GLU(),
                                   conv3x3(channel_num, channel_num), nn.BatchNorm2d(channel_num))
        self._fp0, self._fp1 = _FusedParams(), _FusedParams()

    def forward(self, x):
        if self.training:       # one autograd node: the skip gradient is added in the first conv's data-gradient epilogue
            from .autograd import res_block_train
            return res_block_train(x, self.block[0], self.block[1], self.block[3], self.block[4])
        y = _conv_bn(x, self._fp0, self.block[0], self.block[1], glu=True, training=self.training)
        return _conv_bn(y, self._fp1, self.block[3], self.block[4], residual=x, training=self.training)


# ------------------------------------------------------------------------------------------ text encoder
class RNN_ENCODER(nn.Module):
    """util.py:175-260.  Embedding + bidirectional 1-layer LSTM; forward = tgsr_bilstm_fwd (eval mode: the
    reference's Dropout(0.5) is the identity there).  state_dict keys: `encoder.weight`, `rnn.*_l0[_reverse]`."""

    def __init__(self, ntoken, ninput=300, drop_prob=0.5, nhidden=128, nlayers=1, bidirectional=True):
        super(RNN_ENCODER, self).__init__()
        self.n_steps = cfg.TEXT.WORDS_NUM
        self.ntoken = ntoken
        self.ninput = ninput
        self.drop_prob = drop_prob
        self.nlayers = nlayers
        self.bidirectional = bidirectional
        self.rnn_type = cfg.RNN_TYPE
        if self.rnn_type not in ('LSTM', 'GRU') or nlayers != 1 or not bidirectional:
            raise NotImplementedError("HIP text encoder: 1-layer bidirectional LSTM / GRU (util.py:199-211; every shipped "
                                      "configuration is a 1-layer bidirectional LSTM)")
        self.num_directions = 2
        self.nhidden = nhidden // self.num_directions
        self.encoder = nn.Embedding(self.ntoken, self.ninput)
        self.drop = nn.Dropout(self.drop_prob)
        rnn = nn.LSTM if self.rnn_type == 'LSTM' else nn.GRU             # util.py:199-211 (same parameter names: state_dict keys)
        self.rnn = rnn(self.ninput, self.nhidden, self.nlayers, batch_first=True, bidirectional=True)
        self.init_weights()
        self._key = None
        self._stacked = None
        self._table_key = None
        self._table = None

    def init_weights(self):
        self.encoder.weight.data.uniform_(-0.1, 0.1)  # util.py:214-216

    def init_hidden(self, bsz):
        w = next(self.parameters()).data
        z = w.new_zeros(self.nlayers * self.num_directions, bsz, self.nhidden)
        return (z, z.clone()) if self.rnn_type == 'LSTM' else z          # util.py:222-231

    def _weights(self):
        r = self.rnn
        src = [r.weight_ih_l0, r.weight_ih_l0_reverse, r.weight_hh_l0, r.weight_hh_l0_reverse, r.bias_ih_l0,
               r.bias_ih_l0_reverse, r.bias_hh_l0, r.bias_hh_l0_reverse]
        key = _ver(*src)
        if key != self._key:
            d = [t.detach() for t in src]
            self._stacked = (torch.stack(d[0:2]).contiguous(), torch.stack(d[2:4]).contiguous(),
                             torch.stack(d[4:6]).contiguous(), torch.stack(d[6:8]).contiguous())
            self._key = key
        return self._stacked

    def forward(self, captions, cap_lens, hidden=None, mask=None):
        """captions int64 [B, n_steps] sorted by length (desc), cap_lens [B] -> (words_emb [B, 2H, T_max],
        sent_emb [B, 2H]).  `hidden` must be the zero state of init_hidden (the only use in the reference)."""
        if self.rnn_type == 'GRU':
            return self._forward_gru(captions, cap_lens)
        if self.training:
            # util.py:236-252: emb = drop(encoder(captions)) (torch embedding + dropout: their gradients are
            # autograd's), then the LSTM through its HIP forward / BPTT kernels (autograd.BiLSTM)
            from .autograd import BiLSTM
            lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
            r = self.rnn
            emb = self.drop(self.encoder(captions[:, :max(lens)]))
            return BiLSTM.apply(emb, torch.stack([r.weight_ih_l0, r.weight_ih_l0_reverse]),
                                torch.stack([r.weight_hh_l0, r.weight_hh_l0_reverse]),
                                torch.stack([r.bias_ih_l0, r.bias_ih_l0_reverse]),
                                torch.stack([r.bias_hh_l0, r.bias_hh_l0_reverse]), lens)
        w_ih, w_hh, b_ih, b_hh = self._weights()
        if not self.training:
            # frozen weights: the input projection is a function of the token only -> per-token gate table, built once
            # per weight version with the same GEMM kernel (bit-identical), then ONE recurrence launch per batch
            key = (self._key, _ver(self.encoder.weight))
            if key != self._table_key:
                self._table = C.lstm_gate_table(self.encoder.weight.detach(), w_ih, b_ih, b_hh)
                self._table_key = key
            if torch.is_tensor(cap_lens) and cap_lens.is_cuda:
                # lengths stay on the device (a captured step replayed on new batches): words_emb comes back at the full
                # caption width, zero behind each caption; the caller crops to the batch's longest caption (util.py:250-253)
                lens_d = cap_lens if cap_lens.dtype == torch.int32 else cap_lens.to(torch.int32)
                return C.bilstm_table_static(captions, lens_d.contiguous(), self._table, w_hh)
            lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
            return C.bilstm_table(captions, lens, self._table, w_hh)
        return ops.bilstm(captions, cap_lens, self.encoder.weight, w_ih, w_hh, b_ih, b_hh)


def _rnn_encoder_forward_gru(self, captions, cap_lens):
    """The GRU branch (util.py:207-211, 244-258), eval mode: a per-token gate table built once per weight version, then ONE
    recurrence launch per batch (tgsr_bigru_table_fwd); training mode: autograd.BiGRU (tgsr_bigru_train_fwd / tgsr_bigru_bwd).  No
    shipped configuration selects GRU."""
    if self.training:
        # util.py:236-252 with the GRU cell: emb = drop(encoder(captions)) through torch (their gradients are autograd's), the
        # recurrence and its BPTT on the HIP kernels (autograd.BiGRU)
        from .autograd import BiGRU
        lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
        r = self.rnn
        emb = self.drop(self.encoder(captions[:, :max(lens)]))
        return BiGRU.apply(emb, torch.stack([r.weight_ih_l0, r.weight_ih_l0_reverse]),
                           torch.stack([r.weight_hh_l0, r.weight_hh_l0_reverse]),
                           torch.stack([r.bias_ih_l0, r.bias_ih_l0_reverse]),
                           torch.stack([r.bias_hh_l0, r.bias_hh_l0_reverse]), lens)
    w_ih, w_hh, b_ih, b_hh = self._weights()
    key = (self._key, _ver(self.encoder.weight))
    if key != self._table_key:
        self._table = C.gru_gate_table(self.encoder.weight.detach(), w_ih, b_ih, b_hh)      # (table, b_hn)
        self._table_key = key
    table, b_hn = self._table
    if torch.is_tensor(cap_lens) and cap_lens.is_cuda:
        lens_d = cap_lens if cap_lens.dtype == torch.int32 else cap_lens.to(torch.int32)
        return C.bigru_table_static(captions, lens_d.contiguous(), table, w_hh, b_hn)
    lens = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
    return C.bigru_table(captions, lens, table, w_hh, b_hn)


RNN_ENCODER._forward_gru = _rnn_encoder_forward_gru


_INCEPTION_BLOCKS = ("Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3", "Conv2d_3b_1x1", "Conv2d_4a_3x3", "Mixed_5b", "Mixed_5c",
                     "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e", "Mixed_7a", "Mixed_7b", "Mixed_7c")


class CNN_ENCODER(nn.Module):
    """util.py:263-368.  The frozen Inception-v3 trunk is third-party torchvision arithmetic with downloaded weights
    (util.py:271-275): its blocks are NOT re-implemented here.  Three ways to supply it:
      * nothing: `torchvision.models.inception_v3()` is built like the reference does (raises where torchvision is absent);
      * `inception=` any object exposing the sixteen block attributes the reference copies (util.py:282-298:
        Conv2d_1a_3x3 ... Mixed_7c) - they are registered on this module under the reference's names, so the state_dict
        keys (`Conv2d_1a_3x3.*`, ..., `emb_features.weight`, `emb_cnn_code.*`) equal the reference's image_encoder files,
        and `forward` walks them exactly as util.py:308-362 does (bilinear resize to 299, two max-pools, features after
        Mixed_6e, 8x8 average pool after Mixed_7c);
      * `trunk=` a module images [B,3,H,W] -> (features [B,768,17,17], pooled [B,2048]) replacing the whole walk.
    The two trainable heads (`emb_features`, `emb_cnn_code`; same uniform(-0.1, 0.1) init, util.py:303-306) run on the HIP
    GEMM kernels."""

    def __init__(self, nef, trunk=None, inception=None):
        super(CNN_ENCODER, self).__init__()
        self.nef = nef if cfg.TRAIN.FLAG else 256          # util.py:266-269
        self.trunk = trunk
        if trunk is None:
            model = inception if inception is not None else _torchvision_inception()
            for param in model.parameters():               # util.py:274-275
                param.requires_grad = False
            self.define_module(model)
        else:
            self._define_heads()
        self.init_trainable_weights()

    def define_module(self, model):
        """util.py:281-301."""
        for name in _INCEPTION_BLOCKS:
            setattr(self, name, getattr(model, name))
        self._define_heads()

    def _define_heads(self):
        self.emb_features = conv1x1(768, self.nef)
        self.emb_cnn_code = nn.Linear(2048, self.nef)

    def init_trainable_weights(self):
        self.emb_features.weight.data.uniform_(-0.1, 0.1)
        self.emb_cnn_code.weight.data.uniform_(-0.1, 0.1)

    def frozen_parameters(self):
        """Everything but the two heads (the reference freezes the Inception parameters, util.py:274-275)."""
        heads = {id(p) for m in (self.emb_features, self.emb_cnn_code) for p in m.parameters()}
        return [p for p in self.parameters() if id(p) not in heads]

    def run_trunk(self, x):
        """images -> (features [B,768,17,17], pooled [B,2048]): util.py:308-362 up to (not including) the two heads.
        Third-party blocks run as the torch modules they are; only the order / pooling / resize is the reference's."""
        if self.trunk is not None:
            return self.trunk(x)
        if self._hip_trunk_ok(x):
            # eval mode, frozen (what generator_loss runs, losses.py:375-389): the whole walk on the library's kernels
            # (tgsr_amd/inception.py, csrc/tgsr_igemm.hip); its backward reaches the image only
            from .inception import InceptionTrunk, TrunkFn
            if self._hip_trunk is None:
                self._hip_trunk = InceptionTrunk(self)
            return TrunkFn.apply(x, self._hip_trunk)
        import torch.nn.functional as F
        x = F.interpolate(x, size=(299, 299), mode='bilinear', align_corners=False)   # nn.Upsample(size=(299, 299), 'bilinear')
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x)))              # 149 -> 147 -> 147
        x = F.max_pool2d(x, kernel_size=3, stride=2)                                    # 73
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x))                                   # 73 -> 71
        x = F.max_pool2d(x, kernel_size=3, stride=2)                                    # 35
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(x)))))   # 17 x 17 x 768
        features = x                                                                    # image region features
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))                              # 8 x 8 x 2048
        x = F.avg_pool2d(x, kernel_size=8)
        return features, x.view(x.size(0), -1)

    _hip_trunk = None

    def _hip_trunk_ok(self, x):
        """The HIP walk serves the frozen trunk in eval mode on fp32 HIP images (TGSR_TRUNK=torch: the torch modules, e.g. on
        MIOpen).  In training mode (pretrain_DAMSM.py:49-50 puts the whole encoder in train mode: the trunk's BatchNorm then
        normalises with batch statistics) the blocks run as the torch modules they are."""
        import os
        if os.environ.get("TGSR_TRUNK", "hip") == "torch" or self.training or not (x.is_cuda and x.dtype == torch.float32):
            return False
        if self._hip_trunk is None:
            first = getattr(self, _INCEPTION_BLOCKS[0], None)
            if not (hasattr(first, "conv") and hasattr(first, "bn")):
                return False                                  # not torchvision's layout (e.g. a stub): walk the modules
            if any(p.requires_grad for p in self.frozen_parameters()):
                return False
        return True

    def heads(self, features, pooled):
        """(features [B,768,17,17], pooled [B,2048]) -> (region features [B,nef,17,17], cnn_code [B,nef])."""
        # torch.ops.tgsr.conv1x1 / linear: HIP GEMM kernels, differentiable (their backward GEMMs are the same kernel)
        return (C.conv1x1(features, self.emb_features.weight),
                C.linear(pooled, self.emb_cnn_code.weight, self.emb_cnn_code.bias))

    def forward(self, x):
        features, pooled = self.run_trunk(x)
        return self.heads(features, pooled)


def _torchvision_inception():
    """util.py:271-273: `models.inception_v3()` (the reference then loads the downloaded weights; there is no network
    here, so the caller loads a state_dict)."""
    try:
        from torchvision import models
    except ImportError as e:
        raise ImportError("CNN_ENCODER needs an Inception-v3: torchvision is not installed here; pass `inception=` (an "
                          "object with the blocks Conv2d_1a_3x3 ... Mixed_7c) or `trunk=` (images -> (features "
                          "[B,768,17,17], pooled [B,2048]))") from e
    return models.inception_v3(weights=None, aux_logits=True, init_weights=False)


class CA_NET(nn.Module):
    """util.py:372-400.  One 256->400 Linear + GLU + re-parametrisation on [B,256].  Inference (eval mode under
    no_grad): one HIP launch (tgsr_ca_net_fwd).  Training: tgsr::linear (the library's MFMA GEMM, differentiable) + torch
    pointwise ops - the KL term differentiates through them.  `c_code` is sampled to keep the reference's RNG consumption (util.py:388-396)
    and discarded by the x8 caller."""

    def __init__(self):
        super(CA_NET, self).__init__()
        self.t_dim = cfg.TEXT.EMBEDDING_DIM
        self.c_dim = cfg.GAN.CONDITION_DIM
        self.fc = nn.Linear(self.t_dim, self.c_dim * 4, bias=True)
        self.relu = GLU()

    def encode(self, text_embedding):
        # the Linear on the library's own MFMA GEMM (tgsr::linear = gemm_bias_kernel, with its autograd formula: dW and dx are
        # the same kernel) when the tensors live on the GPU - no rocBLAS / Tensile kernel on the training path
        if text_embedding.is_cuda and text_embedding.dtype == torch.float32:
            y = C.linear(text_embedding, self.fc.weight, self.fc.bias)
        else:
            y = self.fc(text_embedding)
        x = self.relu(y)
        return x[:, :self.c_dim], x[:, self.c_dim:]

    def reparametrize(self, mu, logvar):
        std = logvar.mul(0.5).exp()
        eps = torch.empty_like(std).normal_()
        return eps.mul(std).add_(mu)

    def forward(self, text_embedding):
        if not self.training and not torch.is_grad_enabled():
            # inference: the Linear, the GLU and the re-parametrisation in one HIP launch (tgsr_ca_net_fwd); the normals
            # still come from torch's generator, as many as the reference draws
            eps = None
            if not (text_embedding.is_cuda and torch.cuda.is_current_stream_capturing()):
                # inside a hipGraph capture the draw is skipped (c_code is None): a captured normal_() makes every replay
                # fill two philox-state tensors first, and the x8 / x16 generators discard c_code anyway (model.py:51-52)
                eps = torch.empty(text_embedding.shape[0], self.c_dim, dtype=torch.float32, device=text_embedding.device).normal_()
            c_code, mu, logvar = C.ca_net(text_embedding, self.fc.weight.detach(), self.fc.bias.detach(), self.c_dim, eps)
            if eps is None:
                c_code = None
            return c_code, mu, logvar
        mu, logvar = self.encode(text_embedding)
        return self.reparametrize(mu, logvar), mu, logvar


# ------------------------------------------------------------------------------------------ generator stages
def _make_res_layers(channel_num):
    return nn.Sequential(*[ResBlock(channel_num) for _ in range(cfg.GAN.R_NUM)])


def _up_into(upsample, x, ngf, wide_out):
    """Run the stage's upBlock; with wide_out the result is written as the first half of a fresh
    [B, 2ngf, 2H, 2W] buffer (tagged on the returned view) so the next stage's torch.cat is free."""
    if not wide_out or upsample.training:
        return upsample(x)
    B, _, H, W = x.shape
    wide = torch.empty(B, 2 * ngf, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    out = upsample(x, out=wide[:, :ngf])
    out._tgsr_wide = wide
    return out


class INIT_STAGE_GImgup(nn.Module):
    """util.py:726-777: im2f -> word attention -> cat -> R_NUM ResBlocks -> upBlock.
    The cat never runs: im2f and the attention kernel write the two halves of one [B, 2ngf, H, W] buffer."""

    def __init__(self, ngf, ncf, nef, batchnorm=True):
        super(INIT_STAGE_GImgup, self).__init__()
        if not batchnorm:
            raise NotImplementedError("INIT_STAGE_GImgup(batchnorm=False) belongs to the dead stage1 branch")
        self.gf_dim = ngf
        self.in_dim = cfg.GAN.Z_DIM + ncf
        self.ef_dim = nef
        self.att = ATT_NET(self.gf_dim, self.ef_dim)
        self.im2f = _ConvBnGlu(3, ngf)
        self.upsample = upBlock(ngf * 2, ngf)
        self.residual = _make_res_layers(ngf * 2)

    def forward(self, c_code0, LR, word_embs, mask, wide_out=False, src=None):
        B, _, H, W = LR.shape
        ngf = self.gf_dim
        self.att.applyMask(mask)
        if self.training:   # differentiable path: plain cat (autograd), like the reference (util.py:769-777)
            h_code = self.im2f(LR)
            c_code, att = self.att(h_code, word_embs)
            out_code1 = self.residual(torch.cat((h_code, c_code), 1))
            return self.upsample(out_code1), att
        hc = torch.empty(B, 2 * ngf, H, W, dtype=torch.float32, device=LR.device)
        h_code = self.im2f(LR, out=hc[:, :ngf])
        _, att = self.att(h_code, word_embs, out=hc[:, ngf:], src=src)
        out_code1 = self.residual(hc)
        return _up_into(self.upsample, out_code1, ngf, wide_out), att


class NEXT_STAGE_G(nn.Module):
    """util.py:781-823: word attention -> cat -> R_NUM ResBlocks -> upBlock."""

    def __init__(self, ngf, nef, ncf, weightatten=False):
        super(NEXT_STAGE_G, self).__init__()
        if weightatten:
            raise NotImplementedError("GlobalAttentionGeneral_weight is unused on the shipped path")
        self.gf_dim = ngf
        self.ef_dim = nef
        self.cf_dim = ncf
        self.num_residual = cfg.GAN.R_NUM
        self.att = ATT_NET(ngf, self.ef_dim)
        self.residual = _make_res_layers(ngf * 2)
        self.upsample = upBlock(ngf * 2, ngf)

    def forward(self, h_code, c_code0, word_embs, mask, wide_out=False, src=None):
        B, ngf, H, W = h_code.shape
        self.att.applyMask(mask)
        if self.training:   # util.py:814-823 through autograd
            c_code, att = self.att(h_code, word_embs)
            out_code = self.residual(torch.cat((h_code, c_code), 1))
            return self.upsample(out_code), att
        wide = getattr(h_code, "_tgsr_wide", None)
        if wide is None:  # stand-alone use: build the concatenated buffer (one copy, = the reference's cat)
            wide = torch.empty(B, 2 * ngf, H, W, dtype=torch.float32, device=h_code.device)
            wide[:, :ngf].copy_(h_code)
            h_code = wide[:, :ngf]
        _, att = self.att(h_code, word_embs, out=wide[:, ngf:], src=src)
        out_code = self.residual(wide)
        return _up_into(self.upsample, out_code, ngf, wide_out), att


class GET_IMAGE_G_noAct(nn.Module):
    """util.py:909-919: conv3x3 ngf -> 3, no activation (key `img.0.weight`)."""

    def __init__(self, ngf):
        super(GET_IMAGE_G_noAct, self).__init__()
        self.gf_dim = ngf
        self.img = nn.Sequential(conv3x3(ngf, 3))

    def forward(self, h_code):
        return C.conv_to3(h_code, self.img[0].weight, False, None, 0.0)       # differentiable custom op (register_autograd)


class GET_IMAGE_G(nn.Module):
    """util.py:894-905: conv3x3 ngf -> 3 + Tanh (key `img.0.weight`); the x16 heads (models16.py:14)."""

    def __init__(self, ngf):
        super(GET_IMAGE_G, self).__init__()
        self.gf_dim = ngf
        self.img = nn.Sequential(conv3x3(ngf, 3), nn.Tanh())

    def forward(self, h_code):
        return C.conv_to3(h_code, self.img[0].weight, True, None, 0.0)
