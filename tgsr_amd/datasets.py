"""The data edge of the SR path: the pieces of the reference's datasets.py the hot path touches.

  * `GpuImagePyramid`  - `get_imgs_blur` (datasets.py:151-197) on the GPU: HR pyramid, the pyramid re-grown from the LR
    image, and their GaussianBlur(radius=2) versions, normalised to [-1, 1].  The arithmetic is Pillow's (what
    `transforms.Resize` and `ImageFilter.GaussianBlur` delegate to), restated in integer HIP kernels
    (tgsr_resize_bilinear_u8 / tgsr_gaussian_blur_u8 / tgsr_u8_normalize): byte-identical pyramids, so an end-to-end
    run is no longer bound by the CPU image library.
  * `prepare_data` / `prepare_datablur` - datasets.py:33-109: sort the batch by caption length (descending, the
    pack_padded_sequence order) and move it to the device; same tuple layout as the reference.
  * `load_caption_pickle` - the `[captions, ixtoword, wordtoix]` pickle test1.py:118-127 writes and the datasets read.
Dataset classes, tokenising (nltk) and file walking stay with the caller: CPU-side data preparation, not the hot path.
"""
import math
import pickle

import numpy as np
import torch

from . import ops
from ._lib import TgsrError

_PB = 32 - 8 - 2       # Pillow's PRECISION_BITS


def _resize_tables(in_size: int, out_size: int):
    """Pillow precompute_coeffs (bilinear) in double -> (bounds int32 [out,2], taps int32 [out,ksize], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        tot = sum(w)
        if tot != 0.0:
            w = [v / tot for v in w]
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, -0.5 + kk * (1 << _PB), 0.5 + kk * (1 << _PB)).astype(np.int32)
    return bounds, ik, ksize


def gaussian_box_params(radius: float = 2.0, passes: int = 3):
    """Pillow _gaussian_blur_radius + ImagingHorizontalBoxBlur's weights: (int radius, ww, fw) (float32 like the C code)."""
    f = np.float32
    sigma2 = float(f(radius) * f(radius) / f(passes))
    L = math.sqrt(12.0 * sigma2 + 1.0)
    l = math.floor((L - 1.0) / 2.0)
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2) / (6 * (sigma2 - (l + 1) * (l + 1)))
    fr = f(l + a)
    r = int(fr)
    ww = int(f(1 << 24) / f(fr * f(2) + f(1)))
    return r, ww, ((1 << 24) - (r * 2 + 1) * ww) // 2


class GpuImagePyramid:
    """datasets.py:151-197 for a batch of already cropped HR images [B, 3, S, S] uint8 on the device.

        imgs, bic, imgsblur, bicblur = GpuImagePyramid((32, 64, 128, 256))(hr_u8)

    Each is a list over the scales of float32 [B, 3, s, s] in [-1, 1] (= the reference's `ret, bic, retb, bicb`);
    `u8=True` returns the uint8 pyramids instead."""

    def __init__(self, sizes=(32, 64, 128, 256), blur_radius: float = 2.0, device="cuda"):
        self.sizes = tuple(int(s) for s in sizes)
        self.device = torch.device(device)
        self.blur = gaussian_box_params(blur_radius, 3)
        self._tables = {}

    def _table(self, n_in, n_out):
        key = (n_in, n_out)
        t = self._tables.get(key)
        if t is None:
            b, k, ks = _resize_tables(n_in, n_out)
            t = self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), ks)
        return t

    def resize(self, x: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
        """PIL `resize((out_w, out_h), BILINEAR)` of planar uint8 images [..., H, W]."""
        H, W = x.shape[-2], x.shape[-1]
        return ops.resize_bilinear_u8(x, out_h, out_w, self._table(W, out_w) if out_w != W else None,
                                      self._table(H, out_h) if out_h != H else None)

    def gaussian_blur(self, x: torch.Tensor) -> torch.Tensor:
        """PIL `filter(ImageFilter.GaussianBlur(radius))` of planar uint8 images [..., H, W]."""
        r, ww, fw = self.blur
        return ops.gaussian_blur_u8(x, r, ww, fw, 3)

    @staticmethod
    def normalize(x: torch.Tensor) -> torch.Tensor:
        """ToTensor + Normalize((0.5,)*3, (0.5,)*3) (datasets.py:286-288)."""
        return ops.u8_normalize(x)

    def __call__(self, hr_u8: torch.Tensor, u8: bool = False):
        S = self.sizes[-1]
        if hr_u8.dim() != 4 or hr_u8.shape[1] != 3 or tuple(hr_u8.shape[2:]) != (S, S):
            raise TgsrError("GpuImagePyramid: expected [B,3,%d,%d] uint8, got %s" % (S, S, tuple(hr_u8.shape)))
        lr = self.resize(hr_u8, self.sizes[0], self.sizes[0])                     # `lrimg`, datasets.py:170
        ret, bic, retb, bicb = [], [], [], []
        for i, s in enumerate(self.sizes):
            re = self.resize(hr_u8, s, s) if i < len(self.sizes) - 1 else hr_u8    # :177-181
            bi = self.resize(lr, s, s)                                            # :191
            ret.append(re)
            retb.append(self.gaussian_blur(re))                                   # :186
            bic.append(bi)
            bicb.append(self.gaussian_blur(bi))                                   # :192
        if u8:
            return ret, bic, retb, bicb
        n = self.normalize
        return [n(t) for t in ret], [n(t) for t in bic], [n(t) for t in retb], [n(t) for t in bicb]


def _sorted_to(dev, cap_lens, lists):
    lens, idx = torch.sort(cap_lens, 0, True)
    return lens, idx, [[t[idx].to(dev) for t in lst] for lst in lists]


def prepare_data(data, cfg=None, device=None):
    """datasets.py:33-68: (imgs, captions, cap_lens, class_ids, keys, bic) -> the same six, sorted by caption length
    (descending) and on the device (`cfg.CUDA` picks cuda like the reference unless `device` is given)."""
    imgs, captions, captions_lens, class_ids, keys, bic = data
    dev = torch.device(device if device is not None else ("cuda" if (cfg is None or cfg.CUDA) else "cpu"))
    lens, idx, (real_imgs, real_bic) = _sorted_to(dev, captions_lens, (imgs, bic))
    captions = captions[idx].squeeze().to(dev)
    class_ids = class_ids[idx].numpy() if torch.is_tensor(class_ids) else np.asarray(class_ids)[idx.numpy()]
    keys = [keys[i] for i in idx.numpy()]
    return [real_imgs, captions, lens.to(dev), class_ids, keys, real_bic]


def prepare_datablur(data, cfg=None, device=None):
    """datasets.py:71-109: the eight-tuple form with the blurred pyramids (what gen_exampleSRHL unpacks,
    trainer_objective.py:109)."""
    imgs, captions, captions_lens, class_ids, keys, bic, blur, bicblur = data
    dev = torch.device(device if device is not None else ("cuda" if (cfg is None or cfg.CUDA) else "cpu"))
    lens, idx, (real_imgs, real_blur, real_bic, real_bicblur) = _sorted_to(dev, captions_lens, (imgs, blur, bic, bicblur))
    captions = captions[idx].squeeze().to(dev)
    class_ids = class_ids[idx].numpy() if torch.is_tensor(class_ids) else np.asarray(class_ids)[idx.numpy()]
    keys = [keys[i] for i in idx.numpy()]
    return [real_imgs, captions, lens.to(dev), class_ids, keys, real_bic, real_blur, real_bicblur]


def get_caption(sent_caption, words_num=18, rng=None):
    """datasets.py:461-477: zero-pad a caption to `words_num` tokens; a LONGER caption keeps a random subset of
    `words_num` word positions in their original order (np.random.shuffle of the positions, the first `words_num` of them,
    sorted).  `rng`: None = numpy's global generator, exactly the reference's draw (seed it with np.random.seed like
    test1.py:171 does); a np.random.RandomState / Generator for a private stream.  Returns (x int64 [words_num], x_len)."""
    c = np.asarray(sent_caption).astype('int64')
    x = np.zeros(words_num, dtype='int64')
    n = len(c)
    if n <= words_num:
        x[:n] = c
        return x, n
    ix = list(np.arange(n))
    (np.random if rng is None else rng).shuffle(ix)
    ix = np.sort(ix[:words_num])
    x[:] = c[ix]
    return x, words_num


def load_caption_pickle(path, words_num=18, rng=None):
    """The pickle test1.py:118-127 writes: `[captions (lists of word indices), ixtoword, wordtoix]`.  Returns
    (captions int64 [N, words_num], cap_lens int64 [N], ixtoword, wordtoix), every caption padded / cropped by
    `get_caption` (datasets.py:461-477: captions longer than `words_num` keep a random ordered subset of their words)."""
    with open(path, "rb") as f:
        x = pickle.load(f)
    caps, ixtoword, wordtoix = x[0], x[-2], x[-1]
    out = torch.zeros(len(caps), words_num, dtype=torch.int64)
    lens = torch.zeros(len(caps), dtype=torch.int64)
    for i, c in enumerate(caps):
        row, n = get_caption(list(c), words_num, rng)
        out[i] = torch.from_numpy(row)
        lens[i] = n
    return out, lens, ixtoword, wordtoix
