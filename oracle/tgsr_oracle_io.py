"""CPU restatement of the reference's image-pyramid arithmetic (datasets.py:151-197 `get_imgs_blur`, :236-278).
TEST INFRASTRUCTURE - NOT PRODUCT CODE (only tests/, smoke() and bench.py's cpu_baseline leg may import it).

The reference delegates this arithmetic to third-party libraries:
  * `transforms.Resize(size)` - torchvision (ABSENT from /root/reference and from this image; version unpinned, no
    requirements file).  For PIL images it is `img.resize((ow, oh), Image.BILINEAR)` with the smaller edge matched to
    `size`; the arithmetic is Pillow's `ImagingResample` (Pillow IS installed here: 12.x): separable triangle filter
    whose support widens with the down-scale factor, coefficients normalised in double, converted to 22-bit fixed
    point, horizontal pass then vertical pass, each rounded and clipped to uint8.
  * `img.filter(ImageFilter.GaussianBlur(radius=2))` - Pillow's `ImagingGaussianBlur`: three passes of an "extended box
    blur" of fractional radius 1.375 per axis in 24-bit fixed point with clamped borders.
  * `normalize` = ToTensor + Normalize(0.5, 0.5): (u8 / 255 - 0.5) / 0.5 in float32.
Both integer stages below are pinned BYTE-EXACT against Pillow itself (tests/test_oracle_io.py runs the comparison
wherever Pillow is importable) and against fixtures made by calling the reference's own `get_imgs_blur`
(tests/golden/make_io_golden.py -> tests/golden/io_pyramid.npz).
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resize_coeffs(in_size: int, out_size: int):
    """Pillow `precompute_coeffs` for the bilinear (triangle) filter: per output index (first input index, tap count)
    and the taps as 22-bit fixed-point integers."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)])
        tot = w.sum()
        if tot != 0:
            w = w / tot
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS)).astype(np.int32)
    return bounds, ik


def _resample_axis(a: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    a = np.moveaxis(a, axis, 0)
    bounds, ik = resize_coeffs(a.shape[0], out_size)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx, (xmin, xmax) in enumerate(bounds):
        ss = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            ss = ss + a[xmin + x].astype(np.int64) * int(ik[xx, x])
        out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear(a: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """a [..., H, W] uint8 -> [..., out_h, out_w] like PIL `resize((out_w, out_h), BILINEAR)`: horizontal pass, then
    vertical pass (a pass whose size does not change is skipped, as in Pillow)."""
    if out_w != a.shape[-1]:
        a = _resample_axis(a, out_w, a.ndim - 1)
    if out_h != a.shape[-2]:
        a = _resample_axis(a, out_h, a.ndim - 2)
    return a


def gaussian_box_params(radius: float = 2.0, passes: int = 3):
    """Pillow `_gaussian_blur_radius` + the fixed-point weights of `ImagingHorizontalBoxBlur`: (int radius, ww, fw)."""
    f = np.float32
    sigma2 = float(f(radius) * f(radius) / f(passes))
    L = np.sqrt(12.0 * sigma2 + 1.0)
    l = np.floor((L - 1.0) / 2.0)
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2) / (6 * (sigma2 - (l + 1) * (l + 1)))
    fr = f(l + a)
    r = int(fr)
    ww = int(f(1 << 24) / f(fr * f(2) + f(1)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def _hbox(a: np.ndarray, r: int, ww: int, fw: int) -> np.ndarray:
    """One horizontal extended-box-blur pass along the last axis, clamped borders, uint32 arithmetic."""
    W = a.shape[-1]
    idx = np.arange(W)
    acc = np.zeros(a.shape, np.uint64)
    for d in range(-r, r + 1):
        acc += a[..., np.clip(idx + d, 0, W - 1)].astype(np.uint64)
    far = a[..., np.clip(idx - r - 1, 0, W - 1)].astype(np.uint64) + a[..., np.clip(idx + r + 1, 0, W - 1)].astype(np.uint64)
    bulk = (acc * ww + far * fw) & 0xFFFFFFFF
    return (((bulk + (1 << 23)) & 0xFFFFFFFF) >> 24).astype(np.uint8)


def gaussian_blur(a: np.ndarray, radius: float = 2.0, passes: int = 3) -> np.ndarray:
    """a [..., H, W] uint8 like PIL `filter(ImageFilter.GaussianBlur(radius))`: `passes` horizontal box passes, then the
    same number along the vertical axis."""
    r, ww, fw = gaussian_box_params(radius, passes)
    for _ in range(passes):
        a = _hbox(a, r, ww, fw)
    a = np.swapaxes(a, -1, -2)
    for _ in range(passes):
        a = _hbox(a, r, ww, fw)
    return np.ascontiguousarray(np.swapaxes(a, -1, -2))


def normalize(a: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize((0.5,)*3, (0.5,)*3) (datasets.py:286-288): float32 (u8 / 255 - 0.5) / 0.5."""
    return ((a.astype(np.float32) / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5)


def pyramid(img: np.ndarray, sizes):
    """datasets.py:151-197 `get_imgs_blur` on an already cropped image img [3, S, S] uint8 (S = sizes[-1]):
    returns (ret, bic, retb, bicb) as lists of uint8 arrays [3, s, s]: the HR pyramid, the pyramid re-grown from the
    smallest image (`lrimg`), and their Gaussian-blurred versions.  The last level of `ret` is the image itself."""
    lr = resize_bilinear(img, sizes[0], sizes[0])
    ret, bic, retb, bicb = [], [], [], []
    for i, s in enumerate(sizes):
        re = resize_bilinear(img, s, s) if i < len(sizes) - 1 else img
        bi = resize_bilinear(lr, s, s)
        ret.append(re)
        retb.append(gaussian_blur(re))
        bic.append(bi)
        bicb.append(gaussian_blur(bi))
    return ret, bic, retb, bicb
