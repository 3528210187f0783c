#!/usr/bin/env python3
"""Golden vectors for the image pyramid: runs the reference's own `datasets.get_imgs_blur`
(datasets.py:151-197) on the shipped `data/face/000155.png` and stores inputs + outputs in tests/golden/io_pyramid.npz.
TEST INFRASTRUCTURE - runs only where /root/reference exists (the build container).

datasets.py imports nltk, cv2 and torchvision at module level; none is installed here and none takes part in
get_imgs_blur except `torchvision.transforms`.  Harness-side stubs (the reference's files are imported, never copied):
  * nltk.tokenize / cv2 / easydict: empty modules;
  * torchvision.transforms: `Resize(size)` = Pillow `img.resize(..., Image.BILINEAR)` with torchvision's size rule
    (an int matches the smaller edge), `ToTensor`, `Normalize`, `Compose` with their documented arithmetic.  The
    ARITHMETIC of the pyramid is Pillow's (installed here), which is what the fixture pins.
"""
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

REF = os.environ.get("TGSR_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _stubs():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for name in ("nltk", "nltk.tokenize", "cv2", "pandas_stub"):
        m = types.ModuleType(name)
        sys.modules.setdefault(name, m)
    sys.modules["nltk.tokenize"].RegexpTokenizer = object
    sys.modules["nltk"].tokenize = sys.modules["nltk.tokenize"]

    class Resize:
        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            w, h = img.size
            if isinstance(self.size, int):
                if (w <= h and w == self.size) or (h <= w and h == self.size):
                    return img
                ow, oh = (self.size, int(self.size * h / w)) if w < h else (int(self.size * w / h), self.size)
            else:
                oh, ow = self.size
            return img.resize((ow, oh), Image.BILINEAR)

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean)[:, None, None], torch.tensor(std)[:, None, None]

        def __call__(self, t):
            return (t - self.mean) / self.std

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tr.Resize, tr.ToTensor, tr.Normalize, tr.Compose = Resize, ToTensor, Normalize, Compose
    tv.transforms = tr
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr
    return tr


def main():
    tr = _stubs()
    import datasets                                  # the reference's datasets.py
    cfg = types.SimpleNamespace(GAN=types.SimpleNamespace(B_DCGAN=False), TREE=types.SimpleNamespace(BRANCH_NUM=4))
    path = os.path.join(REF, "data", "face", "000155.png")
    # the callers' `transform` yields a square crop of the final size (test1.py / datasets.py:1558-1562): here a
    # deterministic one - Resize(256 * 76 / 64) then a centre crop of 256
    def transform(img):
        img = tr.Resize(int(256 * 76 / 64))(img)
        w, h = img.size
        l, t = (w - 256) // 2, (h - 256) // 2
        return img.crop([l, t, l + 256, t + 256])
    norm = tr.Compose([tr.ToTensor(), tr.Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))])
    sizes = [32, 64, 128, 256]
    ret, bic, retb, bicb = datasets.get_imgs_blur(path, sizes, None, transform, normalize=norm, cfg=cfg)
    crop = np.asarray(transform(Image.open(path).convert("RGB"))).transpose(2, 0, 1).copy()
    out = {"crop_u8": crop, "sizes": np.array(sizes)}

    def u8(t):
        return np.round((t.numpy() * 0.5 + 0.5) * 255).astype(np.uint8)
    for name, lst in (("ret", ret), ("bic", bic), ("retb", retb), ("bicb", bicb)):
        for i, t in enumerate(lst):
            out["%s%d_u8" % (name, i)] = u8(t)
    out["ret0_f32"] = ret[0].numpy()                 # pins the float normalisation
    out["bicb1_f32"] = bicb[1].numpy()
    np.savez_compressed(os.path.join(OUT, "io_pyramid.npz"), **out)
    print("io_pyramid.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
