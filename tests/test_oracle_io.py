"""CPU: the image-pyramid restatement (oracle/tgsr_oracle_io.py) against (1) the fixture captured from the reference's
own datasets.get_imgs_blur and (2) Pillow itself where it is importable; plus the host-side data helpers."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz
from oracle import tgsr_oracle_io as IO


def test_pyramid_matches_reference_fixture():
    z = load_npz("io_pyramid.npz")
    sizes = [int(s) for s in z["sizes"]]
    ret, bic, retb, bicb = IO.pyramid(z["crop_u8"], sizes)
    for name, lst in (("ret", ret), ("bic", bic), ("retb", retb), ("bicb", bicb)):
        for i, a in enumerate(lst):
            assert np.array_equal(a, z["%s%d_u8" % (name, i)]), "%s[%d] differs from get_imgs_blur" % (name, i)
    assert np.array_equal(IO.normalize(ret[0]), z["ret0_f32"]) and np.array_equal(IO.normalize(bicb[1]), z["bicb1_f32"])


def test_restatement_matches_pillow_on_odd_shapes():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageFilter
    g = np.random.default_rng(3)
    a = g.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    im = Image.fromarray(a)
    for (w, h) in [(16, 16), (53, 20), (90, 37), (7, 5), (106, 74)]:
        ref = np.asarray(im.resize((w, h), Image.BILINEAR)).transpose(2, 0, 1)
        assert np.array_equal(IO.resize_bilinear(a.transpose(2, 0, 1), h, w), ref), (w, h)
    ref = np.asarray(im.filter(ImageFilter.GaussianBlur(radius=2))).transpose(2, 0, 1)
    assert np.array_equal(IO.gaussian_blur(np.ascontiguousarray(a.transpose(2, 0, 1))), ref)
    assert IO.gaussian_box_params(2.0, 3) == (1, 4473924, 1677722)


def test_host_tables_match_the_oracle():
    from tgsr_amd import datasets as D
    for (i, o) in [(256, 32), (32, 256), (178, 64), (64, 65)]:
        b, k = IO.resize_coeffs(i, o)
        b2, k2, ks = D._resize_tables(i, o)
        assert np.array_equal(b, b2) and np.array_equal(k, k2) and ks == k.shape[1]
    assert D.gaussian_box_params() == IO.gaussian_box_params()


def test_prepare_datablur_and_caption_pickle(tmp_path):
    from tgsr_amd import datasets as D
    B = 4
    lens = torch.tensor([5, 9, 3, 9])
    caps = torch.arange(B * 18).reshape(B, 18, 1)
    lists = [[torch.full((B, 3, s, s), float(i)) + torch.arange(B).float()[:, None, None, None] for s in (4, 8)] for i in range(4)]
    data = (lists[0], caps, lens, torch.arange(B), ["a", "b", "c", "d"], lists[1], lists[2], lists[3])
    imgs, captions, cap_lens, class_ids, keys, bic, blur, bicblur = D.prepare_datablur(data, device="cpu")
    assert cap_lens.tolist() == [9, 9, 5, 3] and captions.shape == (B, 18)
    order = [int(c[0, 0]) // 18 for c in captions.unsqueeze(-1)]
    assert sorted(order) == [0, 1, 2, 3] and lens[order].tolist() == [9, 9, 5, 3]
    assert [keys[i] for i in range(B)] == [["a", "b", "c", "d"][j] for j in order] and class_ids.tolist() == order
    assert float(imgs[1][0, 0, 0, 0]) == float(order[0]) and float(bicblur[0][2, 0, 0, 0]) == 3.0 + order[2]
    p = tmp_path / "c.pickle"
    with open(p, "wb") as f:
        pickle.dump([[[3, 4, 5], list(range(1, 25))], {0: "<end>", 3: "x"}, {"<end>": 0, "x": 3}], f, protocol=2)
    cap, ln, ixtoword, wordtoix = D.load_caption_pickle(str(p))
    assert cap.shape == (2, 18) and ln.tolist() == [3, 18] and cap[0, :4].tolist() == [3, 4, 5, 0] and wordtoix["x"] == 3
    shipped = os.path.join(os.path.dirname(GOLDEN), "..", "tests", "golden")   # no reference files are read by tests
    assert os.path.isdir(GOLDEN)
