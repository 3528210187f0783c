"""GPU: the image pyramid kernels against the fixture captured from the reference's datasets.get_imgs_blur and the
numpy restatement of Pillow's arithmetic - byte-exact."""
import numpy as np
import pytest
import torch

from conftest import load_npz
from oracle import tgsr_oracle_io as IO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_gpu_pyramid_is_byte_identical_to_the_reference():
    from tgsr_amd.datasets import GpuImagePyramid
    z = load_npz("io_pyramid.npz")
    sizes = [int(s) for s in z["sizes"]]
    pyr = GpuImagePyramid(sizes, device=DEV)
    hr = torch.from_numpy(z["crop_u8"]).to(DEV)[None].repeat(3, 1, 1, 1)          # a batch of 3 copies
    hr[1] = hr[1].flip(-1)
    got = pyr(hr, u8=True)
    for name, lst in zip(("ret", "bic", "retb", "bicb"), got):
        for i, t in enumerate(lst):
            assert np.array_equal(t[0].cpu().numpy(), z["%s%d_u8" % (name, i)]), "%s[%d]" % (name, i)
            assert np.array_equal(t[2].cpu().numpy(), z["%s%d_u8" % (name, i)])
    fl = pyr(hr)
    assert np.array_equal(fl[0][0][0].cpu().numpy(), z["ret0_f32"]) and np.array_equal(fl[3][1][0].cpu().numpy(), z["bicb1_f32"])
    assert fl[0][3].shape == (3, 3, 256, 256) and fl[0][3].dtype == torch.float32
    # flipped sample vs the restatement
    ret, bic, retb, bicb = IO.pyramid(hr[1].cpu().numpy(), sizes)
    assert np.array_equal(got[2][2][1].cpu().numpy(), retb[2]) and np.array_equal(got[1][3][1].cpu().numpy(), bic[3])


@pytest.mark.parametrize("hin,win,hout,wout", [(37, 53, 16, 16), (37, 53, 74, 106), (64, 64, 64, 31), (20, 300, 45, 300)])
def test_resize_and_blur_odd_shapes(hin, win, hout, wout):
    from tgsr_amd.datasets import GpuImagePyramid
    g = np.random.default_rng(hin + win)
    a = g.integers(0, 256, (2, 3, hin, win), dtype=np.uint8)
    pyr = GpuImagePyramid((32, 64), device=DEV)
    x = torch.from_numpy(a).to(DEV)
    assert np.array_equal(pyr.resize(x, hout, wout).cpu().numpy(), IO.resize_bilinear(a, hout, wout))
    assert np.array_equal(pyr.gaussian_blur(x).cpu().numpy(), IO.gaussian_blur(a))


def test_pyramid_feeds_the_pipeline(face_weights):
    """End to end from uint8 HR crops: pyramid -> LR / blurred LR -> SRPipeline (the order gen_exampleSRHL wires them,
    trainer_objective.py:109-146) -> uint8 SR image."""
    from conftest import split_sd
    from oracle import tgsr_oracle as O
    from tgsr_amd.datasets import GpuImagePyramid
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.trainer import SRPipeline, to_uint8
    cfg_reset()
    cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
    z = load_npz("io_pyramid.npz")
    hr = torch.from_numpy(z["crop_u8"]).to(DEV)[None].repeat(2, 1, 1, 1)
    imgs, bic, imgsblur, bicblur = GpuImagePyramid((32, 64, 128, 256), device=DEV)(hr)
    cap, lens, _LR, _LRb = O.synthetic_batch(2)
    pipe = SRPipeline(41, device=DEV, branch_num=4).load_state_dicts(split_sd(face_weights, "E."), split_sd(face_weights, "GL."),
                                                       split_sd(face_weights, "GH."))
    out = pipe(cap.to(DEV), lens.tolist(), imgs[0], imgsblur[0])
    ref = O.sr_forward(split_sd(face_weights, "E."), split_sd(face_weights, "GL."), split_sd(face_weights, "GH."), cap,
                       lens.tolist(), imgs[0].cpu(), imgsblur[0].cpu())
    np.testing.assert_allclose(out["fine"][2].cpu().numpy(), ref["fine"][2].numpy(), atol=1e-4, rtol=1e-4)
    assert to_uint8(out["fine"][2]).shape == (2, 3, 256, 256)
    cfg_reset()
