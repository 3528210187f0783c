"""GPU parity of the reduced-precision (bf16 / f16) inference path against its CPU model (oracle/tgsr_oracle_lp.py,
same rounding points, fp32 accumulation) and against the fp32 oracle (PSNR bounds stated per dtype).

Elementwise tolerance of an lp kernel against the CPU model: both round the same fp32 value to `dtype`, but the fp32
sums differ in order, so a stored element may land on the neighbouring representable value: |a - b| <= ULP * |b| + tiny,
ULP = 2^-7 (bf16: 8 significant bits) / 2^-10 (f16: 11 bits).
"""
import numpy as np
import pytest
import torch

from oracle import tgsr_oracle as O
from oracle import tgsr_oracle_lp as OL

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [("bf16", torch.bfloat16, 2.0 ** -7), ("f16", torch.float16, 2.0 ** -10)]


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from tgsr_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()


def lp_close(a, b, ulp, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    err = (a - b).abs()
    tol = 1.01 * ulp * b.abs() + 1e-5       # + fp32 summation-order noise near zero
    bad = err > tol
    assert not bool(bad.any()), "%s: %d of %d elements off by more than one %g-ulp (worst %g at value %g)" % (
        what, int(bad.sum()), bad.numel(), ulp, float(err.max()), float(b.flatten()[err.argmax()]))


def test_lp_image_roundtrip():
    from tgsr_amd import lp
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 8, 32, generator=g)
    for name, td, _ in DTYPES:
        img = lp.from_nchw(x.to(DEV), name, cpitch=8, coff=2)
        assert img.shape == (3, 10, 34, 8) and img.dtype == td
        back = lp.to_nchw(img, 5, 2).cpu()
        assert torch.equal(back, OL.rnd(x, td))
        assert float(img[:, 0].abs().max()) == 0 and float(img[:, :, 0].abs().max()) == 0    # border untouched
        assert float(img[..., :2].abs().max()) == 0 and float(img[..., 7:].abs().max()) == 0


CONV_CASES = [
    # B, Cin, Cout, H, W (output), glu, upsample, residual
    (2, 64, 128, 8, 32, True, False, False),
    (1, 64, 64, 16, 32, False, False, True),
    (2, 64, 64, 8, 64, True, True, False),
    (1, 32, 64, 12, 32, True, False, False),
    (2, 32, 32, 8, 32, False, False, True),
    (1, 32, 32, 4, 64, False, False, False),
    (2, 32, 64, 16, 64, True, True, False),
    (3, 64, 128, 32, 64, True, False, False),      # 8-row tiles only from 512 workgroups on; still the 4-row kernel
    (33, 64, 128, 32, 128, True, False, False),    # 33*4*4 = 528 >= 512 tiles of 8 rows: the 8-row kernel
    (33, 32, 32, 32, 128, False, False, True),
    (17, 64, 64, 64, 128, True, True, False),
]


@pytest.mark.parametrize("name,td,ulp", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_lp_conv3x3(case, name, td, ulp):
    from tgsr_amd import lp, ops
    B, Cin, Cout, H, W, glu, up, res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    Hi, Wi = (H // 2, W // 2) if up else (H, W)
    x = OL.rnd(torch.randn(B, Cin, Hi, Wi, generator=g), td)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    scale = 1 + 0.1 * torch.randn(Cout, generator=g)
    shift = 0.1 * torch.randn(Cout, generator=g)
    co = Cout // 2 if glu else Cout
    r = OL.rnd(torch.randn(B, co, H, W, generator=g), td) if res else None
    ref = OL.conv_block(x, w, scale, shift, td, glu=glu, upsample=up, residual=r)
    # input in a wider buffer (cpitch 64 when Cin 32 would also work: the kernel reads channels [0, Cin)); output into a
    # channel slice of a wider image, residual read from a channel offset
    xi = lp.from_nchw(x.to(DEV), name, cpitch=Cin)
    ri = lp.from_nchw(r.to(DEV), name, cpitch=co + 8, coff=8) if res else None
    wp = lp.pack_conv3x3_weight(w.to(DEV), name)
    out = lp.new_image(B, H, W, co + 32, name, DEV)
    lp.conv3x3(xi, wp, Cin, Cout, scale.to(DEV), shift.to(DEV), glu=glu, upsample=up, residual=ri, res_coff=8,
               out=out, out_coff=32)
    torch.cuda.synchronize()
    got = lp.to_nchw(out, co, 32)
    lp_close(got, ref, ulp, "lp conv %s %s" % (name, case))
    assert float(out[..., :32].abs().max()) == 0, "channels outside the slice were written"
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, -1].abs().max()) == 0, "border written"
    assert float(out[:, :, 0].abs().max()) == 0 and float(out[:, :, -1].abs().max()) == 0, "border written"
