"""GPU: the `torch.ops.tgsr.*` custom operators (tgsr_amd/custom_ops.py) - torch.library.opcheck on the two
differentiable ones, agreement with the ctypes wrappers they sit on, and that the drop-in modules dispatch through them."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from tgsr_amd import _lib, custom_ops  # noqa: F401
    _lib.lib()
    assert torch.cuda.is_available()


def test_opcheck_conv_to3_and_conv4x4s2():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 32, 16, 16, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(3, 32, 5, 5, generator=g) / 28.0).to(DEV).requires_grad_(True)
    add = torch.randn(2, 3, 16, 16, generator=g).to(DEV)
    utils = ("test_schema", "test_autograd_registration", "test_faketensor")
    torch.library.opcheck(torch.ops.tgsr.conv_to3.default, (x, w, True, add, 0.5), test_utils=utils)
    torch.library.opcheck(torch.ops.tgsr.conv_to3.default, (x, w[:, :, 1:4, 1:4].contiguous(), False, None, 0.0),
                          test_utils=utils)
    xd = torch.randn(2, 8, 16, 16, generator=g).to(DEV).requires_grad_(True)
    wd = (torch.randn(16, 8, 4, 4, generator=g) / 11.0).to(DEV).requires_grad_(True)
    torch.library.opcheck(torch.ops.tgsr.conv4x4s2.default, (xd, wd, True), test_utils=utils)
    torch.library.opcheck(torch.ops.tgsr.glu.default, (torch.randn(3, 8, 5, generator=g).to(DEV).requires_grad_(True),),
                          test_utils=utils)
    torch.library.opcheck(torch.ops.tgsr.conv3x3_fused_out.default,
                          (torch.randn(1, 32, 8, 32, device=DEV), torch.zeros(32 * 32 * 9, device=DEV), 32, None, None,
                           False, False, None, torch.empty(1, 32, 8, 32, device=DEV)), test_utils=("test_schema", "test_faketensor"))


def test_custom_op_autograd_matches_torch():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 32, 8, 16, generator=g)
    w = torch.randn(3, 32, 3, 3, generator=g) / 17.0
    add = torch.randn(3, 3, 8, 16, generator=g)
    dy = torch.randn(3, 3, 8, 16, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = torch.tanh(F.conv2d(xr, wr, None, 1, 1)) + 0.5 * add
    ref.backward(dy)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    out = torch.ops.tgsr.conv_to3(xd, wd, True, add.to(DEV), 0.5)
    out.backward(dy.to(DEV))
    assert torch.allclose(out.cpu(), ref, atol=1e-5) and torch.allclose(xd.grad.cpu(), xr.grad, atol=1e-5)
    assert torch.allclose(wd.grad.cpu(), wr.grad, atol=1e-4, rtol=1e-4)


def test_modules_dispatch_through_torch_ops(monkeypatch):
    """util's fused blocks call torch.ops.tgsr.* (not the ctypes wrappers directly): count the dispatches of one forward."""
    from tgsr_amd import custom_ops as C, util
    from tgsr_amd.miscc.config import cfg, cfg_reset
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    calls = {"n": 0}
    real = C.conv3x3_wino

    def counting(*a):
        calls["n"] += 1
        return real(*a)
    monkeypatch.setattr(C, "conv3x3_wino", counting)
    rb = util.ResBlock(64).to(DEV).eval()
    y = rb(torch.randn(2, 64, 16, 32, device=DEV))
    torch.cuda.synchronize()
    assert y.shape == (2, 64, 16, 32) and calls["n"] == 2
    assert "tgsr::conv3x3_wino" in str(torch.ops.tgsr.conv3x3_wino.default._schema)
    cfg_reset()


def test_filter_packed_for_another_channel_count_raises():
    """A weight [Cout, Cin', 3, 3] applied to a Cin-channel input is a shape error in torch (the reference raises it for
    NetG_highweight at GF_DIM != 32: ResBlock(channel_num=32) is hard-coded, model.py:258-262).  The HIP kernels walk
    the packed filter by the INPUT's channel count, so the wrappers must refuse instead of reading past the pack."""
    from tgsr_amd import lp, ops
    from tgsr_amd._lib import TgsrError
    x = torch.randn(2, 128, 32, 32, device=DEV)
    w = torch.randn(64, 32, 3, 3, device=DEV)
    with pytest.raises(TgsrError):
        ops.conv3x3_wino(x, ops.pack_wino_weight(w), 64, None, None)
    with pytest.raises(TgsrError):
        ops.conv3x3_fused(x, ops.pack_conv3x3_weight(w), 64, None, None)
    with pytest.raises(TgsrError):
        ops.upwino_glu(x, ops.pack_upwino_weight(w, glu=False), 64, None, None, glu=False)
    xi = lp.from_nchw(x[:, :64].contiguous(), "bf16")
    with pytest.raises(TgsrError):
        lp.conv3x3(xi, lp.pack_conv3x3_weight(w, "bf16"), 64, 64, None, None)
