"""GPU: optim.FlatAdam (tgsr::adam_flat_: one launch over flat parameter / gradient / moment buffers) against torch.optim.Adam - the
reference's optimizer (trainer_objective.py: Adam(lr, betas=(0.5, 0.999))) - on the same gradients."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from tgsr_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()


def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 7, 3), torch.nn.BatchNorm2d(7), torch.nn.Conv2d(7, 5, 1, bias=False),
                               torch.nn.Flatten(), torch.nn.Linear(5 * 6 * 6, 3)).to(DEV)


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adam_follows_torch_adam(wd):
    from tgsr_amd.optim import FlatAdam
    from tgsr_amd.parallel import FlatGradBucket
    a, b = _net(3), _net(3)
    ref = torch.optim.Adam(a.parameters(), lr=2e-3, betas=(0.5, 0.999), eps=1e-8, weight_decay=wd)
    bucket = FlatGradBucket(b.parameters()).attach()
    opt = FlatAdam(bucket.params, bucket.flat, lr=2e-3, betas=(0.5, 0.999), eps=1e-8, weight_decay=wd)
    ptrs = [p.data_ptr() for p in b.parameters()]
    assert all(opt.flat.data_ptr() <= q < opt.flat.data_ptr() + 4 * opt.flat.numel() for q in ptrs)     # the parameters moved in
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        x = torch.randn(4, 3, 8, 8, generator=g).to(DEV)
        a.zero_grad(set_to_none=False)
        a(x).square().mean().backward()
        with torch.no_grad():                                 # the SAME gradients for both optimizers (b's .grad are the bucket's views)
            for p, q in zip(a.parameters(), b.parameters()):
                q.grad.copy_(p.grad)
        ref.step()
        opt.step()
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            assert torch.allclose(p, q, rtol=2e-6, atol=2e-7), (step, n, float((p - q).abs().max()))
        if step == 2:                                         # a learning-rate change is picked up by the next step
            for o in (ref, opt):
                o.param_groups[0]["lr"] = 5e-4
    assert float(opt.counters[0]) == 6.0
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(ref.state[p]["exp_avg"], opt.state[q]["exp_avg"], rtol=1e-5, atol=1e-8)
        assert torch.allclose(ref.state[p]["exp_avg_sq"], opt.state[q]["exp_avg_sq"], rtol=1e-5, atol=1e-10)
    # state_dict / load_state_dict speak torch.optim.Adam's format; loading copies INTO the flat buffers
    sd = copy.deepcopy(opt.state_dict())
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    m_ptr = opt.exp_avg.data_ptr()
    c = _net(3)
    bucket_c = FlatGradBucket(c.parameters()).attach()
    opt_c = FlatAdam(bucket_c.params, bucket_c.flat, lr=1.0, betas=(0.5, 0.999))
    opt_c.load_state_dict(sd)
    assert opt_c.param_groups[0]["lr"] == 5e-4 and torch.equal(opt_c.exp_avg, opt.exp_avg) and torch.equal(opt_c.exp_avg_sq, opt.exp_avg_sq)
    assert float(opt_c.counters[0]) == 6.0 and torch.allclose(opt_c.counters, opt.counters, rtol=1e-6)    # (host pow vs the device's)
    assert opt.exp_avg.data_ptr() == m_ptr


def test_flat_adam_counts_its_steps_on_the_device_across_graph_replays():
    from tgsr_amd.optim import FlatAdam
    from tgsr_amd.parallel import FlatGradBucket
    a, b = _net(5), _net(5)
    ba, bb = FlatGradBucket(a.parameters()).attach(), FlatGradBucket(b.parameters()).attach()
    oa, ob = FlatAdam(ba.params, ba.flat, lr=1e-3, betas=(0.5, 0.999)), FlatAdam(bb.params, bb.flat, lr=1e-3, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(2)
    grads = [torch.randn(ba.flat.numel(), generator=g).to(DEV) for _ in range(4)]
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        ob.step()
    for k in range(4):
        ba.flat.copy_(grads[k])
        bb.flat.copy_(grads[k])
        oa.step()
        gr.replay()
    torch.cuda.synchronize()
    assert float(ob.counters[0]) == 4.0
    assert torch.equal(oa.flat, ob.flat) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)


def test_flat_adam_refuses_parameters_that_left_its_buffer():
    from tgsr_amd.optim import FlatAdam
    from tgsr_amd.parallel import FlatGradBucket
    net = _net(7)
    bucket = FlatGradBucket(net.parameters()).attach()
    opt = FlatAdam(bucket.params, bucket.flat, lr=1e-3, betas=(0.5, 0.999))
    opt.step()                                              # fine: the parameters are the views the optimizer made
    with torch.no_grad():
        last = bucket.params[-1]
        last.data = last.data.clone()                       # what a module.to(...) after construction amounts to
    with pytest.raises(RuntimeError, match="no longer lives"):
        opt.step()
