"""GPU: the data-parallel path on the HIP kernels, rehearsed as 2 ranks on ONE GPU (gloo; each rank a fresh child
process started before this process touches the device).  The CPU gloo test (test_distributed_cpu.py) cannot drive the
HIP path - there is no CPU fallback by design.

  * training: after ONE all-reduce of the flat gradient bucket, every rank holds the mean of the per-shard gradients -
    compared with the same two shards run one after the other in a single process (BatchNorm statistics are per shard in
    both, SURVEY.md 8e: that is the data-parallel semantics, so the comparison is exact up to fp32 summation order);
  * inference: with per-sample masking (`correct_mask=True`) a rank's slice of the batch equals the same rows of the
    un-sharded run (atol = rtol = 1e-4: kernel selection depends on the local batch size)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_ranks(tmp_path, backend):
    """Both ranks as child processes, started before this process has touched a device."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out, backend], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return [torch.load("%s.rank%d.pt" % (out, k)) for k in range(2)]


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_gradients_and_sharded_inference(tmp_path, backend):
    """gloo: both ranks on cuda:0 (always runs).  nccl: one rank per device over RCCL - skipped on a one-GPU box, runs for
    real wherever two devices are visible (torch.cuda.device_count() does not initialise the GPU on this image)."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: %d visible" % torch.cuda.device_count())
    r = _run_ranks(tmp_path, backend)
    assert torch.equal(r[0]["flat"], r[1]["flat"]), "ranks disagree after the all-reduce"

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker as W
    batch = W.build_batch(4)
    tr = W.make_trainer()
    pipe = W.make_pipeline(tr)
    flats = []
    for k in range(2):
        W.shard_grads(tr, batch, r[k]["lo"], r[k]["hi"])
        flats.append(tr.bucket.flat.detach().cpu().clone())
    mean = (flats[0] + flats[1]) / 2
    err = float((r[0]["flat"] - mean).abs().max()) / float(mean.abs().max())
    assert err < 1e-5, "all-reduced gradient differs from the mean of the per-shard gradients: %g" % err
    assert float(mean.abs().max()) > 0

    cap, lens, LR, LRb, _ = batch
    full = pipe(cap.cuda(), lens.tolist(), LR.cuda(), LRb.cuda())["fine"][2].cpu()
    # not bitwise: the kernel selection depends on the local batch (Winograd vs direct below 256 workgroups, util._wino_pays)
    # and the shard's T_max may be shorter; the stated fp32 tolerance of the path applies
    for k in range(2):
        assert torch.allclose(full[r[k]["lo"]:r[k]["hi"]], r[k]["fine"], atol=1e-4, rtol=1e-4), \
            "sharded inference differs from the un-sharded rows"
    # ---- SRTrainer.step under data parallelism: NetG_highweight's gradient range is all-reduced under the tail of backward
    # (train.py `_fire_early`), the rest with the closing collective - every rank leaves the step with the same gradients,
    # running statistics (they ride the bucket's tail) and parameters
    assert torch.equal(r[0]["step_flat"], r[1]["step_flat"]) and float(r[0]["step_flat"].abs().max()) > 0
    assert torch.equal(r[0]["step_params"], r[1]["step_params"])
    # ---- DAMSM on the gathered global batch: both ranks report the loss of the WHOLE batch and hold its
    # gradient - the single-process step on the concatenated batch
    assert abs(r[0]["damsm_loss"] - r[1]["damsm_loss"]) < 1e-6 and torch.equal(r[0]["damsm_flat"], r[1]["damsm_flat"])
    d = W.damsm_case()
    dt = W.make_damsm_trainer()
    one = dt.step_features(d["feats"].cuda(), d["pooled"].cuda(), d["cap"].cuda(), d["lens"].tolist(), d["class_ids"])
    torch.cuda.synchronize()
    assert abs(float(one) - r[0]["damsm_loss"]) < 1e-4 * max(1.0, abs(float(one))), (float(one), r[0]["damsm_loss"])
    ref = dt.bucket.flat.cpu()
    gerr = float((r[0]["damsm_flat"] - ref).abs().max()) / float(ref.abs().max())
    assert gerr < 1e-4, "gathered-batch DAMSM gradient differs from the single-process gradient: %g" % gerr
    # ---- the hipGraph-replayed updates under data parallelism (segments around the bucket's all-reduce) == the eager
    # data-parallel steps, bit for bit, on both ranks; and both ranks hold the same parameters and running statistics after them
    for tag in ("g", "gan"):
        for k in range(2):
            assert r[k]["graph_%s_split" % tag], "rank %d: the %s update was not replayed in segments" % (k, tag)
            assert r[k]["graph_%s_equal" % tag], (tag, k, r[k]["graph_%s_losses" % tag])
        assert torch.equal(r[0]["graph_%s_state" % tag], r[1]["graph_%s_state" % tag])
    # the measured graph policy settles on ONE form for both ranks (the slowest rank's medians decide) and changes no bit
    for k in range(2):
        assert r[k]["graph_auto_settled"] and r[k]["graph_auto_equal"], (k, r[k]["graph_auto_policy"])
    assert r[0]["graph_auto_policy"]["chosen"] == r[1]["graph_auto_policy"]["chosen"]
    assert r[0]["graph_auto_policy"]["eager_ms"] == r[1]["graph_auto_policy"]["eager_ms"]
    if backend == "nccl":
        # G/D alternation over RCCL: four all-reduces per step on four streams; after it every rank holds the same
        # generator gradients and has taken the same Adam steps (the shards differ, so identical discriminator parameters
        # after the step mean their gradients were all-reduced before it)
        assert torch.equal(r[0]["gan_flat"], r[1]["gan_flat"]) and float(r[0]["gan_flat"].abs().max()) > 0
        for a, b in zip(r[0]["gan_params_d"], r[1]["gan_params_d"]):
            assert torch.equal(a, b)
        assert torch.equal(r[0]["gan_params"], r[1]["gan_params"])
        assert all(torch.isfinite(torch.tensor(r[k]["gan_losses"])).all() for k in range(2))


def test_rccl_executes_on_this_box_single_rank(tmp_path):
    """RCCL (backend "nccl" on ROCm) itself, on whatever the box has: a ONE-rank process group - the only shape a one-GPU box
    allows over RCCL (it refuses two ranks on one device) - built the way `parallel.init_distributed`'s nccl branch does, then the collectives
    the data-parallel path issues, on the product's own tensors: the flat bucket's all-reduce (a whole-buffer one and the early
    range + rest pair of SRTrainer) and the DAMSM all-gather.  With one rank every collective is the identity, which is the
    check; what it proves is that librccl loads, builds a communicator on this device and runs on the product's streams."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from tgsr_amd import parallel
torch.cuda.set_device(0)                       # (parallel.init_distributed only builds a group for world > 1: this is its nccl branch)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and parallel.dp_world() == 1
m = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.Linear(32, 8)).cuda()
b = parallel.FlatGradBucket(m.parameters()).attach()
b.flat.copy_(torch.arange(b.numel, dtype=torch.float32, device="cuda"))
want = b.flat.clone()
dist.all_reduce(b.flat_all, op=dist.ReduceOp.SUM)                       # the step's closing collective
h = dist.all_reduce(b.flat_all[:100], op=dist.ReduceOp.SUM, async_op=True)   # the early range, on its own stream
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    h.wait()
torch.cuda.current_stream().wait_stream(side)
x = torch.randn(3, 5, device="cuda", requires_grad=True)
g = parallel._AllGatherCat.apply(x)                                     # DAMSM's gather (autograd-aware)
g.square().sum().backward()
torch.cuda.synchronize()
assert torch.equal(b.flat, want) and torch.equal(g.detach(), x.detach()) and torch.allclose(x.grad, 2 * x.detach())
print("RCCL_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
''' % ROOT
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout + p.stderr


def test_allreduce_flat_through_the_c_abi_single_rank(tmp_path):
    """`tgsr_allreduce_flat` (SURVEY 8b's op list): librccl opened by libtgsr_hip.so itself, a communicator built from a unique id
    (one rank: what a one-GPU box allows), buf <- scale * sum over ranks - in a child process, like every test that builds a
    communicator.  Checked: the scaled identity on a 1.2 M-float bucket, FlatGradBucket(comm=RcclDirect) reducing through it, and a
    second, independent communicator."""
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from tgsr_amd import ops, parallel
assert ops.comm_available()
torch.cuda.set_device(0)
uid = ops.comm_unique_id()
assert len(uid) == 128 and any(uid)
comm = ops.comm_init(uid, 0, 1)
assert ops.comm_count(comm) == (1, 0)            # ncclCommCount / ncclCommUserRank: what RCCL itself reports
x = torch.randn(1191313, device="cuda")
want = (x * 0.25).clone()
ops.allreduce_flat(comm, x, 0.25)
torch.cuda.synchronize()
assert torch.equal(x, want)
ops.comm_destroy(comm)
rc = parallel.RcclDirect.create()
m = torch.nn.Linear(8, 4).cuda()
b = parallel.FlatGradBucket(m.parameters(), comm=rc).attach()
b.flat.copy_(torch.arange(b.numel, dtype=torch.float32, device="cuda"))
import torch.distributed as dist
b.comm.all_reduce_mean_(b.flat_all)
torch.cuda.synchronize()
assert torch.equal(b.flat, torch.arange(b.numel, dtype=torch.float32, device="cuda"))
rc.close()
try:
    ops.allreduce_flat(0, x, 1.0)
    raise SystemExit("a null communicator must be refused")
except ops.TgsrError:
    pass
print("ALLREDUCE_FLAT_OK")
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ALLREDUCE_FLAT_OK" in p.stdout, p.stdout + p.stderr
