"""CPU, gloo, world_size 2: the N>1 logic of tgsr_amd.parallel (sharding, flat-bucket gradient all-reduce)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from tgsr_amd import parallel
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    # --- sharding: every rank derives its slice of the same global batch
    B = 7
    cap_lens = torch.tensor([5, 9, 3, 9, 4, 12, 7])
    captions = torch.arange(B)[:, None].repeat(1, 4)
    imgs = torch.arange(B).float()[:, None]
    c, l, im, sel = parallel.shard_batch(rank, world, captions, cap_lens, imgs)
    assert (l[:-1] >= l[1:]).all()                      # still length-sorted inside the shard
    assert (cap_lens[sel] == l).all() and (imgs[sel] == im).all()
    # --- gradient bucket: mean of per-shard gradients == gradient of the mean loss over the global batch
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    x = torch.randn(8, 6)
    y = torch.randn(8, 3)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref.load_state_dict(model.state_dict())
    ((ref(x) - y) ** 2).mean().backward()
    bucket = parallel.FlatGradBucket(model.parameters()).attach()
    lo, hi = parallel.shard_bounds(8, rank, world)
    ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))   # wrote in place
    bucket.all_reduce_mean()
    err = max(float((p.grad - q_.grad).abs().max()) for p, q_ in zip(model.parameters(), ref.parameters()))
    # --- BatchNorm running statistics ride the same all-reduce: identical on every rank afterwards (= the mean over ranks)
    bnm = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 1), torch.nn.BatchNorm2d(4))
    torch.manual_seed(1)
    xb = torch.randn(8, 2, 3, 3) * 2 + 1
    bb = parallel.FlatGradBucket(bnm.parameters(), buffers=bnm.buffers()).attach()
    assert len(bb.buffers) == 2 and bb.flat.numel() == bb.numel and bb.flat_all.numel() == bb.numel + 8
    bnm(xb[lo:hi]).square().mean().backward()
    mine = [bnm[1].running_mean.clone(), bnm[1].running_var.clone()]
    both = [torch.zeros(2, 4), torch.zeros(2, 4)]
    for k in range(2):
        gathered = [torch.empty(4) for _ in range(world)]
        dist.all_gather(gathered, mine[k])
        both[k] = torch.stack(gathered).mean(0)
    bb.all_reduce_mean()
    berr = max(float((bnm[1].running_mean - both[0]).abs().max()), float((bnm[1].running_var - both[1]).abs().max()))
    assert berr < 1e-6 and int(bnm[1].num_batches_tracked) == 1
    assert float((mine[0] - bnm[1].running_mean).abs().max()) > 1e-4     # the shards' statistics did differ
    # --- async form + gather
    bucket2 = parallel.FlatGradBucket(model.parameters())
    h = bucket2.all_reduce_mean(async_op=True)
    h.wait()
    g = parallel.gather_images(torch.full((2, 1), float(rank)))
    if rank == 0:
        assert g.flatten().tolist() == [0.0, 0.0, 1.0, 1.0]
    q.put((rank, sel.tolist(), err, bucket.numel))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gradient_bucket():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, sel0, e0, n0), (r1, sel1, e1, n1) = res
    assert sorted(sel0 + sel1) == list(range(7)) and len(sel0) == 4 and len(sel1) == 3   # a partition of the batch
    assert e0 < 1e-6 and e1 < 1e-6 and n0 == n1 == 6 * 5 + 5 + 5 * 3 + 3


def test_shard_bounds_partition():
    from tgsr_amd.parallel import shard_bounds
    for n in (1, 7, 16, 33):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
