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
    # (every parameter's region of the bucket starts on a multiple of 4 elements: 30 -> 32, 5 -> 8, 15 -> 16, 3 -> 4)
    assert e0 < 1e-6 and e1 < 1e-6 and n0 == n1 == 32 + 8 + 16 + 4


def _damsm_worker(rank, world, port, q):
    """DAMSM under data parallelism: the product's gather (parallel.gather_damsm_batch, the `x world` rule)
    with the ORACLE's words_loss / sent_loss as the function evaluated on the gathered batch (CPU; the HIP loss kernels have no
    CPU form - tests/test_hip_dp.py runs DAMSMTrainer through them on the GPU)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import tgsr_oracle as O
    from tgsr_amd import parallel
    parallel.init_distributed("gloo")
    torch.manual_seed(3)
    Bg, nef, T = 5, 8, 6                                   # uneven shards: 3 + 2 rows
    feat_in, pooled_in = torch.randn(Bg, 6, 3, 3), torch.randn(Bg, 7)
    words_in = torch.randn(Bg, 5, T)
    lens = torch.tensor([6, 5, 4, 4, 2])                   # globally sorted, descending
    class_ids = [3, 1, 3, 2, 0]

    def nets():
        torch.manual_seed(11)
        return (torch.nn.Conv2d(6, nef, 1), torch.nn.Linear(7, nef), torch.nn.Conv1d(5, nef, 1))

    def loss_of(regions, code, words, sent, ls, ids, B):
        labels = torch.arange(B)
        w0, w1, _ = O.words_loss(regions, words, labels, ls, ids, B, 4.0, 5.0, 10.0)
        s0, s1 = O.sent_loss(code, sent, labels, ids, B, 10.0)
        return w0 + w1 + s0 + s1

    # single process, concatenated batch
    ca, cb, cc = nets()
    words = cc(words_in)
    ref = loss_of(ca(feat_in), cb(pooled_in), words, words.mean(2), lens.tolist(), class_ids, Bg)
    ref.backward()
    ref_grads = [p.grad.clone() for m in (ca, cb, cc) for p in m.parameters()]
    # two ranks: each encodes its shard (its words only up to the shard's longest caption), gathers, evaluates the global loss
    da, db, dc = nets()
    params = [p for m in (da, db, dc) for p in m.parameters()]
    bucket = parallel.FlatGradBucket(params).attach()
    lo, hi = parallel.shard_bounds(Bg, rank, world)
    tmax = int(lens[lo:hi].max())
    w_loc = dc(words_in[lo:hi, :, :tmax])
    sent_loc = dc(words_in[lo:hi]).mean(2)
    g = parallel.gather_damsm_batch(da(feat_in[lo:hi]), db(pooled_in[lo:hi]), w_loc, sent_loc, lens[lo:hi].tolist(),
                                    class_ids[lo:hi], T)
    regions, code, gw, gs, glens, gids, B, scale = g
    assert B == Bg and scale == world and glens == lens.tolist() and list(gids) == class_ids and gw.shape[2] == T
    loss = loss_of(regions, code, gw, gs, glens, list(gids), B)
    (loss * scale).backward()
    bucket.all_reduce_mean()
    gerr = max(float((p.grad - r_).abs().max()) for p, r_ in zip(params, ref_grads))
    # a range of the bucket sent early, the rest later (what SRTrainer does under the tail of backward)
    b2 = parallel.FlatGradBucket(params).attach()
    b2.flat.copy_(torch.arange(b2.numel, dtype=torch.float32) * (rank + 1))
    h = b2.all_reduce_range_async(0, 10)
    h.wait()
    b2.all_reduce_mean(skip=(0, 10))
    want = torch.arange(b2.numel, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    rerr = float((b2.flat - want).abs().max())
    q.put((rank, abs(float(loss) - float(ref)), gerr, rerr))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_damsm_gather_equals_single_process_loss():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_damsm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, lerr, gerr, rerr in res:
        assert lerr < 1e-5, "gathered loss differs from the single-process loss on the concatenated batch: %g" % lerr
        assert gerr < 1e-5, "all-reduced gradient differs from the single-process gradient: %g" % gerr
        assert rerr < 1e-6


def test_shard_bounds_partition():
    from tgsr_amd.parallel import shard_bounds
    for n in (1, 7, 16, 33):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
