"""Data parallelism for the SR path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).

The reference is single-process (trainer_objective.py:31, no DataParallel / distributed anywhere - SURVEY section 2).
The path shards by image:
  * inference: no data-path collective; each rank runs its slice of the (globally length-sorted) minibatch;
    `gather_images` exists only for reporting / saving;
  * training: ONE all-reduce per step over a flat fp32 bucket holding every gradient (G: 1 191 313 params =
    4.77 MB - far below the size where bucketing into several collectives pays on 7 x 153 GB/s xGMI links), then
    x 1/world.  BatchNorm NORMALISES with per-shard batch statistics (the DistributedDataParallel convention; SyncBN
    would be 36 latency-bound collectives per forward).  The RUNNING statistics every rank accumulates from its own
    shards would drift apart - DDP avoids that by broadcasting rank 0's buffers before each forward - so here they ride
    the gradient bucket: FlatGradBucket(params, buffers=...) appends running_mean / running_var to the flat buffer, the
    step's ONE all-reduce averages them with the gradients and they are copied back: every rank holds the same
    buffers after every step (a checkpoint is the same file whichever rank writes it; rank 0 does), each the mean over
    ranks of the per-shard running statistics - for the mean that is the running mean of the global batch, for the
    variance the running average of the per-shard variances, i.e. of the quantity this policy normalises with.
"""
import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: str = None) -> tuple:
    """(rank, local_rank, world) from the torchrun environment; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard_bounds(n: int, rank: int, world: int) -> tuple:
    """Rows [lo, hi) of an n-row batch owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(rank: int, world: int, captions: torch.Tensor, cap_lens: torch.Tensor, *per_sample: torch.Tensor):
    """Global sort by caption length (datasets.py:74-96), then a contiguous row slice per rank.  A slice of a
    descending sequence is still descending, which is all pack_padded_sequence / the BiLSTM kernel need.  Note the
    reference's mask quirk (GlobalAttention.py:109-116) makes a sample's attention depend on the LOCAL batch size:
    bit-parity with a 1-GPU run needs correct_mask=True or equal-length captions."""
    lens, idx = torch.sort(cap_lens, 0, True)
    lo, hi = shard_bounds(captions.shape[0], rank, world)
    sel = idx[lo:hi]
    return (captions[sel], lens[lo:hi]) + tuple(t[sel] for t in per_sample) + (sel,)


# Direct gradient slots: data_ptr of a parameter -> (bucket, index).  The HIP backward Functions (autograd.ConvBnAct, ...)
# ask `grad_slot(param)` for the parameter's region of its flat bucket and let their weight-gradient kernels write
# there; autograd's AccumulateGrad then adopts that tensor as `.grad` without an add kernel (it only adds when a `.grad`
# already exists).  Profile of the round-1 train step: 148 elementwise adds per step (0.78 ms) were exactly these
# accumulations into pre-set `.grad` views.
_SLOTS = {}


def grad_slot(param: torch.Tensor):
    """A fresh view of `param`'s gradient region in its bucket when the bucket runs in direct mode and the region has
    not been written this step; else None (the caller allocates as usual and autograd accumulates)."""
    hit = _SLOTS.get(param.data_ptr())
    if hit is None:
        return None
    bucket, i = hit
    if not bucket.direct or bucket.filled[i]:
        return None
    bucket.filled[i] = True
    off, n = bucket.offsets[i]
    return bucket.flat[off:off + n].view(bucket.params[i].shape)   # a NEW tensor object: autograd may adopt it


class FlatGradBucket:
    """All gradients of `params` in one flat fp32 buffer; `all_reduce_mean()` = one collective per step."""

    def __init__(self, params: Iterable[torch.nn.Parameter], buffers: Iterable[torch.Tensor] = (), comm=None):
        """buffers: module buffers to keep identical across ranks (BatchNorm running_mean / running_var; integer buffers
        such as num_batches_tracked advance in lockstep and are skipped): they travel in the tail of the flat buffer
        and come back as the mean over ranks with every all_reduce_mean() - no collective of their own."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        # every parameter's region starts on a multiple of 4 elements (16 bytes): optim.FlatAdam lays the PARAMETERS out the same
        # way, and the kernels want 16-byte aligned weights; the padding elements stay zero (gradient, moments and update)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append((off, p.numel()))
            off += (p.numel() + 3) & ~3
        self.numel = off
        self.buffers: List[torch.Tensor] = [b for b in buffers if b.is_floating_point() and b.dtype == dt]
        self.flat_all = torch.zeros(self.numel + sum(b.numel() for b in self.buffers), dtype=dt, device=dev)
        self.flat = self.flat_all[:self.numel]                 # the gradients (what begin_step zeroes, what is clipped)
        self.views = [self.flat[o:o + n].view_as(p) for p, (o, n) in zip(self.params, self.offsets)]
        off = self.numel
        self.buf_views = []
        for b in self.buffers:
            self.buf_views.append(self.flat_all[off:off + b.numel()].view_as(b))
            off += b.numel()
        self.direct = False
        self.filled = [False] * len(self.params)
        self.comm = comm          # an RcclDirect: all_reduce_mean() then goes through tgsr_allreduce_flat instead of torch.distributed

    def attach(self):
        """Make every p.grad a view of the flat buffer, so backward writes straight into the bucket (no packing)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self

    def begin_step(self):
        """Direct mode: zero the bucket, clear every `.grad` and open the slots.  The backward kernels then write weight
        gradients straight into the bucket (grad_slot) and autograd adopts those tensors as `.grad`."""
        self.direct = True
        self.flat.zero_()
        for i, p in enumerate(self.params):
            p.grad = None
            self.filled[i] = False
            _SLOTS[p.data_ptr()] = (self, i)

    def end_step(self):
        """After backward: every `.grad` becomes its bucket view - adopted slots already are (same memory), gradients
        that arrived through other paths are copied in, parameters that received none read zeros."""
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                v.copy_(g)
            p.grad = v
            _SLOTS.pop(p.data_ptr(), None)       # the registry only holds buckets with a step in flight
        self.direct = False

    def flush_params(self, i0: int, i1: int):
        """end_step() for the parameters [i0, i1) only, DURING backward, once their gradients are final: their region of the
        flat buffer is then complete and may go out (all_reduce_range_async); end_step() later finds them in place."""
        for i in range(i0, i1):
            p, v = self.params[i], self.views[i]
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                v.copy_(g)
            p.grad = v
            _SLOTS.pop(p.data_ptr(), None)

    def pack(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)

    def unpack(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v.clone() if p.grad is None else p.grad.copy_(v)

    def all_reduce_range_async(self, lo: int, hi: int):
        """Start the all-reduce of flat[lo:hi] (a range of whole parameters whose gradients are final) while backward is
        still producing the rest; returns a handle whose wait() scales the range.  all_reduce_mean(skip=(lo, hi)) then
        reduces what is left.  Used by train.SRTrainer to send NetG_highweight's gradients - complete when backward enters
        G_SR_NET_low - under the tail of backward."""
        if dp_world() == 1:
            return None
        part = self.flat_all[lo:hi]
        work = dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True)
        world = dist.get_world_size()

        class _Done:
            def wait(_s):
                work.wait()
                part.div_(world)
        return _Done()

    def all_reduce_mean(self, async_op: bool = False, skip=None):
        """sum over ranks, x 1/world.  With attach() the gradients are already in place: pack/unpack are no-ops."""
        self.pack()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        world = dist.get_world_size()
        if self.buffers:
            with torch.no_grad():
                torch._foreach_copy_(self.buf_views, self.buffers)      # this step's running statistics -> the tail

        if self.comm is not None and not async_op and skip is None:
            self.comm.all_reduce_mean_(self.flat_all)       # sum and x 1/world inside the C ABI call, on the current stream
            self.unpack()
            if self.buffers:
                with torch.no_grad():
                    torch._foreach_copy_(self.buffers, self.buf_views)
            return None
        if skip is not None:
            # [lo, hi) went out earlier (all_reduce_range_async) and is already averaged: reduce the two pieces around it
            lo, hi = skip
            pieces = [t for t in (self.flat_all[:lo], self.flat_all[hi:]) if t.numel()]
            for t in pieces:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                t.div_(world)
            self.unpack()
            if self.buffers:
                with torch.no_grad():
                    torch._foreach_copy_(self.buffers, self.buf_views)
            return None

        def finish():
            self.flat_all.div_(world)
            self.unpack()
            if self.buffers:
                with torch.no_grad():
                    torch._foreach_copy_(self.buffers, self.buf_views)  # the same values on every rank (bumps versions:
                                                                        # eval-mode packs derived from them are rebuilt)
        if async_op:
            work = dist.all_reduce(self.flat_all, op=dist.ReduceOp.SUM, async_op=True)

            class _Done:
                def wait(_s):
                    work.wait()
                    finish()
            return _Done()
        dist.all_reduce(self.flat_all, op=dist.ReduceOp.SUM)
        finish()
        return None


# DAMSM's contrastive losses under data parallelism: True = every rank evaluates them on the gathered GLOBAL batch (the loss a
# single process would compute on the concatenated batch; gather_damsm_batch), False = on its own shard (B - 1 negatives per
# sample instead of world x B - 1: cheaper - the pair kernel's work grows with world^2 when gathered - but a different loss).
GATHER_NEGATIVES = os.environ.get("TGSR_DP_GATHER_NEGATIVES", "1") != "0"


class RcclDirect:
    """The gradient collective through the library's own C ABI (tgsr_allreduce_flat: librccl opened by libtgsr_hip.so, no
    torch.distributed on the data path).  The 128-byte communicator id still has to travel from rank 0 to the others once:
    `create()` uses the already initialised torch.distributed group (any backend) for that broadcast - a non-Python host would
    use its own channel.  FlatGradBucket(..., comm=RcclDirect.create()) then reduces with it."""

    def __init__(self, comm, rank, world):
        self.comm, self.rank, self.world = comm, rank, world

    @classmethod
    def create(cls):
        from . import ops
        if not ops.comm_available():
            raise RuntimeError("RcclDirect: librccl could not be opened by libtgsr_hip.so")
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            rank, world = dist.get_rank(), dist.get_world_size()
            box = [ops.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
        else:
            rank, world, box = 0, 1, [ops.comm_unique_id()]
        return cls(ops.comm_init(box[0], rank, world), rank, world)

    def all_reduce_mean_(self, flat: torch.Tensor) -> torch.Tensor:
        from . import ops
        return ops.allreduce_flat(self.comm, flat, 1.0 / self.world)

    def count(self) -> tuple:
        """(world, rank) from RCCL itself (ncclCommCount / ncclCommUserRank)."""
        from . import ops
        return ops.comm_count(self.comm)

    def close(self):
        from . import ops
        if self.comm:
            ops.comm_destroy(self.comm)
            self.comm = None


def dp_world() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


class _AllGatherCat(torch.autograd.Function):
    """cat(all_gather(x), 0) whose backward hands every rank the gradient rows of ITS slice.  Correct when every rank goes
    on to evaluate the SAME function of the gathered tensor (the replicated contrastive loss below): rank r's copy of
    dL/d(gathered)[rows of r] is then the whole derivative with respect to its slice - no reduce-scatter needed."""

    @staticmethod
    def forward(ctx, x):
        world, rank = dist.get_world_size(), dist.get_rank()
        x = x.contiguous()
        n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        ns = [int(v) for v in ns]
        m = max(ns)
        if x.shape[0] < m:                                   # shards differ by at most one row (shard_bounds): pad, trim below
            x = torch.cat((x, x.new_zeros((m - x.shape[0],) + tuple(x.shape[1:]))))
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        ctx.lo, ctx.n = sum(ns[:rank]), ns[rank]
        return torch.cat([p_[:k] for p_, k in zip(parts, ns)], 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.lo:ctx.lo + ctx.n].contiguous()


def gather_cat(x: torch.Tensor) -> torch.Tensor:
    """The rows of every rank, in rank order (= the order of the globally length-sorted batch, shard_batch), differentiable."""
    return x if dp_world() == 1 else _AllGatherCat.apply(x)


def gather_damsm_batch(regions, code, words_embs, sent_emb, cap_lens, class_ids, width: int):
    """DAMSM under data parallelism: `words_loss` / `sent_loss` are batch-contrastive (a B x B matching matrix,
    losses.py:45-59, 116-133), so a shard alone would see only its own B - 1 negatives.  This gathers what the two losses
    read - region features [B,nef,17,17], image codes [B,nef], word embeddings [B,nef,T] (padded to `width` words: T = the
    shard's longest caption), sentence codes [B,nef], caption lengths, class ids - from every rank, ~5 MB at 16 x 8 images, so
    that every rank evaluates the loss of the GLOBAL batch (identical on all ranks: the single-process loss on the
    concatenated batch); its gradient flows back into the local rows only.  Returns the gathered arguments, the global batch
    size and `world`: the caller multiplies the term by `world` before backward, because the gradient bucket's all-reduce
    AVERAGES the per-rank parameter gradients while the rank contributions of a replicated loss must be SUMMED."""
    world = dp_world()
    lens = cap_lens.tolist() if torch.is_tensor(cap_lens) else [int(v) for v in cap_lens]
    if world == 1:
        return regions, code, words_embs, sent_emb, lens, class_ids, len(lens), 1
    if words_embs.shape[2] < width:
        words_embs = torch.nn.functional.pad(words_embs, (0, width - words_embs.shape[2]))
    dev = words_embs.device
    meta = torch.tensor(lens, dtype=torch.int64, device=dev)
    glens = [int(v) for v in _AllGatherCat.apply(meta)]
    gids = None
    if class_ids is not None:
        import numpy as np
        ids = torch.as_tensor(np.asarray(class_ids), dtype=torch.int64).to(dev)
        gids = _AllGatherCat.apply(ids).cpu().numpy()
    return (gather_cat(regions), gather_cat(code), gather_cat(words_embs), gather_cat(sent_emb), glens, gids, len(glens), world)


def gather_images(img: torch.Tensor, dst: int = 0):
    """Collect every rank's output slice on `dst` (reporting only; not on the timed path).  Equal slice sizes."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return img
    world = dist.get_world_size()
    out = [torch.empty_like(img) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(img, out, dst=dst)
    return torch.cat(out, 0) if out is not None else None
