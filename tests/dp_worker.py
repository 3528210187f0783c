"""Child process of tests/test_hip_dp.py: one data-parallel rank (gloo, every rank on cuda:0).  Not a test module."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_batch(B, lr=16):
    from oracle import tgsr_oracle as O
    cap, lens, LR, LRb = O.synthetic_batch(B, lr=lr)
    g = torch.Generator().manual_seed(7)
    hr = [torch.rand(B, 3, lr * s, lr * s, generator=g) * 2 - 1 for s in (2, 4, 8)]
    return cap, lens, LR, LRb, hr


def make_trainer(correct_mask=True):
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.train import SRTrainer
    cfg_reset()
    cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 64
    torch.manual_seed(1234)                                   # identical initial weights on every rank
    tr = SRTrainer(41, device="cuda:0")
    for net in (tr.netGL.h_net1, tr.netGL.h_net2, tr.netGL.h_net3):
        net.att.correct_mask = correct_mask                   # per-sample masking: a shard's result must not depend on
    return tr                                                 # the local batch size


def make_pipeline(tr):
    """Inference pipeline on a COPY of the trainer's initial weights - taken before any training forward, whose
    train-mode BatchNorm updates the running statistics with that rank's shard."""
    from tgsr_amd.trainer import SRPipeline
    pipe = SRPipeline(41, device="cuda:0")
    pipe.netGL.load_state_dict(tr.netGL.state_dict())
    pipe.netGH.load_state_dict(tr.netGH.state_dict())
    pipe.text_encoder.load_state_dict(tr.text_encoder.state_dict())
    for net in (pipe.netGL.h_net1, pipe.netGL.h_net2, pipe.netGL.h_net3):
        net.att.correct_mask = True
    return pipe


def shard_grads(tr, batch, lo, hi):
    cap, lens, LR, LRb, hr = batch
    tr._zero(tr.bucket)
    loss, _, _ = tr.loss(cap[lo:hi].cuda(), lens[lo:hi].tolist(), LR[lo:hi].cuda(), LRb[lo:hi].cuda(),
                         [h[lo:hi].cuda() for h in hr])
    loss.backward()
    tr.bucket.end_step()
    return loss.detach()


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from tgsr_amd.parallel import shard_bounds
    B = 4
    batch = build_batch(B)
    tr = make_trainer()
    pipe = make_pipeline(tr)
    lo, hi = shard_bounds(B, rank, world)
    loss = shard_grads(tr, batch, lo, hi)
    tr.bucket.all_reduce_mean()                               # ONE collective over the flat bucket
    cap, lens, LR, LRb, _ = batch                             # inference shard with per-sample masking
    o = pipe(cap[lo:hi].cuda(), lens[lo:hi].tolist(), LR[lo:hi].cuda(), LRb[lo:hi].cuda())
    torch.cuda.synchronize()
    torch.save({"flat": tr.bucket.flat.cpu(), "loss": float(loss), "fine": o["fine"][2].cpu(), "lo": lo, "hi": hi},
               "%s.rank%d.pt" % (out, rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
