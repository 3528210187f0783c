"""Child process of tests/test_hip_dp.py: one data-parallel rank.  `gloo` (default): every rank on cuda:0 - the rehearsal
a one-GPU box allows; `nccl` (argv[2]): one rank per device over RCCL, through tgsr_amd.parallel.init_distributed's nccl
branch - what the driver's multi-GPU bench runs.  Not a test module."""
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


def make_trainer(correct_mask=True, device="cuda:0", discriminators=False):
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.train import SRTrainer
    cfg_reset()
    cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM, cfg.GAN.DF_DIM = 32, 64, 8
    torch.manual_seed(1234)                                   # identical initial weights on every rank
    tr = SRTrainer(41, device=device, discriminators=discriminators)
    for net in (tr.netGL.h_net1, tr.netGL.h_net2, tr.netGL.h_net3):
        net.att.correct_mask = correct_mask                   # per-sample masking: a shard's result must not depend on
    return tr                                                 # the local batch size


def make_pipeline(tr, device="cuda:0"):
    """Inference pipeline on a COPY of the trainer's initial weights - taken before any training forward, whose
    train-mode BatchNorm updates the running statistics with that rank's shard."""
    from tgsr_amd.trainer import SRPipeline
    pipe = SRPipeline(41, device=device, branch_num=4)
    pipe.netGL.load_state_dict(tr.netGL.state_dict())
    pipe.netGH.load_state_dict(tr.netGH.state_dict())
    pipe.text_encoder.load_state_dict(tr.text_encoder.state_dict())
    for net in (pipe.netGL.h_net1, pipe.netGL.h_net2, pipe.netGL.h_net3):
        net.att.correct_mask = True
    return pipe


def damsm_case(B=4):
    from oracle import tgsr_oracle as O
    cap, lens, _LR, _LRb = O.synthetic_batch(B, seed=21)
    g = torch.Generator().manual_seed(5)
    return {"cap": cap, "lens": lens, "feats": torch.randn(B, 768, 17, 17, generator=g), "pooled": torch.randn(B, 2048, generator=g),
            "class_ids": [0, 1, 0, 2][:B]}


def make_damsm_trainer(device="cuda:0"):
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.train import DAMSMTrainer
    cfg_reset()
    cfg.TEXT.EMBEDDING_DIM = 64
    cfg.TRAIN.FLAG = True
    torch.manual_seed(4321)
    tr = DAMSMTrainer(41, device=device, gather_negatives=True)
    tr.text_encoder.drop.p = 0.0 if hasattr(tr.text_encoder, "drop") else 0.0          # no dropout noise in the comparison
    return tr


def shard_grads(tr, batch, lo, hi):
    cap, lens, LR, LRb, hr = batch
    dev = tr.device
    tr._zero(tr.bucket)
    loss, _, _ = tr.loss(cap[lo:hi].to(dev), lens[lo:hi].tolist(), LR[lo:hi].to(dev), LRb[lo:hi].to(dev),
                         [h[lo:hi].to(dev) for h in hr])
    loss.backward()
    tr.bucket.end_step()
    return loss.detach()


def main():
    out = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
    from tgsr_amd import parallel
    if backend == "nccl":
        rank, local_rank, world = parallel.init_distributed("nccl")    # sets the device, RCCL process group on it
        dev = "cuda:%d" % local_rank
        assert dist.get_backend() == "nccl" and torch.cuda.current_device() == local_rank
    else:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        dev = "cuda:0"
    B = 4
    batch = build_batch(B)
    tr = make_trainer(device=dev)
    pipe = make_pipeline(tr, device=dev)
    lo, hi = parallel.shard_bounds(B, rank, world)
    loss = shard_grads(tr, batch, lo, hi)
    tr.bucket.all_reduce_mean()                               # ONE collective over the flat bucket
    cap, lens, LR, LRb, hr = batch                            # inference shard with per-sample masking
    o = pipe(cap[lo:hi].to(dev), lens[lo:hi].tolist(), LR[lo:hi].to(dev), LRb[lo:hi].to(dev))
    torch.cuda.synchronize()
    res = {"flat": tr.bucket.flat.cpu(), "loss": float(loss), "fine": o["fine"][2].cpu(), "lo": lo, "hi": hi}
    # ---- one optimisation step of the trainer itself (tr.step: the early all-reduce of NetG_highweight's range under the
    # tail of backward + the closing one): every rank must leave it with the same parameters and running statistics
    tr.step(cap[lo:hi].to(dev), lens[lo:hi].tolist(), LR[lo:hi].to(dev), LRb[lo:hi].to(dev), [h[lo:hi].to(dev) for h in hr])
    torch.cuda.synchronize()
    res["step_flat"] = tr.bucket.flat_all.cpu()
    res["step_params"] = torch.cat([p.detach().flatten() for p in tr.params]).cpu()
    # ---- DAMSM pre-training step with the contrastive losses of the GLOBAL batch (gathered features / embeddings)
    d = damsm_case()
    dt = make_damsm_trainer(dev)
    dl = dt.step_features(d["feats"][lo:hi].to(dev), d["pooled"][lo:hi].to(dev), d["cap"][lo:hi].to(dev), d["lens"][lo:hi].tolist(),
                          d["class_ids"][lo:hi])
    torch.cuda.synchronize()
    res["damsm_loss"], res["damsm_flat"] = float(dl), dt.bucket.flat.cpu()
    # ---- updates replayed from hipGraphs under data parallelism (train.py `_capture_g` / `_capture_d_update`: segments with the
    # bucket's all-reduce BETWEEN them) against the eager data-parallel step, same initial weights, same shards, same noise: bit
    # for bit - the generator-only step and the G/D alternation
    # (the discriminators are built for 64 / 128 / 256-pixel images: the G/D runs take a 32 x 32 batch)
    cap32, lens32, LR32, LRb32, hr32 = build_batch(B, lr=32)
    for gan in (False, True):
        runs = []
        bc, bl, bLR, bLRb, bhr = (cap32, lens32, LR32, LRb32, hr32) if gan else (cap, lens, LR, LRb, hr)
        for graphs in (False, True):
            t = make_trainer(device=dev, discriminators=gan)
            assert t._graph_capable
            t._graph_g = graphs
            if not graphs:
                t._dsteps = -10 ** 9
            ls = []
            for it in range(6):
                torch.manual_seed(50 + it)
                ls.append(float(t.step(bc[lo:hi].to(dev), bl[lo:hi].tolist(), bLR[lo:hi].to(dev), bLRb[lo:hi].to(dev),
                                       [h[lo:hi].to(dev) for h in bhr])))
            torch.cuda.synchronize()
            sd = [v.detach().flatten().float() for m in [t.netGL, t.netGH] + list(t.netsD) for v in m.state_dict().values()]
            runs.append({"losses": ls, "state": torch.cat(sd).cpu(),
                         "split": bool(t._ggraphs) and all(isinstance(c, dict) and c["opt"] is not None for c in t._ggraphs.values()) and
                         all(isinstance(c, dict) and c["opt"] is not None for c in t._dgraphs)})
            del t
        tag = "gan" if gan else "g"
        res["graph_%s_equal" % tag] = runs[0]["losses"] == runs[1]["losses"] and torch.equal(runs[0]["state"], runs[1]["state"])
        res["graph_%s_split" % tag] = runs[1]["split"] and not runs[0]["split"]
        res["graph_%s_state" % tag] = runs[1]["state"]
        res["graph_%s_losses" % tag] = (runs[0]["losses"], runs[1]["losses"])
    # ---- the MEASURED graph policy (TGSR_GRAPH_G=auto, the default) under data parallelism: the ranks time their eager and replayed
    # steps, take the slowest rank's medians (one all-reduce) and settle on the SAME form; whatever it is, the steps leave the
    # bits of the pinned-eager data-parallel trainer
    from tgsr_amd import train as _train
    runs = []
    for mode in ("eager", "auto"):
        t = make_trainer(device=dev, discriminators=False)
        if mode == "eager":
            t._graph_g = False
        else:
            assert t._auto is not None
        ls = []
        for it in range(_train.GRAPH_G_SETTLED + 1):
            torch.manual_seed(70 + it)
            ls.append(float(t.step(cap[lo:hi].to(dev), lens[lo:hi].tolist(), LR[lo:hi].to(dev), LRb[lo:hi].to(dev),
                                   [h[lo:hi].to(dev) for h in hr])))
        torch.cuda.synchronize()
        runs.append({"losses": ls, "state": torch.cat([p.detach().flatten() for p in t.params]).cpu(), "policy": dict(t.graph_policy),
                     "settled": t._auto is None})
        del t
    res["graph_auto_equal"] = runs[0]["losses"] == runs[1]["losses"] and torch.equal(runs[0]["state"], runs[1]["state"])
    res["graph_auto_policy"], res["graph_auto_settled"] = runs[1]["policy"], runs[1]["settled"]
    if backend == "nccl":
        # the G/D alternation's collectives: each discriminator's bucket is all-reduced on that discriminator's own
        # stream (train.py step_gan), then the generators' bucket on the main one - four RCCL all-reduces per step
        tg = make_trainer(device=dev, discriminators=True)
        errG, errsD = tg.step_gan(cap32[lo:hi].to(dev), lens32[lo:hi].tolist(), LR32[lo:hi].to(dev), LRb32[lo:hi].to(dev),
                                  [h[lo:hi].to(dev) for h in hr32])
        torch.cuda.synchronize()
        res["gan_flat"] = tg.bucket.flat.cpu()
        res["gan_params_d"] = [torch.cat([p.detach().flatten() for p in b.params]).cpu() for b in tg.bucketsD]
        res["gan_losses"] = [float(errG)] + [float(e) for e in errsD]
        res["gan_params"] = torch.cat([p.detach().flatten() for p in tg.params]).cpu()
    torch.save(res, "%s.rank%d.pt" % (out, rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
