#!/usr/bin/env python3
"""Experiment: the G/D alternation (three discriminator updates + the generator update, Adam, EMA) captured in one
hipGraph.  The eager step issues ~1800 launches and is host-bound (kernel time ~13.6 ms of a 29 ms step,
profiles/r02i_train_gan_kernel_stats.csv): how much of the gap does a replay recover?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM, cfg.TREE.BRANCH_NUM = 32, 256, 4
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.train import SRTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gan = (sys.argv[2] if len(sys.argv) > 2 else "gan") == "gan"
tr = SRTrainer(41, device="cuda", discriminators=gan)
tr.opt = torch.optim.Adam(tr.params, lr=cfg.TRAIN.GENERATOR_LR, betas=(0.5, 0.999), capturable=True)
tr.optsD = [torch.optim.Adam(d.parameters(), lr=cfg.TRAIN.DISCRIMINATOR_LR, betas=(0.5, 0.999), capturable=True) for d in tr.netsD]
cap, lens, LR, LRb = synthetic_batch(B, seed=100)
g = torch.Generator().manual_seed(7)
hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).cuda() for s in (64, 128, 256)]
cap, LR, LRb, lens = cap.cuda(), LR.cuda(), LRb.cuda(), lens.tolist()


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print("eager step: %.2f ms" % bench(lambda: tr.step(cap, lens, LR, LRb, hr)), flush=True)
for variant in ("streams", "one-stream"):
    if variant == "one-stream":
        tr._dstreams, tr._wside = [], None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            tr.step(cap, lens, LR, LRb, hr)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    try:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            loss = tr.step(cap, lens, LR, LRb, hr)
        torch.cuda.synchronize()
        print(variant, "captured", flush=True)
        for i in range(3):
            gr.replay()
            torch.cuda.synchronize()
            print("  replay", i, float(loss), flush=True)
        print("%s graphed step: %.2f ms" % (variant, bench(lambda: gr.replay())), flush=True)
    except Exception as e:
        print(variant, "capture failed:", type(e).__name__, str(e)[:400], flush=True)
        torch.cuda.synchronize()
