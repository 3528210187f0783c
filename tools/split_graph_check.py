#!/usr/bin/env python3
"""Experiment: the reduced-precision step as TWO hipGraphs on two streams (text encoder + G_SR_NET_low | NetG_highweight's
trunk) joined by an event before the three heads, against the single two-branch graph (whose branches the runtime runs
one after the other: tools/graph_timeline.sh).   python tools/split_graph_check.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.trainer import SRPipeline, caption_mask
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pipe = SRPipeline(41, device="cuda", dtype="bf16", branch_num=4)
cap, lens, LR, LRb = synthetic_batch(B, seed=100)
cap, LR, LRb, lens = cap.cuda(), LR.cuda(), LRb.cuda(), lens.tolist()
with torch.no_grad():
    ref = pipe(cap, lens, LR, LRb)["fine"][2].clone()
    pipe.capture(cap, lens, LR, LRb)
    def bench(fn, n=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e6
    print("one graph, two branches: %.1f us / step" % bench(lambda: pipe.replay()))
    ex = pipe._lp
    bufs = ex.alloc(B, 32, 32, LR.device)
    A, Bs = torch.cuda.Stream(), torch.cuda.Stream(priority=int(os.environ.get("TRUNK_PRIO", "0")))
    torch.cuda.synchronize()
    gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    hidden = pipe.text_encoder.init_hidden(B)
    with torch.cuda.stream(Bs):
        for _ in range(2):
            feats = ex.high_trunk(bufs, LR, LRb)
    with torch.cuda.stream(A):
        for _ in range(2):
            words, sent = pipe.text_encoder(cap, lens, hidden)
            mask = caption_mask(cap, words.size(2))
            low = ex.low(bufs, LR, sent, words, mask)
    torch.cuda.synchronize()
    with torch.cuda.graph(gB, stream=Bs):
        feats = ex.high_trunk(bufs, LR, LRb)
    with torch.cuda.graph(gA, stream=A):
        words, sent = pipe.text_encoder(cap, lens, hidden)
        mask = caption_mask(cap, words.size(2))
        low = ex.low(bufs, LR, sent, words, mask)
    torch.cuda.synchronize()
    def split():
        with torch.cuda.stream(Bs):
            gB.replay()
        with torch.cuda.stream(A):
            gA.replay()
            A.wait_stream(Bs)
            out = ex.high_heads(feats, low[0])
        return out
    out = split()
    torch.cuda.synchronize()
    print("max |split - reference| on the finest image: %g" % float((out[2] - ref).abs().max()))
    print("two graphs on two streams + eager heads: %.1f us / step" % bench(split))
    def split_eager_trunk():
        with torch.cuda.stream(Bs):
            f = ex.high_trunk(bufs, LR, LRb)
        with torch.cuda.stream(A):
            gA.replay()
            A.wait_stream(Bs)
            return ex.high_heads(f, low[0])
    out = split_eager_trunk()
    torch.cuda.synchronize()
    print("max |eager-trunk split - reference|: %g" % float((out[2] - ref).abs().max()))
    print("low graph on one stream + eager trunk on the other + eager heads: %.1f us / step" % bench(split_eager_trunk))
    def gl_only():
        with torch.cuda.stream(A):
            gA.replay()
    print("the text encoder + G_SR_NET_low graph alone: %.1f us" % bench(gl_only))
    def tr_only():
        with torch.cuda.stream(Bs):
            gB.replay()
    print("the trunk graph alone: %.1f us" % bench(tr_only))
