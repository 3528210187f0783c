#!/usr/bin/env python3
"""Headline benchmark: text-conditioned x8 super-resolution, images / second.

One step = one pass of the SR hot path (BiLSTM text encoder -> G_SR_NET_low -> NetG_highweight, forward, eval-mode
BN, fp32) over one batch of 16 synthetic CelebA-shaped samples (32x32 -> 256x256) that is already resident in HBM
(BASELINE.json configs[1]).  With N GPUs every rank runs the same per-GPU batch on its own images (weak scaling,
independent images, no data-path collective - SURVEY.md 8e); `value` = all images processed / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 16] [--dtype fp32|bf16|f16] [--graph] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
`--gpus N` (N > 1) without a torchrun environment starts that command itself, as a child process, before anything
touches the GPU, and relays its JSON line.  `--dtype bf16|f16` runs BASELINE configs[4]'s reduced-precision path
(fp32 stays the parity configuration and the default).

Prints ONE JSON line (rank 0) carrying `roofline` (dominant 3x3-conv kernel by time, per-launch HIP-event
timing inside the timed region) and `cpu_baseline` (the CPU oracle timed on this host's cores, rank 0, N=1 only).
"""
import argparse
import csv
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)
PEAK_LP_MFMA_TFLOPS = 2500.0    # dense bf16 / f16 MFMA peak (same guide; AMD's 5 PF figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0
# share of the direct-form multiplies a kernel actually issues (Winograd F(2x2,3x3): 16/36, F(4x4,3x3): 36/144; sub-pixel upBlock: 4/9;
# up-sample-aware Winograd: 9 of the 16 positions of a 2x2 output tile = 9/36)
EXECUTED_MAC_FRACTION = {"conv3x3_mfma_kernel": 1.0, "wino_conv3x3_kernel": 16.0 / 36.0, "wino4_conv3x3_kernel": 36.0 / 144.0, "wino4w_conv3x3_kernel": 36.0 / 144.0,
                         "upconv_glu_mfma_kernel": 4.0 / 9.0,
                         "upwino_glu_kernel": 9.0 / 36.0, "upwino4_kernel": 25.0 / 144.0, "lp_conv3x3_kernel": 1.0, "lp_upconv_glu_kernel": 4.0 / 9.0,
                         # training: weight gradients in the Winograd domain (16 | 9 positions per 2x2 outputs), the direct
                         # 9-tap form, the image heads and the discriminators' implicit GEMM (every MAC issued)
                         "wino_wgrad_kernel": 16.0 / 36.0, "upwino_wgrad_kernel": 9.0 / 36.0, "conv3x3_wgrad_kernel": 1.0,
                         "dconv_igemm_kernel": 1.0,
                         # the same GEMMs on the bf16 pipe with three-piece operands (profiles/HISTORY.md 3.18): six bf16 MFMAs (6 x 32 cycles
                         # per 32 x 32 x 16) where the fp32 form issues eight fp32 MFMAs (8 x 64 cycles): 0.375 of its matrix-pipe time
                         "dconv_igemm6_kernel": 6.0 * 32 / (8.0 * 64)}
# bench name of a kernel -> prefix of its name in the rocprofv3 tables under profiles/
PMC_NAME = {"upwino_glu_kernel": "upwino_kernel"}


def pmc_table(dtype):
    """Per-kernel counter averages of the newest `profiles/r*_<dtype>_pmc.csv` (tools/profile_pmc.sh: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in separate passes over `bench.py --serial`, FETCH_SIZE doubled
    per MI355X_MICROARCH.md).  Returns (file name, {kernel name: row dict}); nothing is hard-coded in this file."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.csv" % dtype)))
    if not files:
        return None, {}
    rows = {}
    with open(files[-1]) as f:
        for r in csv.DictReader(line for line in f if not line.startswith("#")):
            rows[r["kernel"]] = r
    return os.path.basename(files[-1]), rows


def pmc_lookup(rows, name):
    """Launch-weighted HBM MB per launch and time-weighted MFMA-busy fraction over the template instances of `name`."""
    pre = PMC_NAME.get(name, name)
    hit = [r for k, r in rows.items() if k.startswith(pre)]
    if not hit:
        return None, None
    n = sum(float(r["launches"]) for r in hit)
    mb = sum(float(r["launches"]) * float(r["avg_HBM_MB_per_launch"]) for r in hit) / n
    busy = None
    if "mfma_busy_frac_of_peak" in hit[0]:
        t = sum(float(r["launches"]) * float(r["avg_us"]) for r in hit)
        busy = sum(float(r["launches"]) * float(r["avg_us"]) * float(r["mfma_busy_frac_of_peak"]) for r in hit) / t
    return mb * 1e6, busy


def maybe_spawn(args):
    """`--gpus N` outside a torchrun environment: launch the N-rank job as a CHILD process (never an exec: nothing in
    this process has touched the GPU yet, and nothing will) and exit with its code."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def batch_pool(B, rank, dev, n=8):
    """`n` different synthetic batches (captions, caption LENGTHS, images: seeds 100 + rank + 1000 i), resident in HBM
    before any timed region starts.  Every timed step runs on the next batch of the pool - a hipGraph replay copies it
    into the graph's static inputs inside the timed loop (one copy launch) - so no figure is the rate of one length vector
    (Q7: T_max and the mask follow each batch, util.py:250-253, trainer_objective.py:136-140)."""
    from tgsr_amd.synthetic import synthetic_batch
    pool = []
    for i in range(n):
        cap, lens, LR, LRb = synthetic_batch(B, seed=100 + rank + 1000 * i)
        pool.append({"cap": cap.to(dev), "lens": lens.tolist(), "lens_dev": lens.to(torch.int32).to(dev),
                     "LR": LR.to(dev), "LRb": LRb.to(dev), "T": int(lens.max())})
    return pool


def replay_on(step, pool, k, lanes=1):
    """Replay a captured step (SRPipeline / GraphedStep) on the pool's next batch(es): new captions, lengths and images
    go into the static inputs by the replay's one copy launch; lengths stay on the device (nothing crosses PCIe)."""
    if lanes == 1:
        b = pool[k % len(pool)]
        return step.replay(b["cap"], b["lens_dev"], b["LR"], b["LRb"], num_words=b["T"])
    bs = [pool[(k * lanes + j) % len(pool)] for j in range(lanes)]
    return step.replay([b["cap"] for b in bs], [b["lens_dev"] for b in bs], [b["LR"] for b in bs],
                       [b["LRb"] for b in bs], num_words=[b["T"] for b in bs])


def load_weights():
    """Shipped x8 face checkpoint (committed as a data fixture) when present, else None -> seeded random init."""
    p = os.path.join(ROOT, "tests", "golden", "face_S8_weights.npz")
    if not os.path.exists(p):
        return None
    z = np.load(p)
    out = {"E.": {}, "GL.": {}, "GH.": {}}
    for k in z.files:
        for pre in out:
            if k.startswith(pre):
                out[pre][k[len(pre):]] = torch.from_numpy(z[k])
    return out


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box hands one
    GPU a share of the host's cores; running oneDNN on every visible core would only oversubscribe it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    try:
        n = min(n, int(os.environ.get("TGSR_CPU_CORES", n)))
    except ValueError:
        pass
    return max(1, min(n, 64))


def cpu_baseline(weights, batch, budget_s=20.0, x16_pipe=None):
    """The oracle (CPU restatement of the reference, plain PyTorch/oneDNN) on this host's cores, bounded sample.
    x16_pipe: the x16 SRPipeline whose (random-init) parameters the x16 oracle runs on (--branch-num != 4)."""
    from oracle import tgsr_oracle as O
    torch.set_num_threads(usable_cores())
    name = "oracle.sr_forward"
    if x16_pipe is not None:
        from oracle import tgsr_oracle_lp as OL
        cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}          # noqa: E731
        sdE, sdL, sdH = cpu(x16_pipe.text_encoder), cpu(x16_pipe.netGL), cpu(x16_pipe.netGH)
        fwd = lambda *a: OL.sr_forward16(*a, dtype=None)                                   # noqa: E731  (None = fp32 oracle)
        name, batch = "oracle sr_forward16 (models16)", min(batch, 4)
    elif weights is None:
        sdE, sdL, sdH = O.random_state(seed=0)
        fwd = O.sr_forward
    else:
        sdE, sdL, sdH = weights["E."], weights["GL."], weights["GH."]
        fwd = O.sr_forward
    cap, lens, LR, LRb = O.synthetic_batch(batch)
    with torch.no_grad():
        fwd(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)       # warm-up
        ts = []
        t_all = time.perf_counter()
        while len(ts) < 5 and (time.perf_counter() - t_all) < budget_s:
            t0 = time.perf_counter()
            fwd(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)
            ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": round(batch / med, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s (CPU PyTorch restatement, fp32, eval BN), batch %d, warm-up 1 + median of %d "
                      "runs, %.2f s/batch" % (name, batch, len(ts), med)}


CONV_GFLOP_PER_IMAGE = 20.5     # direct-form FLOPs of the 36 conv3x3 launches of one forward (profiles/HISTORY.md section 3.1)


def cpu_baseline_train(weights, batch, budget_s=12.0):
    """Generator train step on the CPU: the oracle's forward in train-mode BatchNorm + MSE + KL through torch autograd
    (what the reference would run: plain PyTorch ops), bounded sample of `batch` images."""
    from oracle import tgsr_oracle as O
    torch.set_num_threads(usable_cores())
    sdE, sdL, sdH = O.random_state(seed=0) if weights is None else (weights["E."], weights["GL."], weights["GH."])
    sdL = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sdL.items()}
    sdH = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sdH.items()
           if k != "a"}
    cap, lens, LR, LRb = O.synthetic_batch(batch)
    g = torch.Generator().manual_seed(7)
    hr = [torch.rand(batch, 3, s, s, generator=g) * 2 - 1 for s in (64, 128, 256)]
    with torch.no_grad():
        words, sent = O.rnn_encoder(sdE, cap, lens.tolist())
    mask = (cap == 0)[:, :words.shape[2]]

    def one():
        imgs, _att, mu, logvar = O.g_sr_net_low(sdL, LR, sent, words, mask, training=True)
        fine, _a, _one = O.netg_highweight(sdH, LR, imgs, LRb, "lr", training=True)
        loss = O.mse(imgs, hr) + O.mse(fine, hr) + O.kl_loss(mu, logvar)
        loss.backward()

    one()
    ts, t_all = [], time.perf_counter()
    while len(ts) < 3 and (not ts or (time.perf_counter() - t_all) < budget_s):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": round(batch / med, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle generators in train-mode BN, MSE + KL, torch autograd backward (fp32), batch %d (the configuration's "
                      "own), warm-up 1 + median of %d run(s) within %.0f s, %.2f s/step" % (batch, len(ts), budget_s, med)}


def bench_damsm(args, rank, world, dist, dev):
    """pretrain_DAMSM.py step (RNN_ENCODER + CNN_ENCODER heads on words_loss + sent_loss, Adam, grad clip) on synthetic
    Inception-trunk outputs: every gradient from HIP kernels (DAMSM backward, LSTM BPTT, head GEMMs)."""
    from tgsr_amd.synthetic import synthetic_batch
    from tgsr_amd.train import DAMSMTrainer
    cfg_ = __import__("tgsr_amd.miscc.config", fromlist=["cfg"]).cfg
    cfg_.TRAIN.FLAG = True
    tr = DAMSMTrainer(41, device=dev)
    B = args.batch
    cap, lens, _LR, _LRb = synthetic_batch(B, seed=100 + rank)
    g = torch.Generator().manual_seed(3 + rank)
    feats, pooled = torch.randn(B, 768, 17, 17, generator=g).to(dev), torch.randn(B, 2048, generator=g).to(dev)
    cap, lens = cap.to(dev), lens.tolist()
    for _ in range(args.warmup):
        tr.step_features(feats, pooled, cap, lens)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step_features(feats, pooled, cap, lens)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "DAMSM pre-training caption-image pairs/sec (batch 16 per GPU, fwd+bwd+Adam)",
            "value": round(world * B * args.steps / dt, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic captions and Inception-trunk outputs (the trunk is third-party, frozen)",
            "config": {"workload": "pretrain_DAMSM.py step: RNN_ENCODER.train() + CNN_ENCODER heads, words_loss + "
                                   "sent_loss, Adam, grad-clip 0.25, batch=16 per GPU", "batch_per_gpu": B,
                       "parallelism": "dp%d" % world},
            "final_loss": round(float(loss), 5), "roofline": None}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_train(args, rank, world, dist, dev, weights):
    """`--mode train`: the generator training step (text-enc frozen + G_SR_NET_low + NetG_highweight forward in train-mode
    BN, MSE + KL, HIP backward, one flat-bucket gradient all-reduce (N > 1), Adam, EMA), or with `--gan` the G/D
    alternation with three discriminators.  Synthetic HR targets U(-1, 1).  The roofline object prices the step by the MACs
    its convolution kernels actually execute (train_object), and both forms carry a CPU baseline."""
    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    e = _train_run(args, rank, world, dist, dev, weights, fence, args.gan, args.steps, args.warmup)
    if rank == 0:
        if "error" in e:
            raise SystemExit("train bench failed: " + e["error"])
        B = args.batch
        print(json.dumps({
            "metric": ("SR G/D alternation train images/sec (32->256, batch 16 per GPU, 3 discriminators + generators, Adam)"
                       if args.gan else "SR generator train images/sec (32->256, batch 16 per GPU, fwd+bwd+Adam)"),
            "value": e["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic inputs and HR targets",
            "config": {"workload": "CelebA x8 " + e["workload"] + ", batch=%d per GPU" % B, "batch_per_gpu": B,
                       "parallelism": "dp%d" % world, "grad_bucket_MB": e["grad_bucket_MB"]},
            "final_loss": e["final_loss"], "roofline": e.get("roofline"), "launch": e.get("launch"),
            "graph_policy": e.get("graph_policy"), "device_time": e.get("device_time"),
            **({"cpu_baseline": e["cpu_baseline"]} if "cpu_baseline" in e else {})}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def roofline_objects(agg, nprof, dtype, serial, steps, batch=16, counters=True):
    """`roofline` (dominant conv kernel), per-kernel table and the attention object from the HIP events recorded around
    every launch of the sampled steps.  Nothing here is a literal: HBM traffic and the MFMA-busy share come from the
    newest rocprofv3 table under profiles/ for this dtype (pmc_table), looked up by kernel name."""
    lp = dtype != "fp32"
    peak_mfma = PEAK_LP_MFMA_TFLOPS if lp else PEAK_FP32_MFMA_TFLOPS
    # (counters=False: the committed counter tables were taken on the x8 generators' layer mix - not applied to the x16 line)
    pmc_file, pmc = pmc_table(dtype) if counters else (None, {})
    conv_kernels = [k for k in ("lp_conv3x3_kernel", "lp_upconv_glu_kernel", "wino4w_conv3x3_kernel", "wino4_conv3x3_kernel", "wino_conv3x3_kernel", "upwino_glu_kernel", "upwino4_kernel",
                                "conv3x3_mfma_kernel",
                                "upconv_glu_mfma_kernel") if k in agg]
    dom = max(conv_kernels, key=lambda k: agg[k][3])
    n, fl, by, sec = agg[dom]
    alg_tf, alg_gbs = fl / sec / 1e12, by / sec / 1e9
    exe_tf = alg_tf * EXECUTED_MAC_FRACTION[dom]               # what the MFMA pipe really issued
    traffic, busy = pmc_lookup(pmc, dom)
    # the committed counter tables are taken at the default batch of 16: HBM bytes per launch scale with the batch
    bscale = batch / 16.0
    traffic = None if traffic is None else traffic * bscale
    mfma_frac, hbm_frac = exe_tf / peak_mfma, alg_gbs / PEAK_HBM_GBS
    # the roofline that binds = the one the kernel sits closer to; the 2-byte path is priced against HBM (SURVEY.md 8d:
    # "HBM roofline in bf16"), with the bytes the counters saw when a PMC table for this kernel is committed
    cnt_gbs = None if traffic is None else traffic * n / sec / 1e9
    if lp:
        use = cnt_gbs if cnt_gbs is not None else alg_gbs
        roof = {"bound": "hbm", "achieved": round(use, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(use / PEAK_HBM_GBS, 4), "bytes_from": "counters" if cnt_gbs is not None else "algorithmic"}
    elif mfma_frac >= hbm_frac:
        roof = {"bound": "mfma", "achieved": round(exe_tf, 2), "peak": peak_mfma, "unit": "TFLOP/s",
                "frac": round(mfma_frac, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(alg_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(hbm_frac, 4)}
    roof.update({
        "kernel": dom, "traffic": traffic, "launches_per_step": n // nprof, "avg_launch_us": round(sec / n * 1e6, 2),
        "flop_per_launch": fl / n, "algorithmic_bytes_per_launch": by / n,
        "achieved_algorithmic_TFLOPs": round(alg_tf, 2),          # the reference's direct-form FLOP count / time
        "executed_mac_fraction": round(EXECUTED_MAC_FRACTION[dom], 4),
        "mfma_frac": round(mfma_frac, 4), "hbm_frac_algorithmic": round(hbm_frac, 4),
        "hbm_frac_counter": None if traffic is None else round(traffic * n / sec / 1e9 / PEAK_HBM_GBS, 4),
        "traffic_over_algorithmic": None if traffic is None else round(traffic / (by / n), 3),
        "mfma_busy_frac_measured": None if busy is None else round(busy, 4),
        "counters_from": pmc_file, "counters_batch": 16,
        "timing": "HIP events around every launch of %d of the %d timed steps; those steps run single-stream so each "
                  "launch is timed alone%s" % (nprof, steps, "" if not serial else " (--serial: every step does)")})
    cp = [agg[k] for k in conv_kernels]
    cfl, csec, cby = sum(v[1] for v in cp), sum(v[3] for v in cp), sum(v[2] for v in cp)
    cex = sum(agg[k][1] * EXECUTED_MAC_FRACTION[k] for k in conv_kernels)
    ctr = [(None if pmc_lookup(pmc, k)[0] is None else pmc_lookup(pmc, k)[0] * bscale, agg[k][0]) for k in conv_kernels]
    roof["conv_path"] = {"kernels": conv_kernels, "ms_per_step": round(csec / nprof * 1e3, 4),
                         "achieved_algorithmic_TFLOPs": round(cfl / csec / 1e12, 2),
                         "mfma_frac": round(cex / csec / 1e12 / peak_mfma, 4),
                         "hbm_GBs_algorithmic": round(cby / csec / 1e9, 1),
                         "hbm_frac_algorithmic": round(cby / csec / 1e9 / PEAK_HBM_GBS, 4),
                         "hbm_frac_counter": (None if any(t is None for t, _ in ctr) else
                                              round(sum(t * c for t, c in ctr) / csec / 1e9 / PEAK_HBM_GBS, 4))}
    kern = {k: {"launches_per_step": v[0] // nprof, "ms_per_step": round(v[3] / nprof * 1e3, 4),
                "TFLOPs": round(v[1] / v[3] / 1e12, 2), "GBs_algorithmic": round(v[2] / v[3] / 1e9, 1)}
            for k, v in agg.items() if v[3] > 0}
    if "lp_attention_fused" in agg:      # zero-duration marker: the attention runs inside the kernels that produce h
        v = agg["lp_attention_fused"]
        kern["lp_attention_fused"] = {"launches_per_step": 0, "fused_into": ["lp_stem_kernel", "lp_upconv_glu_kernel"],
                                      "stages_per_step": v[0] // nprof, "GFLOP_per_step": round(v[1] / nprof / 1e9, 3),
                                      "MB_per_step": round(v[2] / nprof / 1e6, 2),
                                      "note": "no launch of its own: its time is part of the producers' (h comes from LDS)"}
    att = None
    ak = "lp_word_attention_kernel" if lp else "word_attention_kernel"
    if ak in agg:
        # BASELINE.json's metric also asks for the attention batched-GEMM's MFMA utilisation: the op is HBM-bound
        # (AI ~ 7 FLOP/B), so both fractions are reported (flops = 4*B*Q*idf*T for the two GEMMs)
        n, fl, by, sec = agg[ak]
        tr, bz = pmc_lookup(pmc, ak)
        tr = None if tr is None else tr * bscale
        att = {"kernel": ak, "bound": "hbm", "launches_per_step": n // nprof, "ms_per_step": round(sec / nprof * 1e3, 4),
               "mfma_util_pct": round(fl / sec / 1e12 / peak_mfma * 100, 2),
               "mfma_busy_pct_measured": None if bz is None else round(bz * 100, 2),
               "hbm_GBs": round(by / sec / 1e9, 1), "hbm_frac": round(by / sec / 1e9 / PEAK_HBM_GBS, 4),
               "hbm_frac_counter": None if tr is None else round(tr * n / sec / 1e9 / PEAK_HBM_GBS, 4)}
    return roof, kern, att


# ------------------------------------------------------------------------------------------------ driver-timed extras
def _all_ok(ok, dist, dev):
    """Every rank agrees whether a section's set-up worked (a rank that failed must not leave the others in a barrier)."""
    if dist is None:
        return ok
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def _max_over_ranks(dt, dist, dev):
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ranks_object(rank, world, dist, dev, ts_local=None):
    """What the collective layer says about a multi-rank run (every rank calls it; rank 0 gets the dict): backend, one all-reduce
    of ones through the process group (= world on every rank), the devices the ranks sit on (all distinct for RCCL), and - from
    `ts_local`, this rank's own region times - the per-rank step time spread (the headline takes the MAX over ranks)."""
    if dist is None:
        return {"world": 1, "backend": None}
    be = dist.get_backend()
    cdev = dev if be == "nccl" else "cpu"
    out = {"world": world, "backend": be}
    x = torch.ones(1 << 16, dtype=torch.float32, device=cdev)
    dist.all_reduce(x)
    out["all_reduce_of_ones"] = float(x[0].item())
    try:
        ident = str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")) or "index %d" % dev.index
    except Exception:          # noqa: BLE001
        ident = "index %d" % dev.index
    box = [None] * world
    dist.all_gather_object(box, "cuda:%d %s" % (dev.index, ident))
    out["devices"], out["distinct_devices"] = box, len(set(box))
    if ts_local:
        t = torch.tensor([float(np.median(ts_local))], dtype=torch.float64, device=cdev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per = [float(v.item()) for v in allt]
        out["per_rank_region_s"] = [round(v, 6) for v in per]
        out["per_rank_spread"] = round(max(per) / min(per), 4) if min(per) > 0 else None
    return out


def rccl_direct_evidence(world, dist, dev, timeout_s=60.0):
    """The library's OWN RCCL communicator (parallel.RcclDirect: librccl opened by libtgsr_hip.so, the id broadcast once through
    the process group): ncclCommCount / ncclCommUserRank as RCCL reports them and one tgsr_allreduce_flat of ones (= world).  Run
    by every rank at the very END of the bench, on a watchdog thread: a second communicator beside torch's cannot be rehearsed
    on a one-GPU box (RCCL refuses two ranks on one device), so if it does not return within `timeout_s` the line is printed
    without it - the measurements are complete by then."""
    if dist is None or dist.get_backend() != "nccl" or os.environ.get("TGSR_BENCH_RCCL_DIRECT", "1") == "0":
        return None
    import threading
    out = {}

    def work():
        try:
            torch.cuda.set_device(dev)
            from tgsr_amd import ops as _ops
            from tgsr_amd import parallel
            rc = parallel.RcclDirect.create()
            out["rccl_ranks"], out["rccl_rank"] = rc.count()
            y = torch.ones(1 << 16, dtype=torch.float32, device=dev)
            _ops.allreduce_flat(rc.comm, y, 1.0)
            torch.cuda.synchronize()
            out["allreduce_flat_of_ones"] = float(y[0].item())
            rc.close()
        except Exception as e:          # noqa: BLE001
            out["error"] = "%s: %s" % (type(e).__name__, e)
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return {"error": "no answer within %.0f s" % timeout_s, "hung": True}
    return out


def time_regions(run_k_steps, fence, reps, dist, dev, local=None):
    """`reps` timed regions of exactly K steps each (run_k_steps(r) issues the K steps of region r), every one bracketed by
    barrier + synchronize on both sides; per region the MAX over ranks; returns (median, all of them).  One K-step region of the
    inference path is 10-30 ms - run-to-run +-4 % on one such region hid real changes - so the reported time per K
    steps is the median over enough regions for >= 0.5 s of device time."""
    ts = []
    for r in range(reps):
        fence()
        t0 = time.perf_counter()
        run_k_steps(r)
        fence()
        ts.append(time.perf_counter() - t0)
    if local is not None:
        local[:] = ts                      # this rank's own times (ranks_object reports the spread)
    if dist is not None:
        t = torch.tensor(ts, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ts = [float(x) for x in t.tolist()]
    return float(np.median(ts)), ts


def reps_for(est_region_s, args, lo=1, hi=40):
    """Number of K-step regions for >= args.min_time seconds of timed device work (the same on every rank: from rank 0's estimate)."""
    if args.repeats > 0:
        return args.repeats
    return int(max(lo, min(hi, np.ceil(args.min_time / max(est_region_s, 1e-6)))))


def _bcast_int(n, dist, dev):
    if dist is None:
        return n
    t = torch.tensor([n], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.broadcast(t, 0)
    return int(t.item())


def _psnr(a, b, peak=2.0):
    """10 log10(peak^2 / mse) of two device tensors (peak = the [-1, 1] image range, as oracle.tgsr_oracle_lp.psnr)."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * float(np.log10(peak * peak / mse))


def lp_object(args, rank, world, dist, dev, weights, fp32_pipe, fence):
    """BASELINE configs[4] inside the default run: the reduced-precision path replayed from a hipGraph, timed in this
    process right after the fp32 region with the same K / W and the same fencing (barrier + synchronize, max over ranks).
    One entry per (storage type, per-GPU batch): bf16 and f16 at batch 16, bf16 at batch 8 (configs[4]'s per-GPU batch:
    64 over 8 GPUs) and at the saturating batch of 128; with N > 1 only the configs[4] entry.  Each entry carries the
    whole-step HBM fraction on SURVEY 8d's compulsory bytes, the conv kernel's HBM (counter bytes) and executed-MFMA
    fractions from HIP events around every launch of one eager step, and the PSNR of the finest image against the fp32
    path of THIS run on the same inputs."""
    from tgsr_amd import ops
    from tgsr_amd.synthetic import synthetic_batch
    from tgsr_amd.trainer import SRPipeline
    plan = [("bf16", 16), ("f16", 16), ("bf16", 8), ("bf16", 128)] if world == 1 else [("bf16", 8)]
    out = {"launch": "hipgraph", "steps": args.steps, "warmup": args.warmup,
           "psnr_reference": "fp32 path of this run, same inputs and weights (finest 256^2 image, peak 2.0); the 2-byte "
                             "kernels themselves are checked against a CPU model of the same rounding points "
                             "(oracle/tgsr_oracle_lp.py, derived from the pinned fp32 oracle) in tests/test_hip_lp.py",
           "runs": []}
    for dtype, B in plan:
        entry = {"dtype": dtype, "batch_per_gpu": B}
        pipe = None
        try:
            pipe = SRPipeline(41, device=dev, low="lr", overlap=True, dtype=dtype)
            if weights is not None:
                pipe.load_state_dicts(weights["E."], weights["GL."], weights["GH."])
            else:
                pipe.netGL.load_state_dict(fp32_pipe.netGL.state_dict())
                pipe.netGH.load_state_dict(fp32_pipe.netGH.state_dict())
                pipe.text_encoder.load_state_dict(fp32_pipe.text_encoder.state_dict())
            pool = batch_pool(B, rank, dev)
            cap, lens, LR, LRb = (pool[0][k] for k in ("cap", "lens", "LR", "LRb"))
            ref = fp32_pipe(cap, lens, LR, LRb)["fine"][2]
            got = pipe(cap, lens, LR, LRb)["fine"][2]
            torch.cuda.synchronize()
            entry["psnr_vs_fp32_dB"] = round(_psnr(got, ref), 2)
            del ref, got
            pipe.capture(cap, lens, LR, LRb)
            for k in range(args.warmup):
                replay_on(pipe, pool, k)
            # the captured step on a batch with OTHER caption lengths == the eager step on that batch, bit for bit
            b = pool[3]
            g = replay_on(pipe, pool, 3)["fine"][2].clone()
            e = pipe(b["cap"], b["lens"], b["LR"], b["LRb"])["fine"][2]
            entry["replay_equals_eager_on_new_lengths"] = bool(torch.equal(g, e))
            del g, e
            ok = True
        except Exception as e:          # noqa: BLE001 - a failed extra must not take the headline line down
            entry["error"] = "%s: %s" % (type(e).__name__, e)
            ok = False
        if not _all_ok(ok, dist, dev):
            entry.setdefault("error", "set-up failed on another rank")
            out["runs"].append(entry)
            continue
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            replay_on(pipe, pool, k)
        fence()
        reps = _bcast_int(reps_for(time.perf_counter() - t0, args), dist, dev)

        def region(r, pipe=pipe, pool=pool):
            for k in range(args.steps):
                replay_on(pipe, pool, r * args.steps + k)   # a different batch (captions, lengths, images) every step
        dt, _ts = time_regions(region, fence, reps, dist, dev)
        entry["repeats"] = reps
        ips = world * B * args.steps / dt
        mb_img = 173.9 / 2
        entry.update({"value": round(ips, 2), "unit": "images/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
                      "step_roofline": {"compulsory_MB_per_image": mb_img,
                                        "hbm_frac": round(ips / world * mb_img * 1e6 / (PEAK_HBM_GBS * 1e9), 4),
                                        "algorithmic_TFLOPs": round(ips / world * 21.07e9 / 1e12, 1)}})
        if B <= 16 and args.steps % 4 == 0:
            # throughput form: four independent batches as parallel branches of ONE graph (what the fp32 headline uses)
            try:
                from tgsr_amd.trainer import GraphedStep
                multi = GraphedStep(pipe, cap, lens, LR, LRb, lanes=4)
                for k in range(max(1, args.warmup // 4)):
                    replay_on(multi, pool, k, 4)
                ok4 = True
            except Exception as e:      # noqa: BLE001
                entry["graph_lanes4"] = {"error": "%s: %s" % (type(e).__name__, e)}
                ok4 = False
            if _all_ok(ok4, dist, dev):
                def region4(r, multi=multi, pool=pool):
                    for k in range(args.steps // 4):
                        replay_on(multi, pool, r * (args.steps // 4) + k, 4)  # four different batches per replay, other ones every replay
                dt4, _ts = time_regions(region4, fence, reps, dist, dev)
                ips4 = world * B * args.steps / dt4
                entry["graph_lanes4"] = {"value": round(ips4, 2), "ms_per_step": round(dt4 / args.steps * 1e3, 4),
                                         "hbm_frac": round(ips4 / world * mb_img * 1e6 / (PEAK_HBM_GBS * 1e9), 4)}
            multi = None
        if rank == 0:
            try:                         # one eager single-stream step with HIP events around every launch
                prof = []
                pipe.overlap = False
                ops.profile = prof
                pipe(cap, lens, LR, LRb)
                ops.profile = None
                torch.cuda.synchronize()
                agg = {}
                for name, flops, nbytes, e0, e1 in prof:
                    a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                    a[0] += 1
                    a[1] += flops
                    a[2] += nbytes
                    a[3] += e0.elapsed_time(e1) * 1e-3
                roof, _kern, att = roofline_objects(agg, 1, dtype, True, 1, B)
                entry["conv_kernel"] = {k: roof[k] for k in ("kernel", "bound", "frac", "achieved", "unit", "bytes_from", "mfma_frac",
                                                             "hbm_frac_algorithmic", "hbm_frac_counter", "avg_launch_us",
                                                             "launches_per_step", "counters_from") if k in roof}
                entry["conv_path"] = roof["conv_path"]
                if att is None and pipe._lp.fuse_attention:
                    # the attention runs inside the producers of h: one more step with the stand-alone launches (same bits)
                    prof2 = []
                    pipe._lp.fuse_attention, ops.profile = False, prof2
                    try:
                        pipe(cap, lens, LR, LRb)
                        torch.cuda.synchronize()
                    finally:
                        pipe._lp.fuse_attention, ops.profile = True, None
                    agg2 = {}
                    for name, flops, nbytes, e0, e1 in prof2:
                        a = agg2.setdefault(name, [0, 0.0, 0.0, 0.0])
                        a[0] += 1
                        a[1] += flops
                        a[2] += nbytes
                        a[3] += e0.elapsed_time(e1) * 1e-3
                    att = roofline_objects(agg2, 1, dtype, True, 1, B)[2]
                    if att is not None:
                        att["measured_with"] = "stand-alone attention launches (one extra step); the timed steps fuse it into the producers of h"
                if att is not None:
                    entry["attention"] = att
            except Exception as e:      # noqa: BLE001
                entry["conv_kernel"] = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                ops.profile = None
        out["runs"].append(entry)
        del pipe
        torch.cuda.empty_cache()
    return out


def _cpu_image_encoder(nef):
    """CNN_ENCODER on the CPU for the baselines: the Inception-v3 topology (tests/inception_v3_arch.py, seeded random weights, frozen,
    eval mode) walked as util.py:308-362 does, then the two heads (conv1x1 768 -> nef, Linear 2048 -> nef, uniform(-0.1, 0.1))."""
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from inception_v3_arch import InceptionV3Arch
    m = InceptionV3Arch(seed=1).eval()
    for q in m.parameters():
        q.requires_grad = False
    g = torch.Generator().manual_seed(3)
    w_f = (torch.rand(nef, 768, 1, 1, generator=g) * 0.2 - 0.1)
    w_c, b_c = (torch.rand(nef, 2048, generator=g) * 0.2 - 0.1), torch.zeros(nef)

    def enc(x):
        x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
        x = m.Conv2d_2b_3x3(m.Conv2d_2a_3x3(m.Conv2d_1a_3x3(x)))
        x = F.max_pool2d(x, kernel_size=3, stride=2)
        x = m.Conv2d_4a_3x3(m.Conv2d_3b_1x1(x))
        x = F.max_pool2d(x, kernel_size=3, stride=2)
        x = m.Mixed_5d(m.Mixed_5c(m.Mixed_5b(x)))
        x = m.Mixed_6e(m.Mixed_6d(m.Mixed_6c(m.Mixed_6b(m.Mixed_6a(x)))))
        feat = x
        x = F.avg_pool2d(m.Mixed_7c(m.Mixed_7b(m.Mixed_7a(x))), kernel_size=8)
        return F.conv2d(feat, w_f), F.linear(x.view(x.size(0), -1), w_c, b_c)
    return enc


def cpu_baseline_gan(weights, batch, budget_s=20.0, encoder=False):
    """One G/D alternation on the CPU (oracle generators + the oracle's torch restatement of the build-declared
    discriminators, DF_DIM 64, torch autograd): the three discriminator losses backward, then the generator loss
    (adversarial + MSE + KL) backward - the arithmetic of SRTrainer.step_gan without the optimizer steps."""
    from oracle import tgsr_oracle as O
    from tgsr_amd import model
    torch.set_num_threads(usable_cores())
    sdE, sdL, sdH = O.random_state(seed=0) if weights is None else (weights["E."], weights["GL."], weights["GH."])
    req = lambda sd: {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v)   # noqa: E731
                      for k, v in sd.items()}
    sdL, sdH = req(sdL), req({k: v for k, v in sdH.items() if k != "a"})
    torch.manual_seed(0)
    sdD = [req({k: v.detach().clone() for k, v in d.state_dict().items()})
           for d in (model.D_NET64(), model.D_NET128(), model.D_NET256())]
    cap, lens, LR, LRb = O.synthetic_batch(batch)
    g = torch.Generator().manual_seed(7)
    hr = [torch.rand(batch, 3, s, s, generator=g) * 2 - 1 for s in (64, 128, 256)]
    with torch.no_grad():
        words, sent = O.rnn_encoder(sdE, cap, lens.tolist())
    mask = (cap == 0)[:, :words.shape[2]]
    rl, fl = torch.ones(batch), torch.zeros(batch)
    enc = _cpu_image_encoder(words.shape[1]) if encoder else None
    from tgsr_amd.miscc.config import cfg as _cfg
    sm = _cfg.TRAIN.SMOOTH

    def one():
        imgs, _att, mu, logvar = O.g_sr_net_low(sdL, LR, sent, words, mask, training=True)
        fine, _a, _one = O.netg_highweight(sdH, LR, imgs, LRb, "lr", training=True)
        for i, sd in enumerate(sdD):
            O.discriminator_loss(sd, hr[i], fine[i].detach(), sent, rl, fl).backward()
        adv = (O.generator_adv_loss(sdD, fine, sent, rl) if enc is None else
               O.generator_loss(sdD, enc, fine, rl, words, sent, torch.arange(batch), lens.tolist(), None, float(sm.GAMMA1),
                                float(sm.GAMMA2), float(sm.GAMMA3), float(sm.LAMBDA)))
        (adv + O.mse(imgs, hr) + O.mse(fine, hr) + O.kl_loss(mu, logvar)).backward()

    ts, t_all = [], time.perf_counter()
    while len(ts) < 3 and (not ts or (time.perf_counter() - t_all) < budget_s):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": round(batch / med, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle generators (train-mode BN) + the torch restatement of D_NET64/128/256 (DF_DIM 64): three "
                      "discriminator_loss backward passes, then generator_loss + MSE + KL backward%s, torch autograd (fp32), "
                      "batch %d (the configuration's own), median of %d run(s) within %.0f s, %.2f s/step (no warm-up run: the first "
                      "step is in the sample)" % (" - with the DAMSM ranking term through the Inception-v3 topology (random weights) and the two "
                                                  "heads" if encoder else "", batch, len(ts), budget_s, med)}


def _train_traffic(gan, conv):
    """HBM bytes per launch (PMC counters, gfx950 corrections: tools/pmc_summary.py) of the train step's dominant convolution
    kernel, from the newest `profiles/r*_train[_gan]_pmc.csv` (tools/profile_pmc.sh over `bench.py --mode train [--gan]`)."""
    if not conv:
        return {"traffic": None}
    dom = max(conv, key=lambda k: conv[k][2])
    fname, rows = pmc_table("train_gan" if gan else "train")
    pre = dom[:-len("_kernel")] if dom.endswith("_kernel") else dom      # wino_wgrad -> wino_wgrad_kernel<..>, wino_wgrad_dma_kernel<..>
    hit = [r for k, r in rows.items() if k.startswith(pre) and "reduce" not in k and "pack" not in k]
    if not hit:
        return {"traffic": None, "traffic_kernel": dom, "traffic_from": fname}
    n = sum(float(r["launches"]) for r in hit)
    mb = sum(float(r["launches"]) * float(r["avg_HBM_MB_per_launch"]) for r in hit) / n
    return {"traffic": round(mb * 1e6), "traffic_kernel": dom, "traffic_from": fname,
            "traffic_note": "launch-weighted mean over the instances of that kernel in the profiled step (batch 16)"}


def train_object(args, rank, world, dist, dev, weights, fence):
    """BASELINE configs[2] inside the default run: the generator train step (fwd + bwd + Adam + EMA on MSE + KL) and the
    G/D alternation (three discriminators), each timed with the same fencing as the headline and priced by the MACs its
    convolution kernels actually EXECUTE (per-launch direct-form FLOPs x the kernel's executed fraction - Winograd's
    saving is not credited), over the whole step time against the fp32 MFMA peak."""
    from tgsr_amd import ops
    from tgsr_amd.synthetic import synthetic_batch
    from tgsr_amd.train import SRTrainer
    from tgsr_amd.miscc.config import cfg
    B = args.batch
    out = {"batch_per_gpu": B, "runs": []}
    for gan, enc in ((False, False), (True, False), (True, True)):
        # runs[2] = BASELINE configs[2] as stated ("G+D+DAMSM loss"): generator_loss's ranking term through CNN_ENCODER
        out["runs"].append(_train_run(args, rank, world, dist, dev, weights, fence, gan, max(2, min(args.steps, 10)), 3, encoder=enc))
    return out


def _train_run(args, rank, world, dist, dev, weights, fence, gan, steps, warmup, encoder=None):
    """One timed train configuration (see train_object): returns its entry dict.  encoder: None = as `--damsm-encoder` says."""
    if encoder is None:
        encoder = bool(getattr(args, "damsm_encoder", False))
    from tgsr_amd import ops
    from tgsr_amd.synthetic import synthetic_batch
    from tgsr_amd.train import SRTrainer
    from tgsr_amd.miscc.config import cfg
    B = args.batch
    entry = {"workload": ("G/D alternation: 3 discriminator updates on (real, fake.detach()), then the generator update "
                          "through them + MSE + KL (losses.py:290-374; D_NET64/128/256 build-declared, DF_DIM %d)"
                          % cfg.GAN.DF_DIM) if gan else
             "generator train step: G_SR_NET_low + NetG_highweight fwd + bwd (train-mode BN), MSE + KL, Adam, EMA",
             "steps": steps, "warmup": warmup}
    tr = None
    try:
        enc = None
        if gan and encoder:
            # generator_loss's DAMSM ranking term (losses.py:375-389) through CNN_ENCODER's real walk.  The trained Inception-v3 is
            # third-party and absent here: the published TOPOLOGY with seeded random weights (tests/inception_v3_arch.py) runs in its
            # place - same layers, same sizes; what it costs, not what it computes, is the point.  In eval mode the walk runs on the
            # library's own kernels (tgsr_amd/inception.py, csrc/tgsr_igemm.hip; TGSR_TRUNK=torch: the torch modules on MIOpen).
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from inception_v3_arch import InceptionV3Arch
            from tgsr_amd.util import CNN_ENCODER
            # (the reference's trainer turns MIOpen's / cuDNN's algorithm search on, trainer_objective.py:32: so does this line)
            torch.backends.cudnn.benchmark = os.environ.get("TGSR_CUDNN_BENCHMARK", "1") != "0"
            enc = CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM, inception=InceptionV3Arch(seed=1)).to(dev).eval()
            for q in enc.parameters():
                q.requires_grad = False
            entry["workload"] += (" + the DAMSM ranking term through CNN_ENCODER (Inception-v3 topology, random weights; trunk on %s)"
                                  % ("torch modules / MIOpen" if os.environ.get("TGSR_TRUNK", "hip") == "torch" else
                                     "the library's own HIP kernels"))
        tr = SRTrainer(41, device=dev, discriminators=gan, image_encoder=enc)
        if weights is not None:
            tr.text_encoder.load_state_dict(weights["E."])
            tr.netGL.load_state_dict(weights["GL."])
            tr.netGH.load_state_dict({k: v for k, v in weights["GH."].items() if k != "a"})
        cap, lens, LR, LRb = synthetic_batch(B, seed=100 + rank)
        g = torch.Generator().manual_seed(7 + rank)
        hr = [(torch.rand(B, 3, s, s, generator=g) * 2 - 1).to(dev) for s in (64, 128, 256)]
        cap, LR, LRb, lens = cap.to(dev), LR.to(dev), LRb.to(dev), lens.tolist()
        from tgsr_amd.train import GRAPH_G_SETTLED
        for _ in range(max(warmup, GRAPH_G_SETTLED if tr._auto is not None else 0)):     # (TGSR_GRAPH_G=auto settles in the warm-up)
            tr.step(cap, lens, LR, LRb, hr)
        entry["graph_policy"] = dict(tr.graph_policy)
        # The train steps are issued by the host about as fast as the device runs them (profiles/HISTORY.md 3.18): whatever this process has
        # alive by now (the earlier objects of the default line, the trainer's modules and buffers) is garbage-collector work on
        # every allocation burst of a step.  Collect once and move the survivors out of the collector's sight - what a training
        # script does after its set-up (`gc.freeze()`); stated in the entry.
        import gc
        gc.collect()
        gc.freeze()
        entry["host"] = "gc.collect() + gc.freeze() after the warm-up steps"
        entry["launch"] = ("generators' half replayed from hipGraphs (fwd | loss + bwd [| all-reduce |] Adam + re-pack + EMA)" if tr._graph_g
                           else "generators' update eager") + (
            "; each discriminator's update replayed from its hipGraph%s" % ("s around its bucket's all-reduce" if world > 1 else "")
            if (tr._graph_d and tr._dstreams) else "")
        ok = True
    except Exception as e:          # noqa: BLE001
        entry["error"] = "%s: %s" % (type(e).__name__, e)
        ok = False
    if not _all_ok(ok, dist, dev):
        entry.setdefault("error", "set-up failed on another rank")
        return entry
    fence()
    t0 = time.perf_counter()
    tr.step(cap, lens, LR, LRb, hr)
    fence()
    reps = _bcast_int(reps_for((time.perf_counter() - t0) * steps, args, hi=10), dist, dev)
    loss = [float("nan")]

    def region(r):
        for _ in range(steps):
            loss[0] = tr.step(cap, lens, LR, LRb, hr)
    dt, _ts = time_regions(region, fence, reps, dist, dev)
    loss = loss[0]
    sec = dt / max(1, steps)
    entry.update({"value": round(world * B / sec, 2), "unit": "images/s", "ms_per_step": round(sec * 1e3, 4), "repeats": reps,
                  "final_loss": round(float(loss), 5), "grad_bucket_MB": round(tr.bucket.numel * 4 / 1e6, 2)})
    # one EAGER step under torch.profiler (roctracer does not report the nodes of a replayed hipGraph; the eager step issues the
    # same kernels): every kernel of the step as the device ran it, summed by family - the library's own kernels (`tgsr::`),
    # BatchNorm passes among them, everything else (aten / MIOpen / copies)
    dterr = None
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        keep = (tr._graph_g, tr._dsteps)
        tr._graph_g, tr._dsteps = False, -10 ** 9
        try:
            with profile(activities=[ProfilerActivity.CUDA]) as tp:
                tr.step(cap, lens, LR, LRb, hr)
                torch.cuda.synchronize()
        finally:
            tr._graph_g, tr._dsteps = keep
        fam = {"tgsr": [0, 0.0], "bn": [0, 0.0], "other": [0, 0.0]}
        top_other = {}
        for ev in tp.events():
            if ev.device_type != torch.autograd.DeviceType.CUDA:
                continue
            nm, us = ev.name, float(getattr(ev, "device_time", 0.0) or getattr(ev, "cuda_time", 0.0))
            own = "tgsr::" in nm
            f = fam["tgsr" if own else "other"]
            f[0] += 1
            f[1] += us
            if own and "::bn_" in nm:
                fam["bn"][0] += 1
                fam["bn"][1] += us
            if not own:
                t = top_other.setdefault(nm[:60], [0, 0.0])
                t[0] += 1
                t[1] += us
        tot = fam["tgsr"][1] + fam["other"][1]
        if rank == 0 and tot > 0:
            entry["device_time"] = {
                "kernels": fam["tgsr"][0] + fam["other"][0], "kernel_ms": round(tot / 1e3, 3),
                "tgsr_ms": round(fam["tgsr"][1] / 1e3, 3), "non_tgsr_ms": round(fam["other"][1] / 1e3, 3),
                "non_tgsr_share": round(fam["other"][1] / tot, 4), "non_tgsr_launches": fam["other"][0],
                "batchnorm_ms": round(fam["bn"][1] / 1e3, 3), "batchnorm_share": round(fam["bn"][1] / tot, 4),
                "batchnorm_launches": fam["bn"][0],
                "largest_non_tgsr": [{"name": k, "launches": v[0], "ms": round(v[1] / 1e3, 3)}
                                     for k, v in sorted(top_other.items(), key=lambda kv: -kv[1][1])[:4]],
                "note": "one eager step under torch.profiler (device activities only), summed kernel durations - streams overlap, "
                        "so kernel_ms exceeds the step time"}
    except Exception as e:          # noqa: BLE001
        dterr = "%s: %s" % (type(e).__name__, e)
    if not _all_ok(dterr is None, dist, dev) and rank == 0:
        entry["device_time"] = {"error": dterr or "failed on another rank"}
    # one more step with HIP events around every convolution launch (single stream).  EVERY rank takes it - the step
    # all-reduces the gradient bucket, a collective rank 0 alone would leave unmatched; only rank 0 records and reports.
    prof = []
    perr = None
    wside, dstreams, graph_g = tr._wside, tr._dstreams, tr._graph_g
    try:
        tr._wside, tr._dstreams, tr._graph_g = None, [], False      # (eager: a replayed graph has no per-launch events to record)
        ops.profile = prof if rank == 0 else None
        tr.step(cap, lens, LR, LRb, hr)
        torch.cuda.synchronize()
    except Exception as e:          # noqa: BLE001
        perr = "%s: %s" % (type(e).__name__, e)
    finally:
        ops.profile = None
        tr._wside, tr._dstreams, tr._graph_g = wside, dstreams, graph_g
    if not _all_ok(perr is None, dist, dev):       # a rank whose profiled step failed may have skipped a collective
        perr = perr or "the profiled step failed on another rank"
    if rank == 0 and perr is not None:
        entry["roofline"] = {"error": perr}
    elif rank == 0:
        try:
            agg = {}
            for name, flops, _nb, e0, e1 in prof:
                a = agg.setdefault(name, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += flops
                a[2] += e0.elapsed_time(e1) * 1e-3
            conv = {k: v for k, v in agg.items() if k in EXECUTED_MAC_FRACTION}
            alg = sum(v[1] for v in conv.values())
            exe = sum(v[1] * EXECUTED_MAC_FRACTION[k] for k, v in conv.items())
            ksec = sum(v[2] for v in conv.values())
            entry["roofline"] = {
                "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_FP32_MFMA_TFLOPS,
                "achieved": round(exe / sec / 1e12, 2), "frac": round(exe / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                "achieved_algorithmic_TFLOPs": round(alg / sec / 1e12, 2),
                "conv_kernels_only": {"ms_per_step": round(ksec * 1e3, 3),
                                      "executed_TFLOPs": round(exe / ksec / 1e12, 2),
                                      "frac": round(exe / ksec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
                "kernels": {k: {"launches": v[0], "ms": round(v[2] * 1e3, 3), "executed_fraction": round(EXECUTED_MAC_FRACTION[k], 4),
                                "executed_TFLOPs": round(v[1] * EXECUTED_MAC_FRACTION[k] / v[2] / 1e12, 2)}
                            for k, v in sorted(conv.items(), key=lambda kv: -kv[1][2])},
                **_train_traffic(gan, conv),
                "note": "`achieved` / `frac`: MACs the convolution kernels of forward, data gradient and weight gradient "
                        "really issue (direct-form FLOPs of every launch x the kernel's executed fraction; for dconv_igemm6_kernel, "
                        "which runs on the bf16 pipe, the fraction is its matrix-pipe time relative to the fp32 MFMA form: `frac` stays "
                        "the share of the matrix pipe's time at the spec clock), over the "
                        "WHOLE step time (BatchNorm passes, losses, optimizer, EMA included); `conv_kernels_only`: the "
                        "same MACs over the summed durations of those launches (HIP events, one single-stream step)"}
            if world == 1 and not args.no_cpu_baseline:
                entry["cpu_baseline"] = (cpu_baseline_gan(weights, B, encoder=bool(gan and encoder)) if gan else
                                         cpu_baseline_train(weights, B))
        except Exception as e:      # noqa: BLE001
            entry["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            ops.profile = None
    del tr
    torch.cuda.empty_cache()
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--dtype", choices=("fp32", "bf16", "f16"), default="fp32",
                    help="fp32 = the parity configuration (BASELINE configs[1]); bf16 / f16 = the reduced-precision "
                         "inference path of configs[4] (2-byte channels-last activations, MFMA bf16/f16, fp32 accumulate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="run the two generators on ONE stream (default: NetG_highweight's trunk overlaps G_SR_NET_low on "
                         "a second HIP stream).  Per-launch kernel timing is only meaningful single-stream, so the "
                         "event-sampled steps always run serial; use this flag to collect a rocprofv3 summary whose "
                         "per-kernel durations are comparable with `roofline`")
    ap.add_argument("--mode", choices=("infer", "train", "damsm"), default="infer",
                    help="infer = the headline (BASELINE configs[1]); train = generator fwd+bwd+Adam step on MSE+KL "
                         "(BASELINE configs[2] without the discriminator / DAMSM terms the reference does not define)")
    ap.add_argument("--gan", action="store_true", help="--mode train with the three discriminators (G/D alternation)")
    ap.add_argument("--damsm-encoder", action="store_true",
                    help="--mode train --gan: add generator_loss's DAMSM ranking term through CNN_ENCODER walking an Inception-v3 "
                         "topology with random weights (tests/inception_v3_arch.py; the trained trunk is third-party)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (inference mode)")
    ap.add_argument("--eager", action="store_true",
                    help="inference: launch every kernel from the host and alternate `--lanes` stream lanes.  Default "
                         "(neither --graph, --eager nor --serial): the step is replayed from ONE hipGraph holding 4 (3, 2: the "
                         "largest that divides --steps) independent batches as parallel branches - measured 10.26 k images/s "
                         "against 9.7-9.8 k for three eager lanes (fp32, batch 16)")
    ap.add_argument("--graph-lanes", type=int, default=1,
                    help="--graph: capture this many independent batches as parallel branches of ONE hipGraph (a replay = "
                         "that many steps; --steps must be a multiple)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="inference: consecutive steps alternate between this many stream lanes (1 = one step at a time)")
    ap.add_argument("--extras", default="lp,train",
                    help="default run (fp32 inference, eager): also time, in this process after the headline region, the "
                         "reduced-precision hipGraph configurations (`lp` object: BASELINE configs[4]) and the train steps "
                         "(`train` object: configs[2]); '' or 'none' = headline only")
    ap.add_argument("--branch-num", type=int, default=4,
                    help="cfg.TREE.BRANCH_NUM: 4 = the x8 generators of model.py (BASELINE configs, the default); anything "
                         "else = the x16 generators of models16.py (trainer_objective.py:74-87: 32 -> 512, weight-tied stages, "
                         "a fourth attention at 256^2 = 65 536 pixels), seeded random-init weights (no x16 checkpoint ships)")
    ap.add_argument("--min-time", type=float, default=0.5,
                    help="every timed figure is the median over enough fenced K-step regions for this many seconds of timed work")
    ap.add_argument("--repeats", type=int, default=0, help="number of fenced K-step regions per figure (0 = from --min-time)")
    ap.add_argument("--profile-every", type=int, default=20,
                    help="bracket every launch of every Nth timed step with HIP events for the roofline (0 = never); "
                         "two events per launch cost ~9 %% of a step, so the timed region samples instead of paying "
                         "it on every step")
    args = ap.parse_args()
    if args.mode == "infer" and not (args.graph or args.eager or args.serial):
        for gl in (4, 3, 2):
            if args.steps % gl == 0:
                args.graph, args.graph_lanes = True, gl
                break
    maybe_spawn(args)          # --gpus N without torchrun: run the N ranks as a child process, relay, exit

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal hook for a 1-GPU box: TGSR_BENCH_REHEARSAL=1 puts every rank on cuda:0 over gloo (RCCL refuses two
        # ranks on one device); the driver's multi-GPU run never sets it
        rehearsal = os.environ.get("TGSR_BENCH_REHEARSAL", "0") == "1"
        dev_index = 0 if rehearsal else local_rank
        torch.cuda.set_device(dev_index)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    from tgsr_amd import _lib, ops
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.synthetic import random_init_, synthetic_batch
    from tgsr_amd.trainer import SRPipeline
    _lib.lib()   # no HIP library -> error, never a fallback
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 256
    cfg.TREE.BRANCH_NUM = args.branch_num
    cfg.TREE.BASE_SIZE = 32
    x16 = args.branch_num != 4

    weights = None if x16 else load_weights()
    if x16:
        args.extras = "none"           # the lp / train objects are BASELINE's x8 configurations
        if args.mode != "infer":
            raise SystemExit("--branch-num applies to the inference bench")
    if args.mode == "train":
        return bench_train(args, rank, world, dist, dev, weights)
    if args.mode == "damsm":
        return bench_damsm(args, rank, world, dist, dev)
    pipe = SRPipeline(41, device=dev, low="lr", overlap=not args.serial, dtype=args.dtype)
    if weights is not None:
        pipe.load_state_dicts(weights["E."], weights["GL."], weights["GH."])
        wdesc = "shipped face_S8 checkpoint (tests/golden fixture), random-init text encoder"
    else:
        for i, m in enumerate((pipe.netGL, pipe.netGH)):
            random_init_(m, seed=i)
        wdesc = "seeded random-init weights"
    B = args.batch
    pool = batch_pool(B, rank, dev)
    cap, lens, LR, LRb = (pool[0][k] for k in ("cap", "lens", "LR", "LRb"))

    glanes = max(1, args.graph_lanes) if args.graph else 1
    if args.graph:      # BASELINE config 5: the step replayed from a captured hipGraph (identical results)
        if args.steps % glanes:
            raise SystemExit("--steps must be a multiple of --graph-lanes (a replay runs that many steps)")
        pipe.capture(cap, lens, LR, LRb)                   # the one-lane step
    multi = None
    if glanes > 1:
        from tgsr_amd.trainer import GraphedStep
        multi = GraphedStep(pipe, cap, lens, LR, LRb, lanes=glanes)   # `glanes` independent batches in one graph

    nstep = [0]

    def step(eager=False):
        """One step on the NEXT batch of the pool (other captions, caption lengths and images than the step before)."""
        k = nstep[0]
        nstep[0] += 1
        if eager or not args.graph:
            b = pool[k % len(pool)]
            return pipe(b["cap"], b["lens"], b["LR"], b["LRb"])
        return replay_on(pipe, pool, k)

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Consecutive steps are independent batches: they alternate between `--lanes` stream lanes (each lane = the two
    # streams of SRPipeline), so the small kernels and tails of one step fill the dispatch gaps of the previous one
    # (~60 dependent launches leave the GPU idle ~10 % of a single-lane step).  Every step still runs completely; the
    # sampled steps run alone (lanes drained before and after) so that a launch is timed in isolation.
    nlanes = 1 if (args.serial or args.graph) else max(1, args.lanes)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(nlanes)] if nlanes > 1 else []
    if lanes:
        torch.cuda.synchronize()            # weight packs / caches of the warm-up steps are complete before fanning out
    for ln in lanes:                        # untimed: every lane allocates its activation buffers once
        with torch.cuda.stream(ln):
            step()
    if args.graph and args.profile_every > 0:
        step(eager=True)                    # untimed: the event-sampled steps run eagerly and own a buffer set too
    prof, nprof = [], [0]
    K = args.steps

    # ---- the strict figure (BASELINE's metric: batch B per GPU, ONE step in flight): K steps issued one at a time, replayed
    # from the one-lane hipGraph (or eager / serial as selected).  The event-sampled steps (every launch bracketed by HIP
    # events on its stream, eager and single-stream so that a launch is timed alone) sit inside the LAST timed region only;
    # the reported time is the median over the regions.
    def region_one(r, last):
        for k in range(K):
            sample = last and args.profile_every > 0 and (K - 1 - k) % args.profile_every == 0
            if sample:
                ops.profile, pipe.overlap = prof, False
                nprof[0] += 1
                step(eager=True)
                ops.profile, pipe.overlap = None, not args.serial
            else:
                step()
    pipe.overlap = not args.serial
    fence()
    t0 = time.perf_counter()
    region_one(0, False)
    fence()
    reps = _bcast_int(reps_for(time.perf_counter() - t0, args), dist, dev)
    ts_local = []
    dt1, ts1 = time_regions(lambda r: region_one(r, r == reps - 1), fence, reps, dist, dev, local=ts_local)
    ranks = ranks_object(rank, world, dist, dev, ts_local)

    # ---- the throughput form: several independent batches in flight (parallel branches of one hipGraph, or eager stream lanes)
    dtm = None
    if multi is not None:
        def region_multi(r):
            for k in range(K // glanes):    # one replay = `glanes` steps on `glanes` different batches, other ones every replay
                replay_on(multi, pool, r * (K // glanes) + k, glanes)
        dtm, _tsm = time_regions(region_multi, fence, reps, dist, dev)
    elif lanes:
        def region_lanes(r):
            for k in range(K):
                with torch.cuda.stream(lanes[k % nlanes]):
                    step()
        dtm, _tsm = time_regions(region_lanes, fence, reps, dist, dev)
    ops.profile = None
    nprof = nprof[0]
    dt = dt1
    form_multi = multi is not None

    extras = {}
    want = [e for e in args.extras.replace("none", "").split(",") if e]
    if want and args.dtype == "fp32" and not args.serial:
        # driver-visible measurements of the other BASELINE configurations, same process, after the headline region
        pipe.overlap = True
        if "lp" in want:
            extras["lp"] = lp_object(args, rank, world, dist, dev, weights, pipe, fence)
        if "train" in want:
            lanes = multi = None
            pipe._graphed = None
            pipe._side = None
            torch.cuda.empty_cache()
            extras["train"] = train_object(args, rank, world, dist, dev, weights, fence)
    if rank == 0:
        # per-kernel totals from the HIP events recorded around every launch of the sampled steps
        agg = {}
        for name, flops, nbytes, e0, e1 in prof:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += nbytes
            a[3] += e0.elapsed_time(e1) * 1e-3
        roof, kern, att = roofline_objects(agg, nprof, args.dtype, args.serial, args.steps, B, counters=not x16) if nprof else (None, {}, None)
        if att is None and nprof and getattr(pipe, "_lp", None) is not None and pipe._lp.fuse_attention:
            # BASELINE's metric asks for the attention GEMM's MFMA utilisation: in the default route it has no launch of its own,
            # so ONE extra eager step with the stand-alone attention launches (same arithmetic, same bits) is event-timed
            try:
                prof2 = []
                pipe._lp.fuse_attention, pipe.overlap, ops.profile = False, False, prof2
                b0 = pool[0]
                pipe(b0["cap"], b0["lens"], b0["LR"], b0["LRb"])
                torch.cuda.synchronize()
                agg2 = {}
                for name, flops, nbytes, e0, e1 in prof2:
                    a = agg2.setdefault(name, [0, 0.0, 0.0, 0.0])
                    a[0] += 1
                    a[1] += flops
                    a[2] += nbytes
                    a[3] += e0.elapsed_time(e1) * 1e-3
                att = roofline_objects(agg2, 1, args.dtype, True, 1, B, counters=not x16)[2]
                if att is not None:
                    att["measured_with"] = ("stand-alone attention launches (TGSR_LP_FUSE_ATT=0), one extra step outside the "
                                            "timed region; the timed steps compute the same arithmetic inside lp_stem_kernel / "
                                            "lp_upconv_glu_kernel, where it costs no launch and no re-read of h")
            except Exception as e:       # noqa: BLE001
                att = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                pipe._lp.fuse_attention, pipe.overlap, ops.profile = True, not args.serial, None
        dname = {"fp32": "f32", "bf16": "bf16", "f16": "f16"}[args.dtype]
        tform = None
        if dtm is not None:
            tform = {"value": round(world * B * K / dtm, 2), "ms_per_step": round(dtm / K * 1e3, 4),
                     "form": ("%d independent batches of %d as parallel branches of ONE hipGraph (%d images in flight), %d replays "
                              "per region" % (glanes, B, B * glanes, K // glanes)) if form_multi else
                             "%d eager stream lanes (consecutive steps alternate between them)" % nlanes,
                     "images_in_flight": B * (glanes if form_multi else nlanes)}
        res = {"metric": "SR images/sec (32->%d, batch %d per GPU, one step in flight)" % (512 if x16 else 256, B),
               "value": round(world * B * K / dt, 2),
               "unit": "images/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
               "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": dname, "data": "synthetic inputs (%d batches, seeds 100 + 1000 i); " % len(pool) + wdesc,
               "value_one_lane": round(world * B * K / dt1, 2),
               "ms_per_step_one_lane": round(dt1 / K * 1e3, 4),
               **({"value_throughput_form": tform["value"]} if tform else {}),
               "config": {"workload": ("CelebA face x16 (32->512, models16.py: TREE.BRANCH_NUM=%d) batch=%d per GPU, text-enc + "
                                       "G_SR_NET_low + NetG_highweight forward, eval BN (the reference's default branch, "
                                       "config.py:28; not a BASELINE config)" % (args.branch_num, B)) if x16 else
                                      "CelebA face x8 (32->256) batch=%d per GPU, text-enc + G_SR_NET_low + "
                                      "NetG_highweight forward, eval BN (BASELINE configs[%d])" %
                                      (B, 1 if args.dtype == "fp32" else 4),
                          "batch_per_gpu": B, "lr": 32, "sr": 512 if x16 else 256, "n_words": 41, "parallelism": "dp%d" % world,
                          "streams": 1 if args.serial else 2, "launch": "hipgraph" if args.graph else "eager",
                          "storage": "fp32 NCHW" if args.dtype == "fp32" else
                          "%s channels-last (zero-bordered), fp32 accumulate; inputs / outputs fp32" % args.dtype,
                          "sampled_steps": nprof,
                          "timing": "median of %d fenced regions of exactly K = %d steps (barrier + synchronize on both sides of "
                                    "each, max over ranks per region; >= %.1f s of timed work in all); regions, s: %s"
                                    % (reps, K, args.min_time, [round(t, 5) for t in ts1]),
                          "repeats": reps,
                          "batches": "%d different resident synthetic batches (captions, caption lengths, images) cycled "
                                     "through the steps: every timed step - every lane of a hipGraph replay - runs on another "
                                     "batch than the one before; a replay copies it into the graph's static inputs inside the "
                                     "timed loop (one copy launch); the captured step is independent of the caption lengths"
                                     % len(pool),
                          **({"throughput_form": tform} if tform else {}),
                          "note": ("`value` = the strict figure of BASELINE's metric: K steps of batch %d issued ONE AT A TIME "
                                   "(%s), %d event-sampled eager single-stream step(s) inside the last region.  "
                                   "`throughput_form` (not the headline): the same K steps with several independent batches in "
                                   "flight, measured right after with the same fencing"
                                   % (B, "replays of the one-lane hipGraph" if args.graph else "eager launches", nprof))},
               "roofline": roof, "kernels": kern}
        # the whole step against SURVEY.md 8(d)'s compulsory conv-path bytes (173.9 MB per image at the reference's layer
        # boundaries in fp32, half of that with 2-byte storage) and 21.07 GFLOP per image: the "fraction of the HBM
        # roofline" BASELINE.json's north_star speaks of, per GPU
        mb_img = 173.9 if args.dtype == "fp32" else 173.9 / 2
        gflop_img = 21.07
        if x16 and nprof:
            # no SURVEY figure exists for the x16 generators: the per-launch accounting of the sampled step(s) stands in
            # (every launch's algorithmic bytes / FLOPs, summed; the same accounting gives 2 782 MB per x8 step at batch 16
            # = SURVEY's 173.9 MB per image)
            mb_img = sum(v[2] for v in agg.values()) / nprof / B / 1e6
            gflop_img = sum(v[1] for v in agg.values()) / nprof / B / 1e9
        res["step_roofline"] = {
            "compulsory_MB_per_image": round(mb_img, 2), "GFLOP_per_image": round(gflop_img, 3),
            "hbm_frac": round(res["value"] / world * mb_img * 1e6 / (PEAK_HBM_GBS * 1e9), 4),
            "algorithmic_TFLOPs": round(res["value"] / world * gflop_img * 1e9 / 1e12, 1),   # direct-form FLOPs (Winograd /
            "mfma_peak_TFLOPs": PEAK_FP32_MFMA_TFLOPS if args.dtype == "fp32" else PEAK_LP_MFMA_TFLOPS,   # sub-pixel forms execute fewer)
            "hbm_frac_one_lane": round(res["value_one_lane"] / world * mb_img * 1e6 / (PEAK_HBM_GBS * 1e9), 4)}
        if att is not None:
            res["attention"] = att
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(weights, B, x16_pipe=pipe if x16 else None)
        res.update(extras)
        res["ranks"] = ranks
    direct = rccl_direct_evidence(world, dist, dev)
    if rank == 0:
        if direct is not None:
            res["ranks"]["rccl_direct"] = direct
            res["ranks"]["rccl_ranks"] = direct.get("rccl_ranks")
        print(json.dumps(res), flush=True)
    if direct is not None:
        # Every rank must leave the same way: a rank stuck inside librccl cannot take part in the closing barrier, and the others
        # would wait in it for ever (a non-zero exit of the launcher although the line is out).  The verdicts travel through the
        # rendezvous store (TCP, no GPU involved); any hung rank -> all of them leave without the interpreter's teardown.
        anyhung = bool(direct.get("hung"))
        try:
            from torch.distributed.distributed_c10d import _get_default_store
            import datetime
            store = _get_default_store()
            store.set("tgsr_direct_%d" % rank, "hung" if anyhung else "ok")
            store.set_timeout(datetime.timedelta(seconds=90))
            for r in range(world):
                anyhung = anyhung or store.get("tgsr_direct_%d" % r) == b"hung"
        except Exception:                # noqa: BLE001 - a rank that never wrote its key is a hung rank
            anyhung = True
        if anyhung:
            sys.stdout.flush()
            os._exit(0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
