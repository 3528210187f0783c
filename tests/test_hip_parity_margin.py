"""GPU: how much of the stated fp32 tolerance the HIP path uses, measured over MANY cases instead of one.

Every case runs the fp32 pipeline at the batch size that switches the F(4x4) Winograd routing on (ops.wino4_wanted /
upwino4_wanted) and is compared with (a) the CPU fp32 oracle at the path's ONE stated tolerance (conftest.FP32_TOL: atol = rtol =
2e-4, every value, no outlier allowance - round 6, see below) and (b) an
fp64 run of the oracle, next to the CPU fp32 oracle's OWN distance from fp64 - the reference's arithmetic against exact
arithmetic is the yardstick, not one fp32 evaluation against another (two correct fp32 evaluations of this network differ by up to
1e-4 at the few pixels whose activations are 20-50x the typical size: profiles/HISTORY.md 3.1g).

Cases: the 8 batches `bench.py` times (shipped face checkpoint, seeds 100 + 1000 i), 4 more seeds, seeded-random weights with
randomised BatchNorm statistics (x8), and the x16 generators.  What the table showed (profiles/r05_parity_margin.txt, DESIGN.md
3.1g) and what is asserted:
  * the MEAN distance from fp64 is stable case to case: HIP 1.27-1.30 x the CPU fp32 oracle's on the final image, 1.53-1.56 x on
    G_SR_NET_low's (round 4's interpolation points and routing: 1.33 / 1.61) -> per case <= MEAN_RATIO;
  * so is a high quantile (the value 0.1 % of the image exceeds) -> per case <= TAIL_RATIO x the CPU's;
  * the MAX is not: it sits on a handful of pixels whose activations are 20-50 x the typical size, where NetG_highweight's
    128^2 section multiplies whatever rounding error it is handed by 10-40 (tools/diag_layer_errors.py).  The CPU fp32 oracle
    itself is up to 8.4e-5 from fp64 there, the HIP path up to 1.8e-4 - on the SAME pixel with every layer on F(2x2), 9.0e-5
    with the direct kernels, i.e. independent of the F(4x4) routing - and per case HIP / CPU ranges from 0.6 to 5.5.  Two correct
    fp32 evaluations can therefore differ by more than 1e-4 at isolated values (the CPU oracle itself needs 2e-4 against the
    reference's own train-mode run, tests/test_oracle_golden.py).  Round 5 asserted 1e-4 "for all but one value per million, none
    beyond 3 tolerances" - a two-tier statement.  Round 6 states ONE tolerance instead, the smallest round figure every value of
    every case meets: atol = rtol = 2e-4 (worst use over the 15 cases: 0.68 of it; 14 of the 15 stay within 1e-4 as well, which
    the report still shows); over the pool the HIP maximum against fp64 must stay within POOL_MAX_RATIO of the CPU's;
  * every STAGE's own error (same input, three ways) is bounded against the CPU op's: the routing rule as an error bound.
"""
import os

import numpy as np
import pytest
import torch

from conftest import FP32_TOL, split_sd
from oracle import tgsr_oracle as O
from oracle import tgsr_oracle_lp as OL

pytestmark = pytest.mark.gpu
DEV = "cuda"
ATOL = RTOL = FP32_TOL   # the path's single stated fp32 tolerance (2e-4): every value of every case
MEAN_RATIO = 1.7          # mean |hip - f64| / mean |cpu32 - f64| per case; measured 1.04 .. 1.56
TAIL_RATIO = 1.75         # the same for the value exceeded by 1e-3 of a case's finest image (~3 000 values: a stable statistic; the
                          # 1e-5 tail - 31 values - already ranges from 0.86 to 2.8 case to case like the maximum)
POOL_MAX_RATIO = 3.0      # max over the pool, HIP vs CPU (measured 2.15 final image, 2.95 G_SR_NET_low's: heavy-tailed, see above)


def _dbl(sd):
    return {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}


def _randomise_bn(sd, seed):
    g = torch.Generator().manual_seed(seed)
    for k in list(sd):
        if k.endswith("running_var"):
            p = k[:-len("running_var")]
            sd[p + "bias"] = 0.1 * torch.randn(sd[k].shape, generator=g)
            sd[p + "running_mean"] = 0.1 * torch.randn(sd[k].shape, generator=g)
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    return sd


def _x16_state(seed=5):
    from tgsr_amd import models16
    from tgsr_amd.synthetic import random_init_
    gl, gh = models16.G_SR_NET_low(), models16.NetG_highweight(weightmap=False, low="lr")
    random_init_(gl, seed), random_init_(gh, seed + 1)
    sdL = _randomise_bn({k: v.detach().clone() for k, v in gl.state_dict().items()}, seed)
    sdH = _randomise_bn({k: v.detach().clone() for k, v in gh.state_dict().items()}, seed + 1)
    return sdL, sdH


def _cases(face_weights):
    E, GL, GH = (split_sd(face_weights, p) for p in ("E.", "GL.", "GH."))
    out = []
    for i in range(8):
        out.append(("face_S8 checkpoint, bench pool batch %d" % i, 4, (E, GL, GH), 16, 100 + 1000 * i))
    for s in (7, 8, 9, 10):
        out.append(("face_S8 checkpoint, seed %d" % s, 4, (E, GL, GH), 16, s))
    for s in (1, 2):
        sdE, sdL, sdH = O.random_state(seed=s)
        out.append(("random weights (seed %d), x8" % s, 4, (sdE, _randomise_bn(sdL, s), _randomise_bn(sdH, s + 10)), 16, 50 + s))
    return out


@pytest.fixture(scope="module")
def margin_table(face_weights):
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.trainer import SRPipeline
    from tgsr_amd import ops
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 256
    rows = []
    cases = _cases(face_weights)
    sdE16 = O.random_state(seed=2)[0]
    sdL16, sdH16 = _x16_state()
    cases.append(("random weights, x16 generators (models16), batch 4", 5, (sdE16, sdL16, sdH16), 4, 9))
    pipes = {}
    for label, branch, (sdE, sdL, sdH), B, seed in cases:
        cap, lens, LR, LRb = O.synthetic_batch(B, seed=seed)
        if branch == 4:
            r32 = O.sr_forward(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)
            r64 = O.sr_forward(_dbl(sdE), _dbl(sdL), _dbl(sdH), cap, lens.tolist(), LR.double(), LRb.double())
        else:
            r32 = OL.sr_forward16(sdE, sdL, sdH, cap, lens.tolist(), LR, LRb)
            r64 = OL.sr_forward16(_dbl(sdE), _dbl(sdL), _dbl(sdH), cap, lens.tolist(), LR.double(), LRb.double())
        cfg.TREE.BRANCH_NUM = branch
        if branch not in pipes:
            pipes[branch] = SRPipeline(41, device=DEV, branch_num=branch)
        p = pipes[branch].load_state_dicts(sdE, sdL, sdH)
        hip = p(cap.to(DEV), lens.tolist(), LR.to(DEV), LRb.to(DEV))
        last = len(r32["fine"]) - 1
        row = {"label": label, "viol": 0.0, "nviol": 0, "nvals": 0}
        for k in ("fake", "fine"):
            for i in range(last + 1):
                h = hip[k][i].cpu()
                v = (h - r32[k][i]).abs() / (ATOL + RTOL * r32[k][i].abs())
                row["viol"] = max(row["viol"], float(v.max()))
                row["nviol"] += int((v > 1).sum())
                row["n1e4"] = row.get("n1e4", 0) + int((v * (ATOL / 1e-4) > 1).sum())      # values beyond round 5's 1e-4
                row["nvals"] += v.numel()
            eh = (hip[k][last].cpu().double() - r64[k][last]).abs()
            ec = (r32[k][last].double() - r64[k][last]).abs()
            kth = max(1, int(eh.numel() * 1e-3))                       # the value 0.1 % of the image exceeds (~3 000 values)
            th, tc = torch.topk(eh.flatten().float(), kth).values, torch.topk(ec.flatten().float(), kth).values
            k4 = max(1, kth // 10)
            row[k] = (float(eh.max()), float(eh.mean()), float(ec.max()), float(ec.mean()), float(th[-1]), float(tc[-1]),
                      float(th[k4 - 1]), float(tc[k4 - 1]))
        rows.append(row)
        del r32, r64, hip
    cfg_reset()
    lines = ["%-52s %-9s  %s" % ("case", "tol used, values beyond it", "finest image: |hip-f64| max/mean   |cpu32-f64| max/mean   ratios HIP / CPU: mean, 1e-3 tail, 1e-4 tail   (G_SR_NET_low's, final)")]
    for r in rows:
        lines.append("%-52s %-4.2f %4d (beyond 1e-4: %d)  " % (r["label"], r["viol"], r["nviol"], r.get("n1e4", 0)) + "   ".join(
            "%.2e/%.1e  %.2e/%.1e  %.2f %.2f %.2f" % (r[k][0], r[k][1], r[k][2], r[k][3], r[k][1] / r[k][3], r[k][4] / r[k][5],
                                                    r[k][6] / r[k][7]) for k in ("fake", "fine")))
    for k in ("fake", "fine"):
        lines.append("pool, %s: max |hip-f64| %.2e   max |cpu32-f64| %.2e   ratio %.2f" % (
            k, max(r[k][0] for r in rows), max(r[k][2] for r in rows), max(r[k][0] for r in rows) / max(r[k][2] for r in rows)))
    text = "\n".join(lines)
    print("\n" + text)
    out = os.environ.get("TGSR_MARGIN_REPORT")
    if out:
        with open(out, "w") as f:
            f.write("routing: wino4=%s min_workgroups=%d\n" % (ops.ROUTING.wino4, ops.ROUTING.min_workgroups) + text + "\n")
    return rows


def test_fp32_parity_margin_stated_tolerance(margin_table):
    """EVERY value of every image of every case within the stated tolerance (atol = rtol = FP32_TOL = 2e-4) of the CPU fp32 oracle:
    no outlier allowance, no second tier."""
    for r in margin_table:
        assert r["nviol"] == 0 and r["viol"] <= 1.0, (r["label"], r["nviol"], r["nvals"], r["viol"])


def test_fp32_parity_margin_mean_and_tail_error_vs_fp64(margin_table):
    """Per case the HIP path's mean distance from exact arithmetic, and the value its 1e-3 tail exceeds, stay within MEAN_RATIO /
    TAIL_RATIO of the reference's own fp32 path's."""
    for r in margin_table:
        for k in ("fake", "fine"):
            assert r[k][1] <= MEAN_RATIO * r[k][3], (r["label"], k, r[k])
            assert r[k][4] <= TAIL_RATIO * r[k][5], (r["label"], k, r[k])


def test_fp32_parity_margin_max_error_vs_fp64_over_the_pool(margin_table):
    """Over the pool the worst value of the HIP path stays within POOL_MAX_RATIO of the CPU fp32 path's worst."""
    x8 = [r for r in margin_table if "x16" not in r["label"]]
    for k in ("fake", "fine"):
        assert max(r[k][0] for r in x8) <= POOL_MAX_RATIO * max(r[k][2] for r in x8), (k, [(r["label"], r[k][0], r[k][2]) for r in x8])


# ------------------------------------------------------------------------------------------ every stage's OWN rounding error
OWN_MEAN_RATIO = 1.5            # HIP module vs the CPU fp32 op on the same input, mean |x - f64|: stages on F(2x2) / direct kernels
                                # (measured 0.69 .. 1.29; NetG_highweight.residual48 with one F(4x4) layer of two: 1.27)
OWN_MEAN_RATIO_F44 = 2.2        # stages whose convolutions are mostly F(4x4): G_SR_NET_low's 64^2 and 128^2 stages (2.02, 1.76) and
                                # NetG_highweight's last upBlock (1.86)


@pytest.fixture(scope="module")
def own_error_table(face_weights):
    """Each stage of the two generators run three ways on the SAME input - the fp64 oracle's activation at that point, rounded
    to fp32: the HIP module, the CPU fp32 oracle function, the fp64 oracle function.  A stage's figure is then its own rounding
    error, not what it inherited: the routing of a layer to a kernel form is justified by THIS number (the error it adds), the
    end-to-end maxima are dominated by what the network does to anybody's rounding error at a handful of outlier pixels."""
    import torch.nn.functional as F
    from tgsr_amd.miscc.config import cfg, cfg_reset
    from tgsr_amd.trainer import SRPipeline
    cfg_reset()
    cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM, cfg.TREE.BRANCH_NUM = 32, 256, 4
    E, L32, H32 = (split_sd(face_weights, p) for p in ("E.", "GL.", "GH."))
    L64, H64 = _dbl(L32), _dbl(H32)
    p = SRPipeline(41, device=DEV, branch_num=4).load_state_dicts(E, L32, H32)
    gl, gh = p.netGL, p.netGH
    rows = []
    for seed in (100, 6100):
        cap, lens, LR, LRb = O.synthetic_batch(16, seed=seed)
        words, sent = O.rnn_encoder(E, cap, lens.tolist())
        mask = (cap == 0)[:, :words.shape[2]]
        wd, md = words.to(DEV), mask.to(DEV)

        def rec(name, yh, yc, y64):
            eh, ec = (yh.cpu().double() - y64).abs(), (yc.double() - y64).abs()
            rows.append((seed, name, float(eh.max()), float(eh.mean()), float(ec.max()), float(ec.mean())))
        with torch.no_grad():
            # ---- G_SR_NET_low: stage by stage (attention + 2 ResBlocks + upBlock), then its image heads
            x64 = LR.double()
            for k in (1, 2, 3):
                pre = "h_net%d." % k
                xin = x64.float()
                if k == 1:
                    fn = lambda x, sd_, w_: O.init_stage(sd_, pre, x, w_, mask)[0]                      # noqa: E731
                    yh = gl.h_net1(None, xin.to(DEV), wd, md)[0]
                else:
                    fn = lambda x, sd_, w_: O.next_stage(sd_, pre, x, w_, mask)[0]                      # noqa: E731
                    yh = getattr(gl, "h_net%d" % k)(xin.to(DEV), None, wd, md)[0]
                rec("G_SR_NET_low.h_net%d" % k, yh, fn(xin, L32, words), fn(xin.double(), L64, words.double()))
                x64 = fn(x64, L64, words.double())
                w3 = L64["img_net%d.img.0.weight" % k]
                xin = x64.float()
                rec("G_SR_NET_low.img_net%d" % k, getattr(gl, "img_net%d" % k)(xin.to(DEV)), F.conv2d(xin, w3.float(), None, 1, 1),
                    F.conv2d(xin.double(), w3, None, 1, 1))
            # ---- NetG_highweight: module by module
            stages = [("convin", gh.convin, lambda x, s: O.conv_bn_glu(x, s, "convin."))]
            for i in range(6):
                stages.append(("residual.%d" % i, gh.residual[i], lambda x, s, i=i: O.res_block(x, s, "residual.%d." % i)))
            stages += [("upscale2x", gh.upscale2x, lambda x, s: O.up_block(x, s, "upscale2x.")),
                       ("residual24", gh.residual24, lambda x, s: O.residual_nosum(x, s, "residual24.")),
                       ("upscale4x", gh.upscale4x, lambda x, s: O.up_block(x, s, "upscale4x.")),
                       ("residual48", gh.residual48, lambda x, s: O.residual_nosum(x, s, "residual48.")),
                       ("upscale8x", gh.upscale8x, lambda x, s: O.up_block(x, s, "upscale8x."))]
            x64 = LR.double()
            for name, mod, fn in stages:
                xin = x64.float()
                rec("NetG_highweight." + name, mod(xin.to(DEV)), fn(xin, H32), fn(xin.double(), H64))
                x64 = fn(x64, H64)
                if name.startswith("upscale"):
                    w5 = H64["conv_output.0.weight"]
                    xin = x64.float()
                    sr = torch.zeros(16, 3, xin.shape[2], xin.shape[3])
                    rec("NetG_highweight.conv_output after " + name, gh._head(xin.to(DEV), sr.to(DEV)),
                        torch.tanh(F.conv2d(xin, w5.float(), None, 1, 2)), torch.tanh(F.conv2d(xin.double(), w5, None, 1, 2)))
    cfg_reset()
    text = "\n".join(["%-6s %-46s %-22s %-22s %s" % ("seed", "stage (own error on the fp64 oracle's input)", "HIP max / mean", "CPU fp32 max / mean", "mean ratio")] +
                     ["%-6d %-46s %.2e / %.2e    %.2e / %.2e    %.2f" % (r[0], r[1], r[2], r[3], r[4], r[5], r[3] / r[5]) for r in rows])
    print("\n" + text)
    out = os.environ.get("TGSR_MARGIN_REPORT")
    if out:
        with open(out.replace(".txt", "_own.txt"), "w") as f:
            f.write(text + "\n")
    return rows


def test_every_stage_adds_no_more_error_than_the_reference_op(own_error_table):
    """The routing rule as an error bound: whatever kernel form a layer is routed to, the stage's own mean distance from exact
    arithmetic stays within OWN_MEAN_RATIO of the CPU fp32 op's on the same input; OWN_MEAN_RATIO_F44 for the three stages that run
    (mostly) on the F(4x4) kernels at batch 16 - F(4x4) executes a quarter of the multiplies and pays for it with about twice the
    rounding error of the direct form per layer, which the end-to-end figures above absorb (mean 1.3 x the reference's)."""
    for seed, name, hmax, hmean, cmax, cmean in own_error_table:
        lim = OWN_MEAN_RATIO_F44 if name in ("G_SR_NET_low.h_net2", "G_SR_NET_low.h_net3", "NetG_highweight.upscale8x") else OWN_MEAN_RATIO
        assert hmean <= lim * cmean, (seed, name, hmean, cmean)
