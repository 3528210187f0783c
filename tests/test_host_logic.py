"""CPU: host-side logic of the drop-in modules (construction, state_dict contract, config, batch prep)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, split_sd


@pytest.fixture()
def cfg32():
    from tgsr_amd.miscc.config import cfg, cfg_reset
    cfg_reset()
    cfg.GAN.GF_DIM = 32
    cfg.TEXT.EMBEDDING_DIM = 256
    yield cfg
    cfg_reset()


def test_state_dict_contract_matches_shipped_checkpoints(cfg32, face_weights):
    from tgsr_amd import model
    man = json.load(open(os.path.join(GOLDEN, "ckpt_manifest.json")))
    gl, gh = model.G_SR_NET_low(), model.NetG_highweight(weightmap=False, low="lr")
    assert {k: list(v.shape) for k, v in gl.state_dict().items()} == {k: v[0] for k, v in man["netG_epoch_7"].items()}
    assert {k: list(v.shape) for k, v in gh.state_dict().items()} == {k: v[0] for k, v in man["netGH_epoch_7"].items()}
    assert "a" not in gh.state_dict()            # model.py:246-248 quirk: never saved
    gl.load_state_dict(split_sd(face_weights, "GL."), strict=True)
    gh.load_state_dict(split_sd(face_weights, "GH."), strict=True)
    enc = model.RNN_ENCODER(41, nhidden=256)
    enc.load_state_dict(split_sd(face_weights, "E."), strict=True)


def test_modules_have_no_cpu_fallback(cfg32):
    from tgsr_amd import util
    from tgsr_amd._lib import TgsrError
    rb = util.ResBlock(64).eval()
    with pytest.raises(TgsrError):
        rb(torch.zeros(1, 64, 8, 8))
    with pytest.raises(TgsrError):                      # training path: same rule, no CPU fallback
        util.ResBlock(64).train()(torch.zeros(1, 64, 8, 8))


def test_cfg_from_file_semantics(tmp_path, cfg32):
    from tgsr_amd.miscc.config import cfg, cfg_from_file
    p = tmp_path / "a.yml"
    p.write_text("TREE:\n    BRANCH_NUM: 4\n    BASE_SIZE: 32\nGAN:\n    GF_DIM: 32\n    R_NUM: 2\nTRAIN:\n    FLAG: False\n")
    cfg_from_file(str(p))
    assert cfg.TREE.BRANCH_NUM == 4 and cfg.GAN.GF_DIM == 32
    p.write_text("NOPE: 1\n")
    with pytest.raises(KeyError):
        cfg_from_file(str(p))
    p.write_text("GAN:\n    GF_DIM: 'x'\n")
    with pytest.raises(ValueError):
        cfg_from_file(str(p))


def test_batch_prep_matches_reference_conventions():
    from tgsr_amd.trainer import caption_mask, sort_by_caption_length
    cap = torch.tensor([[3, 4, 0, 0, 0], [5, 6, 7, 8, 0], [9, 1, 2, 0, 0]])
    lens = torch.tensor([2, 4, 3])
    img = torch.arange(3).float()
    c, l, im, idx = sort_by_caption_length(cap, lens, img)
    assert l.tolist() == [4, 3, 2] and im.tolist() == [1.0, 2.0, 0.0] and c[0].tolist() == [5, 6, 7, 8, 0]
    m = caption_mask(c, 4)
    assert m.shape == (3, 4) and m[2].tolist() == [False, False, True, True]


def test_install_dropin_registers_reference_module_names(cfg32):
    import sys
    import tgsr_amd
    saved = {k: sys.modules.get(k) for k in ("model", "util", "GlobalAttention", "miscc", "miscc.config")}
    try:
        tgsr_amd.install_dropin()
        from model import G_SR_NET_low, NetG_highweight, RNN_ENCODER  # noqa: F401  (trainer_objective.py:8,75-88)
        from GlobalAttention import GlobalAttentionGeneral, func_attention  # noqa: F401
        from miscc.config import cfg  # noqa: F401
        assert G_SR_NET_low.__module__ == "tgsr_amd.model"
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_models16_tied_stages_and_keys(cfg32):
    """models16: one NEXT_STAGE_G / one GET_IMAGE_G object behind the aliases; same state_dict keys as the reference."""
    from conftest import load_npz
    from tgsr_amd import models16
    g = load_npz("nets16_small.npz")
    cfg32.TEXT.EMBEDDING_DIM = 64
    gl = models16.G_SR_NET_low()
    assert gl.h_net2 is gl.h_net3 is gl.h_net4 and gl.img_net1 is gl.img_net4
    assert sorted(gl.state_dict().keys()) == sorted(g["GL.keys"].tolist())
    gh = models16.NetG_highweight(weightmap=False, low="lr")
    assert sorted(gh.state_dict().keys()) == sorted(g["gh16_keys"].tolist())
    assert "a" in gh.state_dict()       # models16.py:126: a registered Parameter here (unlike model.py:246-248)


def test_committed_bench_lines_follow_the_contract():
    """The JSON lines bench.py printed on the MI355X this round (committed under profiles/: r06_*) carry every field of the
    driver's contract, an honest roofline object (frac = achieved / peak <= 1, counters looked up from a committed
    rocprofv3 table) and - on the default N=1 runs - the CPU baseline."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r06_bench_*.json")))
    assert len(files) >= 10, "round-6 bench lines missing"
    saw_cpu = saw_lp = saw_train = False
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, (f, k)
        assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
        assert d["dtype"] in ("f32", "bf16", "f16") and "workload" in d["config"] and "model" not in d["config"]
        per_gpu = d["config"].get("batch_per_gpu", 16)
        assert abs(d["value"] - per_gpu * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (f, k)
        assert "kernel" in r or "kernels" in r, f        # the dominant kernel, or (train lines) the per-kernel table of the step
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1.0
        if "train" in os.path.basename(f):
            saw_train = True
        else:
            assert "value_one_lane" in d and d["value_one_lane"] > 0
        if d["dtype"] != "f32":
            saw_lp = True
            assert "channels-last" in d["config"]["storage"]
            assert r["bound"] == "hbm" and r["peak"] == 8000.0 and "mfma_frac" in r       # HBM roofline in bf16
        if "cpu_baseline" in d:
            saw_cpu = True
            c = d["cpu_baseline"]
            assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert saw_cpu and saw_lp and saw_train
    two = json.loads(open(os.path.join(root, "profiles", "r06_bench_rehearsal_2ranks.json")).read().strip().splitlines()[-1])
    assert two["n_gpus"] == 2                       # `bench.py --gpus 2` launched its two ranks itself
    assert "lp" in two and "train" in two and "error" not in json.dumps(two["train"])   # the extras survive N > 1 (all-reduce matched)
    assert two["ranks"]["world"] == 2 and two["ranks"]["all_reduce_of_ones"] == 2.0     # round 6: what the collective layer itself says
    # the default line (what the driver times): fp32 stays the headline, configs[4] and the train step ride along
    d = json.loads(open(os.path.join(root, "profiles", "r06_bench_fp32.json")).read().strip().splitlines()[-1])
    assert d["dtype"] == "f32" and d["config"]["launch"] == "hipgraph"
    # round 5: `value` is the STRICT figure of BASELINE's metric - batch 16, one step in flight - and says so; the four-lane
    # throughput form rides along; every figure is a median over >= 0.5 s of fenced K-step regions
    assert "one step in flight" in d["metric"] and d["value"] == d["value_one_lane"]
    tf = d["config"]["throughput_form"]
    assert tf["images_in_flight"] == 64 and tf["value"] == d["value_throughput_form"] > d["value"]
    assert d["config"]["repeats"] >= 3 and "median" in d["config"]["timing"]
    runs = {(r["dtype"], r["batch_per_gpu"]): r for r in d["lp"]["runs"]}
    assert set(runs) == {("bf16", 16), ("f16", 16), ("bf16", 8), ("bf16", 128)}
    for r in runs.values():
        # >= 50 dB for the bf16 configuration (met since round 4: the 32x32 trunk of NetG_highweight runs in f16)
        assert r["value"] > 0 and r["psnr_vs_fp32_dB"] >= 50 and 0 < r["step_roofline"]["hbm_frac"] < 1
        assert r["conv_kernel"]["bound"] == "hbm" and "mfma_frac" in r["conv_kernel"]
        # the captured step replayed on a batch with OTHER caption lengths equals the eager step on that batch, bit for bit
        assert r["replay_equals_eager_on_new_lengths"] is True
    assert runs[("bf16", 16)]["value"] >= 28000         # the driver-timed batch-16 bf16 line (29.9-33.3 k depending on the box)
    assert runs[("bf16", 16)]["graph_lanes4"]["hbm_frac"] >= 0.36      # four lanes: 0.39-0.41 of the HBM roofline depending on the box
    assert runs[("bf16", 128)]["step_roofline"]["hbm_frac"] >= 0.40    # north-star: >= 40 % of the HBM roofline on the conv path
    # every timed step runs on another batch than the one before (captions, caption lengths, images)
    assert "different resident synthetic batches" in d["config"]["batches"]
    tr = d["train"]["runs"]
    # round 6: generator step, G/D alternation, and configs[2] as BASELINE states it (G/D + DAMSM through CNN_ENCODER)
    assert len(tr) == 3 and all(0 < t["roofline"]["frac"] < 1 and "executed_fraction" in json.dumps(t["roofline"]) for t in tr)
    # the train steps' CPU baselines run at the configuration's own batch, their rooflines carry counter traffic
    assert all("cpu_baseline" in t and "batch 16" in t["cpu_baseline"]["sample"] for t in tr)
    assert all(t["roofline"]["traffic"] and t["roofline"]["traffic_from"].startswith("r06_train") for t in tr)
    assert d["roofline"]["counters_from"].startswith("r06_fp32")
    assert tr[1]["ms_per_step"] < 22.0              # the G/D alternation (29.1 ms at the end of round 4, 21.1 of round 5)
    assert tr[2]["ms_per_step"] < 34.0 and "CNN_ENCODER" in tr[2]["workload"]      # 35.2 with MIOpen's trunk (round 5)
    # the measured graph policy reports what it timed; the device-time breakdown: non-library kernels <= 8 % of a G/D step
    assert all(t["graph_policy"]["chosen"] in ("eager", "replay") and t["graph_policy"]["eager_ms"] > 0 for t in tr)
    assert tr[1]["device_time"]["non_tgsr_share"] <= 0.08


def test_get_caption_crops_like_the_reference():
    """datasets.py:461-477: short captions are zero padded; a longer one keeps WORDS_NUM word positions drawn by
    np.random.shuffle, in their original order - the same draw as the reference under the same numpy seed."""
    import numpy as np
    from tgsr_amd.datasets import get_caption
    x, n = get_caption([5, 6, 7], 18)
    assert n == 3 and x.tolist() == [5, 6, 7] + [0] * 15
    cap = list(range(1, 31))
    np.random.seed(100)
    x, n = get_caption(cap, 18)
    np.random.seed(100)                       # the reference's lines, verbatim in effect
    ix = list(np.arange(30))
    np.random.shuffle(ix)
    want = np.asarray(cap)[np.sort(ix[:18])]
    assert n == 18 and x.tolist() == want.tolist() and (np.diff(x) > 0).all() and x.tolist() != cap[:18]
    x2, _ = get_caption(cap, 18, rng=np.random.RandomState(3))
    x3, _ = get_caption(cap, 18, rng=np.random.RandomState(3))
    assert x2.tolist() == x3.tolist()


def test_lazy_log_is_opt_in_and_formats_like_the_reference():
    import torch
    from tgsr_amd.miscc.losses import _LazyLog
    log = _LazyLog([("g_loss%d: %%.5f " % 0, torch.tensor(1.5)), ("w_loss: %.5f s_loss: %.5f ", torch.tensor(2.0), torch.tensor(3.0))])
    text = "g_loss0: 1.50000 w_loss: 2.00000 s_loss: 3.00000 "
    assert str(log) == text and log == text and "w_loss" in log and len(log) == len(text) and log[:7] == "g_loss0"
    assert ("" + log) == text and (log + "x") == text + "x" and "%s" % log == text and log._parts is None


def test_wino4_size_policy(monkeypatch):
    """Which conv3x3 layers go to the F(4x4, 3x3) kernels: whole tiles and 64-channel groups; the numerics rule of
    tgsr_winograd4.hip (>= 128 x 128 pixels, at 64 x 64 only 128-channel groups, nothing below); at least 256 workgroups of the
    form the layer takes; TGSR_WINO4=0 switches it off."""
    from tgsr_amd import ops
    monkeypatch.setattr(ops, "ROUTING", ops._Routing(env={}))
    assert ops.wino4_wanted(64, 128, 128, 128, 16) and ops.wino4_wanted(64, 64, 128, 128, 16) and ops.wino4_wanted(64, 64, 256, 256, 2)
    assert ops.wino4_wanted(64, 128, 64, 64, 16) and not ops.wino4_wanted(64, 64, 64, 64, 16)    # 64 x 64: 128-channel groups only
    assert not ops.wino4_wanted(64, 64, 64, 64, 64)                                              # ... whatever the batch
    assert ops.wino4_wanted(64, 64, 128, 128, 4) and not ops.wino4_wanted(64, 64, 128, 128, 2)   # 256 vs 128 four-wave workgroups
    assert ops.wino4_wanted(64, 128, 128, 128, 4) and not ops.wino4_wanted(64, 128, 128, 128, 2)
    assert ops.wino4_wanted(12, 64, 128, 128, 8) and not ops.wino4_wanted(12, 64, 128, 128, 4)   # LDS-fed form: 8-row tiles
    assert not ops.wino4_wanted(64, 128, 32, 32, 64) and not ops.wino4_wanted(128, 256, 32, 64, 64)   # below 64 x 64
    assert not ops.wino4_wanted(64, 32, 128, 128, 16) and not ops.wino4_wanted(3, 64, 128, 128, 16)   # channel groups / stages
    assert not ops.wino4_wanted(64, 64, 128, 160, 16) and not ops.wino4_wanted(64, 64, 132, 128, 16)  # whole tiles
    monkeypatch.setattr(ops.ROUTING, "pin_batch", 16)         # TGSR_WINO4_PIN_BATCH: the routing of batch 16 at every batch size
    assert ops.wino4_wanted(64, 64, 128, 128, 2) and ops.wino4_wanted(64, 128, 64, 64, 1)
    monkeypatch.setattr(ops, "ROUTING", ops._Routing(env={"TGSR_WINO4": "0"}))
    assert not ops.wino4_wanted(64, 128, 128, 128, 16)


def test_upwino4_size_policy(monkeypatch):
    """The upBlocks that go to the F(4x4) form of the up-sample-aware kernel: by OUTPUT size (>= 64 x 64, whole 4 x 64 tiles),
    an even stage count and workgroup count; arguments are the low-resolution input's."""
    from tgsr_amd import ops
    monkeypatch.setattr(ops, "ROUTING", ops._Routing(env={}))
    assert ops.upwino4_wanted(64, 64, 128, 128, 16) and ops.upwino4_wanted(32, 64, 128, 128, 2) and ops.upwino4_wanted(64, 64, 256, 256, 1)
    assert not ops.upwino4_wanted(64, 64, 64, 64, 16) and not ops.upwino4_wanted(64, 64, 32, 32, 64)   # output below 256 x 256: mid-network
    assert ops.upwino4_wanted(64, 64, 128, 128, 1)             # a 256 x 256 output gives 256 workgroups per image
    assert not ops.upwino4_wanted(12, 64, 128, 128, 16)        # three stages: the register-fed form takes an even number
    assert not ops.upwino4_wanted(64, 32, 128, 128, 16) and not ops.upwino4_wanted(64, 64, 129, 128, 16) and not ops.upwino4_wanted(64, 64, 128, 144, 16)
    monkeypatch.setattr(ops.ROUTING, "wino4", False)
    assert not ops.upwino4_wanted(64, 64, 128, 128, 16)


def test_wgrad_kind_only_names_kernels_that_take_the_shape():
    """ops.conv3x3_wgrad_kind: the up-sample-aware Winograd-domain weight gradient has 64-row co-blocks only (round 4's relaxation
    to Cout % 32 would have sent upBlock(32, 16) to a kernel that refuses it - and its host planner into a divide by zero)."""
    from tgsr_amd import ops
    assert ops.conv3x3_wgrad_kind(64, 64, True) == "upwino" and ops.conv3x3_wgrad_kind(32, 128, True) == "upwino"
    assert ops.conv3x3_wgrad_kind(32, 32, True) == "direct" and ops.conv3x3_wgrad_kind(32, 96, True) == "direct"
    assert ops.conv3x3_wgrad_kind(32, 32, False) == "wino" and ops.conv3x3_wgrad_kind(64, 96, False) == "wino"
    assert ops.conv3x3_wgrad_kind(3, 64, False) == "direct" and ops.conv3x3_wgrad_kind(64, 64, True, winograd=False) == "direct"
    from tgsr_amd import _lib
    L = _lib.lib()
    assert L.tgsr_upwino_wgrad_ws_elems(2, 32, 32, 8, 8) == 0 and L.tgsr_upwino_wgrad_ws_elems(2, 32, 96, 8, 8) == 0
    assert L.tgsr_upwino_wgrad_ws_elems(2, 32, 64, 8, 8) > 0


def test_split_form_eligibility_is_host_arithmetic():
    """Which shapes the discriminator GEMMs take on the bf16 pipe (profiles/HISTORY.md 3.18) - pure host logic in the library, no GPU needed:
    whole 16-deep K-chunks (forward: always for the 4x4 form; data gradient: Cout % 4 == 0; 3x3: 9 C % 16 == 0), a weight gradient
    whose output pixels come in whole chunks, the image layer's data gradient on its own kernel, everything off with the switch."""
    from tgsr_amd import _lib
    L = _lib.lib()
    was = L.tgsr_dconv_set_split(1)
    try:
        f4, f3 = L.tgsr_conv4x4s2_split_form, L.tgsr_conv3x3_gemm_split_form
        assert f4(0, 32, 64, 128, 128, 128) == 1 and f4(1, 32, 64, 128, 128, 128) == 1 and f4(2, 32, 64, 128, 128, 128) == 1
        assert f4(0, 2, 3, 16, 16, 8) == 1                      # forward: K = 16 Cin, always whole chunks
        assert f4(1, 2, 3, 16, 16, 8) == 0                      # the image layer's data gradient has its own kernel
        assert f4(1, 2, 6, 8, 8, 10) == 0                       # 4 Cout = 40: a partial chunk
        assert f4(2, 1, 40, 4, 4, 33) == 0 and f4(2, 3, 8, 8, 12, 16) == 0      # 4 and 24 output pixels per image
        assert f4(2, 2, 64, 8, 8, 128) == 1                     # 16 output pixels per image
        assert f4(0, 2, 8, 7, 8, 8) == 0                        # odd height: the 4x4 form itself refuses it
        assert f3(0, 2, 256, 4, 4, 256) == 1 and f3(1, 2, 256, 4, 4, 256) == 1 and f3(2, 2, 256, 4, 4, 256) == 1
        assert f3(0, 3, 300, 5, 7, 260) == 0 and f3(1, 3, 300, 5, 7, 260) == 0 and f3(2, 3, 300, 5, 7, 260) == 0
        assert f3(0, 1, 48, 4, 8, 80) == 1 and f3(1, 1, 48, 4, 8, 80) == 1 and f3(2, 1, 48, 4, 8, 80) == 1
        assert f4(0, 1 << 20, 64, 128, 128, 128) == 0           # a tensor beyond a 4 GB buffer descriptor
        assert L.tgsr_dconv_set_split(0) == 1
        assert f4(0, 32, 64, 128, 128, 128) == 0 and f3(0, 2, 256, 4, 4, 256) == 0
        # the workspace of the default form holds no weight images: slabs (+ the data gradient's fp32 class pack) only
        L.tgsr_dconv_set_split(1)
        assert L.tgsr_conv4x4s2_ws_elems(0, 32, 64, 128, 128, 128) == 1
        assert L.tgsr_conv4x4s2_ws_elems(1, 32, 64, 128, 128, 128) == 16 * 64 * 128
        L.tgsr_dconv_set_split(3)
        assert L.tgsr_conv4x4s2_ws_elems(0, 32, 64, 128, 128, 128) == 128 * 1024 * 3 // 2
    finally:
        L.tgsr_dconv_set_split(was)
