"""Counterpart of the reference's inference caller (condGANTrainer.gen_exampleSRHL, trainer_objective.py:55-165,
and prepare_datablur, datasets.py:71-109): batch preparation, the text-enc -> G_SR_NET_low -> NetG_highweight
wiring and the uint8 epilogue.  Orchestration only; all arithmetic is in the modules' HIP kernels.
"""
import os

import numpy as np

import torch

from .miscc.config import cfg
from . import custom_ops as C
from . import ops
from .model import RNN_ENCODER


def sort_by_caption_length(captions, cap_lens, *per_sample):
    """datasets.py:74-96: sort the batch by caption length, descending (pack_padded_sequence order)."""
    lens, idx = torch.sort(cap_lens, 0, True)
    return (captions[idx], lens) + tuple(t[idx] for t in per_sample) + (idx,)


def caption_mask(captions, num_words):
    """trainer_objective.py:136-140: mask = (captions == 0) cropped to the longest caption of the batch."""
    # one comparison kernel on the cropped view (its output is contiguous: no copy behind it)
    return (captions[:, :num_words] if captions.size(1) > num_words else captions) == 0


def to_uint8(img):
    """trainer_objective.py:153-155: round(clip((x + 1) * 127.5, 0, 255)) as a uint8 numpy array.  Device tensors are
    converted on the GPU (tgsr_to_uint8, byte-identical) so that 4x fewer bytes cross PCIe; host tensors use the
    reference's numpy expression."""
    if img.is_cuda:
        from . import custom_ops as C
        return C.to_uint8(img.detach()).cpu().numpy()
    a = img.detach().cpu().numpy()
    return np.round(np.maximum(0, np.minimum(255, (a + 1.0) * 127.5))).astype(np.uint8)


# where NetG_highweight's stream forks off: at the start of the step (0, the default) or behind the text tail (1).  Measured
# neutral (round 4, hipGraph replays: bf16 31.3 k vs 31.3 k images/s, fp32 10.26 k vs 10.36 k): the step is G_SR_NET_low's
# dependent chain either way, the other branch's kernels competing with the recurrence / text tail do not bound it.
GH_AFTER_TEXT = int(os.environ.get("TGSR_GH_AFTER_TEXT", "-1"))     # -1: by context (see SRPipeline._forward); 0 first, 1 behind the text
                                                                     # tail, 2 issued there but dependent on the start only, 3 last
# fp32 path: NetG_highweight's 5x5 + tanh convolutions on the side stream, only `+ a * SRb` behind G_SR_NET_low (0: the six
# stand-alone heads in the reference's order)
SPLIT_HEADS = os.environ.get("TGSR_SPLIT_HEADS", "1") != "0"


def crop_words(out, num_words):
    """Views of a step's outputs cropped to `num_words` caption columns: what the reference's T_max-sized tensors hold
    (util.py:250-253: words_emb [B, nef, T_max]; trainer_objective.py:136-140: mask [B, T_max]; attention maps
    [B, T_max, r, r]).  The columns cut off are zero (words_emb, attention weights) or masked."""
    T = int(num_words)
    if out["words_emb"].size(2) == T:
        return out
    res = dict(out)
    res["words_emb"] = out["words_emb"][:, :, :T]
    res["mask"] = out["mask"][:, :T]
    res["att"] = [a[:, :T] for a in out["att"]]
    return res


class _ProjWithPack(list):
    """The word projections of a step (a list, one [B, idf, 32] tensor per attention module) that also carry the
    reduced-precision attention pack of tgsr_text_tail_lp_fwd (`.att_pack`) for LpExecutor.low."""

    def __init__(self, proj, att_pack):
        super().__init__(proj)
        self.att_pack = att_pack


def _host_lens(cap_lens):
    return [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]


def distinct_streams(n, dev, avoid=()):
    """n HIP streams whose handles differ from each other and from `avoid`.  `torch.cuda.Stream()` hands out the 32 streams of a
    per-device pool round robin: deep into a long process two "new" stream objects - or a new one and torch's graph-capture
    stream, or the stream a caller is on - can be the SAME hip stream, and a fork / join captured between them is then not the
    topology the code describes (which of them coincide depends on how many streams the process has asked for so far)."""
    seen = {int(h) for h in avoid}
    out = []
    for _ in range(96):
        if len(out) == n:
            break
        st = torch.cuda.Stream(device=dev)
        if int(st.cuda_stream) not in seen:
            seen.add(int(st.cuda_stream))
            out.append(st)
    if len(out) != n:
        raise RuntimeError("could not obtain %d distinct HIP streams" % n)
    return out


class SRPipeline:
    """The three networks of the SR path, built like trainer_objective.py:62-99: TREE.BRANCH_NUM == 4 selects the x8
    generators of model.py, anything else the x16 ones of models16.py (trainer_objective.py:74-87)."""

    def __init__(self, n_words, device="cuda", low="lr", overlap=True, dtype="fp32", branch_num=None):
        """dtype: "fp32" (the parity path: fp32 NCHW kernels) or "bf16" / "f16" (BASELINE configs[4]: the two generators
        run on reduced-precision channels-last images through tgsr_amd.lp_pipeline.LpExecutor; inputs, the text encoder
        and every returned tensor stay fp32).  branch_num: None = cfg.TREE.BRANCH_NUM, read at construction like the
        reference does."""
        self.dtype = dtype
        self._lp = None
        self.branch_num = int(cfg.TREE.BRANCH_NUM if branch_num is None else branch_num)
        if self.branch_num == 4:                                        # trainer_objective.py:74-87
            from .model import G_SR_NET_low, NetG_highweight
        else:
            from .models16 import G_SR_NET_low, NetG_highweight
        self.text_encoder = RNN_ENCODER(n_words, nhidden=cfg.TEXT.EMBEDDING_DIM)
        self.netGL = G_SR_NET_low()
        self.netGH = NetG_highweight(weightmap=False, low=low)
        self.device = torch.device(device)
        # NetG_highweight's trunk does not depend on G_SR_NET_low (only its three heads add the low-frequency
        # images): run it on a second HIP stream so the two networks' small layers and kernel tails overlap
        self.overlap = overlap
        self._side = None
        for m in (self.text_encoder, self.netGL, self.netGH):
            m.to(self.device)
            m.eval()
        if dtype not in ("fp32", "f32", None):
            from .lp_pipeline import LpExecutor
            self._lp = LpExecutor(self.netGL, self.netGH, dtype)

    def load_state_dicts(self, sd_E=None, sd_GL=None, sd_GH=None):
        """strict for E and GL; the x8 GH tolerates only a missing `a` (never saved by the reference, model.py:246-248);
        the x16 GH registers `a` as a parameter (models16.py:126) and loads it."""
        if sd_E is not None:
            self.text_encoder.load_state_dict(sd_E, strict=True)
        if sd_GL is not None:
            self.netGL.load_state_dict(sd_GL, strict=True)
        if sd_GH is not None:
            if self.branch_num == 4:
                sd_GH = {k: v for k, v in sd_GH.items() if k != "a"}
            self.netGH.load_state_dict(sd_GH, strict=True)
        return self

    # ------------------------------------------------------------------ throughput: stream lanes
    def lanes(self, n=3):
        """n HIP streams to alternate consecutive, independent batches over:

            for k, batch in enumerate(batches):
                with torch.cuda.stream(lanes[k % n]):
                    out = pipe(*batch)          # consume `out` on that stream (or synchronise it first)

        The ~60 dependent launches of one forward leave the GPU idle ~10 % of the time; another batch's kernels fill
        those gaps (B=16: 1.74 -> 1.57 ms per step with three lanes).  Each lane keeps its own activation buffers.
        Weight packs and the per-token gate table are built on first use, on whatever stream that forward runs: do
        one warm-up forward and `torch.cuda.synchronize()` after loading / changing weights before fanning out."""
        return distinct_streams(n, self.device, avoid=[torch.cuda.current_stream(self.device).cuda_stream])

    # ------------------------------------------------------------------ hipGraph replay (BASELINE config 5)
    @torch.no_grad()
    def capture(self, captions, cap_lens, LR, LRb, warmup=3, lanes=1):
        """Capture one forward (both streams, ~60 launches) into a hipGraph bound to static input buffers (`lanes` > 1:
        that many independent batches as parallel branches of the graph, see GraphedStep).
        `replay(captions, cap_lens, LR, LRb)` then copies a new batch in and relaunches the whole step with one call.
        The captured step does not depend on the caption lengths (Q7: util.py:250-253, trainer_objective.py:136-140 make
        T_max and the mask follow each batch): inside the graph the text runs at the full caption width
        (cfg.TEXT.WORDS_NUM columns), lengths and mask are device buffers, and the outputs are cropped to the batch's
        longest caption on the host side of the boundary.  Only the shapes (batch, caption width, LR size) are fixed."""
        self._graphed = GraphedStep(self, captions, cap_lens, LR, LRb, warmup=warmup, lanes=lanes)
        return self._graphed.cropped()

    @torch.no_grad()
    def replay(self, captions=None, cap_lens=None, LR=None, LRb=None, num_words=None):
        """Relaunch the captured step, on a new batch when given (same shapes; ANY caption lengths).  The returned
        tensors are views of the graph's static outputs: consume them before the next replay."""
        return self._graphed.replay(captions, cap_lens, LR, LRb, num_words=num_words)

    @torch.no_grad()
    def capture_lanes(self, n, captions, cap_lens, LR, LRb):
        """n independent captured steps, each with its own static buffers and its own stream, for
        `lanes[k % n].replay(...)`: one host call per step (0.17 ms of host time instead of the ~1.3 ms of ~60 eager
        launches).  Measured on ROCm 7.2 the replays of different graphs do NOT overlap the way eager lanes do
        (B=16: 1.70-1.75 ms per step against 1.57 ms with three eager lanes), so bench.py uses ONE graph of several
        lanes (GraphedStep(lanes=...)); this is the option for a host that cannot keep up with the enqueue rate."""
        return [GraphedStep(self, captions, cap_lens, LR, LRb, stream=torch.cuda.Stream(device=self.device))
                for _ in range(n)]

    def invalidate_caches(self):
        """Drop every packed-weight / folded-BatchNorm cache of the three networks (util.invalidate_caches) and of the
        reduced-precision executor: the next call re-derives them from the parameters."""
        from .util import invalidate_caches
        for m in (self.text_encoder, self.netGL, self.netGH):
            invalidate_caches(m)
        if self._lp is not None:
            self._lp.key = None

    @torch.no_grad()
    def __call__(self, captions, cap_lens, LR, LRb, num_words=None):
        """trainer_objective.py:134-146.  Returns the same tensors the reference loop produces.  Runs under the
        pipeline's device (the kernels launch on the CURRENT device's stream; ops refuse tensors that live elsewhere).

        cap_lens on the host (list / CPU tensor, what the reference's callers hold: util.py:239) -> words_emb, mask and the
        attention maps have T_max = max(cap_lens) columns, like the reference's.  cap_lens as a DEVICE tensor -> the
        length-independent form a hipGraph capture needs: no launch argument depends on the lengths' values, the text runs
        at the full caption width with zeros / mask bits behind every caption (bit-identical images: a padded column is
        masked for every sample, its softmax weight is exactly 0), and the outputs are cropped to `num_words` columns when
        the caller says how long the batch's longest caption is (None: left at the full width)."""
        if self.device.type == "cuda" and self.device.index is not None and \
                self.device.index != torch.cuda.current_device():
            with torch.cuda.device(self.device):
                out = self._forward(captions, cap_lens, LR, LRb)
        else:
            out = self._forward(captions, cap_lens, LR, LRb)
        return crop_words(out, num_words) if num_words is not None else out

    def _text_tail(self, words_embs, sent_emb, captions):
        """(word projections, CA_NET outputs, mask) for G_SR_NET_low: ONE launch (ops.text_tail) where the shapes allow -
        before, word_project + a dozen-microsecond ca_net on the second stream behind an event + the comparison kernel of
        the mask + its bool -> uint8 cast.  c_code is not computed (the generators discard it, model.py:51-52); outside a
        hipGraph capture its normals are still drawn, so torch's generator advances as the reference's does
        (util.py:388-396)."""
        GL = self.netGL
        atts = GL.attention_modules() if hasattr(GL, "attention_modules") else None
        fc = GL.ca_net.fc
        T = words_embs.size(2)
        idf = atts[0].conv_context.out_channels if atts else 0
        if (atts is None or GL.training or not words_embs.is_cuda or len(atts) > 4 or T > 32 or fc.in_features % 16
                or captions.dtype != torch.int64 or captions.size(1) < T or idf < 32 or idf % 32
                or any(a.conv_context.out_channels != idf for a in atts)):
            return None, GL.ca_net(sent_emb), caption_mask(captions, T)
        ws = [a.conv_context.weight.detach() for a in atts]
        pack = None
        if self._lp is not None and self._lp.fuse_attention and idf == 32:
            # the reduced-precision generators attend inside the kernels that produce h: the projections also leave as the
            # MFMA fragments (and the mask as the packed rows) those kernels read
            src, mu, logvar, m8, pack = C.text_tail_lp(words_embs, ws, sent_emb, fc.weight.detach(), fc.bias.detach(),
                                                       GL.ca_net.c_dim, captions, self._lp.dtype == torch.bfloat16)
        else:
            src, mu, logvar, m8 = C.text_tail(words_embs, ws, sent_emb, fc.weight.detach(), fc.bias.detach(),
                                              GL.ca_net.c_dim, captions)
        if not torch.cuda.is_current_stream_capturing():
            torch.empty(sent_emb.shape[0], GL.ca_net.c_dim, dtype=torch.float32, device=sent_emb.device).normal_()
        proj = list(src.unbind(0))
        if pack is not None:
            proj = _ProjWithPack(proj, pack)
        return proj, (None, mu, logvar), m8.view(torch.bool)

    def _forward(self, captions, cap_lens, LR, LRb):
        # (the reference passes init_hidden()'s zero state, trainer_objective.py:134; the HIP recurrence starts from zero
        # by construction, so the two fill kernels of building that state are not launched)
        hidden = None
        if self._lp is not None:
            ex = self._lp
            ex.refresh()
            bufs = ex._buffers(LR.shape[0], LR.shape[2], LR.shape[3], LR.device)
            trunk = lambda: ex.high_trunk(bufs, LR, LRb)                                                    # noqa: E731
            low = lambda sent, words, mask, ca, proj: ex.low(bufs, LR, sent, words, mask, ca=ca, proj=proj,  # noqa: E731
                                                             defer_heads=True)
            heads = ex.high_heads
        elif SPLIT_HEADS and not self.netGH.training:
            # NetG_highweight's heads are `tanh(conv5x5(out_k)) + a * SRb_k` (model.py:280-297): only the addition needs
            # G_SR_NET_low.  The convolutions (134 us at batch 16, 92 of them at 256^2) join the trunk on the side stream;
            # what is left behind G_SR_NET_low's last head on the step's critical path is one axpy launch over the three
            # images (~6 us) instead of the 256^2 head.
            trunk = lambda: self.netGH.tanh_heads(self.netGH.trunk(LR, LRb))                                 # noqa: E731
            low = lambda sent, words, mask, ca, proj: self.netGL(LR, sent, words, mask, ca=ca, proj=proj)    # noqa: E731
            heads = self.netGH.finish_heads
        else:
            trunk = lambda: self.netGH.trunk(LR, LRb)                                                        # noqa: E731
            low = lambda sent, words, mask, ca, proj: self.netGL(LR, sent, words, mask, ca=ca, proj=proj)    # noqa: E731
            heads = self.netGH.heads
        feats = side = None

        def fork_trunk(start=None):
            nonlocal feats, side
            main_ = torch.cuda.current_stream(LR.device)
            if self._side is None:
                self._side = {}
            side = self._side.get(main_.cuda_stream)     # one side stream per calling stream (callers may alternate
            if side is None:                             # lanes to overlap consecutive steps)
                side = self._side[main_.cuda_stream] = distinct_streams(1, LR.device, avoid=[main_.cuda_stream])[0]
            if start is not None:
                side.wait_event(start)                   # LR / LRb were ready when `start` was recorded
            else:
                side.wait_stream(main_)                  # LR / LRb are ready on the main stream
            with torch.cuda.stream(side):                # the trunk needs neither the text encoder nor G_SR_NET_low
                feats = trunk()
            return main_

        main = None
        start = None
        # where NetG_highweight's branch is ISSUED.  Eager: first (0), so that the host launches it before G_SR_NET_low's chain.
        # Inside a hipGraph capture: last (3) - the captured graph is a DAG, but ROCm 7.2's executor starts the branch that was
        # created second only when a whole segment of the first one has been submitted: with the trunk first, its 13 small
        # kernels run ALONE for ~110 us (bf16) / ~130 us (fp32) before the recurrence - the head of the step's dependent chain -
        # starts (gpurun_out timelines, profiles/HISTORY.md 3.16); created last, the trunk runs in the shadow of G_SR_NET_low.
        mode = GH_AFTER_TEXT
        if mode < 0:
            mode = 3 if (LR.is_cuda and torch.cuda.is_current_stream_capturing()) else 0
        if self.overlap and LR.is_cuda and mode == 0:
            main = fork_trunk()
        elif self.overlap and LR.is_cuda and mode == 2:
            start = torch.cuda.Event()
            start.record(torch.cuda.current_stream(LR.device))
        words_embs, sent_emb = self.text_encoder(captions, cap_lens, hidden)
        proj, ca, mask = self._text_tail(words_embs, sent_emb, captions)
        if self.overlap and LR.is_cuda and mode in (1, 2):
            # NetG_highweight's branch has slack (G_SR_NET_low's dependent chain is the step): forked BEHIND the text tail, the
            # recurrence and the tail - the head of that chain - run without its kernels competing for the CUs (2: its nodes are
            # ISSUED behind the text tail but depend on the step's start only)
            main = fork_trunk(start)
        gh_last = (self.overlap and LR.is_cuda and mode == 3 and main is None)
        if gh_last:
            start = torch.cuda.Event()
            start.record(torch.cuda.current_stream(LR.device))
        res = low(sent_emb, words_embs, mask, ca, proj)
        if gh_last:
            main = fork_trunk(start)
        fake_imgL, attention_maps, mu, logvar = res[:4]
        pend = res[4:]       # lp path: (partial sums of the low-frequency heads still to be combined,) - heads() finishes them
        if side is not None:
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                for f in feats:
                    f.record_stream(main)                # allocated on the side stream, consumed on the main one
        else:
            feats = trunk()
        fine_im = heads(feats, fake_imgL, *pend)
        return {"words_emb": words_embs, "sent_emb": sent_emb, "mask": mask, "fake": fake_imgL,
                "att": attention_maps, "mu": mu, "logvar": logvar, "fine": fine_im}


class GraphedStep:
    """One captured inference step of an SRPipeline: a hipGraph + the static input / output tensors it is bound to.
    With `stream` the replay (and the copy of new inputs) runs on that stream.

    The graph is independent of the caption lengths: captions [B, W], lengths (int32 [B]), LR and LRb are static DEVICE
    buffers refreshed by one copy launch per replay; inside the graph the text runs at the full width W (the recurrence
    reads each sample's length from memory and stops there, the word projection / attention see zero words behind a
    caption and a mask bit for every padded column, so the softmax weights of those columns are exactly 0); `replay`
    crops words_emb / mask / attention maps to the new batch's longest caption, which is what the reference's per-batch
    T_max produces (util.py:250-253, trainer_objective.py:136-140).

    `lanes` > 1 captures that many INDEPENDENT batches (each with its own static inputs, outputs and activation buffers)
    as parallel branches of the one graph: a replay then runs `lanes` steps whose kernels interleave on the device the
    way eager stream lanes do, without the host cost of ~60 launches per step and without relying on separate graph
    launches overlapping (measured: they do not).  `inputs` / `out` are then lists of length `lanes`."""

    @torch.no_grad()
    def __init__(self, pipe, captions, cap_lens, LR, LRb, stream=None, warmup=3, lanes=1):
        dev = LR.device
        self.stream = stream
        self.lanes = max(1, int(lanes))
        if torch.is_tensor(cap_lens) and cap_lens.is_cuda:
            lens0 = cap_lens.to(torch.int32)
            self.num_words = [captions.size(1)] * self.lanes        # unknown on the host: no crop until a replay says
        else:
            host = _host_lens(cap_lens)
            lens0 = torch.tensor(host, dtype=torch.int32).to(dev)
            self.num_words = [max(host)] * self.lanes
        sets = [(captions.to(torch.int64).clone(), lens0.clone(), LR.clone(), LRb.clone()) for _ in range(self.lanes)]
        lpx = getattr(pipe, "_lp", None)
        bufs = None
        if lpx is not None:                              # reduced-precision path: every lane is bound to its own set of
            lpx.refresh()                                # activation images, allocated (zeroed) outside the graph
            bufs = [lpx.alloc(LR.shape[0], LR.shape[2], LR.shape[3], dev) for _ in range(self.lanes)]
            lpx.force_bufs = bufs[0]
        self.bufs = bufs
        overlap = pipe.overlap
        try:
            # the capture stream, the warm-up stream and the lanes' branch streams: all different hip streams, none of them the
            # caller's (distinct_streams: torch's stream pool wraps around)
            cur = torch.cuda.current_stream(dev)
            cap, s, *branch = distinct_streams(self.lanes + 1, dev, avoid=[cur.cuda_stream, torch.cuda.default_stream(dev).cuda_stream])
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):                   # weight packs, caches and the allocator warm up outside the graph
                for _ in range(warmup):
                    pipe(*sets[0])
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            outs = []
            if self.lanes > 1:
                # every lane is ONE chain: the lanes provide the concurrency the GL / GH stream split provides inside a
                # single step.  (Two streams per lane are a fork nested inside a forked branch: the process segfaults on that
                # topology on ROCm 7.2 - re-checked in round 4 with all streams distinct, §3.15: it still does.)
                pipe.overlap = False
            with torch.cuda.graph(self.graph, stream=cap):
                main = torch.cuda.current_stream(dev)
                for st in branch:                        # fork EVERY branch before any lane's work is captured: a
                    st.wait_stream(main)                 # wait recorded behind lane 0's kernels would make the branches
                for k in range(self.lanes):              # depend on all of lane 0 (they would run after it, not beside it)
                    st = main if k == 0 else branch[k - 1]
                    with torch.cuda.stream(st):
                        if lpx is not None:
                            lpx.force_bufs = bufs[k]
                        outs.append(pipe(*sets[k]))      # lengths = the device tensor: the length-independent form
                for st in branch:
                    main.wait_stream(st)                 # join
        finally:
            if lpx is not None:
                lpx.force_bufs = None
            if self.lanes > 1:
                pipe.overlap = overlap
        # The streams of the capture stay referenced as long as the graph does (CUDA documents a graph as independent of the
        # streams it was captured from; nothing says so for HIP).
        self._capture_streams = [cap, s] + branch
        self.inputs = sets[0] if self.lanes == 1 else sets
        self.out = outs[0] if self.lanes == 1 else outs

    def cropped(self):
        """The static outputs, cropped to the longest caption of the batch each lane last ran."""
        if self.lanes == 1:
            return crop_words(self.out, self.num_words[0])
        return [crop_words(o, t) for o, t in zip(self.out, self.num_words)]

    @torch.no_grad()
    def replay(self, captions=None, cap_lens=None, LR=None, LRb=None, num_words=None):
        """Copy a new batch (same shapes, any caption lengths) into the static buffers and relaunch the step.  Returns
        views of the static outputs cropped to the batch's longest caption: consume them (on `stream`, or after
        synchronising it) before this lane's next replay.

        cap_lens: host list / CPU tensor (T_max = its maximum; its device copy is cached per length vector), or a device int32 / int64
        tensor (copied on the device; pass `num_words` = the batch's longest caption to have the outputs cropped, else they
        keep the width they had), or None (captions unchanged).  With `lanes` > 1 each argument is a list of `lanes`
        entries (or None: keep the buffers' contents)."""
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                return self._go(captions, cap_lens, LR, LRb, num_words)
        return self._go(captions, cap_lens, LR, LRb, num_words)

    def _go(self, captions, cap_lens, LR, LRb, num_words):
        pairs = []
        sets = [self.inputs] if self.lanes == 1 else self.inputs
        one = self.lanes == 1
        if captions is not None and cap_lens is None:
            # (also what an old positional call replay(captions, LR, LRb) of the three-argument signature lands on: its LR would
            # have been taken for the lengths) - new captions with the previous batch's lengths would crop real words away
            raise ValueError("replay: new captions need their cap_lens (host list / tensor, or a device tensor + num_words)")
        for k, dsts in enumerate(sets):
            ln = cap_lens if (one or cap_lens is None) else cap_lens[k]
            nw = num_words if (one or num_words is None or isinstance(num_words, int)) else num_words[k]
            if ln is not None:
                if torch.is_tensor(ln) and ln.is_cuda:
                    if ln.dim() != 1 or ln.numel() != dsts[1].numel() or ln.dtype not in (torch.int32, torch.int64):
                        raise ValueError("replay: device cap_lens must be a 1-D int32 / int64 tensor of %d entries, got %s %s"
                                         % (dsts[1].numel(), tuple(ln.shape), ln.dtype))
                    if ln.dtype != torch.int32:
                        ln = ln.to(torch.int32)
                    # lengths the host does not know: crop to what the caller says, else not at all (never to the PREVIOUS
                    # batch's longest caption)
                    self.num_words[k] = int(nw) if nw is not None else dsts[0].size(1)
                else:
                    host = _host_lens(ln)
                    if min(host) < 1 or max(host) > dsts[0].size(1):
                        raise ValueError("replay: caption lengths %s outside [1, %d]" % (host, dsts[0].size(1)))
                    if len(host) != dsts[1].numel():
                        raise ValueError("replay: %d caption lengths for a captured batch of %d" % (len(host), dsts[1].numel()))
                    self.num_words[k] = max(host) if nw is None else int(nw)
                    # a device copy of this length vector, cached per vector (ops._lens_on_device: one small blocking H2D
                    # copy the first time a vector is seen, none afterwards); it goes into the static buffer with the
                    # other inputs by the replay's one copy launch
                    ln = ops._lens_on_device(tuple(host), dsts[1].device)
            elif nw is not None:
                self.num_words[k] = int(nw)
            lane = lambda a: a if (one or a is None) else a[k]                 # noqa: E731
            for dst, src in zip(dsts, (lane(captions), ln, lane(LR), lane(LRb))):
                if src is not None and src is not dst:
                    pairs.append((dst, src))
        fast = [(d, s_) for d, s_ in pairs
                if s_.is_cuda and s_.dtype == d.dtype and s_.shape == d.shape and s_.is_contiguous() and s_.device == d.device]
        if len(fast) == len(pairs):
            for i in range(0, len(fast), 16):            # the new inputs of every lane: one launch per 16 buffers
                C.multi_copy([d for d, _ in fast[i:i + 16]], [s_ for _, s_ in fast[i:i + 16]])
        else:
            for dst, src in pairs:
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.cropped()
