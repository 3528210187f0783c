"""SR generator training step (the loop the reference never shipped - SURVEY section 3.3).

What the reference pins down and this harness uses: the loss functions and their conventions (`MSE` losses.py:779,
`KL_loss` :806, `generator_loss` / `discriminator_loss` :290-391 when a discriminator is supplied), labels
(`prepare_labels`, trainer_objective.py:43-53), Adam(lr 2e-4, betas (0.5, 0.999)) (config.py:37-38,
pretrain_DAMSM.py:270), the EMA helpers `copy_G_params` / `load_params` (miscc/utils.py:467-474), BatchNorm in
training mode.  What it does NOT pin down (no caller exists): loss weights, update order, the discriminator
architecture (no class anywhere in the reference) and the Inception image encoder (third-party weights).  This
harness therefore trains the two generators on the pixel + KL terms,
    errG = MSE(fake_imgL, HR pyramid) + MSE(fine_im, HR pyramid) + KL(mu, logvar),
and takes the adversarial / DAMSM terms only when the caller supplies `netsD` / `image_encoder` (the DAMSM term
`words_loss + sent_loss` on `image_encoder(fine_im[-1])` is differentiable through the HIP DAMSM backward kernel).
Every forward and backward kernel of the generators is HIP (tgsr_amd.autograd); the text encoder is frozen (eval).
`DAMSMTrainer` is the counterpart of pretrain_DAMSM.py (text encoder + CNN_ENCODER heads on the matching losses).
Data parallel: gradients live in one flat bucket, one all-reduce per step (tgsr_amd.parallel.FlatGradBucket).
"""
import contextlib
import os
import time

import torch

from .miscc import losses
from .miscc.config import cfg
from .miscc.utils import copy_G_params, load_params  # noqa: F401  (miscc/utils.py:467-474: the generator EMA helpers)
from .model import CNN_ENCODER, G_SR_NET_low, NetG_highweight, RNN_ENCODER
from .parallel import FlatGradBucket
from .trainer import caption_mask, distinct_streams


def prepare_labels(batch_size, device):
    """trainer_objective.py:43-53."""
    return (torch.ones(batch_size, device=device), torch.zeros(batch_size, device=device),
            torch.arange(batch_size, device=device))


# eager G/D steps before the discriminator updates are captured (allocator, streams and Adam state warm)
GRAPH_D_WARMUP = 3
# eager steps before the generators' update is captured (the same, plus every weight pack of the step in the PackCache)
GRAPH_G_WARMUP = 3
# TGSR_GRAPH_G=auto: steps of each form (eager, replayed) the trainer times before it settles on the faster one
GRAPH_G_TRIALS = 3
# ... and the number of steps after which that choice has been made (warm-up, eager trials, the capturing step, replayed trials)
GRAPH_G_SETTLED = GRAPH_G_WARMUP + 2 * GRAPH_G_TRIALS + 1


class SRTrainer:
    def __init__(self, n_words, device="cuda", low="lr", lr=None, ema_decay=0.999, image_encoder=None,
                 discriminators=False, d_lr=None, gather_negatives=None):
        """image_encoder: optional frozen module image [B,3,256,256] -> (region features [B,nef,17,17], cnn_code
        [B,nef]) (a CNN_ENCODER with its trunk): adds the DAMSM ranking term of generator_loss (losses.py:375-386)
        on the finest image, x TRAIN.SMOOTH.LAMBDA.
        discriminators: True builds one discriminator per output scale (model.D_NET64 / 128 / 256 for the x8
        generators' 64 / 128 / 256 images) or pass a list of modules exposing COND_DNET / UNCOND_DNET; `step()` then
        alternates the discriminator update (discriminator_loss, losses.py:290-316) and the generator update
        (generator_loss :351-391 + MSE + KL), each discriminator with its own Adam(DISCRIMINATOR_LR, betas (0.5, 0.999))
        and flat gradient bucket.  The reference defines the two loss functions but neither the discriminators nor the
        loop: architecture and update order (D first, then G on the same fake images, as in the AttnGAN
        trainer TGSR was forked from) are the build's declaration."""
        self.device = torch.device(device)
        # data parallel: the DAMSM ranking term on the gathered global batch (parallel.GATHER_NEGATIVES, default on) or per shard
        from . import parallel as _par
        self.gather_negatives = _par.GATHER_NEGATIVES if gather_negatives is None else bool(gather_negatives)
        # the generators' weight gradients run on a side stream beside the data-gradient chain while a step's backward
        # is in flight (12.7 -> 11.7 ms per step at B=16: the small layers' weight-gradient kernels and the slab sums
        # fill a fraction of the CUs); TGSR_WGRAD_SIDE=0 keeps everything on one stream
        from . import autograd as _ag
        self._packs = _ag.PackCache() if (self.device.type == "cuda" and os.environ.get("TGSR_PACK_CACHE", "1") != "0") else None
        # (distinct_streams: torch hands out pool streams round robin - two "new" streams can be the same hip stream)
        cur = [torch.cuda.current_stream(self.device).cuda_stream] if self.device.type == "cuda" else []
        self._wside = distinct_streams(1, self.device, avoid=cur)[0] \
            if self.device.type == "cuda" and os.environ.get("TGSR_WGRAD_SIDE", "1") != "0" else None
        self.image_encoder = image_encoder
        self.text_encoder = RNN_ENCODER(n_words, nhidden=cfg.TEXT.EMBEDDING_DIM).to(self.device).eval()
        for p in self.text_encoder.parameters():
            p.requires_grad = False
        self.netGL = G_SR_NET_low().to(self.device).train()
        self.netGH = NetG_highweight(weightmap=False, low=low).to(self.device).train()
        self.params = list(self.netGL.parameters()) + list(self.netGH.parameters())
        # (BatchNorm's running statistics ride the gradient bucket's all-reduce: identical on every rank, parallel.py.)
        # Bucket layout [NetG_highweight | G_SR_NET_low | buffers]: backward runs through NetG_highweight first (its nodes
        # are the younger ones), so its gradients are final while G_SR_NET_low's backward still runs - that range goes out
        # early (`_fire_early`), the rest with the step's closing all-reduce: two collectives, the first under backward.
        gh_params = [p for p in self.netGH.parameters() if p.requires_grad]
        self._bucket_bufs = list(self.netGL.buffers()) + list(self.netGH.buffers())
        self._gh_params = gh_params
        self._early_n = len(gh_params)                                   # parameters of the early range
        self._early_hi = None                                            # ... = flat[0:_early_hi], set once the bucket exists
        self._early_on = os.environ.get("TGSR_EARLY_ALLREDUCE", "1") != "0"
        self._early_left, self._early = -1, None
        self._comm = distinct_streams(1, self.device, avoid=cur + ([self._wside.cuda_stream] if self._wside is not None else []))[0] \
            if self.device.type == "cuda" else None
        for p in gh_params:
            p.register_post_accumulate_grad_hook(self._gh_grad_done)
        self._fused_adam = self.device.type == "cuda" and os.environ.get("TGSR_FUSED_ADAM", "1") != "0"
        # The generators' update - forward, losses, backward, Adam, re-pack, EMA - holds no host decision once the text encoder has
        # produced the embeddings: it is replayed from hipGraphs (one per batch shape), in segments with the gradient all-reduce
        # BETWEEN them, so the replayed step also exists with more than one rank (`_capture_g`).  TGSR_GRAPH_G=0: eager.
        # Default ("auto"): replay where it measured at least as fast as the eager step on an idle host - the G/D alternation
        # (20.6 vs 20.9 ms); the generator-only step (10.0-10.3 vs 9.9 ms) and the step with the Inception encoder (38.8 vs 35.2 ms:
        # ~1 000 more small kernels, each dependent node of a replay costs a few microseconds more than a launch from a host that
        # keeps ahead) stay eager: the steps are DEVICE-bound (kernel time 12.1 ms, busy 9.6 ms of a 9.9 ms generator step), so
        # taking the host out buys nothing there.  TGSR_GRAPH_G=1 replays all of them (a loaded or slower host: under rocprofv3
        # the replayed generator step runs 10.7 ms, the eager one 15.7), 0 none.
        # Which of the two wins is a property of the HOST the process lands on (round 6: the same tree ran the eager generator
        # step in 9.9 ms on one box and 12.0 ms on another, whose replays would have taken 10.3), so "auto" MEASURES: after the
        # warm-up it times GRAPH_G_TRIALS eager steps, captures, times as many replayed ones and keeps the faster form (with
        # several ranks: the slowest rank's times, so that every rank takes the same one).  The rule above is only the form the
        # first steps take; assigning `_graph_g` by hand ends the measurement and pins the form.
        mode = os.environ.get("TGSR_GRAPH_G", "auto")
        self._auto = None
        self._graph_g = self.device.type == "cuda" and (mode == "1" or (mode == "auto" and bool(discriminators) and
                                                                         image_encoder is None))
        self._graph_capable = self.device.type == "cuda" and mode != "0"
        if mode == "auto" and self.device.type == "cuda":
            self._auto = {"eager_s": [], "replay_s": [], "t0": None, "form": None}
        self.graph_policy = {"mode": mode, "prior": "replay" if self._graph_g else "eager"}
        self._ggraphs, self._gsteps, self._ghyper, self._g_bump = {}, 0, None, None
        # TGSR_FLAT_ADAM (default on, HIP only): parameters and moments re-homed into flat buffers beside the flat gradient bucket,
        # the update ONE launch of tgsr::adam_flat_ (optim.FlatAdam; the same rule as torch.optim.Adam); 0 = torch's fused Adam
        self._flat_adam = self.device.type == "cuda" and os.environ.get("TGSR_FLAT_ADAM", "1") != "0"
        self._g_lr = lr or cfg.TRAIN.GENERATOR_LR
        self.ema_decay = ema_decay
        self.avg_param_G = copy_G_params(self.netGL) + copy_G_params(self.netGH)
        self.netsD, self.optsD, self.bucketsD = [], [], []
        self._graph_d, self._dgraphs, self._dsteps, self._d_bump = False, [], 0, []
        if discriminators:
            from . import model
            self.netsD = list(discriminators) if not isinstance(discriminators, bool) else \
                [model.D_NET64(), model.D_NET128(), model.D_NET256()]
            # A discriminator's update - forward on (real, fake.detach()), loss, backward, Adam - is a closed piece of device work
            # with no host decision in it: replayed from a hipGraph per discriminator once the step has run `GRAPH_D_WARMUP` times
            # (with more than one rank as two graphs around the bucket's all-reduce: `_capture_d_update`).  ~1 000 of a step's ~1 570
            # launches leave the host that way; the step was issued no faster than 21-25 ms (profiles/HISTORY.md 3.18).  TGSR_GRAPH_D=0: eager.
            self._graph_d = self.device.type == "cuda" and os.environ.get("TGSR_GRAPH_D", "1") != "0"
            self._dgraphs, self._dsteps, self._d_bump = [None] * len(self.netsD), 0, [None] * len(self.netsD)
            for d in self.netsD:
                d.to(self.device).train()
                self.bucketsD.append(FlatGradBucket(d.parameters(), buffers=d.buffers()).attach())
                # (fused: one pass over a discriminator's ~70 M parameters and their moments instead of the ~10 of the
                # multi-tensor form - 2.0 ms of a G/D step were Adam kernels running alone on the device)
                if self._flat_adam:
                    from .optim import FlatAdam
                    self.optsD.append(FlatAdam(self.bucketsD[-1].params, self.bucketsD[-1].flat, lr=d_lr or cfg.TRAIN.DISCRIMINATOR_LR,
                                               betas=(0.5, 0.999)))
                else:
                    self.optsD.append(torch.optim.Adam(d.parameters(), lr=d_lr or cfg.TRAIN.DISCRIMINATOR_LR, betas=(0.5, 0.999),
                                                       capturable=self._graph_d, fused=self._fused_adam))
        if self.netsD:
            # the generator loss runs the train-mode discriminators on the fake images once more (g_loss): their running
            # statistics move again, per rank, AFTER their own bucket's all-reduce - so they also ride the generators' bucket
            # and every rank leaves the step with the same discriminator buffers (a snapshot is the same file on every rank)
            self._bucket_bufs = self._bucket_bufs + [b for d in self.netsD for b in d.buffers()]
        self.bucket = FlatGradBucket(gh_params + list(self.netGL.parameters()), buffers=self._bucket_bufs).attach()
        self._early_hi = self.bucket.offsets[self._early_n][0] if self._early_n < len(self.bucket.params) else self.bucket.numel
        if self._flat_adam:
            from .optim import FlatAdam
            self.opt = FlatAdam(self.bucket.params, self.bucket.flat, lr=self._g_lr, betas=(0.5, 0.999))
        else:
            self.opt = torch.optim.Adam(self.params, lr=self._g_lr, betas=(0.5, 0.999), fused=self._fused_adam,
                                        capturable=self._graph_capable)
        # TGSR_COMM=direct: the buckets' closing all-reduce through the library's own RCCL communicator (tgsr_allreduce_flat,
        # parallel.RcclDirect) instead of torch.distributed's; the early range and the DAMSM gather stay on the process group
        self._rccl = None
        if os.environ.get("TGSR_COMM", "") == "direct" and self.device.type == "cuda":
            from .parallel import RcclDirect, dp_world
            if dp_world() > 1:
                self._rccl = RcclDirect.create()
                self._early_on = False
                for b in [self.bucket] + self.bucketsD:
                    b.comm = self._rccl
        taken = cur + ([self._wside.cuda_stream] if self._wside is not None else []) + \
            ([self._comm.cuda_stream] if self._comm is not None else [])
        self._dstreams = distinct_streams(len(self.netsD), self.device, avoid=taken) \
            if self.device.type == "cuda" and self.netsD and os.environ.get("TGSR_D_STREAMS", "1") != "0" else []
        taken = taken + [st.cuda_stream for st in self._dstreams]
        # TGSR_D_WGRAD_SIDE=1 (opt-in): each discriminator's weight gradients on a side stream of its own, beside its data-gradient
        # chain - the 256^2 discriminator's update is the longest dependent chain of a G/D step and a quarter of its backward
        # kernels are weight gradients nothing waits for until Adam.  Built, bit-identical (the gan / dp suites pass with it), and
        # SLOWER on this chip: 20.8 against 19.6 ms per G/D step, 31.9 against 30.3 with the ranking term (same box) - the three
        # updates already run side by side, and six streams compete for four hardware queues (GPU_MAX_HW_QUEUES=8 is worse
        # still: 43.6 ms).  Left off.
        self._dwside = distinct_streams(len(self._dstreams), self.device, avoid=taken) \
            if self._dstreams and os.environ.get("TGSR_D_WGRAD_SIDE", "0") == "1" else []
        taken = taken + [st.cuda_stream for st in self._dwside]
        # TGSR_ENC_EARLY=1 (opt-in): generator_loss's image encoder (CNN_ENCODER: ~190 launches of small GEMMs forward) reads the fake
        # image only - not the discriminators - so it can be issued BEFORE the discriminator updates, on a stream of its own (as a
        # hipGraph of its own when the step is replayed), and run beside them; its backward stays where it was.  Built, parity-green
        # (the early forms, eager and replayed, are bit-identical; against the late form the fake image's gradient adds its terms in
        # another order) and NOT faster: 29.6 / 29.8 against 28.7 / 29.6 ms (two pairs, one box) - "one kernel in flight" during the
        # 256^2 discriminator's update does not mean idle CUs: its GEMMs fill the chip, and the encoder's kernels only lengthen them.
        self._encst = distinct_streams(1, self.device, avoid=taken)[0] \
            if (self.device.type == "cuda" and image_encoder is not None and self.netsD and
                os.environ.get("TGSR_ENC_EARLY", "0") == "1") else None
        if self._encst is not None:
            taken = taken + [self._encst.cuda_stream]
        # the stream the generators' graphs are captured on, and the branch their re-pack launches fork onto
        self._gcap, self._gpack = distinct_streams(2, self.device, avoid=taken) if self._graph_capable else (None, None)

    @property
    def _graph_g(self):
        return self._graph_g_value

    @_graph_g.setter
    def _graph_g(self, v):
        self._graph_g_value = bool(v)
        self._auto = None                    # an explicit choice ends the measured policy

    # ------------------------------------------------------------------ TGSR_GRAPH_G=auto: the faster form, measured
    def _auto_begin(self):
        """Called at the top of a step: which form this step takes while the policy is still measuring (None = settled), and the
        start of its clock.  Steps [WARMUP, WARMUP + TRIALS) are timed eager, step WARMUP + TRIALS captures (untimed), the next
        TRIALS are timed replays; then `_auto_end` decides."""
        a = self._auto
        if a is None:
            return
        k = self._gsteps
        a["form"] = "eager" if k < GRAPH_G_WARMUP + GRAPH_G_TRIALS else "replay"
        timed = GRAPH_G_WARMUP <= k and k != GRAPH_G_WARMUP + GRAPH_G_TRIALS
        if timed:
            torch.cuda.synchronize(self.device)
            a["t0"] = time.perf_counter()
        else:
            a["t0"] = None

    def _auto_end(self, replayed):
        a = self._auto
        if a is None:
            return
        if a["t0"] is not None:
            torch.cuda.synchronize(self.device)
            dt = time.perf_counter() - a["t0"]
            if a["form"] == "replay":
                a["replay_s"].append(dt if replayed else float("inf"))     # (the capture failed or the configuration has none)
            else:
                a["eager_s"].append(dt)
        if self._gsteps < GRAPH_G_SETTLED:
            return
        import statistics
        inf = float("inf")
        te = statistics.median(a["eager_s"]) if a["eager_s"] else inf      # (steps that raised may have left a form untimed)
        tp = statistics.median(a["replay_s"]) if a["replay_s"] else inf
        from .parallel import dp_world
        if dp_world() > 1:
            import torch.distributed as dist
            t = torch.tensor([te, tp], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            te, tp = float(t[0]), float(t[1])
        if te == inf:                                               # nothing to compare with: the first guess stands
            self._graph_g = self._graph_g_value
            self.graph_policy["chosen"] = "replay" if self._graph_g_value else "eager"
            return
        self._graph_g = tp <= te                                            # (the setter ends the measurement)
        self.graph_policy.update({"eager_ms": round(te * 1e3, 3), "replay_ms": None if tp == float("inf") else round(tp * 1e3, 3),
                                  "chosen": "replay" if tp <= te else "eager",
                                  "trials": "median of %d steps of each form, device idle on both sides" % GRAPH_G_TRIALS})

    # ------------------------------------------------------------------ gradient all-reduce under the tail of backward
    def _arm_early(self):
        from .parallel import dp_world
        self._early = None
        self._early_left = self._early_n if (self._early_on and dp_world() > 1 and self._comm is not None) else -1

    def _gh_grad_done(self, _p):
        """post-accumulate hook of every NetG_highweight parameter: when the last one has its gradient, that range of the
        bucket is final - flush it and start its all-reduce on the communication stream while G_SR_NET_low's backward goes on."""
        if self._early_left <= 0:
            return
        self._early_left -= 1
        if self._early_left == 0:
            self._fire_early()

    def _fire_early(self):
        cur = torch.cuda.current_stream(self.device)
        self.bucket.flush_params(0, self._early_n)
        self._comm.wait_stream(cur)
        if self._wside is not None:
            self._comm.wait_stream(self._wside)            # the weight-gradient kernels of this range run there
        with torch.cuda.stream(self._comm):
            self._early = self.bucket.all_reduce_range_async(0, self._early_hi)

    def _all_reduce(self):
        """The step's closing collective: everything the early one did not take (all of it when none was started)."""
        if self._early is None:
            self._early_left = -1
            self.bucket.all_reduce_mean()
            return
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._comm):
            self._early.wait()
        cur.wait_stream(self._comm)
        self._early, self._early_left = None, -1
        self.bucket.all_reduce_mean(skip=(0, self._early_hi))

    def loss(self, captions, cap_lens, LR, LRb, hr_pyramid, class_ids=None):
        """hr_pyramid: the 3 target scales [B,3,2s,2s], [B,3,4s,4s], [B,3,8s,8s]."""
        fake_imgL, fine_im, mu, logvar, words_embs, sent_emb = self.forward_G(captions, cap_lens, LR, LRb)
        errG = self._loss_from(fake_imgL, fine_im, mu, logvar, words_embs, sent_emb, cap_lens, hr_pyramid, class_ids)
        return errG, fake_imgL, fine_im

    @staticmethod
    def _zero(bucket):
        """Open a step: one memset of the flat bucket; `.grad` cleared and the gradient slots opened, so the weight-
        gradient kernels write straight into the bucket (parallel.grad_slot) instead of autograd adding into views."""
        bucket.begin_step()

    def _text(self, captions, cap_lens):
        """The frozen text encoder and the caption mask: the only part of a step that reads host data (the caption lengths)."""
        with torch.no_grad():
            words_embs, sent_emb = self.text_encoder(captions, cap_lens, self.text_encoder.init_hidden(captions.shape[0]))
        return words_embs, sent_emb, caption_mask(captions, words_embs.size(2))

    def _forward_nets(self, LR, LRb, words_embs, sent_emb, mask):
        # (NetG_highweight's trunk on a second stream beside G_SR_NET_low, forward and backward, was measured: 11.9 ms
        # against 11.6 ms on one stream once the weight gradients have their side stream - not kept)
        fake_imgL, _att, mu, logvar = self.netGL(LR, sent_emb, words_embs, mask)
        fine_im, _a, _one = self.netGH(LR, fake_imgL, LRb)
        return fake_imgL, fine_im, mu, logvar

    def forward_G(self, captions, cap_lens, LR, LRb):
        """Text encoder (frozen) + both generators in training mode: (fake_imgL, fine_im, mu, logvar, words, sent)."""
        words_embs, sent_emb, mask = self._text(captions, cap_lens)
        fake_imgL, fine_im, mu, logvar = self._forward_nets(LR, LRb, words_embs, sent_emb, mask)
        return fake_imgL, fine_im, mu, logvar, words_embs, sent_emb

    def d_losses(self, fine_im, hr_pyramid, sent_emb):
        """discriminator_loss (losses.py:290-316) of every scale: real = HR pyramid, fake = the generators' output."""
        B = sent_emb.shape[0]
        real_labels, fake_labels, _ = prepare_labels(B, self.device)
        return [losses.discriminator_loss(d, hr_pyramid[i], fine_im[i], sent_emb, real_labels, fake_labels)
                for i, d in enumerate(self.netsD)]

    def g_loss(self, fake_imgL, fine_im, mu, logvar, words_embs, sent_emb, cap_lens, hr_pyramid, class_ids=None, enc_out=None):
        """generator_loss (losses.py:351-391) on the fine images + the pixel and KL terms of `loss`.  enc_out: the image encoder's
        outputs on the finest fake image when `_encode_early` has computed them already."""
        B = sent_emb.shape[0]
        real_labels, _fake, match_labels = prepare_labels(B, self.device)
        adv, _log = losses.generator_loss(self.netsD, self.image_encoder, fine_im, real_labels, words_embs, sent_emb,
                                          match_labels, cap_lens, class_ids, streams=self._dstreams or None, lazy_log=True,
                                          gather_negatives=self.gather_negatives, enc_out=enc_out)
        return adv + self._pixel_kl(fake_imgL, fine_im, mu, logvar, hr_pyramid)

    def _encode_early(self, image):
        """`self.image_encoder(image)` on the encoder's own stream, forked from the current one (the image is ready there); the
        caller joins with `_encode_join` before it uses the outputs.  None when the early form is off."""
        if self._encst is None:
            return None
        main = torch.cuda.current_stream(self.device)
        self._encst.wait_stream(main)
        with torch.cuda.stream(self._encst):
            if not torch.cuda.is_current_stream_capturing():
                image.record_stream(self._encst)
            return self.image_encoder(image)

    def _encode_join(self, enc_out):
        if enc_out is None:
            return
        main = torch.cuda.current_stream(self.device)
        main.wait_stream(self._encst)
        if not torch.cuda.is_current_stream_capturing():
            for t in enc_out:
                t.record_stream(main)

    @contextlib.contextmanager
    def _use_packs(self):
        """Scope in which the conv blocks take their packed weights from this trainer's autograd.PackCache."""
        from . import autograd
        prev, autograd._PACKS = autograd._PACKS, self._packs
        try:
            yield
        finally:
            autograd._PACKS = prev

    @contextlib.contextmanager
    def _wgrad_side(self):
        """Scope in which autograd.ConvBnAct issues its weight gradients on this trainer's side stream; the stream is
        joined on exit, before anything reads the gradients."""
        from . import autograd
        if self._wside is None:
            yield
            return
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        autograd.WGRAD_SIDE[idx] = self._wside
        ok = False
        try:
            yield
            ok = True
        finally:
            autograd.WGRAD_SIDE.pop(idx, None)
            torch.cuda.current_stream(self.device).wait_stream(self._wside)      # the join comes first ...
            if not ok:
                autograd._ADOPTED.clear()
        # ... then the check that autograd adopted every side-stream gradient in place.  A failure means the step's gradients
        # are INVALID (an accumulation kernel read a slot the side stream was still writing): the bucket is zeroed so that
        # nothing downstream (all-reduce, optimizer) can consume them, and the error propagates - the caller skips the step.
        try:
            autograd.check_adopted()
        except Exception:
            self.bucket.flat.zero_()
            raise

    @contextlib.contextmanager
    def _d_wgrad_side(self, i):
        """`_wgrad_side` for discriminator i's update: its weight gradients on the discriminator's side stream, joined into the
        CURRENT stream (the discriminator's own) on exit, before the bucket is closed, reduced or read."""
        from . import autograd
        if not self._dwside:
            yield
            return
        side = self._dwside[i]
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        prev = autograd.WGRAD_SIDE.get(idx)
        autograd.WGRAD_SIDE[idx] = side
        ok = False
        try:
            yield
            ok = True
        finally:
            if prev is None:
                autograd.WGRAD_SIDE.pop(idx, None)
            else:
                autograd.WGRAD_SIDE[idx] = prev
            torch.cuda.current_stream(self.device).wait_stream(side)
            if not ok:
                autograd._ADOPTED.clear()
        try:
            autograd.check_adopted()
        except Exception:
            self.bucketsD[i].flat.zero_()
            raise

    # ------------------------------------------------------------------ updates replayed from hipGraphs
    @staticmethod
    def _opt_key(o):
        """What a captured optimizer step has baked in: the optimizer object, the addresses of its moment / step tensors (a
        load_state_dict replaces them) and its scalar hyper-parameters."""
        first = o.param_groups[0]["params"][0]
        st = o.state.get(first, {})
        ids = tuple(int(st[k].data_ptr()) for k in ("exp_avg", "exp_avg_sq", "step") if torch.is_tensor(st.get(k)))
        return (id(o), ids) + tuple((g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]) for g in o.param_groups)

    def _d_hyper(self, i):
        return self._opt_key(self.optsD[i])

    def _bump_d(self, i):
        """A replayed (or fused-Adam) update wrote parameters and running statistics without telling autograd's version counters,
        which every cache of derived tensors keys on (util._FusedParams, PackCache.get, ...)."""
        if self._d_bump[i] is None:
            self._d_bump[i] = list(self.bucketsD[i].params) + [b for b in self.netsD[i].buffers()]
        torch.autograd.graph.increment_version(self._d_bump[i])

    def _bump_g(self):
        if self._g_bump is None:
            self._g_bump = list(self.params) + list(self._bucket_bufs)
        torch.autograd.graph.increment_version(self._g_bump)

    def _d_update_eager(self, i, fake, real, sent, real_labels, fake_labels):
        d, b, o = self.netsD[i], self.bucketsD[i], self.optsD[i]
        b.begin_step()
        e = losses.discriminator_loss(d, real, fake, sent, real_labels, fake_labels)
        with self._d_wgrad_side(i):
            e.backward()
        b.end_step()
        b.all_reduce_mean()
        o.step()
        self._bump_d(i)
        return e

    def _d_update_graphed(self, i, fake, real, sent, real_labels, fake_labels):
        """Discriminator i's update from its hipGraphs (captured on first use, on the discriminator's own stream, which is the
        current one): the inputs are copied into the capture's buffers; segment "fb" zeroes the gradient bucket and runs forward,
        loss and backward, the bucket's all-reduce follows on the same stream when there is more than one rank, segment "opt" is
        Adam (one rank: one graph holds both).  Returns the loss (a buffer of the capture: valid until the next replay).  A batch
        of another shape, or a capture that failed once, takes the eager update."""
        g = self._dgraphs[i]
        if g not in (None, False) and g["hyper"] != self._d_hyper(i):
            g = self._dgraphs[i] = None                     # lr / betas / eps / the moment tensors changed since the capture: capture again
        if g not in (None, False) and (tuple(fake.shape) != tuple(g["fake"].shape) or tuple(sent.shape) != tuple(g["sent"].shape)):
            return self._d_update_eager(i, fake, real, sent, real_labels, fake_labels)
        if g is None:
            g = self._dgraphs[i] = self._capture_d_update(i, fake, real, sent, real_labels, fake_labels)
        if g is False:                                       # the capture failed once: eager from then on
            return self._d_update_eager(i, fake, real, sent, real_labels, fake_labels)
        with torch.no_grad():
            torch._foreach_copy_([g["fake"], g["real"], g["sent"]], [fake.detach(), real, sent.detach()])
        g["fb"].replay()
        if g["opt"] is not None:
            self.bucketsD[i].all_reduce_mean()
            g["opt"].replay()
        self._bump_d(i)
        return g["err"]

    def reset_d_graphs(self):
        """Forget the captured updates (they are also dropped by themselves when an optimizer's hyper-parameters or state
        tensors change); the next steps capture again."""
        self._dgraphs = [None] * len(self.netsD)
        self._ggraphs = {}

    def _capture_d_update(self, i, fake, real, sent, real_labels, fake_labels):
        from .parallel import dp_world
        d, b, o, st = self.netsD[i], self.bucketsD[i], self.optsD[i], self._dstreams[i]
        buf = {"fake": fake.detach().clone(), "real": real.clone(), "sent": sent.detach().clone(),
               "rl": real_labels.clone(), "fl": fake_labels.clone(), "hyper": self._d_hyper(i), "opt": None}
        split = dp_world() > 1                               # the all-reduce sits between the two segments
        pool = torch.cuda.graph_pool_handle()
        fb = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(fb, stream=st, pool=pool):
                b.begin_step()
                e = losses.discriminator_loss(d, buf["real"], buf["fake"], buf["sent"], buf["rl"], buf["fl"])
                with self._d_wgrad_side(i):
                    e.backward()
                b.end_step()
                if not split:
                    o.step()
                buf["err"] = e.detach()
            if split:
                buf["opt"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(buf["opt"], stream=st, pool=pool):
                    o.step()
        except Exception as ex:                               # noqa: BLE001 - the eager path is always there
            import warnings
            warnings.warn("discriminator %d: the update could not be captured (%s: %s); it stays eager" % (i, type(ex).__name__, ex))
            b.end_step()                                      # close whatever begin_step opened
            return False
        buf["fb"] = fb
        return buf

    # ------------------------------------------------------------------ the generators' update from hipGraphs
    def _g_key(self, gan, LR, words_embs, cap_lens, class_ids):
        from .parallel import dp_world
        key = (bool(gan), tuple(LR.shape), int(words_embs.shape[2]), dp_world())
        if self.image_encoder is not None:
            # the DAMSM kernels take the caption lengths (and the class mask) as launch arguments: part of what a capture bakes in
            import numpy as np
            key += (tuple(int(v) for v in cap_lens),
                    None if class_ids is None else tuple(int(v) for v in np.asarray(class_ids).ravel()))
        return key

    def _g_graphs(self, gan, LR, LRb, hr_pyramid, words_embs, sent_emb, mask, cap_lens, class_ids):
        """The captured update for this step's shapes: a dict of graphs and their static buffers, or None = take the eager step
        (warm-up, switched off, a configuration that needs a collective inside the loss, or a capture that failed)."""
        from .parallel import dp_world
        use = (self._graph_g if self._auto is None else self._auto["form"] == "replay") and self._gsteps >= GRAPH_G_WARMUP
        self._gsteps += 1
        if not use or (self.image_encoder is not None and self.gather_negatives and dp_world() > 1):
            return None                     # (DAMSM on the gathered global batch all-gathers inside generator_loss)
        hyper = self._opt_key(self.opt) + (self.ema_decay,)
        if hyper != self._ghyper:
            self._ggraphs, self._ghyper = {}, hyper
        key = self._g_key(gan, LR, words_embs, cap_lens, class_ids)
        g = self._ggraphs.get(key)
        if g is None:
            g = self._ggraphs[key] = self._capture_g(gan, LR, LRb, hr_pyramid, words_embs, sent_emb, mask, cap_lens, class_ids)
        return g or None

    def _pixel_kl(self, fake_imgL, fine_im, mu, logvar, hr_pyramid):
        return losses.MSE(fake_imgL, hr_pyramid) + losses.MSE(fine_im, hr_pyramid) + losses.KL_loss(mu, logvar)

    def _loss_from(self, fake_imgL, fine_im, mu, logvar, words_embs, sent_emb, cap_lens, hr_pyramid, class_ids):
        """The generator-only step's loss on the networks' outputs (see `loss`)."""
        errG = self._pixel_kl(fake_imgL, fine_im, mu, logvar, hr_pyramid)
        if self.image_encoder is not None:
            region_features, cnn_code = self.image_encoder(fine_im[-1])
            w0, w1, s0, s1, scale, _ = losses.damsm_terms(region_features, cnn_code, words_embs, sent_emb, cap_lens, class_ids,
                                                          gather=self.gather_negatives)
            errG = errG + (w0 + w1 + s0 + s1) * (cfg.TRAIN.SMOOTH.LAMBDA * scale)
        return errG

    def _g_backward(self, errG, early=True):
        """errG.backward() with the packs and the weight-gradient side stream; the discriminators only pass the gradient through
        to the images (their own parameter gradients would be discarded: not computed)."""
        if early:
            self._arm_early()
        else:
            self._early, self._early_left = None, -1
        trunk = getattr(self.image_encoder, "_hip_trunk", None)
        if trunk is not None:
            trunk.bwd_stream = torch.cuda.current_stream(self.device)      # (inception.TrunkFn.backward: where the walk belongs)
        with self._use_packs(), self._wgrad_side():
            errG.backward()

    def _g_finish(self, captured=False):
        """What follows the gradient all-reduce: Adam, the re-pack of every cached weight pack, the EMA of the parameters."""
        self.opt.step()
        if captured:
            if self._packs is not None:
                self._packs.repack_captured(self._gpack)
        elif self._packs is not None:
            self._packs.repack(force=True)      # the optimizer has just run: every pack is stale, whatever the version counters say
        with torch.no_grad():
            torch._foreach_mul_(self.avg_param_G, self.ema_decay)
            torch._foreach_add_(self.avg_param_G, [p.data for p in self.params], alpha=1.0 - self.ema_decay)
        if captured and self._packs is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._gpack)      # join the pack branch before the capture ends

    def _capture_g(self, gan, LR, LRb, hr_pyramid, words_embs, sent_emb, mask, cap_lens, class_ids):
        """Capture the generators' update for one batch shape.  Nothing executes here (stream capture records); `_g_run` replays.
        Segments, all in one memory pool, the autograd graph of "fwd" alive while "fb" is recorded:
            "fwd" (G/D alternation only)  both generators forward - the discriminator updates run between it and "fb";
            "fb"   zero the bucket, [forward,] losses, backward (weight gradients on the side branch), gradients in place;
            -- the bucket's all-reduce, eager, when there is more than one rank --
            "opt"  fused Adam, every weight pack re-derived (a branch of its own), EMA      (one rank: part of "fb")."""
        from .parallel import dp_world
        split = dp_world() > 1
        st = self._gcap
        buf = {"LR": LR.clone(), "LRb": LRb.clone(), "hr": [h.clone() for h in hr_pyramid], "words": words_embs.clone(),
               "sent": sent_emb.clone(), "mask": mask.clone(), "fwd": None, "opt": None, "enc": None}
        buf["dst"] = [buf["LR"], buf["LRb"], buf["words"], buf["sent"], buf["mask"]] + buf["hr"]
        pool = torch.cuda.graph_pool_handle()
        cap_lens = [int(v) for v in cap_lens]
        d_params = [p for b in self.bucketsD for p in b.params]
        if self._packs is not None:
            self._packs.settle(self.device)
        opened = False
        try:
            if gan:
                buf["fwd"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(buf["fwd"], stream=st, pool=pool):
                    with self._use_packs():
                        nets = self._forward_nets(buf["LR"], buf["LRb"], buf["words"], buf["sent"], buf["mask"])
                buf["fine"] = nets[1]
                if self._encst is not None:
                    # the image encoder's forward as a graph of its own, captured on ITS stream as the origin (its branch forks are
                    # then plain diamonds) and replayed beside the discriminators' graphs; its autograd node stays alive for "fb"
                    buf["enc"] = torch.cuda.CUDAGraph()
                    self._encst.wait_stream(st)
                    with torch.cuda.graph(buf["enc"], stream=self._encst, pool=pool):
                        buf["enc_out"] = self.image_encoder(nets[1][len(self.netsD) - 1])
                    st.wait_stream(self._encst)
            fb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(fb, stream=st, pool=pool):
                self.bucket.begin_step()
                opened = True
                if gan:
                    for p in d_params:
                        p.requires_grad_(False)
                    errG = self.g_loss(nets[0], nets[1], nets[2], nets[3], buf["words"], buf["sent"], cap_lens, buf["hr"], class_ids,
                                       enc_out=buf.get("enc_out"))
                else:
                    with self._use_packs():
                        nets = self._forward_nets(buf["LR"], buf["LRb"], buf["words"], buf["sent"], buf["mask"])
                        errG = self._loss_from(nets[0], nets[1], nets[2], nets[3], buf["words"], buf["sent"], cap_lens, buf["hr"],
                                               class_ids)
                self._g_backward(errG, early=False)
                self.bucket.end_step()
                opened = False
                if not split:
                    self._g_finish(captured=True)
                buf["err"] = errG.detach()
            if split:
                buf["opt"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(buf["opt"], stream=st, pool=pool):
                    self._g_finish(captured=True)
        except Exception as ex:                               # noqa: BLE001 - the eager path is always there
            import warnings
            warnings.warn("the generators' update could not be captured (%s: %s); it stays eager" % (type(ex).__name__, ex))
            if opened:
                self.bucket.end_step()
            return False
        finally:
            for p in d_params:
                p.requires_grad_(True)
        buf["fb"] = fb
        del nets, errG
        return buf

    def _g_load(self, g, LR, LRb, hr_pyramid, words_embs, sent_emb, mask):
        with torch.no_grad():
            torch._foreach_copy_(g["dst"], [LR, LRb, words_embs, sent_emb, mask] + list(hr_pyramid))
        if self._packs is not None:
            self._packs.settle(self.device)

    def _g_update_replay(self, g):
        """Segments "fb" [-> all-reduce] -> "opt" of a captured update; returns the loss (a buffer of the capture)."""
        g["fb"].replay()
        if g["opt"] is not None:
            self._early, self._early_left = None, -1
            self.bucket.all_reduce_mean()
            g["opt"].replay()
        self._bump_g()
        if self._packs is not None:
            self._packs.mark_fresh()
        return g["err"]

    def _d_updates(self, fine_im, hr_pyramid, sent_emb):
        """Every discriminator's update on (real, fake.detach()): each on a stream of its own - replayed from its hipGraphs once
        the step has run GRAPH_D_WARMUP times."""
        graphed = bool(self._dstreams) and self._graph_d and self._dsteps >= GRAPH_D_WARMUP
        self._dsteps += 1
        B = sent_emb.shape[0]
        real_labels, fake_labels, _ = prepare_labels(B, self.device)
        if not self._dstreams:
            return [self._d_update_eager(i, fine_im[i], hr_pyramid[i], sent_emb, real_labels, fake_labels)
                    for i in range(len(self.netsD))]
        # the three discriminators are independent of each other: each one's forward, backward, all-reduce and Adam
        # step run on a stream of their own (the 64^2 / 128^2 discriminators' layers leave most CUs idle)
        main = torch.cuda.current_stream(self.device)
        errsD = []
        for i, st in enumerate(self._dstreams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for t in (fine_im[i], hr_pyramid[i], sent_emb):
                    t.record_stream(st)
                upd = self._d_update_graphed if graphed else self._d_update_eager
                errsD.append(upd(i, fine_im[i], hr_pyramid[i], sent_emb, real_labels, fake_labels))
        for st in self._dstreams:
            main.wait_stream(st)
        return errsD

    def step_gan(self, captions, cap_lens, LR, LRb, hr_pyramid, class_ids=None):
        """One G/D alternation: forward the generators once; update every discriminator on (real, fake.detach());
        then update the generators through the UPDATED discriminators on the same fake images.  Returns
        (errG, [errD_i]) as detached tensors (buffers of the captures when the step is replayed: valid until the next step)."""
        self._auto_begin()
        words_embs, sent_emb, mask = self._text(captions, cap_lens)
        g = self._g_graphs(True, LR, LRb, hr_pyramid, words_embs, sent_emb, mask, cap_lens, class_ids)
        if g is not None:
            self._g_load(g, LR, LRb, hr_pyramid, words_embs, sent_emb, mask)
            g["fwd"].replay()
            if g["enc"] is not None:                        # the image encoder's forward beside the discriminator updates
                main = torch.cuda.current_stream(self.device)
                self._encst.wait_stream(main)
                with torch.cuda.stream(self._encst):
                    g["enc"].replay()
            errsD = self._d_updates(g["fine"], g["hr"], g["sent"])
            if g["enc"] is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._encst)
            errG = self._g_update_replay(g)
            self._auto_end(True)
            return errG, [e.detach() for e in errsD]
        with self._use_packs():
            fake_imgL, fine_im, mu, logvar = self._forward_nets(LR, LRb, words_embs, sent_emb, mask)
        enc_out = self._encode_early(fine_im[len(self.netsD) - 1])          # beside the discriminator updates
        errsD = self._d_updates(fine_im, hr_pyramid, sent_emb)
        self._encode_join(enc_out)
        self._zero(self.bucket)
        # the discriminators only pass the gradient through to the images here: their own parameter gradients would be
        # discarded (the next discriminator update zeroes its bucket first), so they are not computed
        d_params = [p for b in self.bucketsD for p in b.params]
        for p in d_params:
            p.requires_grad_(False)
        try:
            errG = self.g_loss(fake_imgL, fine_im, mu, logvar, words_embs, sent_emb, cap_lens, hr_pyramid, class_ids, enc_out=enc_out)
            self._g_backward(errG)
        finally:
            for p in d_params:
                p.requires_grad_(True)
            self.bucket.end_step()
        self._all_reduce()
        self._g_finish()
        self._bump_g()
        self._auto_end(False)
        return errG.detach(), [e.detach() for e in errsD]

    def step(self, captions, cap_lens, LR, LRb, hr_pyramid, class_ids=None):
        """forward + backward + gradient all-reduce (if distributed) + Adam + EMA.  Returns the loss tensor.  With
        discriminators this is `step_gan` (the generator loss is returned).  After GRAPH_G_WARMUP eager steps the update is
        replayed from hipGraphs (`_capture_g`), bit-identical to the eager one."""
        if self.netsD:
            return self.step_gan(captions, cap_lens, LR, LRb, hr_pyramid, class_ids)[0]
        self._auto_begin()
        words_embs, sent_emb, mask = self._text(captions, cap_lens)
        g = self._g_graphs(False, LR, LRb, hr_pyramid, words_embs, sent_emb, mask, cap_lens, class_ids)
        if g is not None:
            self._g_load(g, LR, LRb, hr_pyramid, words_embs, sent_emb, mask)
            err = self._g_update_replay(g)
            self._auto_end(True)
            return err
        self._zero(self.bucket)
        try:
            with self._use_packs():
                nets = self._forward_nets(LR, LRb, words_embs, sent_emb, mask)
                errG = self._loss_from(nets[0], nets[1], nets[2], nets[3], words_embs, sent_emb, cap_lens, hr_pyramid, class_ids)
            self._g_backward(errG)
        finally:
            self.bucket.end_step()               # also after a failed step: `.grad` views restored, slots closed
        self._all_reduce()
        self._g_finish()
        self._bump_g()
        self._auto_end(False)
        return errG.detach()


class DAMSMTrainer:
    """pretrain_DAMSM.py:48-125, 262-284: joint training of RNN_ENCODER and the CNN_ENCODER heads on
    words_loss + sent_loss.  Every gradient comes from HIP kernels: DAMSM backward (tgsr_damsm_words_bwd), LSTM BPTT
    (tgsr_bilstm_bwd) and the GEMMs of the heads; the Inception trunk is the caller's frozen module
    (CNN_ENCODER(trunk=...)) or pre-extracted features via `step_features`.  Like the reference: a fresh
    Adam(lr, betas (0.5, 0.999)) per epoch, lr x 0.98 per epoch down to ENCODER_LR / 10, gradient-norm clip
    RNN_GRAD_CLIP on the text encoder only.  Data parallel: one flat gradient bucket, one all-reduce per step; the
    contrastive losses are those of the GLOBAL batch (features and embeddings all-gathered, parallel.gather_damsm_batch;
    `gather_negatives=False` / TGSR_DP_GATHER_NEGATIVES=0: the local shard's negatives only - SURVEY section 8e (2))."""

    def __init__(self, n_words, device="cuda", trunk=None, lr=None, gather_negatives=None):
        self.device = torch.device(device)
        from . import parallel as _par
        self.gather_negatives = _par.GATHER_NEGATIVES if gather_negatives is None else bool(gather_negatives)
        self.text_encoder = RNN_ENCODER(n_words, nhidden=cfg.TEXT.EMBEDDING_DIM).to(self.device).train()
        self.image_encoder = CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM,
                                         trunk=trunk if trunk is not None else torch.nn.Identity()).to(self.device)
        self.image_encoder.train()
        for p in self.image_encoder.frozen_parameters():
            p.requires_grad = False                                  # util.py:274-275
        self.params = list(self.text_encoder.parameters()) + [p for p in self.image_encoder.parameters()
                                                              if p.requires_grad]
        self.bucket = FlatGradBucket(self.params).attach()
        self.base_lr = self.lr = lr or cfg.TRAIN.ENCODER_LR
        self.start_epoch()

    def start_epoch(self):
        """pretrain_DAMSM.py:270: the optimizer (and its moments) is rebuilt every epoch; train() puts both encoders back
        in training mode (:49-50; evaluate() leaves them in eval mode)."""
        self.text_encoder.train()
        self.image_encoder.train()
        self.opt = torch.optim.Adam(self.params, lr=self.lr, betas=(0.5, 0.999))

    # ------------------------------------------------------------------ validation, snapshots, resume
    @torch.no_grad()
    def evaluate_features(self, batches):
        """pretrain_DAMSM.py:133-163 on trunk outputs: `batches` yields (features, pooled, captions, cap_lens, class_ids);
        both encoders in eval mode (and left there, as in the reference), at most 51 batches (`if step == 50: break`),
        returns (s_cur_loss, w_cur_loss) = the summed sentence / word losses divided by the LAST STEP INDEX - the
        reference's `s_total_loss[0] / step` (:160-161), not by the number of batches: N batches (N <= 50) are divided by
        N - 1, a single batch by 0 (inf), 51 or more by 50.  Kept as is: the numbers it prints are the ones a user of
        the reference compares against."""
        self.text_encoder.eval()
        self.image_encoder.eval()
        s_total = torch.zeros((), dtype=torch.float32, device=self.device)
        w_total = torch.zeros((), dtype=torch.float32, device=self.device)
        step = -1
        for step, (features, pooled, captions, cap_lens, class_ids) in enumerate(batches):
            B = captions.shape[0]
            labels = torch.arange(B, device=self.device)
            words_features, sent_code = self.image_encoder.heads(features, pooled)
            words_emb, sent_emb = self.text_encoder(captions, cap_lens, self.text_encoder.init_hidden(B))
            w0, w1, _att = losses.words_loss(words_features, words_emb, labels, cap_lens, class_ids, B)
            s0, s1 = losses.sent_loss(sent_code, sent_emb, labels, class_ids, B)
            w_total += (w0 + w1).detach()
            s_total += (s0 + s1).detach()
            if step == 50:
                break
        if step < 0:
            raise ValueError("evaluate: no validation batch (the reference only evaluates when len(dataloader_val) > 0)")
        s, w = float(s_total), float(w_total)
        return (s / step, w / step) if step > 0 else (float("inf") * (1 if s >= 0 else -1), float("inf") * (1 if w >= 0 else -1))

    @torch.no_grad()
    def evaluate(self, batches):
        """pretrain_DAMSM.py:133-163 with the images through the (frozen) trunk: `batches` yields
        (imgs, captions, cap_lens, class_ids) - imgs = the data loader's real_imgs[-1]."""
        def through_trunk():
            for imgs, captions, cap_lens, class_ids in batches:
                features, pooled = self.image_encoder.run_trunk(imgs)
                yield features, pooled, captions, cap_lens, class_ids
        return self.evaluate_features(through_trunk())

    def snapshot_due(self, epoch, max_epoch=None):
        """pretrain_DAMSM.py:286-287."""
        max_epoch = cfg.TRAIN.MAX_EPOCH if max_epoch is None else max_epoch
        return epoch % cfg.TRAIN.SNAPSHOT_INTERVAL == 0 or epoch == max_epoch

    def snapshot(self, model_dir, epoch):
        """pretrain_DAMSM.py:288-291: `image_encoder%d.pth` / `text_encoder%d.pth` state_dicts (no optimizer state: the
        reference rebuilds Adam every epoch anyway).  Under data parallelism call it on rank 0 (parameters are identical
        on every rank after the all-reduced step)."""
        import os
        os.makedirs(model_dir, exist_ok=True)
        pi, pt = "%s/image_encoder%d.pth" % (model_dir, epoch), "%s/text_encoder%d.pth" % (model_dir, epoch)
        torch.save(self.image_encoder.state_dict(), pi)
        torch.save(self.text_encoder.state_dict(), pt)
        return pi, pt

    def resume(self, net_e=None):
        """pretrain_DAMSM.py:172-186: load `cfg.TRAIN.NET_E` (a text_encoder snapshot), the image encoder from the same
        name with 'text_encoder' -> 'image_encoder', and take the epoch to continue from out of the file name
        (`istart = rfind('_') + 8`: the digits behind 'text_encoder').  Returns start_epoch = that epoch + 1; the learning
        rate restarts at ENCODER_LR as in the reference (the decayed value is not saved)."""
        net_e = cfg.TRAIN.NET_E if net_e is None else net_e
        if net_e == '':
            return 0
        self.text_encoder.load_state_dict(torch.load(net_e, map_location=self.device))
        self.image_encoder.load_state_dict(torch.load(net_e.replace('text_encoder', 'image_encoder'),
                                                      map_location=self.device))
        istart, iend = net_e.rfind('_') + 8, net_e.rfind('.')
        self.start_epoch()
        return int(net_e[istart:iend]) + 1

    def end_epoch(self):
        """pretrain_DAMSM.py:283-284."""
        if self.lr > self.base_lr / 10.:
            self.lr *= 0.98

    def loss_from_features(self, features, pooled, captions, cap_lens, class_ids=None):
        B = captions.shape[0]
        labels = torch.arange(B, device=self.device)
        words_features, sent_code = self.image_encoder.heads(features, pooled)
        words_emb, sent_emb = self.text_encoder(captions, cap_lens, self.text_encoder.init_hidden(B))
        # data parallel (gather_negatives): the losses of the GLOBAL batch, identical on every rank; step_features multiplies by
        # `_bw_scale` = world for backward (the bucket's all-reduce averages what a replicated loss needs summed)
        w0, w1, s0, s1, scale, att = losses.damsm_terms(words_features, sent_code, words_emb, sent_emb, cap_lens, class_ids,
                                                        gather=self.gather_negatives)
        self._bw_scale = scale
        return w0 + w1 + s0 + s1, (w0.detach(), w1.detach(), s0.detach(), s1.detach()), att

    def step_features(self, features, pooled, captions, cap_lens, class_ids=None):
        """One optimisation step on trunk outputs (features [B,768,17,17], pooled [B,2048]).  Returns the loss."""
        self.bucket.flat.zero_()
        for p, v in zip(self.bucket.params, self.bucket.views):
            p.grad = v
        loss, _parts, _att = self.loss_from_features(features, pooled, captions, cap_lens, class_ids)
        (loss * self._bw_scale if self._bw_scale != 1 else loss).backward()
        self.bucket.all_reduce_mean()
        torch.nn.utils.clip_grad_norm_(self.text_encoder.parameters(), cfg.TRAIN.RNN_GRAD_CLIP)   # :96-97
        self.opt.step()
        return loss.detach()

    def step(self, imgs, captions, cap_lens, class_ids=None):
        """pretrain_DAMSM.py:66-98 with the image through the (frozen) trunk."""
        with torch.no_grad():
            features, pooled = self.image_encoder.run_trunk(imgs)
        return self.step_features(features, pooled, captions, cap_lens, class_ids)
