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
        self.bucket = FlatGradBucket(gh_params + list(self.netGL.parameters()), buffers=self._bucket_bufs).attach()
        self._early_n = len(gh_params)                                   # parameters of the early range
        self._early_hi = sum(p.numel() for p in gh_params)               # ... = flat[0:_early_hi]
        self._early_on = os.environ.get("TGSR_EARLY_ALLREDUCE", "1") != "0"
        self._early_left, self._early = -1, None
        self._comm = distinct_streams(1, self.device, avoid=cur + ([self._wside.cuda_stream] if self._wside is not None else []))[0] \
            if self.device.type == "cuda" else None
        for p in gh_params:
            p.register_post_accumulate_grad_hook(self._gh_grad_done)
        self._fused_adam = self.device.type == "cuda" and os.environ.get("TGSR_FUSED_ADAM", "1") != "0"
        self.opt = torch.optim.Adam(self.params, lr=lr or cfg.TRAIN.GENERATOR_LR, betas=(0.5, 0.999), fused=self._fused_adam)
        self.ema_decay = ema_decay
        self.avg_param_G = copy_G_params(self.netGL) + copy_G_params(self.netGH)
        self.netsD, self.optsD, self.bucketsD = [], [], []
        self._graph_d, self._dgraphs, self._dsteps = False, [], 0
        if discriminators:
            from . import model
            self.netsD = list(discriminators) if not isinstance(discriminators, bool) else \
                [model.D_NET64(), model.D_NET128(), model.D_NET256()]
            # A discriminator's update - forward on (real, fake.detach()), loss, backward, Adam - is a closed piece of device work
            # with no host decision in it: replayed from a hipGraph per discriminator once the step has run `GRAPH_D_WARMUP` times
            # (single process only: the captured region would have to hold the gradient all-reduce).  ~1 000 of a step's ~1 570
            # launches leave the host that way; the step was issued no faster than 21-25 ms.  TGSR_GRAPH_D=0: eager.
            self._graph_d = self.device.type == "cuda" and os.environ.get("TGSR_GRAPH_D", "1") != "0"
            self._dgraphs, self._dsteps = [None] * len(self.netsD), 0
            for d in self.netsD:
                d.to(self.device).train()
                self.bucketsD.append(FlatGradBucket(d.parameters(), buffers=d.buffers()).attach())
                # (fused: one pass over a discriminator's ~70 M parameters and their moments instead of the ~10 of the
                # multi-tensor form - 2.0 ms of a G/D step were Adam kernels running alone on the device)
                self.optsD.append(torch.optim.Adam(d.parameters(), lr=d_lr or cfg.TRAIN.DISCRIMINATOR_LR, betas=(0.5, 0.999),
                                                   capturable=self._graph_d, fused=self._fused_adam))
        if self.netsD:
            # the generator loss runs the train-mode discriminators on the fake images once more (g_loss): their running
            # statistics move again, per rank, AFTER their own bucket's all-reduce - so they also ride the generators' bucket
            # and every rank leaves the step with the same discriminator buffers (a snapshot is the same file on every rank)
            self._bucket_bufs = self._bucket_bufs + [b for d in self.netsD for b in d.buffers()]
            self.bucket = FlatGradBucket(gh_params + list(self.netGL.parameters()), buffers=self._bucket_bufs).attach()
        self._dstreams = distinct_streams(len(self.netsD), self.device,
                                          avoid=cur + ([self._wside.cuda_stream] if self._wside is not None else []) +
                                          ([self._comm.cuda_stream] if self._comm is not None else [])) \
            if self.device.type == "cuda" and self.netsD and os.environ.get("TGSR_D_STREAMS", "1") != "0" else []

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
        errG = losses.MSE(fake_imgL, hr_pyramid) + losses.MSE(fine_im, hr_pyramid) + losses.KL_loss(mu, logvar)
        if self.image_encoder is not None:
            B = captions.shape[0]
            match_labels = torch.arange(B, device=self.device)
            region_features, cnn_code = self.image_encoder(fine_im[-1])
            w0, w1, s0, s1, scale, _ = losses.damsm_terms(region_features, cnn_code, words_embs, sent_emb, cap_lens, class_ids,
                                                          gather=self.gather_negatives)
            errG = errG + (w0 + w1 + s0 + s1) * (cfg.TRAIN.SMOOTH.LAMBDA * scale)
        return errG, fake_imgL, fine_im

    @staticmethod
    def _zero(bucket):
        """Open a step: one memset of the flat bucket; `.grad` cleared and the gradient slots opened, so the weight-
        gradient kernels write straight into the bucket (parallel.grad_slot) instead of autograd adding into views."""
        bucket.begin_step()

    def forward_G(self, captions, cap_lens, LR, LRb):
        """Text encoder (frozen) + both generators in training mode: (fake_imgL, fine_im, mu, logvar, words, sent)."""
        with torch.no_grad():
            words_embs, sent_emb = self.text_encoder(captions, cap_lens, self.text_encoder.init_hidden(captions.shape[0]))
        mask = caption_mask(captions, words_embs.size(2))
        # (NetG_highweight's trunk on a second stream beside G_SR_NET_low, forward and backward, was measured: 11.9 ms
        # against 11.6 ms on one stream once the weight gradients have their side stream - not kept)
        fake_imgL, _att, mu, logvar = self.netGL(LR, sent_emb, words_embs, mask)
        fine_im, _a, _one = self.netGH(LR, fake_imgL, LRb)
        return fake_imgL, fine_im, mu, logvar, words_embs, sent_emb

    def d_losses(self, fine_im, hr_pyramid, sent_emb):
        """discriminator_loss (losses.py:290-316) of every scale: real = HR pyramid, fake = the generators' output."""
        B = sent_emb.shape[0]
        real_labels, fake_labels, _ = prepare_labels(B, self.device)
        return [losses.discriminator_loss(d, hr_pyramid[i], fine_im[i], sent_emb, real_labels, fake_labels)
                for i, d in enumerate(self.netsD)]

    def g_loss(self, fake_imgL, fine_im, mu, logvar, words_embs, sent_emb, cap_lens, hr_pyramid, class_ids=None):
        """generator_loss (losses.py:351-391) on the fine images + the pixel and KL terms of `loss`."""
        B = sent_emb.shape[0]
        real_labels, _fake, match_labels = prepare_labels(B, self.device)
        adv, _log = losses.generator_loss(self.netsD, self.image_encoder, fine_im, real_labels, words_embs, sent_emb,
                                          match_labels, cap_lens, class_ids, streams=self._dstreams or None, lazy_log=True,
                                          gather_negatives=self.gather_negatives)
        return adv + losses.MSE(fake_imgL, hr_pyramid) + losses.MSE(fine_im, hr_pyramid) + losses.KL_loss(mu, logvar)

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

    def _d_update_graphed(self, i, fake, real, sent, real_labels, fake_labels):
        """Discriminator i's update from its hipGraph (captured on first use, on the discriminator's own stream, which is the
        current one): the inputs are copied into the capture's buffers, the replay zeroes the gradient bucket, runs forward, loss,
        backward and Adam.  Returns the loss (a buffer of the capture: valid until the next replay)."""
        g = self._dgraphs[i]
        if g not in (None, False) and g["hyper"] != self._d_hyper(i):
            g = self._dgraphs[i] = None                     # lr / betas / eps changed since the capture (they are baked into it): capture again
        if g is None:
            g = self._dgraphs[i] = self._capture_d_update(i, fake, real, sent, real_labels, fake_labels)
        if g is False:                                       # the capture failed once: eager from then on
            d, b, o = self.netsD[i], self.bucketsD[i], self.optsD[i]
            b.begin_step()
            e = losses.discriminator_loss(d, real, fake, sent, real_labels, fake_labels)
            e.backward()
            b.end_step()
            o.step()
            return e
        if tuple(fake.shape) != tuple(g["fake"].shape) or tuple(sent.shape) != tuple(g["sent"].shape):
            raise ValueError("the discriminator update was captured for a batch of %d: a step with another batch size needs "
                             "TGSR_GRAPH_D=0 (or a new trainer)" % g["fake"].shape[0])
        with torch.no_grad():
            torch._foreach_copy_([g["fake"], g["real"], g["sent"]], [fake.detach(), real, sent.detach()])
        g["graph"].replay()
        return g["err"]

    def _d_hyper(self, i):
        """What a captured update has baked in besides the tensors: the optimizer object and its scalar hyper-parameters."""
        o = self.optsD[i]
        return (id(o),) + tuple((g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]) for g in o.param_groups)

    def reset_d_graphs(self):
        """Forget the captured discriminator updates (after replacing an optimizer or loading its state: the captures hold the
        old moment tensors); the next steps capture again."""
        self._dgraphs = [None] * len(self.netsD)

    def _capture_d_update(self, i, fake, real, sent, real_labels, fake_labels):
        d, b, o, st = self.netsD[i], self.bucketsD[i], self.optsD[i], self._dstreams[i]
        buf = {"fake": fake.detach().clone(), "real": real.clone(), "sent": sent.detach().clone(),
               "rl": real_labels.clone(), "fl": fake_labels.clone(), "hyper": self._d_hyper(i)}
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, stream=st):
                b.begin_step()
                e = losses.discriminator_loss(d, buf["real"], buf["fake"], buf["sent"], buf["rl"], buf["fl"])
                e.backward()
                b.end_step()
                o.step()
                buf["err"] = e.detach()
        except Exception as ex:                               # noqa: BLE001 - the eager path is always there
            import warnings
            warnings.warn("discriminator %d: the update could not be captured (%s: %s); it stays eager" % (i, type(ex).__name__, ex))
            b.end_step()                                      # close whatever begin_step opened
            return False
        buf["graph"] = graph
        return buf

    def step_gan(self, captions, cap_lens, LR, LRb, hr_pyramid, class_ids=None):
        """One G/D alternation: forward the generators once; update every discriminator on (real, fake.detach());
        then update the generators through the UPDATED discriminators on the same fake images.  Returns
        (errG, [errD_i]) as detached tensors."""
        with self._use_packs():
            fake_imgL, fine_im, mu, logvar, words_embs, sent_emb = self.forward_G(captions, cap_lens, LR, LRb)
        from .parallel import dp_world
        graphed = bool(self._dstreams) and self._graph_d and dp_world() == 1 and self._dsteps >= GRAPH_D_WARMUP
        self._dsteps += 1
        if not graphed:
            for b in self.bucketsD:
                self._zero(b)
        if self._dstreams:
            # the three discriminators are independent of each other: each one's forward, backward, all-reduce and Adam
            # step run on a stream of their own (the 64^2 / 128^2 discriminators' layers leave most CUs idle)
            B = sent_emb.shape[0]
            real_labels, fake_labels, _ = prepare_labels(B, self.device)
            main = torch.cuda.current_stream(self.device)
            errsD = []
            for i, (d, b, o, st) in enumerate(zip(self.netsD, self.bucketsD, self.optsD, self._dstreams)):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    for t in (fine_im[i], hr_pyramid[i], sent_emb):
                        t.record_stream(st)
                    if graphed:
                        e = self._d_update_graphed(i, fine_im[i], hr_pyramid[i], sent_emb, real_labels, fake_labels)
                    else:
                        e = losses.discriminator_loss(d, hr_pyramid[i], fine_im[i], sent_emb, real_labels, fake_labels)
                        e.backward()
                        b.end_step()
                        b.all_reduce_mean()
                        o.step()
                errsD.append(e)
            for st in self._dstreams:
                main.wait_stream(st)
        else:
            errsD = self.d_losses(fine_im, hr_pyramid, sent_emb)
            for e, b, o in zip(errsD, self.bucketsD, self.optsD):
                e.backward()
                b.end_step()
                b.all_reduce_mean()
                o.step()
        self._zero(self.bucket)
        for b in self.bucketsD:                       # the generator step also deposits gradients in the discriminators'
            self._zero(b)                             # parameters; they are discarded (zeroed again next step)
        # the discriminators only pass the gradient through to the images here: their own parameter gradients would be
        # discarded (netsD[i].zero_grad() opens the next discriminator step), so they are not computed
        for b in self.bucketsD:
            for p in b.params:
                p.requires_grad_(False)
        try:
            errG = self.g_loss(fake_imgL, fine_im, mu, logvar, words_embs, sent_emb, cap_lens, hr_pyramid, class_ids)
            self._arm_early()
            with self._use_packs(), self._wgrad_side():
                errG.backward()
        finally:
            for b in self.bucketsD:
                for p in b.params:
                    p.requires_grad_(True)
        self.bucket.end_step()
        for b in self.bucketsD:
            b.end_step()
        self._all_reduce()
        self.opt.step()
        if self._packs is not None:
            self._packs.repack(force=True)      # the optimizer has just run: every pack is stale, whatever the version counters say
        with torch.no_grad():
            torch._foreach_mul_(self.avg_param_G, self.ema_decay)
            torch._foreach_add_(self.avg_param_G, [p.data for p in self.params], alpha=1.0 - self.ema_decay)
        return errG.detach(), [e.detach() for e in errsD]

    def step(self, captions, cap_lens, LR, LRb, hr_pyramid):
        """forward + backward + gradient all-reduce (if distributed) + Adam + EMA.  Returns the loss tensor.  With
        discriminators this is `step_gan` (the generator loss is returned)."""
        if self.netsD:
            return self.step_gan(captions, cap_lens, LR, LRb, hr_pyramid)[0]
        self._zero(self.bucket)
        try:
            with self._use_packs():
                errG, _, _ = self.loss(captions, cap_lens, LR, LRb, hr_pyramid)
                self._arm_early()
                with self._wgrad_side():
                    errG.backward()
        finally:
            self.bucket.end_step()               # also after a failed step: `.grad` views restored, slots closed
        self._all_reduce()
        self.opt.step()
        if self._packs is not None:
            self._packs.repack(force=True)      # the optimizer has just run: every pack is stale, whatever the version counters say                 # next step's packed weights, off the critical stream
        with torch.no_grad():
            torch._foreach_mul_(self.avg_param_G, self.ema_decay)
            torch._foreach_add_(self.avg_param_G, [p.data for p in self.params], alpha=1.0 - self.ema_decay)
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
