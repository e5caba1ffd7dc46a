"""The GAN training step of the reference on synthetic tensors (BASELINE.json configs[2], configs[3]).

``WeatherTransferStep`` is the harness counterpart of ``WeatherTransfer.update_discriminator`` /
``update_inference`` / ``evaluation`` (class-conditioned: t_cls_train.py:226-367; soft-label / estimator-conditioned:
t_est_train.py:214-332): same order of forwards / backwards, same losses (ops.py), same optimisers
(Adam lr 1e-4, betas (0, 0.999), weight_decay lr/20, t_cls_train.py:184-185), same switches
(``--supervised`` t_cls_train.py:232-235,260-262,294-297,419-421; ``--cross_ent`` :247-251,256,436).  Datasets and
TensorBoard are out of scope.  The estimator is any frozen ``nn.Module (N,3,H,W) -> (N,nc)`` RAW outputs
(what the scripts call ``self.estimator_``) through which gradients flow to the generator: ``wu.resnet.ResNet101Estimator``
(the architecture of classifier.py:106 / estimator.py:143 on the HIP kernels) or the small ``StandInEstimator``.

Differences from the reference loop, all numerically neutral:
  * update_discriminator runs G under no_grad (the reference builds G's graph and then detaches it,
    t_cls_train.py:302-303);
  * no ``.item()`` host syncs inside the step (the reference forces six per step, :275-282,310-312): losses are
    returned as device tensors;
  * ``step()`` evaluates the frozen, eval-mode estimator ONCE on ``cat(rand_images, images)``: the reference runs it on
    ``rand_images`` (:424) and then twice on the same ``images`` (:297 in update_discriminator, :237 in update_inference), three
    no-grad forwards of a deterministic function of which two are identical; the one call on the concatenated batch returns the
    same numbers (no cross-sample coupling in eval-mode BatchNorm) with half the launches and twice the GEMM rows;
  * step() computes the generator's ENCODER once per iteration: the reference runs G twice on the same images with the same weights
    (t_cls_train.py:302 in update_discriminator, :242 in update_inference); the encoder (cunet.py:45-54) has no Dropout and does not see
    the labels, so both runs compute the same encoder activations bit for bit -- the second forward reuses the first one's and runs the
    decoder (AdaIN, Dropout with its own masks, up-convs) only; its backward differentiates through the shared activations as usual;
  * evaluation() runs its B transfers as ONE (B*B)-image pass of G / estimator instead of B passes; D joins
    the batched pass in eval mode and runs per pass in train mode (one power iteration per forward, as in the reference);
  * data parallel (new): G and D gradients are averaged by two ``GradBucketReducer``s; the D
    gradients that g_loss.backward() deposits (and the next d_opt.zero_grad() discards, t_cls_train.py:291) are
    not all-reduced.
"""
import os
import warnings

import torch
import torch.nn as nn

import ops
from cunet import Conditional_UNet
from disc import SNDisc
import disc as _disc
from wu.ddp import GradBucketReducer, broadcast_buffers, is_distributed, ready_order


# update_inference: the discriminator's pass over the fake batch on a second stream, beside the estimator's (A/B switch)
OVERLAP_D_WITH_ESTIMATOR = os.environ.get("WU_GAN_OVERLAP", "1") == "1"
# update_discriminator: the discriminator's real-batch and fake-batch passes on two streams (A/B switch)
OVERLAP_D_PASSES = os.environ.get("WU_GAN_OVERLAP_D", "1") == "1"
# ... also when the discriminator's gradients go through a bucketed reducer (multi-GPU); WU_GAN_OVERLAP_D_DDP=0: sequential passes then
OVERLAP_D_WITH_REDUCER = os.environ.get("WU_GAN_OVERLAP_D_DDP", "1") == "1"
# step(): the generator's encoder once per iteration instead of twice (A/B switch)
SHARE_ENCODER = os.environ.get("WU_GAN_SHARE_ENCODER", "1") == "1"
# step(): the frozen ResNet-101 estimator's no-grad pass over cat(rand_images, images) replayed from a hipGraph (A/B switch; bit-identical)
GRAPH_ESTIMATOR = os.environ.get("WU_GAN_GRAPH_EST", "1") == "1"
# with a pass on the second stream, a parameter's AccumulateGrad node (created on the main stream) receives gradients produced on the
# other one: autograd synchronises the two correctly and says so once per backward (a note about CUDA-graph capture, not an error)
warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")


def disc_batched_sn():
    """SNDisc normalises its ten weights in one batched call at the top of forward (and records SNDisc.sn_done there)."""
    return _disc.BATCHED_SPECTRAL_NORM


class StandInEstimator(nn.Module):
    """Small frozen stand-in for the pre-trained ResNet-101 (classifier.py:106, estimator.py:143) for tests and for step
    timings that should exclude the estimator: average pool to 32x32, two dense layers, RAW outputs (no softmax)."""

    def __init__(self, num_classes, softmax=False):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(32)
        self.features = nn.Sequential(nn.Flatten(), nn.Linear(3 * 32 * 32, 64), nn.ReLU(inplace=True), nn.Linear(64, num_classes))
        self.softmax = softmax            # legacy switch: fold the classifier's softmax head into the module itself
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        y = self.features(self.pool(x))
        return torch.softmax(y, dim=1) if self.softmax else y


class WeatherTransferStep:
    """``mode="cls"``: t_cls_train.py (``self.estimator`` = Softmax(estimator_), :174-178); ``mode="est"``: t_est_train.py
    (``self.estimator`` = the raw 5-signal regressor, :165-166).  ``supervised`` / ``cross_ent``: the t_cls_train flags."""

    def __init__(self, num_classes=5, mode="cls", precision="bf16", lr=1e-4, device="cuda", ddp=None, seed=0,
                 supervised=False, cross_ent=False, estimator=None):
        if mode not in ("cls", "est"):
            raise ValueError("mode must be 'cls' (t_cls_train.py) or 'est' (t_est_train.py)")
        if mode == "est" and (supervised or cross_ent):
            raise ValueError("--supervised / --cross_ent exist in t_cls_train.py only")
        self.mode, self.num_classes = mode, num_classes
        self.supervised, self.cross_ent = bool(supervised), bool(cross_ent)
        dev = torch.device(device)
        torch.manual_seed(seed)
        self.inference = Conditional_UNet(num_classes, precision=precision).to(dev)
        self.discriminator = SNDisc(num_classes, precision=precision).to(dev)
        # estimator_ = raw outputs; estimator = what the script calls self.estimator (t_cls_train.py:172-178)
        self.estimator_ = (estimator if estimator is not None else StandInEstimator(num_classes)).to(dev).eval()
        for p in self.estimator_.parameters():
            p.requires_grad_(False)
        self._est_graphs = {}           # batch shape -> GraphedEstimatorPass (step(): the estimator's no-grad pass)
        self.inference.train()
        self.discriminator.train()
        wd = lr / 20
        self.g_opt = torch.optim.Adam(self.inference.parameters(), lr, betas=(0.0, 0.999), weight_decay=wd, fused=True)
        self.d_opt = torch.optim.Adam(self.discriminator.parameters(), lr, betas=(0.0, 0.999), weight_decay=wd, fused=True)
        self.ddp = is_distributed() if ddp is None else ddp
        self.g_red = self.d_red = None
        if self.ddp:
            self.g_red = GradBucketReducer(ready_order(self.inference), bucket_mb=32.0, ready_order=True, tail_mb=2.0).attach(self.inference)
            self.d_red = GradBucketReducer(list(self.discriminator.parameters()), bucket_mb=32.0, tail_mb=2.0)
            broadcast_buffers(self.discriminator)      # SN weight_u / weight_v identical on every rank

    def _estimator_nograd(self, rand_images, images):
        """estimator_(cat(rand_images, images)) under no_grad.  The HIP ResNet-101 on fp32 device images: one hipGraph replay per iteration
        (wu.resnet.GraphedEstimatorPass, captured per batch shape, again when the estimator's state moves); anything else: the eager call."""
        from wu.resnet import GraphedEstimatorPass, ResNet101Estimator
        est = self.estimator_
        if not (GRAPH_ESTIMATOR and isinstance(est, ResNet101Estimator) and images.is_cuda and images.dtype == torch.float32
                and rand_images.dtype == torch.float32 and images.shape[1:] == rand_images.shape[1:]
                and not torch.cuda.is_current_stream_capturing()):
            return est(torch.cat([rand_images, images]))
        shape = (rand_images.shape[0] + images.shape[0],) + tuple(images.shape[1:])
        g = self._est_graphs.get(shape)
        if g is None or g.stale():
            g = self._est_graphs[shape] = GraphedEstimatorPass(est, shape)
        return g((rand_images, images))

    def _estimator_grad(self, x):
        """estimator_(x) with the generator's graph attached (t_cls_train.py:247-250).  The HIP ResNet-101: forward replayed from a hipGraph
        whose static buffers hold the activations, eager backward (wu.resnet.GraphedEstimatorGradPass; update_inference has exactly one such
        forward alive per iteration); anything else: the eager call."""
        from wu.resnet import GraphedEstimatorGradPass, ResNet101Estimator
        est = self.estimator_
        if not (GRAPH_ESTIMATOR and isinstance(est, ResNet101Estimator) and x.is_cuda and x.dtype == torch.float32 and x.requires_grad
                and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()):
            return est(x)
        key = ("grad",) + tuple(x.shape)
        g = self._est_graphs.get(key)
        if g is None or g.stale():
            g = self._est_graphs[key] = GraphedEstimatorGradPass(est, tuple(x.shape))
        return g(x)

    def _d_stream(self, dev):
        """The probed second stream of the fused U-Net graph (a stream that really runs beside the current one: HIP maps streams
        onto a few hardware queues, unet_graph._side_stream).  G's backward, its other user, starts after both passes have ended."""
        from wu.unet_graph import _side_stream
        return _side_stream(dev)

    def estimator(self, x):
        """``self.estimator`` of the scripts: softmax head in t_cls_train (:174-178), raw in t_est_train."""
        y = self.estimator_(x)
        return torch.softmax(y, dim=1) if self.mode == "cls" else y

    # ------------------------------------------------------------------ t_cls_train.py:288-312 / t_est_train.py:261-283
    def update_discriminator(self, images, labels, c_d=None, pred_labels=None, encoder_cache=None):
        """``pred_labels``: ``self.estimator(images)`` if the caller already has it (step() does); computed here otherwise.
        ``encoder_cache``: step()'s per-iteration dict shared with update_inference (Conditional_UNet.forward)."""
        if self.d_red is not None:
            self.d_red.enabled = True
            self.d_red.zero_grad()
        else:
            self.d_opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            if self.supervised:
                pred_labels = c_d                                                        # :294-295
            elif pred_labels is None:
                pred_labels = self.estimator(images)                                     # :297
        if OVERLAP_D_PASSES and (self.d_red is None or OVERLAP_D_WITH_REDUCER) and images.is_cuda and disc_batched_sn():
            # The pass over the REAL batch needs the images and their labels only: it runs on the second stream, beside the generator's
            # no-grad forward and then beside the pass over the FAKE batch (the kernels of one pass at B = 32 leave much of the chip
            # idle; the two backward chains overlap the same way: autograd replays a node on the stream of its forward).  The only
            # coupling between the passes is the power-iteration state -- the fake pass iterates the u / v the real pass's iteration
            # left (disc.py: one iteration per forward, real first: t_cls_train.py:299,303) -- so the fake pass waits for the real
            # pass's batched normalisation (SNDisc.sn_done), not for its convs.  With a gradient reducer too (round 4): a parameter's
            # two contributions (one per pass, produced on two streams) are summed by autograd's input buffer on the parameter's own
            # stream BEFORE its post-accumulate hook fires, so the bucket's all-reduce is ordered behind both producers
            # (tests/test_gpu_round3.py::test_two_process_data_parallel_real_kernels runs this schedule on two ranks).
            main, side = torch.cuda.current_stream(images.device), self._d_stream(images.device)
            side.wait_stream(main)                           # the labels' producer, last step's optimizer
            with torch.cuda.stream(side):
                real_d_out_pred = self.discriminator(images, pred_labels)[0]             # :299
            real_d_out_pred.record_stream(main)
            with torch.no_grad():
                fake_out = self.inference(images, labels, encoder_cache)                 # :302-303
            if getattr(self.discriminator, "sn_batched_last", False):
                main.wait_event(self.discriminator.sn_done)  # recorded on `side` after the real pass's normalisation
            else:                                            # per-layer normalisation (mixed modes): the whole real pass first
                main.wait_stream(side)
            fake_d_out = self.discriminator(fake_out, labels)[0]
            main.wait_stream(side)
        else:
            with torch.no_grad():
                fake_out = self.inference(images, labels, encoder_cache)                 # :302-303
            real_d_out_pred = self.discriminator(images, pred_labels)[0]                 # :299
            fake_d_out = self.discriminator(fake_out, labels)[0]
        d_loss = ops.dis_hinge(fake_d_out, real_d_out_pred)                              # :305
        d_loss.backward()
        if self.d_red is not None:
            self.d_red.finalize()
        self.d_opt.step()
        return d_loss.detach()

    # ------------------------------------------------------------------ t_cls_train.py:226-286 / t_est_train.py:214-259
    def update_inference(self, images, r_labels, d_labels=None, r_labels_=None, pred_labels=None, encoder_cache=None):
        if self.g_red is not None:
            self.g_red.zero_grad()
            self.d_red.enabled = False             # D's gradients from this backward are discarded, not reduced
        else:
            self.g_opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            if self.supervised:
                pred_labels = d_labels                                                   # :232-235
            elif pred_labels is None:
                pred_labels = self.estimator(images)                                     # :237
        # D's parameter gradients from this backward would be thrown away by the next d_opt.zero_grad()
        # (t_cls_train.py:291): do not compute them (only the data-gradient path through D is needed)
        d_params = list(self.discriminator.parameters())
        for p in d_params:
            p.requires_grad_(False)
        if self.cross_ent and r_labels_ is None:
            raise ValueError("update_inference: --cross_ent needs the class indices r_labels_ (t_cls_train.py:434-438)")
        # D(fake) and estimator(fake) both hang off fake_out and nothing else: D runs on a second stream beside the estimator
        # (autograd replays each node on the stream of its forward, so the two data-gradient chains overlap in backward too).
        # Same kernels, same numbers: the sum into fake_out's gradient follows graph order, not completion order.
        side = self._d_stream(images.device) if OVERLAP_D_WITH_ESTIMATOR and images.is_cuda else None
        try:
            fake_out = self.inference(images, r_labels, encoder_cache)                   # :242
            if side is not None:
                main = torch.cuda.current_stream(images.device)
                side.wait_stream(main)
                fake_out.record_stream(side)
                with torch.cuda.stream(side):
                    fake_d_out = self.discriminator(fake_out, r_labels)[0]               # :243-244
                fake_d_out.record_stream(main)
            else:
                fake_d_out = self.discriminator(fake_out, r_labels)[0]                   # :243-244
        finally:
            for p in d_params:
                p.requires_grad_(True)
        raw_fake = self._estimator_grad(fake_out)
        if self.cross_ent:
            fake_c_out = raw_fake                                                        # :248 last layer is not softmax
        else:
            fake_c_out = torch.softmax(raw_fake, dim=1) if self.mode == "cls" else raw_fake   # :250 (self.estimator: softmax head in t_cls_train)
            r_labels_ = r_labels                                                         # :251
        if side is not None:
            main.wait_stream(side)
        g_loss_adv = ops.gen_hinge(fake_d_out)                                           # :254 adversarial
        g_loss_w = ops.pred_loss(fake_c_out, r_labels_, one_hot=self.cross_ent)          # :256 weather prediction
        diff = torch.mean(torch.abs(fake_out - images), [1, 2, 3])
        lmda = torch.mean(torch.abs(pred_labels - r_labels), 1)
        loss_con = torch.mean(diff / (lmda + (1e-2 if self.supervised else 1e-7)))       # :260-266 reconstruction
        g_loss = g_loss_adv + loss_con + g_loss_w                                        # :268-270
        g_loss.backward()
        if self.g_red is not None:
            self.g_red.finalize()
        self.g_opt.step()
        return g_loss.detach(), g_loss_adv.detach(), loss_con.detach(), g_loss_w.detach()

    # ------------------------------------------------------------------ t_cls_train.py:414-438 (one iteration of train())
    def step(self, images, rand_images, c_d=None, c_r=None):
        """One iteration.  ``c_d`` / ``c_r``: class indices of the two batches (needed with ``supervised``; ``c_r`` also
        feeds ``--cross_ent`` in the i2w branch, :437-438 -- without it the flicker branch :435-436 is taken)."""
        nc = self.num_classes
        if self.supervised:
            if c_d is None or c_r is None:
                raise ValueError("step: --supervised needs the class indices c_d and c_r (t_cls_train.py:419-421)")
            eye = torch.eye(nc, device=images.device)
            rand_labels, d_labels = eye[c_r], eye[c_d]                                   # :420-421
            r_idx = c_r                                                                  # :432
            pred_labels = None
        else:
            with torch.no_grad():
                # estimator(rand_images) (:424) and estimator(images) (:297, :237) in one pass over the concatenated batch
                raw_all = self._estimator_nograd(rand_images, images)
                raw, raw_img = raw_all[:rand_images.shape[0]], raw_all[rand_images.shape[0]:]
                rand_labels = torch.softmax(raw, dim=1) if self.mode == "cls" else raw   # :423
                pred_labels = torch.softmax(raw_img, dim=1) if self.mode == "cls" else raw_img
            d_labels = None
            r_idx = (c_r if c_r is not None else torch.argmax(raw, dim=1)) if self.cross_ent else None   # :436,438
        # both generator forwards of an iteration see the same images and the same weights (G is updated last): the encoder is computed
        # by the first and reused by the second (Conditional_UNet.forward, encoder_cache)
        enc = {} if SHARE_ENCODER and getattr(self.inference, "fused", False) else None
        d_loss = self.update_discriminator(images, rand_labels, d_labels, pred_labels, enc)   # :429
        g_losses = self.update_inference(images, rand_labels, d_labels, r_idx, pred_labels, enc)   # :432-438
        return (d_loss,) + g_losses

    # ------------------------------------------------------------------ t_cls_train.py:314-367 / t_est_train.py:285-332
    @torch.no_grad()
    def evaluation(self, images, labels, ref_labels, max_images=1024):
        """The test-time sweep: every image of the batch transferred to every reference row's conditioning, averaged losses.
        The reference runs B passes of G / estimator / D over the B-image batch; here G and the estimator run ONE pass over
        the B*B (image, condition) pairs (chunked to ``max_images`` images).  G / D stay in whatever train / eval mode they
        are in (the reference never calls .eval(): Dropout and the power iteration are active there), and D follows its mode:

        * D in EVAL mode: D(images, labels) once and D on all fakes in the batched pass -- identical means, since every
          per-pass loss is a mean over equally sized blocks;
        * D in TRAIN mode (the reference's actual mode, t_cls_train.py:331-341): each of the reference's 2*B discriminator
          forwards runs one power iteration, so every pass sees its own W/sigma and D's ``weight_u`` / ``weight_v`` -- training
          state -- have moved 2*B iterations when the sweep returns.  That is reproduced exactly: D runs per pass, real batch
          then fake batch, in the reference's order (2*B forwards of B images: the same FLOPs as the batched form, more
          launches; G and the estimator, 95 % of the work, stay batched).

        Returns (dict of device scalars g_loss_adv, g_loss_l1, g_loss_w, d_loss; fake images (B, B, 3, H, W): [i] = transfers
        to ref_labels[i])."""
        bs, nc = images.shape[0], ref_labels.shape[1]
        if labels.dim() == 1:                                                            # :327-329 (--one_hot)
            labels = torch.eye(nc, device=images.device)[labels]
        d_train = self.discriminator.training
        real_d = None if d_train else self.discriminator(images, labels)[0]              # :340
        est = self.estimator_ if self.mode == "cls" else self.estimator                  # :338 / t_est_train.py:309
        rows = max(1, max_images // bs)
        fakes, fake_d, fake_c = [], [], []
        for i0 in range(0, bs, rows):
            r = ref_labels[i0:i0 + rows]
            cond = r.repeat_interleave(bs, dim=0)                                        # :336 ref_labels[i] tiled B times
            x = images.repeat(r.shape[0], 1, 1, 1)
            f = self.inference(x, cond)                                                  # :337
            fakes.append(f)
            fake_c.append(est(f))                                                        # :338
            if not d_train:
                fake_d.append(self.discriminator(f, cond)[0])                            # :341
        fake = torch.cat(fakes)
        cond_all = ref_labels.repeat_interleave(bs, dim=0)
        if d_train:
            d_terms = []
            for i in range(bs):                                                          # the reference's order, pass by pass
                real_i = self.discriminator(images, labels)[0]                           # :340 power iteration 2i+1
                fd_i = self.discriminator(fake[i * bs:(i + 1) * bs], cond_all[i * bs:(i + 1) * bs])[0]   # :341 iteration 2i+2
                fake_d.append(fd_i)
                d_terms.append(ops.dis_hinge(fd_i, real_i))                              # :352
            d_loss = torch.stack(d_terms).mean()
        fd = torch.cat(fake_d)
        if not d_train:
            d_loss = ops.dis_hinge(fd, real_d)                                           # :352
        losses = {
            "g_loss_adv": ops.gen_hinge(fd),                                             # :349
            "g_loss_l1": ops.l1_loss(fake, images.repeat(bs, 1, 1, 1)),                  # :350
            "g_loss_w": ops.pred_loss(torch.cat(fake_c), cond_all),                      # :351
            "d_loss": d_loss,
        }
        return losses, fake.view(bs, bs, *images.shape[1:])
