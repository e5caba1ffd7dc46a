"""The GAN training step of the reference on synthetic tensors (BASELINE.json configs[2], configs[3]).

``WeatherTransferStep`` is the harness counterpart of ``WeatherTransfer.update_discriminator`` /
``update_inference`` (class-conditioned: t_cls_train.py:226-312; soft-label / estimator-conditioned:
t_est_train.py:214-283): same order of forwards / backwards, same losses (ops.py), same optimisers
(Adam lr 1e-4, betas (0, 0.999), weight_decay lr/20, t_cls_train.py:184-185).  Datasets, TensorBoard and the
pickled ResNet-101 estimator are out of scope: the estimator is a small frozen stand-in
``(N,3,H,W) -> (N,nc)`` through which gradients flow to the generator, as they do through the real one.

Differences from the reference loop, all numerically neutral:
  * update_discriminator runs G under no_grad (the reference builds G's graph and then detaches it,
    t_cls_train.py:302-303);
  * no ``.item()`` host syncs inside the step (the reference forces six per step, :275-282,310-312): losses are
    returned as device tensors;
  * data parallel (new): G and D gradients are averaged by two ``GradBucketReducer``s; the D
    gradients that g_loss.backward() deposits (and the next d_opt.zero_grad() discards, t_cls_train.py:291) are
    not all-reduced.
"""
import torch
import torch.nn as nn

import ops
from cunet import Conditional_UNet
from disc import SNDisc
from wu.ddp import GradBucketReducer, broadcast_buffers, is_distributed, ready_order


class StandInEstimator(nn.Module):
    """Frozen stand-in for the pre-trained ResNet-101 classifier / estimator (classifier.py:106, estimator.py:143;
    OUT of scope).  ``softmax=True`` mimics the classifier head used by t_cls_train (``self.estimator``, softmax
    output), ``False`` the 5-signal regression estimator of t_est_train."""

    def __init__(self, num_classes, softmax=True):
        super().__init__()
        # The stand-in must not show up in the step time (the real ResNet-101 is out of scope): average pool to 32x32, then two
        # dense layers -- plain GEMMs, no MIOpen convolution (its im2col / naive fallbacks cost ~3 ms per GAN iteration).
        self.pool = nn.AdaptiveAvgPool2d(32)
        self.features = nn.Sequential(nn.Flatten(), nn.Linear(3 * 32 * 32, 64), nn.ReLU(inplace=True), nn.Linear(64, num_classes))
        self.softmax = softmax
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        y = self.features(self.pool(x))
        return torch.softmax(y, dim=1) if self.softmax else y


class WeatherTransferStep:
    def __init__(self, num_classes=5, mode="cls", precision="bf16", lr=1e-4, device="cuda", ddp=None, seed=0):
        if mode not in ("cls", "est"):
            raise ValueError("mode must be 'cls' (t_cls_train.py) or 'est' (t_est_train.py)")
        self.mode, self.num_classes = mode, num_classes
        dev = torch.device(device)
        torch.manual_seed(seed)
        self.inference = Conditional_UNet(num_classes, precision=precision).to(dev)
        self.discriminator = SNDisc(num_classes, precision=precision).to(dev)
        self.estimator = StandInEstimator(num_classes, softmax=(mode == "cls")).to(dev).eval()
        self.inference.train()
        self.discriminator.train()
        wd = lr / 20
        self.g_opt = torch.optim.Adam(self.inference.parameters(), lr, betas=(0.0, 0.999), weight_decay=wd, fused=True)
        self.d_opt = torch.optim.Adam(self.discriminator.parameters(), lr, betas=(0.0, 0.999), weight_decay=wd, fused=True)
        self.ddp = is_distributed() if ddp is None else ddp
        self.g_red = self.d_red = None
        if self.ddp:
            self.g_red = GradBucketReducer(ready_order(self.inference), bucket_mb=12.0, ready_order=True).attach(self.inference)
            self.d_red = GradBucketReducer(list(self.discriminator.parameters()), bucket_mb=12.0)
            broadcast_buffers(self.discriminator)      # SN weight_u / weight_v identical on every rank

    # ------------------------------------------------------------------ t_cls_train.py:288-312 / t_est_train.py:261-283
    def update_discriminator(self, images, labels):
        if self.d_red is not None:
            self.d_red.enabled = True
            self.d_red.zero_grad()
        else:
            self.d_opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            pred_labels = self.estimator(images)
            fake_out = self.inference(images, labels)
        real_d_out_pred = self.discriminator(images, pred_labels)[0]
        fake_d_out = self.discriminator(fake_out, labels)[0]
        d_loss = ops.dis_hinge(fake_d_out, real_d_out_pred)
        d_loss.backward()
        if self.d_red is not None:
            self.d_red.finalize()
        self.d_opt.step()
        return d_loss.detach()

    # ------------------------------------------------------------------ t_cls_train.py:226-286 / t_est_train.py:214-259
    def update_inference(self, images, r_labels):
        if self.g_red is not None:
            self.g_red.zero_grad()
            self.d_red.enabled = False             # D's gradients from this backward are discarded, not reduced
        else:
            self.g_opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            pred_labels = self.estimator(images)
        # D's parameter gradients from this backward would be thrown away by the next d_opt.zero_grad()
        # (t_cls_train.py:291): do not compute them (only the data-gradient path through D is needed)
        d_params = list(self.discriminator.parameters())
        for p in d_params:
            p.requires_grad_(False)
        try:
            fake_out = self.inference(images, r_labels)
            fake_d_out = self.discriminator(fake_out, r_labels)[0]
        finally:
            for p in d_params:
                p.requires_grad_(True)
        fake_c_out = self.estimator(fake_out)
        g_loss_adv = ops.gen_hinge(fake_d_out)                                   # adversarial
        g_loss_w = ops.pred_loss(fake_c_out, r_labels)                           # weather prediction (MSE)
        diff = torch.mean(torch.abs(fake_out - images), [1, 2, 3])
        lmda = torch.mean(torch.abs(pred_labels - r_labels), 1)
        loss_con = torch.mean(diff / (lmda + 1e-7))                              # reconstruction
        g_loss = g_loss_adv + loss_con + g_loss_w
        g_loss.backward()
        if self.g_red is not None:
            self.g_red.finalize()
        self.g_opt.step()
        return g_loss.detach(), g_loss_adv.detach(), loss_con.detach(), g_loss_w.detach()

    # ------------------------------------------------------------------ t_cls_train.py:424-438 (one iteration of train())
    def step(self, images, rand_images):
        with torch.no_grad():
            rand_labels = self.estimator(rand_images)
        d_loss = self.update_discriminator(images, rand_labels)
        g_losses = self.update_inference(images, rand_labels)
        return (d_loss,) + g_losses
