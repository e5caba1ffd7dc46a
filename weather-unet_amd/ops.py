"""Losses and label helpers behind the reference's ``ops`` module name (reference ops.py:14-83).

The five loss one-liners sit inside the GAN step: tiny reductions on (N,1) / (N,nc) / (N,3,H,W) tensors
that stay fp32 torch ops.  The label / image helpers are host-side conveniences (off the hot path); they keep
the reference's names and call signatures because its scripts reach them through ``from ops import *``, which also leaks
``F``, ``Variable``, ``np``, ``nn``, ``torch`` (t_cls_train.py:328 uses ``F`` obtained that way) -- hence ``__all__`` below.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Variable

xp = np

__all__ = ["F", "Variable", "np", "nn", "torch", "xp",
           "adv_loss", "l1_loss", "feat_loss", "pred_loss", "dis_hinge", "gen_hinge",
           "soft_transform", "vector_to_one_hot", "get_rand_labels", "get_sequential_labels", "Variable_Float", "make_table_img"]


def _same_size(a, b):
    assert a.size() == b.size(), 'The size of a and b is different.{}!={}'.format(a.size(), b.size())


# ---- losses used inside the step (t_cls_train.py:254-270,305; t_est_train.py:232-243,274) -----------------------------------
def adv_loss(a, b):
    """ops.py:18-20: mean squared error between two same-sized tensors."""
    _same_size(a, b)
    return F.mse_loss(a, b)


def l1_loss(a, b):
    """ops.py:22-24: mean absolute error (logged as g_loss_l1, not part of g_loss).  fp32 device tensors take the fused
    one-pass kernel (loss and its gradient together); anything else is the stock torch op."""
    _same_size(a, b)
    if a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.numel() > 0:
        # the fused kernel reads 16-byte groups: a contiguous view at an odd storage offset (x.flatten()[1:], a row slice of an
        # odd-width tensor) is legal input for the reference's F.l1_loss and takes the stock op here
        ac, bc = a.contiguous(), b.contiguous()
        if ac.data_ptr() % 16 == 0 and bc.data_ptr() % 16 == 0:
            from wu.functional import l1_mean
            return l1_mean(ac, bc)
    return F.l1_loss(a, b)


def feat_loss(a, b):
    """ops.py:26-27: mean over feature-map pairs of their L1 distance (SNDisc's c1..c4 arrive in the compute dtype: fp32 here)."""
    per_map = [F.l1_loss(u.float(), v.float()) for u, v in zip(a, b)]
    return torch.stack(per_map).mean()


def pred_loss(preds, labels, one_hot=False):
    """ops.py:29-40: weather-prediction loss -- cross entropy on un-softmaxed logits vs class indices (``--cross_ent``),
    else MSE against the (soft) label rows."""
    return F.cross_entropy(preds, labels) if one_hot else F.mse_loss(preds, labels)


def dis_hinge(dis_fake, dis_real):
    """ops.py:42-45: hinge loss of the discriminator."""
    return F.relu(1. - dis_real).mean() + F.relu(1. + dis_fake).mean()


def gen_hinge(dis_fake):
    """ops.py:47-48: hinge loss of the generator."""
    return (-dis_fake).mean()


# ---- host-side helpers (off the hot path; names kept for `from ops import *`) -----------------------------------------------
def soft_transform(x, std=0.05):
    """Label smoothing by additive Gaussian noise (ops.py:14-16)."""
    return x + std * torch.randn_like(x)


def vector_to_one_hot(vec):
    """One-hot of the arg-max along dim 0, same shape / dtype as `vec` (ops.py:50-54)."""
    idx = vec.argmax(dim=0)
    return torch.movedim(F.one_hot(idx, vec.shape[0]), -1, 0).to(vec.dtype)


def get_rand_labels(num_classes, batch_size, one_hot=False):
    """Random conditioning rows on the GPU (ops.py:56-60): U(-1, 1) signal rows, or -- with one_hot -- one-hot rows of uniformly
    drawn classes (the reference passes its float tensor to F.one_hot there, which raises; never called with one_hot=True)."""
    if one_hot:
        cls = torch.randint(num_classes, (batch_size,))
        return F.one_hot(cls, num_classes).float().to('cuda')
    return (torch.rand(batch_size, num_classes) * 2 - 1).to('cuda')


def get_sequential_labels(num_classes, batch_size, one_hot=False):
    """Classes 0, 1, .., nc-1, 0, 1, .. for a batch (ops.py:62-71): float class ids, or their one-hot rows."""
    cls = torch.arange(batch_size) % num_classes
    out = torch.eye(num_classes)[cls] if one_hot else cls.float()
    return out.to('cuda')


def Variable_Float(x, batch_size):
    """(batch_size, 1) constant on the GPU (ops.py:73-74; the real / fake targets of t_cls_train.py:77-78)."""
    return torch.full((batch_size, 1), float(x), device='cuda', requires_grad=False)


def make_table_img(images, ref_images, results):
    """Input batch and its transfers stacked along the height axis (ops.py:77-83; `ref_images` is unused there too)."""
    return torch.cat([images, *results], dim=2)
