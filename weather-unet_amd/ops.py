"""Losses and label helpers -- same names and behaviour as the reference's ops.py (ops.py:14-83).

These are tiny reductions on (N,1) / (N,nc) / (N,3,H,W) tensors: they stay torch ops in fp32
(SURVEY.md 8a12).  ``from ops import *`` also leaks ``F``, ``Variable``, ``np``, ``nn``, ``torch`` exactly
as the reference's does (t_cls_train.py:328 relies on ``F``)."""
import os  # noqa: F401

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Variable
import numpy as np

xp = np


def soft_transform(x, std=0.05):
    dist = torch.zeros_like(x).normal_(0, std=std)
    return x + dist


def adv_loss(a, b):
    assert a.size() == b.size(), 'The size of a and b is different.{}!={}'.format(a.size(), b.size())
    return F.mse_loss(a, b)


def l1_loss(a, b):
    assert a.size() == b.size(), 'The size of a and b is different.{}!={}'.format(a.size(), b.size())
    return F.l1_loss(a, b)


def feat_loss(a, b):
    return torch.mean(torch.stack([F.l1_loss(a_.float(), b_.float()) for a_, b_ in zip(a, b)]))


def pred_loss(preds, labels, one_hot=False):
    if one_hot:
        criterion = nn.CrossEntropyLoss()
    else:
        criterion = nn.MSELoss()
    return criterion(preds, labels)


def dis_hinge(dis_fake, dis_real):
    loss = torch.mean(torch.relu(1. - dis_real)) + \
        torch.mean(torch.relu(1. + dis_fake))
    return loss


def gen_hinge(dis_fake):
    return torch.mean(-dis_fake)


def vector_to_one_hot(vec):
    arg = torch.argmax(vec, 0, keepdim=True)
    one_hot = torch.zeros_like(vec)
    one_hot.scatter_(0, arg, 1).float()
    return one_hot


def get_rand_labels(num_classes, batch_size, one_hot=False):
    label = torch.FloatTensor(batch_size, num_classes).uniform_(-1, 1)
    if one_hot:
        label = F.one_hot(label, num_classes)
    return label.to('cuda')


def get_sequential_labels(num_classes, batch_size, one_hot=False):
    rep = batch_size // num_classes + 1
    if one_hot:
        arr = xp.eye(num_classes, dtype=xp.float32)
        arr = xp.tile(arr, (rep, 1))[:batch_size]
        return torch.from_numpy(arr).float().to('cuda')
    else:
        arr = torch.arange(num_classes, dtype=torch.float32)
        arr = arr.repeat(rep)[:batch_size]
        return arr.to('cuda')


def Variable_Float(x, batch_size):
    return Variable(torch.full((batch_size, 1), float(x), device='cuda'), requires_grad=False)


def make_table_img(images, ref_images, results):
    in_out_img = torch.cat([images] + results, dim=2)
    return in_out_img
