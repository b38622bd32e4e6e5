"""Image losses of the reference (python/losses.py) on torch tensors.  `img` and `ref_img` have the
same shape; every loss is normalised by `dr.width(img)` = the number of entries."""
from __future__ import annotations

import math

import torch


def average(img, ref_img=None, shape=None):
    return img.sum() / img.numel()


def l1(img, ref_img, shape=None):
    return (img - ref_img).abs().sum() / img.numel()


def l2(img, ref_img, shape=None):
    return ((img - ref_img) ** 2).sum() / img.numel()


def root_mean_squared_error(*args, **kwargs):
    return torch.sqrt(l2(*args, **kwargs))


def huber(img, ref_img, shape=None, delta=1.0):
    # NB the reference compares the signed residual with delta (losses.py:18), kept as is
    residual = img - ref_img
    loss = torch.where(residual < delta, 0.5 * residual ** 2, delta * residual.abs() - 0.5 * delta)
    return loss.sum() / img.numel()


def mean_relative_absolute_error(img, ref_img, shape=None, epsilon=1e-2):
    return ((img - ref_img).abs() / (ref_img.abs() + epsilon)).sum() / img.numel()


def mean_relative_squared_error(img, ref_img, shape=None, epsilon=1e-2):
    return ((img - ref_img) ** 2 / (ref_img ** 2 + epsilon)).sum() / img.numel()


def root_mean_relative_squared_error(*args, **kwargs):
    return torch.sqrt(mean_relative_squared_error(*args, **kwargs))


def psnr(img, ref_img, max_value=1.0, shape=None):
    mse = ((img - ref_img) ** 2).sum() / img.numel()
    return 20.0 * (math.log(max_value) / math.log(10.0)) - (10.0 / math.log(10.0)) * torch.log(mse)
