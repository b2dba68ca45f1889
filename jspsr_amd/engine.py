"""Layer-level operators of the hot path, all on the GPU.

Activations are torch tensors of logical shape (B,C,H,W) held in channels-last memory (NHWC):
the layout the implicit-GEMM kernels consume.  One-channel tensors are identical in both
layouts, so the module boundary (reference: contiguous NCHW fp32, utils/utils.py:156-179)
needs no conversion for the DEM and the output.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops

CL = torch.channels_last


def _gpu(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("jspsr_amd runs on the GPU only: move the module and its inputs to cuda "
                           "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")
    return x


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    return _gpu(x).contiguous(memory_format=CL)


def conv2d(x, weight, bias=None, stride=1, padding=0):
    return F.conv2d(to_nhwc(x), weight, bias, stride, padding)


def conv_transpose2d(x, weight):
    """ConvTranspose2d k3 s2 p1 op1, no bias (basics.py:69-77)."""
    return F.conv_transpose2d(to_nhwc(x), weight, None, 2, 1, 1)


def batch_norm(x, bn: torch.nn.BatchNorm2d, relu=False, residual=None, res_scale=1.0):
    """BatchNorm2d (+ `* res_scale + residual`) (+ ReLU): basics.py:113-123."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
    if residual is not None:
        y = y * res_scale + residual if res_scale != 1.0 else y + residual
    return F.relu(y) if relu else y


def bias_act(x, relu=True):
    return F.relu(x) if relu else x


def channel_gate(x, w1, w2):
    """x * sigmoid(MLP(avgpool x) + MLP(maxpool x)), resnet_cbam.py:49-53 + basics.py:57-58."""
    avg = x.mean((2, 3), keepdim=True)
    mx = x.amax((2, 3), keepdim=True)
    mlp = lambda v: F.conv2d(F.relu(F.conv2d(v, w1)), w2)
    return x * torch.sigmoid(mlp(avg) + mlp(mx))


def cat(tensors):
    return torch.cat(tensors, 1)


def sigmoid(x):
    return torch.sigmoid(x)


def propagate(dem, weight, offset, w, b, scale=1.0):
    """K1 wants planar (NCHW) weight / offset: one coalesced stream per tap plane."""
    return ops.propagate(dem.contiguous(), weight.contiguous(), offset.contiguous(), w, b, scale)
