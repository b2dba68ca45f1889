"""Layer-level operators of the hot path.  Every one of them runs HIP kernels from
libjspsr_hip.so (jspsr_amd.ops / jspsr_amd.kernels); there is no CPU or eager fallback.

Activations travel as NHWC tensors (B, H, W, C) in the compute dtype (fp32, or bf16 storage with
fp32 accumulation and fp32 statistics).  The module boundary is the reference's: contiguous fp32
NCHW tensors on the device (utils/utils.py:156-179); one-channel tensors are identical in both
layouts, so the DEM and the output need no conversion.
"""
from __future__ import annotations

import os

import torch

from . import kernels as K
from . import ops

_compute_dtype = torch.float32
_gate_sync = None   # set by jspsr_amd.tiling during sharded inference (scene-wide gate statistics)
_offset_probe = None   # a list during sharded inference: the models append their learned-offset tensors (halo check)


class compute_dtype:
    """Context manager: storage dtype of activations (torch.float32 or torch.bfloat16)."""

    def __init__(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self.dtype = dtype

    def __enter__(self):
        global _compute_dtype
        self.prev, _compute_dtype = _compute_dtype, self.dtype

    def __exit__(self, *a):
        global _compute_dtype
        _compute_dtype = self.prev


def _gpu(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("jspsr_amd runs on the GPU only: move the module and its inputs to cuda "
                           "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")
    return x


def from_nchw(x: torch.Tensor) -> torch.Tensor:
    """Boundary tensor (B,C,H,W) fp32 -> NHWC activations in the compute dtype, channel-padded."""
    _gpu(x)
    B, C, H, W = x.shape
    e = K.epc(_compute_dtype)
    if x.dtype == torch.float32 and x.is_contiguous() and not (x.requires_grad and torch.is_grad_enabled()):
        return K.nchw_to_nhwc(x, _compute_dtype, (C + e - 1) // e * e)      # cast + channels-last + pad in one launch
    x = x.reshape(B, H, W, 1) if C == 1 else x.permute(0, 2, 3, 1)           # inputs that carry a gradient: torch ops
    return ops.pad_channels(x.to(_compute_dtype), e).contiguous()


def to_nchw_f32(x: torch.Tensor) -> torch.Tensor:
    """NHWC activations -> planar fp32 (B,C,H,W) (what K1 and the caller consume)."""
    B, H, W, C = x.shape
    if C == 1:
        return x.float().reshape(B, 1, H, W)
    return x.permute(0, 3, 1, 2).float().contiguous()


SliceBuffer = ops.SliceBuffer


def conv2d(x, weight, bias=None, stride=1, padding=0, relu=False, dest=None):
    return ops.conv2d(x, weight, bias, stride, padding, relu, dest)


def conv_transpose2d(x, weight):
    """ConvTranspose2d k3 s2 p1 op1, no bias (basics.py:69-77)."""
    return ops.conv_transpose2d(x, weight)


_nbt_pending = None   # inside `count_batches()`: the BatchNorm step counters to bump, once, at the end


class count_batches:
    """BatchNorm2d bumps `num_batches_tracked` once per training forward (one tiny kernel per layer, 70 per
    step for JSPSR).  Inside this context the counters are collected and bumped by one multi-tensor add."""

    def __enter__(self):
        global _nbt_pending
        self.prev, _nbt_pending = _nbt_pending, []
        return self

    def __exit__(self, *a):
        global _nbt_pending
        todo, _nbt_pending = _nbt_pending, self.prev
        if todo and a[0] is None:
            seen = {}
            for t in todo:                       # a layer applied k times in one forward counts k batches
                seen[id(t)] = (t, seen.get(id(t), (t, 0))[1] + 1)
            once = [t for t, k in seen.values() if k == 1]
            if once:
                torch._foreach_add_(once, 1)
            for t, k in seen.values():
                if k > 1:
                    t += k
        return False


def _count_batch(bn):
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _nbt_pending is not None:
            _nbt_pending.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1


def batch_norm(x, bn: torch.nn.BatchNorm2d, relu=False, residual=None, res_scale=1.0, partial=None, dest=None):
    """BatchNorm2d (+ `* res_scale + residual`) (+ ReLU): basics.py:113-123.  dest = (SliceBuffer, channel):
    write the result into that channel slice instead of a fresh tensor."""
    training = bn.training or bn.running_mean is None
    _count_batch(bn)
    return ops.batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, training,
                          relu, residual, res_scale, partial, dest)


def conv_bn(x, weight, bn: torch.nn.BatchNorm2d, stride=1, padding=0, relu=False, residual=None, res_scale=1.0, dest=None):
    """Bias-free conv -> BatchNorm (+residual, +ReLU).  In training mode the batch statistics come out of the
    convolution's epilogue (fp32 accumulators), so BatchNorm makes one pass over the tensor instead of two."""
    if bn.training or bn.running_mean is None:
        y, st = ops.conv2d_with_stats(x, weight, stride, padding)
        return batch_norm(y, bn, relu, residual, res_scale, partial=st, dest=dest)
    if not torch.is_grad_enabled():     # inference: BatchNorm folded into the conv epilogue, one launch
        return ops.conv_bn_infer(x, weight, bn.weight, bn.bias, _bn_state(bn), stride, padding, relu, residual,
                                 res_scale, dest)
    return batch_norm(ops.conv2d(x, weight, None, stride, padding), bn, relu, residual, res_scale, dest=dest)


def _bn_state(bn: torch.nn.BatchNorm2d):
    training = bn.training or bn.running_mean is None
    _count_batch(bn)
    return (bn.running_mean, bn.running_var, bn.momentum, bn.eps, training)


def res_unit(x, conv1, bn1, conv2, bn2, downsample, stride, scale, act, dest=None, grad_extra=None):
    """BasicBlock (basics.py:88-123) as one autograd node (ops._ResUnit)."""
    bns = [_bn_state(bn1), _bn_state(bn2)]
    wd = gd = bd = None
    if downsample is not None:
        wd, gd, bd = downsample[0].weight, downsample[1].weight, downsample[1].bias
        bns.append(_bn_state(downsample[1]))
    return ops.res_unit(x, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias, wd, gd, bd,
                        stride, scale, act, tuple(bns), dest, grad_extra)


def channel_gate(x, w1, w2):
    """x * sigmoid(MLP(avgpool x) + MLP(maxpool x)), resnet_cbam.py:49-53 + basics.py:57-58."""
    if _gate_sync is not None:   # sharded inference: pooled statistics come from all strips
        avg, mx = _gate_sync.pool(x.contiguous())
        s = ops._gate_mlp(avg, mx, w1.detach().float(), w2.detach().float()).contiguous()
        return K.gate_scale(x.contiguous(), s)
    return ops.channel_gate(x, w1, w2)


def propagate_head(dem, head, w, b, scale=1.0):
    """K1h: sigmoid + propagation straight from the merged 1x1 head's NHWC output (ops.propagate_head)."""
    return ops.propagate_head(dem.contiguous(), head, w, b, scale)


# JSPSR_HEAD_PLANES=0: the models feed the propagation step from the 32-channel NHWC head again (K1h / K1hd, rounds 2-3)
planar_heads = os.environ.get("JSPSR_HEAD_PLANES", "1") != "0"


def heads_propagate(dem, feature, conv_weight, conv_offset, w, b, scale=1.0):
    """The two 1x1 heads (spn.py:66-68; LRRU.py:238-247) + Sigmoid + zero centre offset + PostProcessor.forward
    (spn.py:43,69-73,99-118) as the models run them.  dem (B,1,H,W) fp32 detached, feature NHWC in the compute dtype,
    conv_weight / conv_offset the reference-named nn.Conv2d modules (9 and 16 rows).
    Round 4: the heads write the (B,25,H,W) fp32 planes the propagation kernel of the PUBLIC boundary reads (K1c,
    ops.head_planes -> ops.propagate_logits): the in-model step is the roofline kernel itself, nothing is padded or
    transposed, and the learned offsets / affinity logits stay fp32 whatever the storage type of the network."""
    dem = dem.contiguous()
    if planar_heads and ops.head_planes_ok(feature):
        planes = ops.head_planes(feature, conv_weight.weight, conv_weight.bias, conv_offset.weight, conv_offset.bias)
        if _offset_probe is not None:
            _offset_probe.append(planes[:, 9:].permute(0, 2, 3, 1))
        return ops.propagate_logits(dem, planes, w, b, scale)
    w_all, b_all = ops.merge_heads(conv_weight.weight, conv_weight.bias, conv_offset.weight, conv_offset.bias)
    head = conv2d(feature, w_all, b_all)
    if _offset_probe is not None:
        _offset_probe.append(ops.split_head(head)[1])
    return ops.propagate_head(dem, head, w, b, scale)


def cat(tensors):
    return torch.cat(tuple(tensors), 3)


def sigmoid(x):
    return torch.sigmoid(x)


def propagate(dem, weight, offset, w, b, scale=1.0):
    """K1: planar fp32 operands, one coalesced stream per tap plane."""
    return ops.propagate(dem.contiguous(), weight.contiguous(), offset.contiguous(), w, b, scale)
