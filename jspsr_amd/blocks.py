"""Parameter containers whose state_dict keys equal the reference's, with
forwards expressed through jspsr_amd.engine (HIP-backed operators).

Key patterns reproduced: ``<unit>.conv.0.{weight,bias}``, ``<unit>.conv.bn.*``,
``<unit>.camb.fc.{0,2}.weight`` (reference Basic2d, models/components/basics.py:23-60);
``<up>.dconv.{0,1,bn}`` (Basic2dTrans, :63-85); ``<block>.{conv1,bn1,conv2,bn2,downsample.{0,1}}``
(BasicBlock, :88-123).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine as E
from . import ops


class HotPathModule(nn.Module):
    """Base of the drop-in model classes.  The reference's loop clears gradients with
    ``model.zero_grad(set_to_none=True)`` (train/train_utils.py:210); when a ``GradReducer`` has been attached
    (``reducer.attach(model)``) the gradients alias its flat communication buffer, so clearing them means zeroing that
    buffer and keeping the aliases."""

    def zero_grad(self, set_to_none: bool = True):
        red = self.__dict__.get("_grad_reducer")
        if red is not None:
            red.zero_grad()
        else:
            super().zero_grad(set_to_none)

    # The conv kernels read re-laid ("packed") copies of the weights, cached per Parameter and validated by torch's
    # version counter (ops._packed).  Writes that bypass the counter (`p.data.mul_()`, `p.data.copy_()`: EMA, clipping,
    # re-initialisation) would leave stale copies in use; the usual places such writes are followed by -- a train() /
    # eval() switch, load_state_dict() -- therefore drop every cached copy.  Anything else that writes through `.data`
    # must call ops.invalidate_packed_weights() itself.
    def train(self, mode: bool = True):
        ops.invalidate_packed_weights()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        ops.invalidate_packed_weights()
        return out


class ChannelGate(nn.Module):
    """Holds the shared bias-free 1x1 MLP of ChannelAttention (resnet_cbam.py:36-53)."""

    def __init__(self, channels: int, ratio: int = 16):
        super().__init__()
        self.fc = nn.Sequential()
        self.fc.add_module("0", nn.Conv2d(channels, channels // ratio, 1, bias=False))
        self.fc.add_module("2", nn.Conv2d(channels // ratio, channels, 1, bias=False))

    def forward(self, x):
        return E.channel_gate(x, self.fc[0].weight, self.fc[1].weight)


class ConvUnit(nn.Module):
    """[channel gate ->] conv kxk (bias iff no BN) [-> BN] [-> ReLU]."""

    def __init__(self, cin, cout, kernel_size=3, bn=True, relu=True, gate=False):
        super().__init__()
        if gate:
            self.camb = ChannelGate(cin)
        self.conv = nn.Sequential()
        self.conv.add_module("0", nn.Conv2d(cin, cout, kernel_size, 1, kernel_size // 2, bias=not bn))
        if bn:
            self.conv.add_module("bn", nn.BatchNorm2d(cout))
        self.k = kernel_size
        self.relu = relu

    def forward(self, x, dest=None):
        if hasattr(self, "camb"):
            x = self.camb(x)
        c = self.conv[0]
        if hasattr(self.conv, "bn"):
            return E.conv_bn(x, c.weight, self.conv.bn, 1, self.k // 2, relu=self.relu, dest=dest)
        return E.conv2d(x, c.weight, c.bias, 1, self.k // 2, relu=self.relu, dest=dest)  # bias + ReLU in the epilogue


class UpUnit(nn.Module):
    """ConvUnit(gate, BN, ReLU) -> ConvTranspose2d k3 s2 p1 op1 -> BN -> ReLU."""

    def __init__(self, cin, cout):
        super().__init__()
        self.dconv = nn.Sequential()
        self.dconv.add_module("0", ConvUnit(cin, cout, 3, bn=True, relu=True, gate=True))
        self.dconv.add_module("1", nn.ConvTranspose2d(cout, cout, 3, 2, 1, 1, bias=False))
        self.dconv.add_module("bn", nn.BatchNorm2d(cout))

    def forward(self, x, dest=None):
        y = self.dconv[0](x)
        bn = self.dconv.bn
        if not torch.is_grad_enabled() and not (bn.training or bn.running_mean is None):   # inference: one launch per phase
            return ops.conv_transpose_bn_infer(y, self.dconv[1].weight, bn.weight, bn.bias, E._bn_state(bn), True, dest)
        y = E.conv_transpose2d(y, self.dconv[1].weight)
        return E.batch_norm(y, bn, relu=True, dest=dest)


class ResUnit(nn.Module):
    """relu(bn2(conv3x3(relu(bn1(conv3x3_s(x))))) * scale + shortcut(x))."""

    def __init__(self, cin, cout, stride=1, project=False, act=True, scale=1.0):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.stride, self.act, self.scale = stride, act, scale

    fused = True   # one autograd node per block (ops._ResUnit); False = the op-by-op composition below

    def forward(self, x, dest=None, grad_extra=None):
        if self.fused:
            return E.res_unit(x, self.conv1, self.bn1, self.conv2, self.bn2, self.downsample, self.stride, self.scale,
                              self.act, dest, grad_extra)
        if grad_extra is not None:
            raise RuntimeError("grad_extra needs the fused block")
        y = E.conv_bn(x, self.conv1.weight, self.bn1, self.stride, 1, relu=True)
        if self.downsample is not None:
            r = E.conv_bn(x, self.downsample[0].weight, self.downsample[1], self.stride, 0)
        else:
            r = x
        return E.conv_bn(y, self.conv2.weight, self.bn2, 1, 1, relu=self.act, residual=r, res_scale=self.scale,
                         dest=dest)
