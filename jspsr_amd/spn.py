"""Affinity/offset generator and propagation step -- the inner operator boundary of the hot
path (reference: models/components/spn.py, ``Generator`` :8-75 and ``PostProcessor`` :79-118).
Same class names, constructor arguments, parameter names and call signatures.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine as E
from . import ops
from .blocks import ConvUnit, ResUnit


class Generator(nn.Module):
    """(dem, context) -> (weight (B,9,H,W) in (0,1), offset (B,18,H,W)); spn.py:54-75."""

    def __init__(self, in_channels, kernel_size, block=None, bc=16, leaky=False):
        super().__init__()
        if kernel_size != 3 or leaky:
            raise NotImplementedError("only the 3x3, ReLU generator the reference instantiates is built")
        self.kernel_size = kernel_size
        self.num = kernel_size * kernel_size - 1
        self.idx_ref = self.num // 2
        self.convd1 = ConvUnit(1, bc * 2, 3, bn=False)
        self.convd2 = ConvUnit(bc * 2, bc * 2, 3, bn=False)
        self.convf1 = ConvUnit(in_channels, bc * 2, 3, bn=False)
        self.convf2 = ConvUnit(bc * 2, bc * 2, 3, bn=False)
        self.conv = ConvUnit(bc * 4, bc * 4, 3, bn=False)
        self.block = ResUnit(bc * 4, bc * 4)
        self.conv_weight = nn.Sequential(nn.Conv2d(bc * 4, kernel_size**2, 1))
        self.conv_offset = ConvUnit(bc * 4, 2 * self.num, 1, bn=False, relu=False)

    def features(self, dem, context):
        """dem, context: NHWC activations (engine.from_nchw)."""
        B, H, W = context.shape[:3]
        nd, nc = self.convd2.conv[0].out_channels, self.convf2.conv[0].out_channels
        buf = E.SliceBuffer(B, H, W, nd + nc, context.dtype, context.device)  # cat((d, f)) without the copy
        d = self.convd2(self.convd1(dem), dest=(buf, 0))
        f = self.convf2(self.convf1(context), dest=(buf, nd))
        return self.block(self.conv(buf.join((d, f))))

    def heads(self, feature):
        """NHWC weight (B,H,W,9) after the sigmoid and the 16 learned offset channels (B,H,W,16)."""
        cw, co = self.conv_weight[0], self.conv_offset.conv[0]
        # the two 1x1 heads (spn.py:66-68) read the same full-resolution feature: one conv with 9 + 16 (+ 7 zero)
        # output channels reads it once, and its backward is one data-gradient launch instead of two plus an add
        nw, no = cw.weight.shape[0], co.weight.shape[0]
        pad = (-(nw + no)) % 8
        w_all = torch.cat((cw.weight, co.weight, cw.weight.new_zeros((pad,) + tuple(cw.weight.shape[1:]))), 0)
        b_all = torch.cat((cw.bias, co.bias, cw.bias.new_zeros(pad)))
        y = E.conv2d(feature, w_all, b_all)
        return E.sigmoid(y[..., :nw]), y[..., nw:nw + no]

    def head(self, feature):
        """The two 1x1 heads as ONE 32-channel convolution in the tap-major order K1h reads (ops.merge_heads): affinity
        LOGITS (the Sigmoid of spn.py:43 is applied inside the propagation kernel, in fp32) and learned offsets."""
        cw, co = self.conv_weight[0], self.conv_offset.conv[0]
        w_all, b_all = ops.merge_heads(cw.weight, cw.bias, co.weight, co.bias)
        return E.conv2d(feature, w_all, b_all)

    def forward(self, dem, context):
        """Reference contract: NCHW in, (weight (B,9,H,W), offset (B,18,H,W)) NCHW fp32 out."""
        weight, off16 = self.heads(self.features(E.from_nchw(dem), E.from_nchw(context)))
        weight, off16 = E.to_nchw_f32(weight), E.to_nchw_f32(off16)
        B, _, H, W = off16.shape
        zero = torch.zeros(B, 2, H, W, dtype=off16.dtype, device=off16.device)
        i = 2 * self.idx_ref
        offset = torch.cat((off16[:, :i], zero, off16[:, i:]), 1)  # centre tap, spn.py:69-73
        return weight.contiguous(), offset.contiguous()


class PostProcessor(nn.Module):
    """out = b + sum_k w_k (a_k - mean a) S_k + scale * dem; spn.py:99-118 (residual form only)."""

    def __init__(self, kernel_size=3, residual=True, scale=1.0):
        super().__init__()
        if kernel_size != 3 or not residual:
            raise NotImplementedError("only kernel_size=3, residual=True (what JSPSR/LRRU/EDSR use) is built")
        self.residual = residual
        self.w = nn.Parameter(torch.ones((1, 1, kernel_size, kernel_size)))
        self.b = nn.Parameter(torch.zeros(1))
        self.scale = scale
        if self.scale != 1:
            print("Warning: The scale factor is not 1. This may lead to unexpected results.")

    def forward(self, init_dem, weight, offset):
        """offset: (B,18,H,W) torchvision layout, or (B,16,H,W) without the zero centre pair."""
        return E.propagate(init_dem, weight, offset, self.w, self.b, self.scale)

    def from_feature(self, init_dem, feature, generator):
        """Inside the models: the generator's last feature -> heads -> propagation (engine.heads_propagate)."""
        return E.heads_propagate(init_dem, feature, generator.conv_weight[0], generator.conv_offset.conv[0], self.w, self.b, self.scale)

    def from_head(self, init_dem, head):
        """Inside the models: the merged head's NHWC output (Generator.head) goes straight into the kernel."""
        return E.propagate_head(init_dem, head, self.w, self.b, self.scale)
