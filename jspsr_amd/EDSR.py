"""``EDSR`` -- drop-in for the reference's ``models.EDSR.EDSR`` in the configuration that shares the hot
path (models/EDSR.py:66-137 with ``scale=1, spn=True``): BN-free residual trunk feeding the same
affinity/offset generator and propagation step as JSPSR.  Same constructor arguments, ``forward(x)``
with ``x = cat(dem, guides)`` (B,C,H,W) and ``state_dict`` keys.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import engine as E
from . import ops
from .blocks import HotPathModule
from .spn import Generator, PostProcessor


class ResBlock(nn.Module):
    """conv3x3 - ReLU - conv3x3, times res_scale, plus the input (EDSR.py:13-44); keys ``body.{0,2}``."""

    def __init__(self, n_feat, res_scale):
        super().__init__()
        self.body = nn.Sequential()
        self.body.add_module("0", nn.Conv2d(n_feat, n_feat, 3, padding=1))
        self.body.add_module("2", nn.Conv2d(n_feat, n_feat, 3, padding=1))
        self.res_scale = res_scale

    def forward(self, x):
        a, b = self.body[0], self.body[1]
        r = E.conv2d(x, a.weight, a.bias, 1, 1, relu=True)   # bias + ReLU in the conv epilogue
        r = E.conv2d(r, b.weight, b.bias, 1, 1)
        return r * self.res_scale + x


class EDSR(HotPathModule):
    receptive_radius = None      # sharded inference (tiling.py): not certified for this network (check_reach=False only)

    def __init__(self, in_channels=3, out_channels=3, n_resblocks=16, n_features=64, scale=2, res_scale=0.1,
                 spn=False):
        super().__init__()
        if scale != 1 or not spn:
            raise NotImplementedError("only scale=1, spn=True (the configuration on the JSPSR hot path) is built")
        self.url = r"./models/pretrained/EDSR-b32f128x2.bin"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.res_scale, self.spn = res_scale, spn
        self.compute_dtype = torch.float32
        self.entry = nn.Conv2d(in_channels, n_features, 3, padding=1)
        blocks = [ResBlock(n_features, res_scale) for _ in range(n_resblocks)]
        blocks.append(nn.Conv2d(n_features, n_features, 3, padding=1))
        self.encoder = nn.Sequential(*blocks)
        self.generator = Generator(n_features, 3, bc=n_features // 2)
        self.post_layer = PostProcessor(3, True)
        for m in self.modules():  # EDSR.py:109-117
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        with E.compute_dtype(self.compute_dtype), E.count_batches():
            dem = x[:, 0:1].detach().contiguous()
            xs = E.conv2d(E.from_nchw(x), self.entry.weight, self.entry.bias, 1, 1)
            h = xs
            for blk in self.encoder[:-1]:
                h = blk(h)
            tail = self.encoder[-1]
            h = E.conv2d(h, tail.weight, tail.bias, 1, 1) + self.res_scale * xs
            return self.post_layer.from_feature(dem.float(), self.generator.features(E.from_nchw(dem), h), self.generator)
