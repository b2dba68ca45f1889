"""``Model`` -- drop-in for the reference's ``models.LRRU.Model`` (models/LRRU.py:301-581): the LRRU
baseline re-targeted to DEMs, i.e. the N = 4 user of the propagation kernel (four
generator -> K1 steps on the detached running estimate).  Same ``args`` contract
(``input_channels, output_channels, kernel_size, bc, prob, dkn_residual``), ``forward(*in_tensor)``
and ``state_dict`` keys; every operator is a HIP kernel from jspsr_amd.engine.

Only what the reference's factory instantiates is built (utils/common_config.py:57-69):
kernel_size 3, prob 1.0 (stochastic depth degenerates to plain residual blocks), dkn_residual True.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import engine as E
from . import ops
from .blocks import ConvUnit, HotPathModule, ResUnit
from .JSPSR import Model as _JSPSR


class UpT(nn.Module):
    """ConvTranspose2d k3 s2 p1 op1 -> BN -> ReLU (LRRU.py:67-88); keys ``conv``, ``bn``."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.ConvTranspose2d(cin, cout, 3, 2, 1, 1, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return E.batch_norm(E.conv_transpose2d(x, self.conv.weight), self.bn, relu=True)


class FuseConv(nn.Module):
    """cat(feat, weight) -> conv3x3 -> BN -> ReLU (Guide, LRRU.py:188-200); key ``conv``."""

    def __init__(self, planes):
        super().__init__()
        self.conv = ConvUnit(planes * 2, planes, 3, bn=True)

    def forward(self, feat, weight):
        return self.conv(E.cat((feat, weight)))


class DepthEncoder(nn.Module):
    """BasicDepthEncoder (LRRU.py:203-247): same arithmetic as the JSPSR generator, other key names."""

    def __init__(self, bc):
        super().__init__()
        c = bc * 2
        self.convd1 = ConvUnit(1, c, 3, bn=False)
        self.convd2 = ConvUnit(c, c, 3, bn=False)
        self.convf1 = ConvUnit(c, c, 3, bn=False)
        self.convf2 = ConvUnit(c, c, 3, bn=False)
        self.conv = ConvUnit(2 * c, 2 * c, 3, bn=False)
        self.ref = ResUnit(2 * c, 2 * c, act=False)
        self.conv_weight = nn.Conv2d(2 * c, 9, 1)
        self.conv_offset = nn.Conv2d(2 * c, 16, 1)

    def forward(self, depth, context):
        """NHWC activations in, NHWC (B,H,W,9) affinities (after sigmoid) and (B,H,W,16) offsets out."""
        d = self.convd2(self.convd1(depth))
        f = self.convf2(self.convf1(context))
        x = self.ref(self.conv(E.cat((d, f))))
        weight = E.sigmoid(E.conv2d(x, self.conv_weight.weight, self.conv_weight.bias))
        return weight, E.conv2d(x, self.conv_offset.weight, self.conv_offset.bias)

    def features(self, depth, context):
        """The 2c-channel feature the two 1x1 heads read (LRRU.py:226-237)."""
        d = self.convd2(self.convd1(depth))
        f = self.convf2(self.convf1(context))
        return self.ref(self.conv(E.cat((d, f))))

    def head(self, depth, context):
        """Same features, the two 1x1 heads as one tap-major 32-channel convolution (affinity logits + offsets) for
        the head-fed propagation kernel (ops.merge_heads / ops.propagate_head)."""
        d = self.convd2(self.convd1(depth))
        f = self.convf2(self.convf1(context))
        x = self.ref(self.conv(E.cat((d, f))))
        w_all, b_all = ops.merge_heads(self.conv_weight.weight, self.conv_weight.bias, self.conv_offset.weight,
                                       self.conv_offset.bias)
        return E.conv2d(x, w_all, b_all)


class PostProcess(nn.Module):
    """Post_process_deconv (LRRU.py:250-298), residual form."""

    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.ones((1, 1, 3, 3)))
        self.b = nn.Parameter(torch.zeros(1))

    def forward(self, depth, weight, offset):
        return E.propagate(depth, weight, offset, self.w, self.b, 1.0)


class Model(HotPathModule):
    # sharded inference (tiling.py): no certified receptive radius for this network (five stride-2 stages and four
    # propagation steps applied to their own output) -> strips run only with check_reach=False; the steps still report
    # their learned offsets, and their reaches add up
    receptive_radius = None
    offsets_chain = True

    def __init__(self, args, layers=(2, 2, 2, 2, 2)):
        super().__init__()
        self.args = args
        self.in_channels = args.input_channels
        self.out_channels = args.output_channels
        assert len(self.in_channels) > 1, "At least 2 input data are required"
        if args.kernel_size != 3 or args.prob != 1.0 or not args.dkn_residual:
            raise NotImplementedError("only kernel_size=3, prob=1.0, dkn_residual=True (the reference's factory) is built")
        self.kernel_size = 3
        self.preserve_input = True
        self.compute_dtype = torch.float32
        bc = args.bc
        c = bc * 2
        self.conv_img = ConvUnit(3, c, 5, bn=True)
        self.conv_lidar = ConvUnit(1, c, 5, bn=False)
        planes = (2 * c, 4 * c, 8 * c, 8 * c, 8 * c)
        inpl = c
        for s in range(5):
            stride = 1 if s == 0 else 2
            for br in ("img", "lidar"):
                units = [ResUnit(inpl, planes[s], stride, project=(stride != 1 or inpl != planes[s]))]
                units += [ResUnit(planes[s], planes[s]) for _ in range(1, layers[s])]
                setattr(self, f"layer{s + 1}_{br}", nn.Sequential(*units))
            if s < 4:
                setattr(self, f"guide{s + 1}", FuseConv(planes[s]))
            inpl = planes[s]
        self.layer4d = UpT(8 * c, 8 * c)
        self.upproj0 = nn.Sequential(UpT(8 * c, 4 * c), UpT(4 * c, 2 * c), UpT(2 * c, c))
        self.weight_offset0 = DepthEncoder(bc)
        self.layer3d = UpT(8 * c, 8 * c)
        self.upproj1 = nn.Sequential(UpT(8 * c, 4 * c), UpT(4 * c, c))
        self.weight_offset1 = DepthEncoder(bc)
        self.layer2d = UpT(8 * c, 4 * c)
        self.upproj2 = nn.Sequential(UpT(4 * c, c))
        self.weight_offset2 = DepthEncoder(bc)
        self.layer1d = UpT(4 * c, 2 * c)
        self.conv = ConvUnit(2 * c, c, 3, bn=True)
        self.weight_offset3 = DepthEncoder(bc)
        self.Post_process = PostProcess()
        self._initialize_weights()

    def _initialize_weights(self):
        """Conv2d only (LRRU.py:560-581): truncated normal, sigma = sqrt(2.6 / (k*k*C_in)); bias 0."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.in_channels
                std = math.sqrt(1.3 * 2.0 / n)
                nn.init.trunc_normal_(m.weight, 0.0, std, -2 * std, 2 * std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, *in_tensor):
        depth, img, _, _, _ = _JSPSR.parse_input(True, False, False, False, *in_tensor)
        if depth.shape[2] % 16 or depth.shape[3] % 16:
            raise ValueError("LRRU needs H and W to be multiples of 16 (five stride-2 stages)")
        with E.compute_dtype(self.compute_dtype), E.count_batches():
            return self._forward(depth, img)

    def _step(self, current, context, enc):
        """One propagation step on the detached running estimate (LRRU.py:453-455 etc.)."""
        current = current.detach().float().contiguous()
        return E.heads_propagate(current, enc.features(E.from_nchw(current), context), enc.conv_weight, enc.conv_offset,
                                 self.Post_process.w, self.Post_process.b, 1.0)

    def _keep_input(self, out, d_clear):
        """preserve_input blend (LRRU.py:447-450): valid input pixels overwrite the estimate."""
        mask = ((d_clear > 0.0).sum(1, keepdim=True) > 0.0).type_as(d_clear)
        return (1.0 - mask) * out + mask * d_clear

    def _forward(self, depth, img):
        d_clear = depth
        c0_img = self.conv_img(E.from_nchw(img))
        c0_lidar = self.conv_lidar(E.from_nchw(depth))
        fi, fl = c0_img, c0_lidar
        dyn = []
        for s in range(1, 6):
            fi = getattr(self, f"layer{s}_img")(fi)
            fl = getattr(self, f"layer{s}_lidar")(fl)
            if s < 5:
                fl = getattr(self, f"guide{s}")(fl, fi)
                dyn.append(fl)
        c4 = self.layer4d(fi + fl) + dyn[3]
        out = self._step(self._keep_input(depth, d_clear), self.upproj0(c4), self.weight_offset0)
        c3 = self.layer3d(c4) + dyn[2]
        out = self._step(self._keep_input(out, d_clear), self.upproj1(c3), self.weight_offset1)
        c2 = self.layer2d(c3) + dyn[1]
        out = self._step(self._keep_input(out, d_clear), self.upproj2(c2), self.weight_offset2)
        c1 = self.layer1d(c2) + dyn[0]
        c0 = self.conv(c1) + c0_lidar
        return self._step(self._keep_input(out, d_clear), c0, self.weight_offset3)
