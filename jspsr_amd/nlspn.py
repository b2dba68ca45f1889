"""``NLSPN`` -- drop-in for the reference's ``models.components.nlspn.NLSPN`` (nlspn.py:8-235): the N-iteration,
fixed-affinity user of the propagation kernel.  Same constructor, parameter names
(``conv_offset_aff``, ``aff_scale_const``, ``w``, ``b``, ``w_conf``), forward signature and return tuple.

What runs where: the guidance convolution is the MFMA implicit-GEMM kernel; the confidence sampling (the reference's
eight 1x1 ``deform_conv2d`` calls) and the ``prop_time`` propagation steps -- forward, and backward including the
gradient with respect to the propagated raster -- are the HIP step kernels (jspsr_prop_step_*); the once-per-forward
affinity normalisation on the (B,8,H,W) tensor (tanh, abs-sum, clamp: nlspn.py:92-173) is a handful of element-wise
torch operators.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine as E
from . import ops


class NLSPN(nn.Module):
    def __init__(self, args, ch_g, ch_f, k_g, k_f):
        super().__init__()
        assert ch_f == 1, "only tested with ch_f == 1 but {}".format(ch_f)
        assert (k_g % 2) == 1, "only odd kernel is supported but k_g = {}".format(k_g)
        assert (k_f % 2) == 1, "only odd kernel is supported but k_f = {}".format(k_f)
        if k_f != 3:
            raise NotImplementedError("the propagation kernels are built for the 3x3 window (k_f = 3) the reference uses")
        self.args = args
        self.prop_time = args.prop_time
        self.affinity = args.affinity
        self.ch_g, self.ch_f, self.k_g, self.k_f = ch_g, ch_f, k_g, k_f
        self.num = k_f * k_f - 1
        self.idx_ref = self.num // 2
        if self.affinity not in ("AS", "ASS", "TC", "TGASS"):
            raise NotImplementedError
        self.conv_offset_aff = nn.Conv2d(ch_g, 3 * self.num, kernel_size=k_g, stride=1, padding=(k_g - 1) // 2, bias=True)
        self.conv_offset_aff.weight.data.zero_()
        self.conv_offset_aff.bias.data.zero_()
        if self.affinity == "TC":
            self.aff_scale_const = nn.Parameter(self.num * torch.ones(1), requires_grad=False)
        elif self.affinity == "TGASS":
            self.aff_scale_const = nn.Parameter(args.affinity_gamma * self.num * torch.ones(1))
        else:
            self.aff_scale_const = nn.Parameter(torch.ones(1), requires_grad=False)
        self.w = nn.Parameter(torch.ones((ch_f, 1, k_f, k_f)), requires_grad=False)     # gathering weights: constants
        self.b = nn.Parameter(torch.zeros(ch_f), requires_grad=False)
        self.w_conf = nn.Parameter(torch.ones((1, 1, 1, 1)), requires_grad=False)
        self.compute_dtype = torch.float32

    def _get_offset_affinity(self, guidance, confidence=None, rgb=None):
        """nlspn.py:77-175 -> offset (B,18,H,W) with the zero reference pair, aff (B,9,H,W) normalised."""
        B, _, H, W = guidance.shape
        c = self.conv_offset_aff
        with E.compute_dtype(self.compute_dtype):
            raw = E.to_nchw_f32(E.conv2d(E.from_nchw(guidance), c.weight, c.bias, 1, c.padding[0]))   # (B,24,H,W)
        off16, aff = raw[:, :2 * self.num], raw[:, 2 * self.num:]
        zero = torch.zeros(B, 2, H, W, dtype=raw.dtype, device=raw.device)
        i = 2 * self.idx_ref
        offset = torch.cat((off16[:, :i], zero, off16[:, i:]), 1)
        if self.affinity == "TC":
            aff = torch.tanh(aff / 100) / self.aff_scale_const
        elif self.affinity == "TGASS":
            aff = torch.tanh(aff / 100) / (self.aff_scale_const + 1e-8)
        if self.args.conf_prop:
            aff = aff * ops.sample_taps(confidence.float(), off16.detach(), legacy=bool(getattr(self.args, "legacy", False)))
        aff_abs_sum = aff.abs().sum(1, keepdim=True) + 1e-4
        if self.affinity in ("ASS", "TGASS"):
            aff_abs_sum = torch.where(aff_abs_sum < 1.0, torch.ones_like(aff_abs_sum), aff_abs_sum)
        if self.affinity in ("AS", "ASS", "TGASS"):
            aff = aff / aff_abs_sum
        aff_ref = 1.0 - aff.sum(1, keepdim=True)
        aff = torch.cat((aff[:, :self.idx_ref], aff_ref, aff[:, self.idx_ref:]), 1)
        return offset, aff

    def forward(self, feat_init, guidance, confidence=None, feat_fix=None, rgb=None):
        assert self.ch_g == guidance.shape[1]
        assert self.ch_f == feat_init.shape[1]
        if self.args.conf_prop:
            assert confidence is not None
        offset, aff = self._get_offset_affinity(guidance, confidence if self.args.conf_prop else None, rgb)
        mask_fix = fix = None
        if self.args.preserve_input:
            assert feat_init.shape == feat_fix.shape
            mask_fix = ((feat_fix > 0.0).sum(1, keepdim=True).detach() > 0.0).type_as(feat_fix)
            fix = feat_fix.float()
        list_feat = ops.propagate_steps(feat_init.float(), aff.contiguous(), offset.contiguous(), self.prop_time, mask_fix, fix)
        return list_feat[-1], list_feat, offset, aff, self.aff_scale_const.data
