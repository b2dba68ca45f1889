"""Evaluation scores of the reference, computed where the prediction lives (no host round trip per
sample): PSNR, RMSE, median error, NMAD, LE95 on de-scaled elevations.

Restates evaluation/metrics.py: ``MeterBase._prepare`` :147-199 (border crop by a fraction of the
size, clamp of the prediction to [0,1]); PSNR :229-235 (piq.psnr(data_range=1, reduction="mean"):
10*log10(1 / (mse + 1e-8)) per sample -- epsilon recalled from piq's public source, unpinned);
RMSE :361-384; median :444-455; NMAD :499-512 (1.4826 * median|dh - median dh|); LE95 :556-570
(k-th smallest |dh| with k = 1 + round(0.95 (n-1))), and data/data_utils.py:441-457
``ToDEM.descale_data`` / :289-312 ``ToTensor.scale_data``.  The reference feeds batches of one tile.
"""
from __future__ import annotations

from math import log

import torch

from . import _lib


def _elev_hip(t, descale, elev_min, elev_max, elev_log, base_elev=0.0):
    """fp32 device rasters: one HIP pass (csrc/tiles.hip, jspsr_elev_scale_f32)."""
    t = t.contiguous()
    out = torch.empty_like(t)
    _lib.check(_lib.load().jspsr_elev_scale_f32(t.data_ptr(), out.data_ptr(), t.numel(), int(descale), int(bool(elev_log)), float(elev_min),
                                                float(elev_max), float(base_elev), torch.cuda.current_stream().cuda_stream),
               "jspsr_elev_scale_f32")
    return out


def scale_data(z, elev_min, elev_max, elev_log=False, base_elev=0.0):
    """metres -> network range (data_utils.py:289-312)."""
    if z.is_cuda and z.dtype == torch.float32 and z.numel() and not z.requires_grad:
        return _elev_hip(z, False, elev_min, elev_max, elev_log, base_elev)
    z = z - base_elev if base_elev != 0 else z
    if elev_log:
        return torch.log(z - elev_min) / log(elev_max - elev_min) + 1e-8
    return (z - elev_min) / (elev_max - elev_min)


def descale_data(v, elev_min, elev_max, elev_log=False):
    """network range -> metres (data_utils.py:441-457)."""
    if v.is_cuda and v.dtype == torch.float32 and v.numel() and not v.requires_grad:
        return _elev_hip(v, True, elev_min, elev_max, elev_log)
    if elev_log:
        return torch.exp(v * log(elev_max - elev_min)) + elev_min
    return v * (elev_max - elev_min) + elev_min


def prepare(pred, gt, border=0.0):
    """metrics.py:147-199: crop `border` (fraction) on every side, clamp the prediction to [0,1]."""
    assert pred.shape == gt.shape, f"{pred.shape} {gt.shape}"
    if border != 0:
        h, w = pred.shape[-2:]
        bh, bw = int(h * border), int(w * border)
        pred, gt = pred[..., bh:h - bh, bw:w - bw], gt[..., bh:h - bh, bw:w - bw]
    return pred.clamp(0.0, 1.0), gt


def psnr(pred, gt):
    """piq.psnr(gt, pred, data_range=1, reduction='mean') on [0,1] tensors (B,1,H,W)."""
    mse = ((pred - gt) ** 2).mean((1, 2, 3))
    return (-10.0 * torch.log10(mse + 1e-8)).mean()


def rmse(dh):
    return torch.sqrt((dh * dh).sum() / dh.numel())


def median(dh):
    return torch.median(dh)


def nmad(dh):
    return 1.4826 * torch.median((dh - torch.median(dh)).abs())


def le95(dh):
    k = 1 + round(0.95 * (dh.numel() - 1))
    return torch.kthvalue(dh.abs().flatten(), k).values


def tile_scores(pred, gt, value_min, value_max, border=0.05, elev_log=True):
    """All five scores of ONE tile (1,1,H,W) on the GPU in one C-ABI call (jspsr_metrics_forward: fused crop / clamp /
    de-scale / reductions + an exact radix select for the three order statistics) -> device tensor
    (PSNR, RMSE, median, NMAD, LE95).  No host synchronisation."""
    if not pred.is_cuda or pred.shape != gt.shape or pred.numel() != pred.shape[-1] * pred.shape[-2]:
        raise ValueError("tile_scores: one (1,1,H,W) GPU tile per call")
    H, W = pred.shape[-2:]
    p, g = pred.float().contiguous(), gt.float().contiguous()
    lib = _lib.load()
    ws = torch.empty(lib.jspsr_metrics_workspace_bytes(H, W), dtype=torch.uint8, device=pred.device)
    out = torch.empty(5, dtype=torch.float32, device=pred.device)
    _lib.check(lib.jspsr_metrics_forward(p.data_ptr(), g.data_ptr(), H, W, float(border), float(value_min), float(value_max),
                                         int(bool(elev_log)), out.data_ptr(), ws.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "jspsr_metrics_forward")
    return out


class Meter:
    """Running per-sample averages of all five scores (what PerformanceMeter.get_score reports,
    evaluation/evaluate_utils.py:26-47).  Accumulates on the device; one host sync in ``scores()``.
    GPU tiles of batch size 1 (how the reference evaluates) go through the fused HIP path (`tile_scores`); CPU tensors
    and larger batches through the same formulas as torch operators."""

    NAMES = ("PSNR", "RMSE", "Median", "NMAD", "LE95")

    def __init__(self, value_min, value_max, border=0.05, elev_log=True):
        self.vmin, self.vmax, self.border, self.elev_log = value_min, value_max, border, elev_log
        self.sums, self.n = None, 0

    @torch.no_grad()
    def update(self, pred, gt):
        if pred.is_cuda and pred.dim() == 4 and pred.shape[0] == 1 and pred.shape[1] == 1:
            vals = tile_scores(pred, gt, self.vmin, self.vmax, self.border, self.elev_log)
            self.sums = vals if self.sums is None else self.sums + vals
            self.n += 1
            return
        p, g = prepare(pred.float(), gt.float(), self.border)
        dh = descale_data(p, self.vmin, self.vmax, self.elev_log) - descale_data(g, self.vmin, self.vmax, self.elev_log)
        vals = torch.stack((psnr(p, g), rmse(dh), median(dh), nmad(dh), le95(dh)))
        self.sums = vals if self.sums is None else self.sums + vals
        self.n += 1

    def scores(self):
        v = (self.sums / max(self.n, 1)).tolist()
        return dict(zip(self.NAMES, v))
