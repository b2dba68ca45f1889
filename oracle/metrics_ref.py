"""TEST INFRASTRUCTURE ONLY -- numpy (fp64) restatement of the reference's evaluation scores and input scaling.
Only tests/ may import this module.

Restates:
  * MeterBase._prepare            evaluation/metrics.py:147-199  (border crop by int(h*border), clamp pred to [0,1])
  * MeterRMSE / Median / NMAD / LE95 ("local")   evaluation/metrics.py:361-384, 444-455, 499-512, 556-570
    (torch.median = the LOWER middle element; kthvalue with k = 1 + round(0.95 (n-1)), Python's round)
  * MeterPSNR via piq.psnr(data_range=1, reduction="mean")   evaluation/metrics.py:225-235
  * ToTensor.scale_data / ToDEM.descale_data   data/data_utils.py:289-312, 441-457
  * MultiLoss bookkeeping   losses/loss_schemes.py:55-72

Pin: tests/golden/g7_host_side.npz was produced by the reference's own Meter classes, scale/descale functions,
TileCrop and MultiLoss (oracle/gen_golden.py::gen_host_side) -- everything above except PSNR, whose piq dependency
is absent here: that one formula (-10 log10(mse + 1e-8), per sample, then mean) is recalled from piq's public
source and stays UNPINNED.
"""
from math import log

import numpy as np


def prepare(pred, gt, border=0.0):
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    assert pred.shape == gt.shape
    if border != 0:
        h, w = pred.shape[-2:]
        bh, bw = int(h * border), int(w * border)
        pred, gt = pred[..., bh:h - bh, bw:w - bw], gt[..., bh:h - bh, bw:w - bw]
    return np.clip(pred, 0.0, 1.0), gt


def scale_data(z, vmin, vmax, elev_log=False, base_elev=0.0):
    z = np.asarray(z, np.float64)
    if base_elev != 0:
        z = z - base_elev
    if elev_log:
        return np.log(z - vmin) / log(vmax - vmin) + 1e-8
    return (z - vmin) / (vmax - vmin)


def descale_data(v, vmin, vmax, elev_log=False):
    v = np.asarray(v, np.float64)
    if elev_log:
        return np.exp(v * log(vmax - vmin)) + vmin
    return v * (vmax - vmin) + vmin


def _lower_median(a):
    s = np.sort(np.ravel(a))
    return s[(s.size - 1) // 2]


def scores(pred, gt, vmin, vmax, border=0.05, elev_log=True):
    """One tile (1,1,H,W) -> dict of the five scores."""
    p, g = prepare(pred, gt, border)
    dh = (descale_data(p, vmin, vmax, elev_log) - descale_data(g, vmin, vmax, elev_log)).ravel()
    med = _lower_median(dh)
    k = 1 + round(0.95 * (dh.size - 1))
    return {
        "PSNR": float(-10.0 * np.log10(((p - g) ** 2).mean() + 1e-8)),   # piq: unpinned (see header)
        "RMSE": float(np.sqrt((dh ** 2).sum() / dh.size)),
        "Median": float(med),
        "NMAD": float(1.4826 * _lower_median(np.abs(dh - med))),
        "LE95": float(np.sort(np.abs(dh))[k - 1]),
    }


def mean_scores(preds, gts, vmin, vmax, border=0.05, elev_log=True):
    """Per-sample average over tiles -- what Meter*.get_score returns (sum of per-tile values / count)."""
    acc = None
    for p, g in zip(preds, gts):
        s = scores(p[None], g[None], vmin, vmax, border, elev_log)
        acc = s if acc is None else {k: acc[k] + s[k] for k in s}
    return {k: v / len(preds) for k, v in acc.items()}


def l1_l2(pred, gt):
    d = np.asarray(pred, np.float64) - np.asarray(gt, np.float64)
    return float(np.abs(d).mean()), float((d ** 2).mean()), np.sign(d) / d.size + 2 * d / d.size
