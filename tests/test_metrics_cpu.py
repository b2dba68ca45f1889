"""CPU: evaluation scores restated from evaluation/metrics.py against literal numpy transcriptions
of the cited formulas (the reference module itself cannot be imported: piq/kornia/richdem absent)."""
import numpy as np
import torch

from jspsr_amd import metrics as M


def test_scores_against_numpy_formulas():
    g = torch.Generator().manual_seed(0)
    vmin, vmax = -80.0, 929.0
    z = 200 * torch.rand(1, 1, 100, 120, generator=g)                 # metres above the tile minimum
    gt = M.scale_data(z, vmin, vmax, True)
    pred = M.scale_data((z + torch.randn(z.shape, generator=g)).clamp_min(-70), vmin, vmax, True)
    pred[0, 0, 50, 60] = 1.7                                           # exercised by the clamp
    m = M.Meter(vmin, vmax, border=0.05, elev_log=True)
    m.update(pred, gt)
    s = m.scores()
    p, q = pred.numpy().astype(np.float64)[0, 0, 5:95, 6:114], gt.numpy().astype(np.float64)[0, 0, 5:95, 6:114]
    p = np.clip(p, 0, 1)
    d = (np.exp(p * np.log(vmax - vmin)) + vmin) - (np.exp(q * np.log(vmax - vmin)) + vmin)
    assert abs(s["PSNR"] - (-10 * np.log10(((p - q) ** 2).mean() + 1e-8))) < 1e-3
    assert abs(s["RMSE"] - np.sqrt((d ** 2).mean())) < 1e-3
    srt = np.sort(d.ravel())
    assert abs(s["Median"] - srt[(srt.size - 1) // 2]) < 1e-3          # torch.median: lower middle
    ad = np.sort(np.abs(d.ravel() - srt[(srt.size - 1) // 2]))
    assert abs(s["NMAD"] - 1.4826 * ad[(ad.size - 1) // 2]) < 1e-3
    k = 1 + round(0.95 * (d.size - 1))
    assert abs(s["LE95"] - np.sort(np.abs(d.ravel()))[k - 1]) < 1e-3


def test_scale_descale_round_trip():
    z = torch.linspace(-70, 900, 50)
    for lg in (True, False):
        v = M.scale_data(z, -80.0, 929.0, lg)
        assert torch.allclose(M.descale_data(v - (1e-8 if lg else 0), -80.0, 929.0, lg), z, atol=1e-2)
    assert 0 <= v.min() and v.max() <= 1
