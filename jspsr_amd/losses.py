"""Training loss of the reference configs (configs/*.yml:67-70): L1 + L2 + 0.1 * Sobel-L1
(losses/loss_schemes.py:55-72, losses/loss_functions.py:171-185), evaluated on the GPU."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def sobel_gradient(x: torch.Tensor) -> torch.Tensor:
    """Normalised Sobel d/dx, d/dy with replicate padding: (B,C,H,W) -> (B,C,2,H,W).
    Separable form on shifted views (smooth [1,2,1]/4 across, central difference /2 along)."""
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate")
    sy = (xp[:, :, :-2, :] + 2 * xp[:, :, 1:-1, :] + xp[:, :, 2:, :]) * 0.25   # smoothed over rows
    sx = (xp[:, :, :, :-2] + 2 * xp[:, :, :, 1:-1] + xp[:, :, :, 2:]) * 0.25   # smoothed over cols
    gx = (sy[:, :, :, 2:] - sy[:, :, :, :-2]) * 0.5
    gy = (sx[:, :, 2:, :] - sx[:, :, :-2, :]) * 0.5
    return torch.stack((gx, gy), 2)


class MultiLoss(torch.nn.Module):
    """Returns the reference's dict {"L1","L2","Grad","Total"} (loss_schemes.py:61-72)."""

    def __init__(self, l1=1.0, l2=1.0, grad=0.1):
        super().__init__()
        self.weights = {"L1": l1, "L2": l2, "Grad": grad}

    def forward(self, pred, gt):
        d = pred - gt
        out = {"L1": d.abs().mean(), "L2": (d * d).mean()}
        if self.weights["Grad"]:
            out["Grad"] = (sobel_gradient(pred) - sobel_gradient(gt)).abs().mean()
        out["Total"] = sum(self.weights[k] * v for k, v in out.items())
        return out
