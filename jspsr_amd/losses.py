"""Training loss of the reference configs (configs/*.yml:67-70): L1 + L2 + 0.1 * Sobel-L1
(losses/loss_schemes.py:55-72, losses/loss_functions.py:171-185) as one fused HIP forward and one
fused HIP backward over the (B,1,H,W) prediction (jspsr_loss_forward / jspsr_loss_backward)."""
from __future__ import annotations

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, w1, w2, wg):
        if not pred.is_cuda:
            raise RuntimeError("jspsr_amd losses run on the GPU only (no CPU fallback)")
        if pred.shape != gt.shape or pred.dim() != 4:
            raise ValueError(f"loss: expected equal (B,C,H,W) shapes, got {tuple(pred.shape)} {tuple(gt.shape)}")
        pred_c, gt_c = pred.float().contiguous(), gt.float().contiguous()
        B, C, H, W = pred_c.shape
        lib = _lib.load()
        ws = torch.empty(lib.jspsr_loss_workspace_bytes(B * C, H, W), dtype=torch.uint8, device=pred.device)
        losses = torch.empty(4, dtype=torch.float32, device=pred.device)
        _lib.check(lib.jspsr_loss_forward(pred_c.data_ptr(), gt_c.data_ptr(), w1, w2, wg, losses.data_ptr(),
                                          ws.data_ptr(), B * C, H, W, _stream()), "jspsr_loss_forward")
        ctx.save_for_backward(pred_c, gt_c, ws)
        ctx.w = (w1, w2, wg)
        ctx.mark_non_differentiable(gt)
        return losses

    @staticmethod
    def backward(ctx, glosses):
        pred, gt, ws = ctx.saved_tensors
        w1, w2, wg = ctx.w
        B, C, H, W = pred.shape
        g = glosses.float().contiguous()
        gp = torch.empty_like(pred)
        lib = _lib.load()
        # only "Total" carries gradient (MultiLoss detaches the three components)
        _lib.check(lib.jspsr_loss_backward(pred.data_ptr(), gt.data_ptr(), g[3:4].contiguous().data_ptr(), w1, w2, wg,
                                           gp.data_ptr(), ws.data_ptr(), B * C, H, W, _stream()), "jspsr_loss_backward")
        return gp, None, None, None, None


class MultiLoss(torch.nn.Module):
    """Returns the reference's dict {"L1","L2","Grad","Total"} (loss_schemes.py:61-72).  Gradients flow
    through "Total" (what the reference back-propagates, train/train_utils.py:217)."""

    def __init__(self, l1=1.0, l2=1.0, grad=0.1):
        super().__init__()
        self.weights = (float(l1), float(l2), float(grad))
        self.out = {}

    def forward(self, pred, gt):
        v = _FusedLoss.apply(pred, gt, *self.weights)
        self.out = {"L1": v[0].detach(), "L2": v[1].detach(), "Grad": v[2].detach(), "Total": v[3]}
        return self.out

    def reset(self):
        """The reference's train loop calls criterion.reset() every iteration (train/train_utils.py:206;
        loss_schemes.py:74-75)."""
        self.out = {}

    def __str__(self):
        return f"{self.__class__.__name__}:: ['L1', 'L2', 'Grad'], {list(self.weights)}, fused HIP (jspsr_loss_forward/backward)"
