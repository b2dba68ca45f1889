"""torch.autograd bridges onto the C ABI (include/jspsr_hip.h).

Tensors are only carriers of device memory: each op hands raw device pointers, sizes and the
current HIP stream to libjspsr_hip.so.  CPU tensors are rejected (no fallback).
"""
from __future__ import annotations

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("jspsr_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32, got {t.dtype}")


class _Propagate(torch.autograd.Function):
    """PostProcessor.forward (models/components/spn.py:99-118) as one HIP kernel each way."""

    @staticmethod
    def forward(ctx, dem, weight, offset, w, b, scale):
        _need_gpu(dem, weight, offset, w, b)
        B, one, H, W = dem.shape
        oc = offset.shape[1]
        if one != 1 or weight.shape != (B, 9, H, W) or offset.shape != (B, oc, H, W) or oc not in (16, 18):
            raise ValueError(f"propagate: bad shapes dem {tuple(dem.shape)} weight {tuple(weight.shape)} offset {tuple(offset.shape)}")
        if w.numel() != 9 or b.numel() != 1:
            raise ValueError("propagate: w must have 9 elements and b 1")
        dem, weight, offset = dem.contiguous(), weight.contiguous(), offset.contiguous()
        w, b = w.contiguous(), b.contiguous()
        out = torch.empty_like(dem)
        lib = _lib.load()
        _lib.check(lib.jspsr_prop_forward_f32(dem.data_ptr(), weight.data_ptr(), offset.data_ptr(), oc,
                                              w.data_ptr(), b.data_ptr(), float(scale), out.data_ptr(),
                                              B, H, W, _stream()), "jspsr_prop_forward_f32")
        ctx.save_for_backward(dem, weight, offset, w)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dem, weight, offset, w = ctx.saved_tensors
        B, _, H, W = dem.shape
        oc = offset.shape[1]
        grad_out = grad_out.contiguous()
        gweight = torch.empty_like(weight)
        goffset = torch.empty_like(offset)
        gw = torch.empty_like(w)
        gb = torch.empty(1, device=dem.device, dtype=dem.dtype)
        lib = _lib.load()
        ws = torch.empty(lib.jspsr_prop_backward_workspace_bytes(B, H, W), dtype=torch.uint8, device=dem.device)
        _lib.check(lib.jspsr_prop_backward_f32(grad_out.data_ptr(), dem.data_ptr(), weight.data_ptr(),
                                               offset.data_ptr(), oc, w.data_ptr(), gweight.data_ptr(),
                                               goffset.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(),
                                               B, H, W, _stream()), "jspsr_prop_backward_f32")
        return None, gweight, goffset, gw, gb, None


def propagate(dem, weight, offset, w, b, scale: float = 1.0):
    """out = b + sum_k w_k (weight_k - mean weight) bilinear(dem, p_k + offset_k) + scale*dem.

    dem (B,1,H,W); weight (B,9,H,W); offset (B,18,H,W) in torchvision's deform_conv2d layout or
    (B,16,H,W) without the all-zero centre pair; w (1,1,3,3); b (1,).  No gradient flows to dem
    (the reference detaches it: models/JSPSR.py:372).
    """
    return _Propagate.apply(dem, weight, offset, w, b, scale)
