"""torch.autograd bridges onto the C ABI (include/jspsr_hip.h).

Tensors are only carriers of device memory: each op hands raw device pointers, sizes and the
current HIP stream to libjspsr_hip.so.  CPU tensors are rejected (no fallback).
"""
from __future__ import annotations

import collections
import os
import weakref

import torch

from . import _lib
from . import kernels as K


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("jspsr_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32, got {t.dtype}")


def prop_forward_raw(dem, weight, offset, w, b, scale, out):
    """Launch K1 forward on raw, contiguous fp32 operands (no autograd bookkeeping)."""
    B, _, H, W = dem.shape
    lib = _lib.load()
    _lib.check(lib.jspsr_prop_forward_f32(dem.data_ptr(), weight.data_ptr(), offset.data_ptr(), offset.shape[1],
                                          w.data_ptr(), b.data_ptr(), float(scale), out.data_ptr(), B, H, W, _stream()),
               "jspsr_prop_forward_f32")


def prop_backward_raw(grad_out, dem, weight, offset, w, gweight, goffset, gw, gb, ws):
    """gw = gb = None: the streaming kernel alone (partial rows stay in `ws`, see prop_backward_fold_raw)."""
    B, _, H, W = dem.shape
    lib = _lib.load()
    _lib.check(lib.jspsr_prop_backward_f32(grad_out.data_ptr(), dem.data_ptr(), weight.data_ptr(), offset.data_ptr(),
                                           offset.shape[1], w.data_ptr(), gweight.data_ptr(), goffset.data_ptr(),
                                           gw.data_ptr() if gw is not None else None,
                                           gb.data_ptr() if gb is not None else None, ws.data_ptr(), B, H, W, _stream()),
               "jspsr_prop_backward_f32")


def prop_backward_fold_raw(ws, B, H, W, gw, gb):
    _lib.check(_lib.load().jspsr_prop_backward_fold_f32(ws.data_ptr(), B, H, W, gw.data_ptr(), gb.data_ptr(), _stream()),
               "jspsr_prop_backward_fold_f32")


def prop_backward_workspace(B, H, W, device):
    n = _lib.load().jspsr_prop_backward_workspace_bytes(B, H, W)
    return torch.empty(max(n, 16), dtype=torch.uint8, device=device)


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
    if dem.requires_grad:
        raise RuntimeError("propagate: dem requires grad, but this operator produces no gradient with respect to the DEM "
                           "(every reference caller detaches it); use propagate_steps for chains that need it")
    return _Propagate.apply(dem, weight, offset, w, b, scale)


# ---- K1s: chains of propagation steps (NLSPN-style: fixed affinities / offsets, N iterations) -----------------------
def _step_forward(dem, weight, offset, w, b, scale, normalize, out):
    B, _, H, W = dem.shape
    lib = _lib.load()
    _lib.check(lib.jspsr_prop_step_forward_f32(dem.data_ptr(), weight.data_ptr(), offset.data_ptr(), offset.shape[1],
                                               w.data_ptr(), b.data_ptr(), float(scale), int(normalize), out.data_ptr(),
                                               B, H, W, _stream()), "jspsr_prop_step_forward_f32")
    return out


def _step_backward(gout, dem, weight, offset, w, scale, normalize, accumulate, gweight, goffset, gdem, ws):
    B, _, H, W = dem.shape
    lib = _lib.load()
    _lib.check(lib.jspsr_prop_step_backward_f32(gout.data_ptr(), dem.data_ptr(), weight.data_ptr(), offset.data_ptr(),
                                                offset.shape[1], w.data_ptr(), float(scale), int(normalize), int(accumulate),
                                                gweight.data_ptr(), goffset.data_ptr(),
                                                gdem.data_ptr() if gdem is not None else None, None, None, ws.data_ptr(),
                                                B, H, W, _stream()), "jspsr_prop_step_backward_f32")


def _step_workspace(B, H, W, device):
    n = _lib.load().jspsr_prop_step_backward_workspace_bytes(B, H, W)
    return torch.empty(max(n, 16), dtype=torch.uint8, device=device)


class _PropagateSteps(torch.autograd.Function):
    """feat_{i+1} = sum_k aff_k bilinear(feat_i, p_k + offset_k), i = 0..n-1, with the SAME affinities and offsets in
    every step (NLSPN.forward, models/components/nlspn.py:219-233; `_propagate_once` :177-187), optionally re-imposing
    known pixels before every step (preserve_input, :221-224).  Returns every step's result (the reference's
    `list_feat`).  Backward: the steps in reverse; each yields the gradient with respect to its input raster (bilinear
    scatter, jspsr_prop_step_backward_f32) and ADDS its share to the gradients of the shared affinities / offsets."""

    @staticmethod
    def forward(ctx, feat, aff, offset, n_steps, mask_fix, feat_fix):
        _need_gpu(feat, aff, offset)
        B, one, H, W = feat.shape
        oc = offset.shape[1]
        if one != 1 or tuple(aff.shape) != (B, 9, H, W) or tuple(offset.shape) != (B, oc, H, W) or oc not in (16, 18):
            raise ValueError(f"propagate_steps: bad shapes feat {tuple(feat.shape)} aff {tuple(aff.shape)} offset {tuple(offset.shape)}")
        if n_steps < 1:
            raise ValueError("propagate_steps: n_steps must be >= 1")
        aff, offset = aff.contiguous(), offset.contiguous()
        ones = torch.ones(9, device=feat.device)
        zero = torch.zeros(1, device=feat.device)
        cur, inputs, outs = feat.contiguous(), [], []
        for _ in range(n_steps):
            if mask_fix is not None:
                cur = ((1.0 - mask_fix) * cur + mask_fix * feat_fix).contiguous()
            inputs.append(cur)
            cur = _step_forward(cur, aff, offset, ones, zero, 0.0, 0, torch.empty_like(cur))
            outs.append(cur)
        ctx.save_for_backward(aff, offset, ones, mask_fix, *inputs)
        ctx.n = n_steps
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        aff, offset, ones, mask_fix = ctx.saved_tensors[:4]
        inputs = ctx.saved_tensors[4:]
        B, _, H, W = inputs[0].shape
        gaff, goff = torch.empty_like(aff), torch.empty_like(offset)
        ws = _step_workspace(B, H, W, aff.device)
        g, gfix = None, None
        first = _PropagateSteps._last(ctx, gouts)     # this step overwrites the shared gradient buffers, earlier ones add
        if first < 0:
            return None, None, None, None, None, None
        for i in range(ctx.n - 1, -1, -1):
            gi = gouts[i]
            if gi is not None:
                g = gi.contiguous() if g is None else g + gi
            if g is None:
                continue
            gin = torch.zeros_like(inputs[i])
            _step_backward(g.contiguous(), inputs[i], aff, offset, ones, 0.0, 0, i != first, gaff, goff, gin, ws)
            if mask_fix is not None:
                gfix = mask_fix * gin if gfix is None else gfix + mask_fix * gin
                gin = (1.0 - mask_fix) * gin
            g = gin
        return g, gaff, goff, None, None, gfix

    @staticmethod
    def _last(ctx, gouts):
        """Index of the first step processed in backward (the last one with an incoming gradient): it overwrites the
        shared gradient buffers, the earlier steps accumulate."""
        for i in range(ctx.n - 1, -1, -1):
            if gouts[i] is not None:
                return i
        return -1


def propagate_steps(feat, aff, offset, n_steps, mask_fix=None, feat_fix=None):
    """-> list of the n_steps rasters.  feat (B,1,H,W) fp32 (differentiable), aff (B,9,H,W) normalised affinities incl.
    the reference tap, offset (B,18,H,W) or (B,16,H,W)."""
    if (mask_fix is None) != (feat_fix is None):
        raise ValueError("propagate_steps: mask_fix and feat_fix go together")
    return list(_PropagateSteps.apply(feat, aff, offset, int(n_steps), mask_fix, feat_fix))


class _SampleTaps(torch.autograd.Function):
    """conf_k = bilinear(conf, (y, x) + offset_k) for the 8 learned taps -- the 1x1 deform_conv2d calls that modulate
    the affinities by the confidence (nlspn.py:104-154; `legacy`: the tap's window displacement is added, :118-125).
    Eight launches of the step kernel with a one-hot tap weight; offsets are constants (the reference detaches them)."""

    @staticmethod
    def forward(ctx, conf, off16, legacy):
        _need_gpu(conf, off16)
        B, _, H, W = conf.shape
        conf = conf.contiguous()
        off = off16.detach().clone().contiguous()
        if not legacy:      # the kernel adds the 3x3 window displacement itself: cancel it
            for t in range(8):
                k = t if t < 4 else t + 1
                off[:, 2 * t] -= (k // 3 - 1)
                off[:, 2 * t + 1] -= (k % 3 - 1)
        ones9 = torch.ones(B, 9, H, W, device=conf.device)
        zero = torch.zeros(1, device=conf.device)
        out = torch.empty(8, B, 1, H, W, device=conf.device)     # one contiguous (B,1,H,W) raster per tap
        hots = []
        for t in range(8):
            hot = torch.zeros(9, device=conf.device)
            hot[t if t < 4 else t + 1] = 1.0
            hots.append(hot)
            _step_forward(conf, ones9, off, hot, zero, 0.0, 0, out[t])
        ctx.save_for_backward(conf, off, ones9, *hots)
        return out.permute(1, 0, 2, 3, 4).reshape(B, 8, H, W).contiguous()

    @staticmethod
    def backward(ctx, g):
        conf, off, ones9 = ctx.saved_tensors[:3]
        hots = ctx.saved_tensors[3:]
        B, _, H, W = conf.shape
        gconf = torch.zeros_like(conf)
        scratch_w, scratch_o = torch.empty_like(ones9), torch.empty_like(off)
        ws = _step_workspace(B, H, W, conf.device)
        for t in range(8):
            _step_backward(g[:, t:t + 1].contiguous(), conf, ones9, off, hots[t], 0.0, 0, 0, scratch_w, scratch_o, gconf, ws)
        return gconf, None, None


def sample_taps(conf, off16, legacy=False):
    return _SampleTaps.apply(conf, off16, bool(legacy))


# ---- K1h: propagation fed straight from the merged 1x1 head's NHWC output ---------------------------------------------
HEAD_CHANNELS = 32


def head_rows():
    """Row order of the merged head convolution ("tap-major", include/jspsr_hip.h K1h): for the 8 learned taps t
    (k = t < 4 ? t : t + 1): [affinity row k, offset row 2t (dy), offset row 2t+1 (dx), extra], extra = the centre
    tap's affinity row (k = 4) for t == 0, a zero row otherwise.  Returns (kind, index) pairs: ("w", k), ("o", j),
    ("z", 0)."""
    rows = []
    for t in range(8):
        k = t if t < 4 else t + 1
        rows += [("w", k), ("o", 2 * t), ("o", 2 * t + 1), ("w", 4) if t == 0 else ("z", 0)]
    return rows


def merge_heads(w_weight, b_weight, w_offset, b_offset):
    """(9,C,1,1)/(9,) affinity head + (16,C,1,1)/(16,) offset head -> (32,C,1,1)/(32,) tap-major merged head
    (differentiable: the gradients of the merged rows flow back to the two reference-named parameters)."""
    # one gather from [affinity rows | offset rows | a zero row] (a row-by-row concatenation costs ~100 tiny kernels
    # per step, forward and backward, for the same 32 rows)
    nw, no = w_weight.shape[0], w_offset.shape[0]
    base = {"w": 0, "o": nw, "z": nw + no}
    key = (nw, no, w_weight.device)
    idx = _head_index.get(key)
    if idx is None:
        idx = _head_index[key] = torch.tensor([base[kind] + i for kind, i in head_rows()], device=w_weight.device)
    stack_w = torch.cat((w_weight, w_offset, w_weight.new_zeros((1,) + tuple(w_weight.shape[1:]))), 0)
    stack_b = torch.cat((b_weight, b_offset, b_weight.new_zeros(1)), 0)
    return stack_w.index_select(0, idx), stack_b.index_select(0, idx)


_head_index = {}


def split_head(head):
    """(B,H,W,32) tap-major head -> (affinity logits (B,H,W,9) in window order, offsets (B,H,W,16)) -- views/gathers
    for callers that need the reference's two tensors (tests, the halo check of sharded inference)."""
    h = head.reshape(*head.shape[:3], 8, 4)
    logits = torch.cat((h[..., :4, 0], h[..., 0:1, 3], h[..., 4:, 0]), -1)
    return logits, h[..., 1:3].reshape(*head.shape[:3], 16)


class _PropagateHead(torch.autograd.Function):
    """Sigmoid (spn.py:43) + zero centre offset (spn.py:69-73) + PostProcessor.forward (spn.py:99-118) on the merged
    head's output, one HIP kernel each way (jspsr_prop_head_forward / _backward)."""

    @staticmethod
    def forward(ctx, dem, head, w, b, scale):
        _need_gpu(dem, w, b)
        if not head.is_cuda or head.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("propagate_head: head must be a float32 or bfloat16 GPU tensor")
        B, one, H, W = dem.shape
        if one != 1 or tuple(head.shape) != (B, H, W, HEAD_CHANNELS):
            raise ValueError(f"propagate_head: dem {tuple(dem.shape)} against head {tuple(head.shape)} (want (B,H,W,{HEAD_CHANNELS}))")
        if w.numel() != 9 or b.numel() != 1:
            raise ValueError("propagate_head: w must have 9 elements and b 1")
        dem, head, w, b = dem.contiguous(), head.contiguous(), w.contiguous(), b.contiguous()
        out = torch.empty_like(dem)
        lib = _lib.load()
        _lib.check(lib.jspsr_prop_head_forward(K._dt(head), dem.data_ptr(), head.data_ptr(), w.data_ptr(), b.data_ptr(),
                                               float(scale), out.data_ptr(), B, H, W, _stream()), "jspsr_prop_head_forward")
        ctx.save_for_backward(dem, head, w)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dem, head, w = ctx.saved_tensors
        B, _, H, W = dem.shape
        grad_out = grad_out.contiguous()
        ghead = torch.empty_like(head)
        gw = torch.empty_like(w)
        gb = torch.empty(1, device=dem.device, dtype=dem.dtype)
        lib = _lib.load()
        ws = torch.empty(max(lib.jspsr_prop_head_backward_workspace_bytes(B, H, W), 16), dtype=torch.uint8, device=dem.device)
        _lib.check(lib.jspsr_prop_head_backward(K._dt(head), grad_out.data_ptr(), dem.data_ptr(), head.data_ptr(),
                                                w.data_ptr(), ghead.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(),
                                                B, H, W, _stream()), "jspsr_prop_head_backward")
        return None, ghead, gw, gb, None


def propagate_head(dem, head, w, b, scale: float = 1.0):
    """out = b + sum_k w_k (a_k - mean a) bilinear(dem, p_k + offset_k) + scale*dem with a = sigmoid(affinity logits),
    operands in the merged head's tap-major NHWC layout (`merge_heads`).  dem (B,1,H,W) fp32, head (B,H,W,32) fp32 or
    bf16.  No gradient flows to dem (detached by every caller: models/JSPSR.py:372, LRRU.py:453-496)."""
    if dem.requires_grad:
        raise RuntimeError("propagate_head: dem must be detached (no gradient with respect to the DEM is produced)")
    return _PropagateHead.apply(dem, head, w, b, scale)


# ---- K1c + K1 in the models (round 4): the heads write PLANES, the propagation step reads them -------------------------
HEAD_PLANES = 25          # 9 affinity logits + 16 learned offsets (Generator's order: spn.py:41-52,66-68)


def head_planes_ok(x) -> bool:
    """Can the planar head route take this feature tensor?  (H*W a multiple of 32, 32 / 64 / 128 channels, fp32 | bf16.)"""
    if not x.is_cuda or x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    B, H, W, C = x.shape
    return bool(_lib.load().jspsr_head_ok(K._dt(x), B, H, W, C))


class _HeadPlanes(torch.autograd.Function):
    """The two 1x1 heads of Generator.forward (spn.py:66-68, without the Sigmoid) as ONE convolution writing the
    (B,25,H,W) fp32 operand of the propagation step: jspsr_head_forward; backward = jspsr_head_backward (data gradient,
    bias gradient, and the NHWC copy of the gradient the weight-gradient kernel reads) + jspsr_conv2d_wgrad."""

    @staticmethod
    def forward(ctx, x, w25, b25):
        x = K.nhwc(x)
        B, H, W, C = x.shape
        w = w25.detach().reshape(HEAD_PLANES, C).contiguous().float()
        b = b25.detach().contiguous().float()
        planes = torch.empty((B, HEAD_PLANES, H, W), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        _lib.check(lib.jspsr_head_forward(K._dt(x), x.data_ptr(), K.pitch(x), 0, C, w.data_ptr(), b.data_ptr(), planes.data_ptr(),
                                          B, H, W, _stream()), "jspsr_head_forward")
        ctx.save_for_backward(x, w)
        return planes

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B, H, W, C = x.shape
        g = g.contiguous()
        lib = _lib.load()
        dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device) if ctx.needs_input_grad[0] else None
        gn = torch.empty((B, H, W, 32), dtype=x.dtype, device=x.device)
        db = torch.empty(HEAD_PLANES, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.jspsr_head_backward_workspace_bytes(B, H, W), dtype=torch.uint8, device=x.device)
        _lib.check(lib.jspsr_head_backward(K._dt(x), g.data_ptr(), w.data_ptr(), C, dx.data_ptr() if dx is not None else None,
                                           C, 0, gn.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, W, _stream()),
                   "jspsr_head_backward")
        dW = None
        if ctx.needs_input_grad[1]:
            dW = _wgrad_async(None, gn, x, 32, C, 1, 1, 1, 0)[:HEAD_PLANES]
        return dx, dW, db if ctx.needs_input_grad[2] else None


def head_planes(x, w_weight, b_weight, w_offset, b_offset):
    """(B,H,W,C) NHWC feature -> (B,25,H,W) fp32 planes [9 affinity logits | 16 offsets]; differentiable with respect to
    the feature and the two reference-named heads' parameters."""
    w25 = torch.cat((w_weight, w_offset), 0)
    b25 = torch.cat((b_weight, b_offset), 0)
    return _HeadPlanes.apply(x, w25, b25)


class _PropagateLogits(torch.autograd.Function):
    """Sigmoid (spn.py:43) + zero centre offset (spn.py:69-73) + PostProcessor.forward (spn.py:99-118) on the planar
    head: jspsr_prop_logits_forward_f32 / _backward_f32 -- the kernels of the public PostProcessor boundary."""

    @staticmethod
    def forward(ctx, dem, head, w, b, scale):
        _need_gpu(dem, head, w, b)
        B, one, H, W = dem.shape
        if one != 1 or tuple(head.shape) != (B, HEAD_PLANES, H, W) or head.dtype != torch.float32:
            raise ValueError(f"propagate_logits: dem {tuple(dem.shape)} against head {tuple(head.shape)} {head.dtype} (want fp32 (B,{HEAD_PLANES},H,W))")
        if w.numel() != 9 or b.numel() != 1:
            raise ValueError("propagate_logits: w must have 9 elements and b 1")
        dem, head, w, b = dem.contiguous(), head.contiguous(), w.contiguous(), b.contiguous()
        out = torch.empty_like(dem)
        lib = _lib.load()
        _lib.check(lib.jspsr_prop_logits_forward_f32(dem.data_ptr(), head.data_ptr(), w.data_ptr(), b.data_ptr(), float(scale),
                                                     out.data_ptr(), B, H, W, _stream()), "jspsr_prop_logits_forward_f32")
        ctx.save_for_backward(dem, head, w)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dem, head, w = ctx.saved_tensors
        B, _, H, W = dem.shape
        grad_out = grad_out.contiguous()
        ghead = torch.empty_like(head)
        gw = torch.empty_like(w)
        gb = torch.empty(1, device=dem.device, dtype=dem.dtype)
        lib = _lib.load()
        ws = torch.empty(max(lib.jspsr_prop_backward_workspace_bytes(B, H, W), 16), dtype=torch.uint8, device=dem.device)
        _lib.check(lib.jspsr_prop_logits_backward_f32(grad_out.data_ptr(), dem.data_ptr(), head.data_ptr(), w.data_ptr(),
                                                      ghead.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), B, H, W,
                                                      _stream()), "jspsr_prop_logits_backward_f32")
        return None, ghead, gw, gb, None


def propagate_logits(dem, head, w, b, scale: float = 1.0):
    """out = b + sum_k w_k (a_k - mean a) bilinear(dem, p_k + offset_k) + scale*dem, a = sigmoid(head[:, :9]), offsets
    head[:, 9:] (the eight learned taps; the centre tap's offset is zero).  dem (B,1,H,W) fp32, head (B,25,H,W) fp32."""
    if dem.requires_grad:
        raise RuntimeError("propagate_logits: dem must be detached (no gradient with respect to the DEM is produced)")
    return _PropagateLogits.apply(dem, head, w, b, scale)


# =============================================================================================
# NHWC layer operators: torch.autograd.Function shells around jspsr_amd.kernels (HIP).
# Tensors are (B, H, W, C) contiguous, fp32 or bf16; parameters stay fp32 masters.
# =============================================================================================


# ---- packed weights: re-laid once per weight update, not once per use -------------------------------------------------
# The kernels read weights k-contiguous per output channel in the compute dtype (jspsr_pack_weight).  Weights change
# only in the optimizer step, so the packed copy of a Parameter is kept on the Parameter object and re-made when the
# values may have changed: torch's in-place version counter (optimizers, load_state_dict, copy_) or this module's
# `weights epoch`, which the writers that bypass the counter bump (FlatAdamW's HIP kernel, broadcast_module's writes
# through .data).  Tensors that are not Parameters (merged heads, test operands) are packed on every use.
_weights_epoch = 0


def invalidate_packed_weights():
    """Call after modifying parameters in a way torch's version counter does not see (raw-pointer kernels, `.data`)."""
    global _weights_epoch
    _weights_epoch += 1


# Re-making all packed copies in ONE launch right after the optimizer step removes 152 small launches per step, but
# measured no faster (65.4 vs 65.2 ms at 8 tiles, 21.1 vs 21.0 ms at 2): the per-use pack kernels run on the branch /
# weight-gradient streams beside real work, the single launch sits on the main stream's critical path.  Off by default.
repack_after_step = os.environ.get("JSPSR_REPACK_ALL", "0") != "0"
_pack_registry = {}      # id(param) -> weakref(param): every Parameter that holds cached packs
_pack_table = None       # (signature, device descriptor tensor, grand_total) of the last repack_all()


def _packed(param, w, mode, c_pad, dtype):
    if not isinstance(param, torch.nn.Parameter):
        return K.pack_weight(w, mode, c_pad, dtype)
    cache = param.__dict__.setdefault("_jspsr_packs", {})
    key = (mode, c_pad, dtype, w.data_ptr())
    hit = cache.get(key)
    stamp = (param._version, _weights_epoch)
    if hit is not None and hit[1] == stamp:
        return hit[0]
    if len(cache) > 8:
        cache.clear()
    packed = K.pack_weight(w, mode, c_pad, dtype)
    cache[key] = (packed, stamp)
    if id(param) not in _pack_registry:
        _pack_registry[id(param)] = weakref.ref(param, lambda _, k=id(param): _pack_registry.pop(k, None))
    return packed


def repack_all():
    """Re-lay EVERY cached packed weight from its (just updated) master in one launch and mark the copies current.
    FlatAdamW.step() calls this right after its kernel: the 2 x 76 per-use pack launches of the next step disappear
    (jspsr_pack_weights_multi).  Runs on the current stream, which by then is ordered after every stream that read the
    old copies (GradReducer.finish() joins them before the optimizer)."""
    global _pack_table
    import numpy as np
    entries = []
    for ref in list(_pack_registry.values()):
        p = ref()
        if p is None or not p.is_cuda:
            continue
        for (mode, c_pad, dtype, ptr), (packed, _) in p.__dict__.get("_jspsr_packs", {}).items():
            if ptr == p.data_ptr() and p.dim() == 4 and p.is_contiguous():
                entries.append((p, mode, c_pad, dtype, packed))
    if not entries:
        return 0
    sig = tuple((p.data_ptr(), packed.data_ptr(), mode, c_pad) for p, mode, c_pad, _, packed in entries)
    if _pack_table is None or _pack_table[0] != sig:
        desc = np.zeros(len(entries), dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("start", "<i8"), ("total", "<i8"),
                                                      ("O", "<i4"), ("I", "<i4"), ("KH", "<i4"), ("KW", "<i4"), ("mode", "<i4"),
                                                      ("c_pad", "<i4"), ("dtype", "<i4"), ("reserved", "<i4")]))
        start, chunk = 0, _lib.load().jspsr_pack_chunk()
        for i, (p, mode, c_pad, dtype, packed) in enumerate(entries):
            O, I, KH, KW = p.shape
            total = packed.numel()
            desc[i] = (p.data_ptr(), packed.data_ptr(), start, total, O, I, KH, KW, mode, c_pad, K._dt(packed), 0)
            start += (total + chunk - 1) // chunk            # workgroups serving this descriptor
        table = torch.from_numpy(desc.view(np.uint8).copy()).to(entries[0][0].device)
        _pack_table = (sig, table, start)
    _, table, grand = _pack_table
    _lib.check(_lib.load().jspsr_pack_weights_multi(table.data_ptr(), len(entries), grand, _stream()), "jspsr_pack_weights_multi")
    for p, mode, c_pad, dtype, packed in entries:
        p.__dict__["_jspsr_packs"][(mode, c_pad, dtype, p.data_ptr())] = (packed, (p._version, _weights_epoch))
    return len(entries)


def pad_channels(x: torch.Tensor, mult: int) -> torch.Tensor:
    """Zero-pad the channel (last) dim to a multiple of `mult` (16-byte chunks for the gathers)."""
    c = x.shape[-1]
    p = (-c) % mult
    return x if p == 0 else torch.nn.functional.pad(x, (0, p))


def _wgrad_into(param, G, X, R, C, KH, KW, stride, pad, **kw):
    """Weight gradient of one conv.  If the parameter's .grad lives in a GradReducer's flat buffer (opted in through
    `_jspsr_direct_grad`), the ordered slab reduction adds straight into it and the autograd node reports no
    gradient for that input -- the per-parameter `grad += dW` pass disappears; the reducer's bucket bookkeeping is
    notified by hand.  Otherwise returns dW for autograd to accumulate."""
    gbuf = param.grad if param is not None and getattr(param, "_jspsr_direct_grad", False) else None
    if (gbuf is not None and gbuf.dtype == torch.float32 and gbuf.is_contiguous() and gbuf.is_cuda
            and tuple(gbuf.shape) == (R, C, KH, KW)):
        K.conv2d_wgrad(G, X, R, C, KH, KW, stride, pad, out=gbuf, accumulate=True, **kw)
        ready = getattr(param, "_jspsr_grad_ready", None)
        if ready is not None:
            ready(param)
        return None
    return K.conv2d_wgrad(G, X, R, C, KH, KW, stride, pad, **kw)


# ---- weight gradients beside the data-gradient chain -------------------------------------------------------------
# A conv's weight gradient depends only on (dz, x) and nothing downstream in the backward pass waits for it, while the
# chain dz -> data gradient -> BatchNorm backward -> ... is what the next layer needs.  The wgrad kernels (MFMA-bound)
# therefore go to an auxiliary stream per home stream and overlap the memory-bound BatchNorm passes of the chain.
_aux_streams = {}
_marks = {}          # home stream -> events marking recent weight-gradient forks (run-ahead throttle)
RUN_AHEAD = int(os.environ.get("JSPSR_RUN_AHEAD", "16"))
wgrad_async = os.environ.get("JSPSR_WGRAD_ASYNC", "1") != "0"
wgrad_after_dgrad = os.environ.get("JSPSR_WGRAD_AFTER_DGRAD", "0") != "0"   # lab: weight gradients forked behind their layer's data gradient
wgrad_one_stream = os.environ.get("JSPSR_WGRAD_ONE_STREAM", "0") != "0"   # lab: ONE weight-gradient stream for all home streams
bn_reduce_fused = os.environ.get("JSPSR_BN_REDUCE_FUSE", "1") != "0"      # bn1's backward reduce in conv2's data-gradient epilogue


def aux_streams():
    """Every auxiliary stream created so far (GradReducer orders its collectives / finish() after them)."""
    return list(_aux_streams.values())


def _wgrad_async(param, G, X, R, C, KH, KW, stride, pad, **kw):
    if not wgrad_async:
        return _wgrad_into(param, G, X, R, C, KH, KW, stride, pad, **kw)
    cur = torch.cuda.current_stream()
    key = 0 if wgrad_one_stream else cur.cuda_stream
    aux = _aux_streams.get(key)
    if aux is None:
        aux = _aux_streams[key] = torch.cuda.Stream(device=G.device)
    aux.wait_stream(cur)                     # G and X are complete on the home stream
    with torch.cuda.stream(aux):
        dW = _wgrad_into(param, G, X, R, C, KH, KW, stride, pad, **kw)
    for t_ in kw.values():
        if isinstance(t_, torch.Tensor):
            t_.record_stream(aux)
    G.record_stream(aux)                     # home-pool tensors read by queued aux work
    X.record_stream(aux)
    # Run-ahead throttle.  A block released while another stream still has work queued is handed out again only after
    # that stream has drained everything queued up to the release; the host enqueues a whole backward pass in a
    # fraction of the time the GPU needs for it, so without a bound almost nothing freed during a backward pass is
    # reusable within it (measured: 160 GiB reserved for a 24 GiB working set).  The host therefore never gets more
    # than RUN_AHEAD weight-gradient launches (about five residual blocks) ahead of the GPU -- still far more queued
    # work than the launch latency needs.
    if not torch.cuda.is_current_stream_capturing():     # (a captured step has a fixed memory plan and no host to throttle)
        marks = _marks.setdefault(cur.cuda_stream, collections.deque())
        marks.append(cur.record_event())
        if len(marks) > RUN_AHEAD:
            marks.popleft().synchronize()
    if dW is not None:                       # handed to autograd, which consumes it on the home stream
        cur.wait_stream(aux)
        dW.record_stream(cur)
    return dW


def _bn_sink(gparam, bparam):
    """(dgamma_buf, dbeta_buf) if both BatchNorm parameters' .grad live in a GradReducer's flat buffer, else None;
    see _wgrad_into."""
    bufs = []
    for p_ in (gparam, bparam):
        g_ = p_.grad if p_ is not None and getattr(p_, "_jspsr_direct_grad", False) else None
        if g_ is None or g_.dtype != torch.float32 or not g_.is_contiguous() or not g_.is_cuda or g_.dim() != 1:
            return None
        bufs.append(g_)
    return tuple(bufs)


def _bn_ready(gparam, bparam):
    for p_ in (gparam, bparam):
        ready = getattr(p_, "_jspsr_grad_ready", None)
        if ready is not None:
            ready(p_)


class SliceBuffer:
    """A wide NHWC buffer that several operators fill channel slice by channel slice -- the reference's
    ``torch.cat((a, b, c), 1)`` (Guide, basics.py:134; skip concats JSPSR.py:355-368; spn.py:63) without the copy.
    Producers take ``dest=(buffer, first_channel)`` and return the slice they wrote; ``join`` returns the joined
    tensor as an autograd node over the producers."""

    def __init__(self, B, H, W, C, dtype, device):
        self.buf = torch.empty((B, H, W, C), dtype=dtype, device=device)
        self.deferred = None     # see join(defer=...)
        self.deferred_event = None      # recorded where the parked gradient was produced
        self.deposit_events = []        # recorded by every res_unit that added its input gradient into the parked one
        self.expected_deposits = 0      # how many such units the forward pass set up
        self.pending_deposits = 0       # ... of which this backward pass still expects (re-armed by every _Join.backward)

    def take_deferred(self):
        """The parked gradient, for the unit that adds it in its data-gradient epilogue.  Units that DEPOSIT into it
        (res_unit(grad_extra=(self, first_channel, channels)), possibly on other streams) must all have run."""
        g, self.deferred = self.deferred, None
        if g is not None:
            if len(self.deposit_events) != self.pending_deposits:
                raise RuntimeError(f"SliceBuffer: {len(self.deposit_events)} of {self.pending_deposits} deposits into the parked "
                                   "gradient had run when it was taken (autograd order changed?)")
            cur = torch.cuda.current_stream()
            for ev in self.deposit_events:
                cur.wait_event(ev)
            self.deposit_events = []
        return g

    def slice(self, lo, n, shape=None):
        """The (B,H,W,n) channel slice starting at `lo`; `shape` = the (B,H,W) the producer is about to write."""
        if lo < 0 or lo + n > self.buf.shape[3]:
            raise ValueError(f"SliceBuffer: channels [{lo}, {lo + n}) outside a {self.buf.shape[3]}-channel buffer")
        if shape is not None and tuple(shape) != tuple(self.buf.shape[:3]):
            raise ValueError(f"SliceBuffer: producer shape {tuple(shape)} != buffer {tuple(self.buf.shape[:3])}")
        v = self.buf.narrow(3, lo, n).detach()
        if not (v.is_contiguous() or K.is_slice(v)):
            raise ValueError("SliceBuffer: slice start / buffer width must be a multiple of 16 bytes")
        return v

    def join(self, parts, lo=0, defer=0):
        """parts: the tensors returned by the producers, in channel order starting at `lo`.
        defer = n: the gradient of the LAST n parts is not returned to them by this join; it is parked in
        `self.deferred` (a channel-slice view of the incoming gradient) for the other consumer of those parts -- a
        `res_unit(..., grad_extra=self)` that reads the same slices through another join and runs later in the
        backward pass -- to add in its data-gradient epilogue.  The two gradients of a tensor with two consumers then
        meet inside that kernel instead of in an extra pass over the tensor."""
        return _Join.apply(self, lo, defer, *parts)


class _Join(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder, lo, defer, *parts):
        ctx.holder, ctx.defer = holder, int(defer)
        ctx.widths = [p.shape[3] for p in parts]
        n = sum(ctx.widths)
        for p, w in zip(parts, ctx.widths):   # the producers must really have written into this buffer
            if p.untyped_storage().data_ptr() != holder.buf.untyped_storage().data_ptr():
                raise RuntimeError("SliceBuffer.join: a part does not live in this buffer")
        return holder.buf.narrow(3, lo, n).detach()

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        keep = len(ctx.widths) - ctx.defer
        for i, w in enumerate(ctx.widths):
            outs.append(g.narrow(3, off, w) if i < keep else None)
            if i == keep:
                ctx.holder.deferred = g.narrow(3, off, sum(ctx.widths[keep:]))
                ctx.holder.deferred_event = torch.cuda.current_stream().record_event()
                ctx.holder.deposit_events = []
                ctx.holder.pending_deposits = ctx.holder.expected_deposits     # per backward pass: a second backward over a retained graph starts afresh
            off += w
        return (None, None, None) + tuple(outs)


class _Conv(torch.autograd.Function):
    """nn.Conv2d (basics.py:11-20,39-47) or nn.ConvTranspose2d k3 s2 p1 op1 (basics.py:69-77),
    with the bias + ReLU of a BN-free Basic2d fused into the epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu, transposed, want_stats=False, dest=None):
        x = K.nhwc(x)
        cdt = x.dtype
        e = K.epc(cdt)
        B, H, W, Cp = x.shape
        if Cp % e:
            raise ValueError(f"conv: input channels {Cp} must be padded to a multiple of {e} (ops.pad_channels)")
        w = weight.detach().contiguous()
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None
        bias_d = bias.detach().contiguous() if bias is not None else None
        if not transposed:
            O, I, KH, KW = w.shape
            if Cp < I:
                raise ValueError("conv: input has fewer channels than the weight")
            out = None
            if dest is not None:
                out = dest[0].slice(dest[1], O, (B, (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1))
            y = K.conv2d_forward(x, _packed(weight, w, 0, Cp, cdt), bias_d, stride, pad, relu, stats=want_stats, out=out)
            if want_stats:
                y, st = y
                ctx.cfg = (stride, pad, relu, transposed, bias is not None)
                ctx.save_for_backward(x, w, None)
                ctx.mark_non_differentiable(st)
                return y, st
        else:
            I, O, KH, KW = w.shape  # ConvTranspose2d weight layout
            if Cp != I or stride != 2 or pad != 1 or KH != 3:
                raise ValueError("conv_transpose: only k3 s2 p1 op1 without channel padding is built")
            if dest is not None:
                raise ValueError("conv_transpose: dest is not supported")
            y = K.conv2d_dgrad(x, _packed(weight, w, 1, Cp, cdt), (2 * H, 2 * W), 2, 1, bias=bias_d, relu=relu)
        ctx.cfg = (stride, pad, relu, transposed, bias is not None)
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy, *unused):
        stride, pad, relu, transposed, has_bias = ctx.cfg
        x, w, y = ctx.saved_tensors
        cdt = x.dtype
        e = K.epc(cdt)
        dy = K.nhwc(dy)
        Co = dy.shape[3]
        Cg = (Co + e - 1) // e * e
        dbias = None
        if Co % e == 0 and (relu or has_bias):
            dz, dbias = K.act_backward(dy, y, relu, want_dz=relu, want_dbias=has_bias)
            if dz is None:
                dz = dy
        else:  # tiny heads (9 channels): pad the channel pitch to a chunk, then the same kernel
            dz = pad_channels(dy * (y > 0) if relu else dy, e).contiguous()
            if has_bias:
                dbias = K.act_backward(dz, None, False, want_dz=False, want_dbias=True)[1][:Co]
        B, H, W, Cp = x.shape
        dx = dW = None
        if not transposed:
            O, I, KH, KW = w.shape
            if ctx.needs_input_grad[0]:
                if Cp != I:
                    raise RuntimeError("conv backward: gradient w.r.t. a channel-padded input is not supported")
                dx = K.conv2d_dgrad(dz, _packed(ctx.wparam, w, 1, Cg, cdt), (H, W), stride, pad)
            if ctx.needs_input_grad[1]:
                dW = _wgrad_async(ctx.wparam, dz, x, O, I, KH, KW, stride, pad)
        else:
            I, O, KH, KW = w.shape
            if ctx.needs_input_grad[0]:
                # d/dx of a transposed conv = ordinary stride-2 conv of the fine-grid gradient
                dx = K.conv2d_forward(dz, _packed(ctx.wparam, w, 0, Cg, cdt), None, 2, 1, False)
            if ctx.needs_input_grad[1]:
                dW = _wgrad_async(ctx.wparam, x, dz, I, O, KH, KW, 2, 1)
        return dx, dW, dbias, None, None, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, pad=0, relu=False, dest=None):
    return _Conv.apply(x, weight, bias, stride, pad, relu, False, False, dest)


def conv2d_with_stats(x, weight, stride=1, pad=0):
    """Bias-free conv that also returns the BatchNorm partial statistics of its output (from the fp32
    accumulators in the epilogue) -> feeds batch_norm(..., partial=...)."""
    return _Conv.apply(x, weight, None, stride, pad, False, False, True)


def conv_transpose2d(x, weight):
    return _Conv.apply(x, weight, None, 2, 1, False, True)


class _BatchNorm(torch.autograd.Function):
    """y = [relu](BatchNorm2d(x) * res_scale + res), basics.py:111-123."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, res, res_scale,
                partial=None, dest=None):
        x = K.nhwc(x)
        res_c = K.nhwc(res) if res is not None else None
        rs = float(res_scale) if res is not None else 1.0
        out = dest[0].slice(dest[1], x.shape[3], x.shape[:3]) if dest is not None else None
        y, mean, invstd = K.bn_forward(x, gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps,
                                       training, relu, res_c, rs, partial=partial if training else None, out=out)
        # without a residual the ReLU mask is a function of x alone: the backward recomputes it (mode 2)
        # instead of reading the saved output
        mode = 0 if not relu else (1 if res is not None else 2)
        ctx.gb = tuple(p if isinstance(p, torch.nn.Parameter) else None for p in (gamma, beta))
        ctx.cfg = (training, mode, rs, res is not None)
        ctx.save_for_backward(x, y if mode == 1 else None, gamma.detach(), beta.detach(), mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        training, relu, rs, has_res = ctx.cfg
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        sink = _bn_sink(*ctx.gb)
        dx, dres, dgamma, dbeta = K.bn_backward(K.nhwc(dy), y, x, gamma, mean, invstd, training, relu, rs,
                                                want_dres=has_res and ctx.needs_input_grad[9], beta=beta, grads_into=sink)
        if sink is not None:
            _bn_ready(*ctx.gb)
        if has_res and ctx.needs_input_grad[9] and dres is None:
            dres = dy
        return dx, dgamma, dbeta, None, None, None, None, None, None, dres, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu=False, res=None, res_scale=1.0,
               partial=None, dest=None):
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, res, res_scale,
                            partial, dest)


fold_shortcut_bn = os.environ.get("JSPSR_BN_FOLD_SHORTCUT", "1") != "0"   # A/B switch (Switches)
# conv1 -> bn1 -> ReLU -> conv2 of a BasicBlock without the normalised activation: conv2 (and its weight gradient) read
# conv1's RAW output and apply bn1's (scale | shift) + ReLU while staging their operand (jspsr_conv2d_forward: in_affine,
# jspsr_conv2d_wgrad: x_affine).  Where the patch kernels do not apply (channel counts not a multiple of 32 / 64) the
# block falls back to the separate BatchNorm pass.
fuse_bn1_input = os.environ.get("JSPSR_FUSE_BN1", "1") != "0"


class _ResUnit(torch.autograd.Function):
    """One BasicBlock (basics.py:88-123) as a single autograd node:

        out = [relu](bn2(conv3x3(relu(bn1(conv3x3_s(x))))) * scale + shortcut(x)),  shortcut = x | bn_d(conv1x1_s(x))

    Same kernels as the unfused composition (conv with BN statistics in its epilogue, BN apply, and their
    backward passes); what the fusion buys is in the backward: the gradient that reaches `x` along the shortcut is
    handed to the conv1 data-gradient kernel as its epilogue addend, so the two paths meet inside that kernel
    instead of in a separate add over the whole tensor (autograd's accumulation), and the node count seen by
    the autograd engine drops from 5-8 to 1."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, stride, scale, act, bns, dest, grad_extra=None):
        ctx.grad_extra = grad_extra
        x = K.nhwc(x)
        cdt, e = x.dtype, K.epc(x.dtype)
        B, H, W, Cp = x.shape
        O, I = w1.shape[:2]
        if Cp != I or Cp % e or O % e:
            raise ValueError(f"res_unit: {Cp} input channels against weight {tuple(w1.shape)} (multiples of {e} needed)")
        has_d = wd is not None
        (rm1, rv1, mom1, eps1, tr1), (rm2, rv2, mom2, eps2, tr2) = bns[0], bns[1]
        w1d, w2d = w1.detach().contiguous(), w2.detach().contiguous()
        g1d, b1d, g2d, b2d = g1.detach(), b1.detach(), g2.detach(), b2.detach()

        def conv_bn(inp, wt, k, st, pad, gam, bet, rm, rv, mom, eps, tr, relu, res=None, rs=1.0, out=None, par=None,
                    res_affine=None, stats_only=False, in_affine=None):
            z = K.conv2d_forward(inp, _packed(par, wt, 0, inp.shape[3], cdt), None, st, pad, False, stats=tr,
                                 in_affine=in_affine, in_relu=in_affine is not None)
            z, part = z if tr else (z, None)
            y, mean, invstd = K.bn_forward(z, gam, bet, rm, rv, mom, eps, tr, relu, res, rs, partial=part, out=out,
                                           res_affine=res_affine, stats_only=stats_only)
            return z, y, mean, invstd

        OH1, OW1 = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        fuse1 = fuse_bn1_input and K.fused_input_ok(cdt, B, OH1, OW1, O, O, 3, 3, 1, 1)
        if fuse1:    # statistics + (scale | shift) of bn1 only: relu(bn1(z1)) is formed inside conv2's / wgrad2's staging
            z1, aff1, m1, i1 = conv_bn(x, w1d, 3, stride, 1, g1d, b1d, rm1, rv1, mom1, eps1, tr1, True, par=w1, stats_only=True)
            y1 = None
        else:
            z1, y1, m1, i1 = conv_bn(x, w1d, 3, stride, 1, g1d, b1d, rm1, rv1, mom1, eps1, tr1, True, par=w1)
            aff1 = None
        if has_d:
            rmd, rvd, momd, epsd, trd = bns[2]
            wdd, gdd, bdd = wd.detach().contiguous(), gd.detach(), bd.detach()
            # the projection's BatchNorm is not applied in a pass of its own: its (scale | shift) rides into bn2's apply,
            # which reads the raw 1x1 output as the residual operand (jspsr_bn_forward: res_affine / affine_out)
            if fold_shortcut_bn:
                zd, r_aff, md, idd = conv_bn(x, wdd, 1, stride, 0, gdd, bdd, rmd, rvd, momd, epsd, trd, False, par=wd, stats_only=True)
                r = zd
            else:
                zd, r, md, idd = conv_bn(x, wdd, 1, stride, 0, gdd, bdd, rmd, rvd, momd, epsd, trd, False, par=wd)
                r_aff = None
        else:
            r_aff = None
            if stride != 1 or O != I:
                raise ValueError("res_unit: identity shortcut needs stride 1 and equal channel counts")
            wdd = gdd = bdd = zd = md = idd = None
            trd = False
            r = x
        out_v = dest[0].slice(dest[1], O, z1.shape[:3]) if dest is not None else None
        z2, out, m2, i2 = conv_bn(z1 if fuse1 else y1, w2d, 3, 1, 1, g2d, b2d, rm2, rv2, mom2, eps2, tr2, bool(act), r, float(scale),
                                  out_v, par=w2, res_affine=r_aff, in_affine=aff1)
        ctx.cfg = (stride, float(scale), bool(act), has_d, tr1, tr2, trd)
        ctx.wparams = tuple(p if isinstance(p, torch.nn.Parameter) else None for p in (w1, w2, wd))
        ctx.bparams = tuple(p if isinstance(p, torch.nn.Parameter) else None for p in (g1, b1, g2, b2, gd, bd))
        ctx.save_for_backward(x, w1d, g1d, b1d, z1, m1, i1, y1, w2d, g2d, b2d, z2, m2, i2, out if act else None,
                              wdd, gdd, bdd, zd, md, idd, aff1)
        return out

    @staticmethod
    def backward(ctx, dout):
        stride, scale, act, has_d, tr1, tr2, trd = ctx.cfg
        (x, w1, g1, b1, z1, m1, i1, y1, w2, g2, b2, z2, m2, i2, out, wd, gd, bd, zd, md, idd, aff1) = ctx.saved_tensors
        cdt = x.dtype
        B, H, W, Cin = x.shape
        O = w1.shape[0]
        dout = K.nhwc(dout)
        need_x = ctx.needs_input_grad[0]
        # bn2 (+ residual split): dz2 for the conv branch, dres for the shortcut
        pg1, pb1, pg2, pb2, pgd, pbd = ctx.bparams
        sink2, sink1 = _bn_sink(pg2, pb2), _bn_sink(pg1, pb1)
        dz2, dres, dg2, db2 = K.bn_backward(dout, out, z2, g2, m2, i2, tr2, 1 if act else 0, scale,
                                            want_dres=(need_x or has_d), beta=b2, grads_into=sink2)
        if sink2 is not None:
            _bn_ready(pg2, pb2)
        if dres is None:
            dres = dout
        p1, p2, pd = ctx.wparams

        def wgrad2():
            if aff1 is not None:      # conv2's input was never materialised: its weight gradient re-forms relu(bn1(z1)) while staging
                return _wgrad_async(p2, dz2, z1, O, O, 3, 3, 1, 1, x_affine=aff1, x_relu=True)
            return _wgrad_async(p2, dz2, y1, O, O, 3, 3, 1, 1)

        # lab (JSPSR_WGRAD_AFTER_DGRAD=1): fork each weight gradient BEHIND the data gradient of the same layer, so that it starts
        # beside the HBM-bound BatchNorm passes that follow instead of beside an MFMA-bound data gradient
        if not wgrad_after_dgrad:
            dW2 = wgrad2()
        # bn1's backward reduce (sum dz, sum dz xhat) rides in the epilogue of the data gradient that produces dy1, where the
        # launch is the patch kernel (the 64-channel bf16 layers on K2r keep the separate pass)
        Bz, Hz, Wz, _ = z1.shape
        part1 = None
        if bn_reduce_fused and tr1 and K.dgrad_reduce_ok(cdt, Bz, Hz, Wz, O, O, 3, 3, 1, 1) and z1.is_contiguous():
            par1 = K.bn_reduce_params(g1, b1, m1, i1)
            dy1, part1 = K.conv2d_dgrad(dz2, _packed(p2, w2, 1, O, cdt), z1.shape[1:3], 1, 1, red=(z1, par1))
        else:
            dy1 = K.conv2d_dgrad(dz2, _packed(p2, w2, 1, O, cdt), z1.shape[1:3], 1, 1)
        if wgrad_after_dgrad:
            dW2 = wgrad2()
        del dz2
        dz1, _, dg1, db1 = K.bn_backward(dy1, None, z1, g1, m1, i1, tr1, 2, 1.0, beta=b1, grads_into=sink1, ext_partial=part1)
        if sink1 is not None:
            _bn_ready(pg1, pb1)
        del dy1
        dW1 = None if wgrad_after_dgrad else _wgrad_async(p1, dz1, x, O, Cin, 3, 3, stride, 1)
        dWd = dgd = dbd = None
        side = dres                      # what reaches x along the shortcut
        # gradient of x parked by its other consumer (SliceBuffer.join(defer=...)): rides along as an addend too
        deposit = None
        if isinstance(ctx.grad_extra, tuple):
            # (holder, first channel, channels): `x` is one part of a joined tensor whose OTHER consumer parked its
            # gradient in `holder`; a third consumer of the joined tensor takes the parked gradient later (take_deferred).
            # This unit adds its own gradient of `x` into the parked slice in place -- out = addend's slice of the final
            # data-gradient launch -- and reports no gradient: autograd's full-tensor add where the two gradients of `x`
            # would meet does not happen (2 x 3 passes over full-resolution tensors per step)
            holder, off, n = ctx.grad_extra
            if holder.deferred is not None and has_d and need_x:
                cur = torch.cuda.current_stream()
                cur.wait_event(holder.deferred_event)
                deposit = holder.deferred.narrow(3, off, n)
                if tuple(deposit.shape) != tuple(x.shape):
                    raise RuntimeError(f"res_unit: parked gradient slice {tuple(deposit.shape)} does not match the input {tuple(x.shape)}")
                if deposit.dtype != x.dtype or not (deposit.is_contiguous() or K.is_slice(deposit)):
                    # autograd handed the join a gradient that is not a dense NHWC tensor of the compute dtype: the kernel
                    # would write it with the wrong pitch -- fall back to the ordinary gradient of x (autograd adds)
                    holder.pending_deposits -= 1
                    deposit = None
                else:
                    deposit.record_stream(cur)
            elif holder.deferred is not None:
                holder.pending_deposits -= 1           # this unit cannot deposit (no projection shortcut): ordinary gradient
        extra = ctx.grad_extra.take_deferred() if (ctx.grad_extra is not None and not isinstance(ctx.grad_extra, tuple)) else deposit
        if extra is not None:
            extra = K.nhwc(extra)
            if tuple(extra.shape) != tuple(x.shape):
                raise RuntimeError(f"res_unit: deferred gradient {tuple(extra.shape)} does not match the input {tuple(x.shape)}")
            if not (has_d and need_x):
                side, extra = side + extra, None
        if has_d:
            sinkd = _bn_sink(pgd, pbd)
            dzd, _, dgd, dbd = K.bn_backward(dres, None, zd, gd, md, idd, trd, 0, 1.0, beta=bd, grads_into=sinkd)
            if sinkd is not None:
                _bn_ready(pgd, pbd)
            dWd = _wgrad_async(pd, dzd, x, O, Cin, 1, 1, stride, 0)
            side = K.conv2d_dgrad(dzd, _packed(pd, wd, 1, O, cdt), (H, W), stride, 0, addend=extra) if need_x else None
        dx = None
        if need_x and deposit is not None:
            K.conv2d_dgrad(dz1, _packed(p1, w1, 1, O, cdt), (H, W), stride, 1, addend=K.nhwc(side), out=deposit)
            ctx.grad_extra[0].deposit_events.append(torch.cuda.current_stream().record_event())
        elif need_x:
            dx = K.conv2d_dgrad(dz1, _packed(p1, w1, 1, O, cdt), (H, W), stride, 1, addend=K.nhwc(side))
        if wgrad_after_dgrad:
            dW1 = _wgrad_async(p1, dz1, x, O, Cin, 3, 3, stride, 1)
        return (dx, dW1, dg1, db1, dW2, dg2, db2, dWd, dgd, dbd, None, None, None, None, None, None)


def conv_bn_infer(x, weight, gamma, beta, bn, stride, pad, relu=False, res=None, res_scale=1.0, dest=None):
    """Inference (no autograd, BatchNorm on its running statistics): conv -> BN (-> * res_scale + res) (-> ReLU) as
    ONE launch -- the normalisation is a per-channel affine folded into the conv epilogue, the shortcut is its addend.
    bn = (running_mean, running_var, momentum, eps, training)."""
    x = K.nhwc(x)
    rm, rv, _, eps, training = bn
    if training or rm is None:
        raise RuntimeError("conv_bn_infer: BatchNorm must be in eval mode with running statistics")
    sc, sh = K.bn_fold(gamma.detach(), beta.detach(), rm, rv, eps, res_scale if res is not None else 1.0)
    O, I, KH, KW = weight.shape
    B, H, W, Cp = x.shape
    out = None
    if dest is not None:
        out = dest[0].slice(dest[1], O, (B, (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1))
    return K.conv2d_forward(x, _packed(weight, weight.detach().contiguous(), 0, Cp, x.dtype), sh, stride, pad, relu, out=out,
                            scale=sc, addend=K.nhwc(res) if res is not None else None)


def conv_transpose_bn_infer(x, weight, gamma, beta, bn, relu=True, dest=None):
    """Inference: ConvTranspose2d(k3 s2 p1 op1) -> BatchNorm(eval) (-> ReLU) of the decoder (basics.py:69-85), one
    launch per stride phase with the normalisation folded into the epilogue."""
    x = K.nhwc(x)
    rm, rv, _, eps, training = bn
    if training or rm is None:
        raise RuntimeError("conv_transpose_bn_infer: BatchNorm must be in eval mode with running statistics")
    I, O, KH, KW = weight.shape
    B, H, W, Cp = x.shape
    if Cp != I or KH != 3:
        raise ValueError("conv_transpose: only k3 s2 p1 op1 without channel padding is built")
    sc, sh = K.bn_fold(gamma.detach(), beta.detach(), rm, rv, eps, 1.0)
    out = dest[0].slice(dest[1], O, (B, 2 * H, 2 * W)) if dest is not None else None
    return K.conv2d_dgrad(x, _packed(weight, weight.detach().contiguous(), 1, Cp, x.dtype), (2 * H, 2 * W), 2, 1, bias=sh,
                          relu=relu, out=out, scale=sc)


def res_unit_infer(x, w1, g1, b1, w2, g2, b2, wd, gd, bd, stride, scale, act, bns, dest=None):
    """BasicBlock in inference mode: three (two without projection) launches, nothing else touches the tensors."""
    y1 = conv_bn_infer(x, w1, g1, b1, bns[0], stride, 1, relu=True)
    r = conv_bn_infer(x, wd, gd, bd, bns[2], stride, 0) if wd is not None else x
    return conv_bn_infer(y1, w2, g2, b2, bns[1], 1, 1, relu=bool(act), res=r, res_scale=scale, dest=dest)


def res_unit(x, w1, g1, b1, w2, g2, b2, wd, gd, bd, stride, scale, act, bns, dest=None, grad_extra=None):
    """bns: ((running_mean, running_var, momentum, eps, training), ...) for bn1, bn2[, downsample bn].
    grad_extra: a SliceBuffer whose `deferred` gradient (see SliceBuffer.join) belongs to `x`."""
    if not torch.is_grad_enabled() and not any(b[4] for b in bns):
        return res_unit_infer(x, w1, g1, b1, w2, g2, b2, wd, gd, bd, stride, scale, act, bns, dest)
    return _ResUnit.apply(x, w1, g1, b1, w2, g2, b2, wd, gd, bd, stride, scale, act, bns, dest, grad_extra)


def _gate_mlp(avg, mx, w1, w2):
    f = torch.nn.functional
    h = lambda v: f.linear(f.relu(f.linear(v, w1.flatten(1))), w2.flatten(1))
    return torch.sigmoid(h(avg) + h(mx))


class _ChannelGate(torch.autograd.Function):
    """x * sigmoid(MLP(avgpool x) + MLP(maxpool x)); resnet_cbam.py:49-53 + basics.py:57-58.
    Pooling / scaling / their backward are HIP kernels over the NHWC tensor; the C -> C/16 -> C MLP
    acts on (B, C) vectors and is evaluated (and differentiated) by the host library."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        x = x.contiguous()
        avg, mx, amax = K.gate_pool(x)
        w1f, w2f = w1.detach().float().flatten(1).contiguous(), w2.detach().float().flatten(1).contiguous()
        s, hid = K.gate_mlp_forward(avg, mx, w1f, w2f)          # one launch (torch: 10 library kernels)
        y = K.gate_scale(x, s)
        ctx.save_for_backward(x, avg, mx, amax, s, hid, w1f, w2f)
        ctx.wshapes = (w1.shape, w2.shape, w1.dtype, w2.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, avg, mx, amax, s, hid, w1f, w2f = ctx.saved_tensors
        dy = dy.contiguous()
        ds = K.gate_backward_reduce(dy, x)
        davg, dmax, dw1, dw2 = K.gate_mlp_backward(ds, s, hid, avg, mx, w1f, w2f)      # two launches (torch: ~35)
        dx = K.gate_backward_apply(dy, s, davg, dmax, amax)
        s1, s2, t1, t2 = ctx.wshapes
        return dx, dw1.reshape(s1).to(t1), dw2.reshape(s2).to(t2)


def channel_gate(x, w1, w2):
    return _ChannelGate.apply(x, w1, w2)
