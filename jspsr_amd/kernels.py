"""Thin, autograd-free Python wrappers over the C ABI (include/jspsr_hip.h).

Activations are NHWC torch tensors of shape (B, H, W, C), contiguous, fp32 or bf16, on the GPU;
torch only owns the memory.  Every function launches on the current HIP stream.
"""
from __future__ import annotations

import torch

from . import _lib

F32, BF16 = 0, 1


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def epc(dtype: torch.dtype) -> int:
    """Channel granularity of gathered tensors: elements per 16-byte chunk."""
    return 4 if dtype == torch.float32 else 8


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: jspsr_amd kernels run on the GPU only (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")


def is_slice(t: torch.Tensor) -> bool:
    """(B,H,W,C) tensor that is a channel slice of a wider dense NHWC buffer: unit channel stride, one common
    pixel pitch, 16-byte aligned rows.  The kernels address such a tensor as (pointer, pitch) -- no copy."""
    if t.dim() != 4 or t.stride(3) != 1:
        return False
    B, H, W, C = t.shape
    p = t.stride(2)
    if p < C or (H > 1 and t.stride(1) != W * p) or (B > 1 and t.stride(0) != H * W * p) or W == 1:
        return False
    return t.data_ptr() % 16 == 0 and (p * t.element_size()) % 16 == 0


def pitch(t: torch.Tensor) -> int:
    """Channel pitch (elements between consecutive pixels)."""
    return t.shape[3] if t.is_contiguous() else t.stride(2)


def nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return `t` itself if the kernels can address it in place, else a dense copy."""
    return t if (t.is_contiguous() or is_slice(t)) else t.contiguous()


def _chk_s(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: jspsr_amd kernels run on the GPU only (no CPU fallback)")
    if not (t.is_contiguous() or is_slice(t)):
        raise ValueError(f"{name}: expected a dense NHWC tensor or a channel slice of one")


def pack_weight(w: torch.Tensor, mode: int, c_pad: int, dtype: torch.dtype) -> torch.Tensor:
    """(O,I,KH,KW) fp32 master -> [O][KH][KW][c_pad] (mode 0) or [I][KH][KW][c_pad] (mode 1)."""
    _chk(w, "pack_weight")
    O, I, KH, KW = w.shape
    n = O if mode == 0 else I
    out = torch.empty((n, KH, KW, c_pad), dtype=dtype, device=w.device)
    lib = _lib.load()
    _lib.check(lib.jspsr_pack_weight(_dt(out), w.data_ptr(), out.data_ptr(), O, I, KH, KW, mode, c_pad, _stream()),
               "jspsr_pack_weight")
    return out


def fused_input_ok(dtype, B, OH, OW, Cin, Cout, KH, KW, stride, pad) -> bool:
    """Can a conv with this geometry read its input through (scale | shift) + ReLU -- in the forward (in_affine) AND in
    its weight gradient (x_affine)?  (The patch kernels: 3x3, stride 1, pad 1, Cin a multiple of 32 fp32 / 64 bf16.)"""
    lib = _lib.load()
    dt = F32 if dtype == torch.float32 else BF16
    return bool(lib.jspsr_conv2d_in_affine_ok(dt, Cin, KH, KW, stride)) and \
        bool(lib.jspsr_conv2d_wgrad_x_affine_ok(dt, B, OH, OW, Cout, Cin, KH, KW, stride, pad))


def conv2d_forward(x, wpack, bias, stride, pad, relu=False, out=None, out_coff=0, cin=None, in_coff=0, stats=False,
                   scale=None, addend=None, in_affine=None, in_relu=False):
    """x (B,IH,IW,Cs) NHWC, wpack [Cout][KH][KW][Cin] -> (B,OH,OW,Cout) (or a slice of `out`).
    stats=True (no bias / ReLU): also returns the BatchNorm partial statistics (rows, 2, Cout) fp32 taken from
    the accumulators in the epilogue.  scale (Cout,) fp32 / addend (B,OH,OW,Cout): inference epilogue
    out = [relu](acc * scale + bias + addend)."""
    _chk_s(x, "conv2d_forward")
    B, IH, IW, _ = x.shape
    Cs = pitch(x)
    Cout, KH, KW, Cin = wpack.shape
    if cin is not None and cin != Cin:
        raise ValueError("conv2d_forward: cin mismatch")
    OH = (IH + 2 * pad - KH) // stride + 1
    OW = (IW + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty((B, OH, OW, Cout), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    st = None
    if stats:
        st = torch.empty((lib.jspsr_conv2d_stats_rows(B, OH, OW), 2, Cout), dtype=torch.float32, device=x.device)
    if addend is not None:
        _chk_s(addend, "conv2d_forward addend")
        if tuple(addend.shape) != (B, OH, OW, Cout) or addend.dtype != x.dtype:
            raise ValueError(f"conv2d_forward: addend {tuple(addend.shape)} {addend.dtype} does not match the result")
    _lib.check(lib.jspsr_conv2d_forward(_dt(x), x.data_ptr(), wpack.data_ptr(),
                                        bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                        B, IH, IW, Cin, Cs, in_coff, Cout, pitch(out), out_coff,
                                        KH, KW, stride, pad, int(relu), st.data_ptr() if st is not None else None,
                                        scale.data_ptr() if scale is not None else None,
                                        addend.data_ptr() if addend is not None else None,
                                        pitch(addend) if addend is not None else 0,
                                        in_affine.data_ptr() if in_affine is not None else None, int(in_relu),
                                        _stream()), "jspsr_conv2d_forward")
    return (out, st) if stats else out


def dgrad_reduce_ok(dtype, B, IH, IW, Cg, Cin, KH, KW, stride, pad) -> bool:
    """Can this data gradient carry the reduce pass of the BatchNorm behind its input (conv2d_dgrad(red=...))?"""
    dt = F32 if dtype == torch.float32 else BF16
    return bool(_lib.load().jspsr_conv2d_dgrad_reduce_ok(dt, B, IH, IW, Cg, Cin, KH, KW, stride, pad))


def bn_reduce_params(gamma, beta, mean, invstd):
    """[4][C] fp32 (a | b | is | mis) for conv2d_dgrad(red=...): mask = a z + b > 0, xhat = z is + mis."""
    C = gamma.numel()
    par = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    _lib.check(_lib.load().jspsr_bn_reduce_params(gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(), C,
                                                  par.data_ptr(), _stream()), "jspsr_bn_reduce_params")
    return par


def conv2d_dgrad(g, wpack_t, in_hw, stride, pad, bias=None, relu=False, out=None, out_coff=0, g_coff=0, addend=None,
                 scale=None, red=None):
    """g (B,OH,OW,Cgs) NHWC, wpack_t [Cin][KH][KW][Cg] -> (B,IH,IW,Cin): data gradient of a conv /
    forward of a transposed conv.  addend (B,IH,IW,Cin): added in the epilogue (a gradient arriving along
    another path)."""
    _chk_s(g, "conv2d_dgrad")
    B, OH, OW, _ = g.shape
    Cgs = pitch(g)
    Cin, KH, KW, Cg = wpack_t.shape
    IH, IW = in_hw
    if out is None:
        out = torch.empty((B, IH, IW, Cin), dtype=g.dtype, device=g.device)
    else:
        _chk_s(out, "conv2d_dgrad out")         # a dense NHWC tensor or a 16-byte-aligned channel slice of one
        if tuple(out.shape[:3]) != (B, IH, IW) or out.shape[3] < out_coff + Cin or out.dtype != g.dtype:
            raise ValueError(f"conv2d_dgrad: out {tuple(out.shape)} {out.dtype} cannot take a ({B},{IH},{IW},{Cin}) {g.dtype} result at channel {out_coff}")
    if addend is not None:
        _chk_s(addend, "conv2d_dgrad addend")
        if tuple(addend.shape) != (B, IH, IW, Cin) or addend.dtype != g.dtype:
            raise ValueError(f"conv2d_dgrad: addend {tuple(addend.shape)} {addend.dtype} does not match the result")
    lib = _lib.load()
    red_x = red_par = red_out = None
    if red is not None:
        # red = (x, par): the BatchNorm's saved input on the result's grid and jspsr_bn_reduce_params' table -> the partial
        # rows of its backward reduce come back beside the result
        red_x, red_par = red
        _chk_s(red_x, "conv2d_dgrad red_x")
        if tuple(red_x.shape) != (B, IH, IW, Cin) or red_x.dtype != g.dtype:
            raise ValueError(f"conv2d_dgrad: red_x {tuple(red_x.shape)} {red_x.dtype} does not match the result")
        rows = lib.jspsr_conv2d_stats_rows(B, IH, IW)
        red_out = torch.empty((rows, 2, Cin), dtype=torch.float32, device=g.device)
    _lib.check(lib.jspsr_conv2d_dgrad(_dt(g), g.data_ptr(), wpack_t.data_ptr(),
                                      bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                      B, OH, OW, Cg, Cgs, g_coff, IH, IW, Cin, pitch(out), out_coff,
                                      KH, KW, stride, pad, int(relu),
                                      addend.data_ptr() if addend is not None else None,
                                      pitch(addend) if addend is not None else 0,
                                      scale.data_ptr() if scale is not None else None,
                                      red_x.data_ptr() if red is not None else None, pitch(red_x) if red is not None else 0,
                                      red_par.data_ptr() if red is not None else None,
                                      red_out.data_ptr() if red is not None else None, _stream()), "jspsr_conv2d_dgrad")
    return (out, red_out) if red is not None else out


def conv2d_wgrad(G, X, R, C, KH, KW, stride, pad, out=None, accumulate=False, g_coff=0, cg=None, x_coff=0, cx=None,
                 x_affine=None, x_relu=False):
    """dW (R,C,KH,KW) fp32 = sum_pixels G[.., r] * X[shifted.., c].  G (B,OH,OW,Cgs), X (B,IH,IW,Cxs) NHWC."""
    _chk_s(G, "conv2d_wgrad")
    _chk_s(X, "conv2d_wgrad")
    B, OH, OW, Cg_ = G.shape
    _, IH, IW, Cx_ = X.shape
    Cgs, Cxs = pitch(G), pitch(X)
    cg = Cg_ if cg is None else cg
    cx = Cx_ if cx is None else cx
    if out is None:
        out = torch.empty((R, C, KH, KW), dtype=torch.float32, device=G.device)
    lib = _lib.load()
    dt = _dt(G)
    nbytes = lib.jspsr_conv2d_wgrad_workspace_bytes(dt, B, OH, OW, cg, cx, KH, KW)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=G.device)
    _lib.check(lib.jspsr_conv2d_wgrad(dt, G.data_ptr(), cg, Cgs, g_coff, X.data_ptr(), cx, Cxs, x_coff,
                                      out.data_ptr(), R, C, B, OH, OW, IH, IW, KH, KW, stride, pad,
                                      int(accumulate), x_affine.data_ptr() if x_affine is not None else None, int(x_relu),
                                      ws.data_ptr(), _stream()), "jspsr_conv2d_wgrad")
    return out


# ---------------------------------------------------------------------------------------------
# per-channel operators (elementwise.hip)
# ---------------------------------------------------------------------------------------------

_ws_cache = {}


def _workspace(dtype_code: int, C: int, nseg: int, device) -> torch.Tensor:
    """Scratch for the reduction kernels; cached per (stream, size class) -- stream-ordered reuse."""
    lib = _lib.load()
    n = lib.jspsr_reduce_workspace_bytes(dtype_code, C, nseg)
    key = (device, _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def bn_forward(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu=False, res=None,
               res_scale=1.0, out=None, out_coff=0, partial=None, res_affine=None, stats_only=False):
    """x (B,H,W,C) -> y = [relu](bn(x)*res_scale + res); returns (y, save_mean, save_invstd).
    partial: (rows, 2, C) statistics from conv2d_forward(stats=True) -- skips the statistics pass.
    stats_only: no output tensor; returns (affine, save_mean, save_invstd) with affine (2, C) fp32 = this BatchNorm's
    per-channel (scale | shift) for a consumer to apply.  res_affine: such an affine for the residual operand."""
    _chk_s(x, "bn_forward")
    B, H, W, C = x.shape
    affine = torch.empty((2, C), dtype=torch.float32, device=x.device) if stats_only else None
    if out is None and not stats_only:
        out = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    dt = _dt(x)
    ws = _workspace(dt, C, 1, x.device)
    lib = _lib.load()
    _lib.check(lib.jspsr_bn_forward(dt, x.data_ptr(), pitch(x), 0, res.data_ptr() if res is not None else None,
                                    pitch(res) if res is not None else 0, 0, out.data_ptr() if out is not None else None,
                                    pitch(out) if out is not None else 0, out_coff,
                                    gamma.data_ptr(), beta.data_ptr(),
                                    running_mean.data_ptr() if running_mean is not None else None,
                                    running_var.data_ptr() if running_var is not None else None,
                                    float(momentum), float(eps), int(training), int(relu), float(res_scale),
                                    mean.data_ptr(), invstd.data_ptr(), B * H * W, C,
                                    partial.data_ptr() if partial is not None else None,
                                    partial.shape[0] if partial is not None else 0,
                                    res_affine.data_ptr() if res_affine is not None else None,
                                    affine.data_ptr() if affine is not None else None, ws.data_ptr(), _stream()),
               "jspsr_bn_forward")
    return (affine if stats_only else out), mean, invstd


def bn_fold(gamma, beta, running_mean, running_var, eps, res_scale=1.0):
    """Eval-mode BatchNorm as (scale, shift) fp32 vectors for a conv epilogue."""
    C = gamma.numel()
    scale = torch.empty(C, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    _lib.check(_lib.load().jspsr_bn_fold(gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                                         float(eps), float(res_scale), C, scale.data_ptr(), shift.data_ptr(), _stream()),
               "jspsr_bn_fold")
    return scale, shift


def bn_backward(dy, y, x, gamma, mean, invstd, training, relu, res_scale=1.0, want_dres=False, beta=None,
                grads_into=None, ext_partial=None):
    """-> (dx, dres or None, dgamma, dbeta).  relu: False/0, True/1 (mask from y) or 2 (mask from x, needs beta).
    grads_into = (dgamma_buf, dbeta_buf): add the parameter gradients to those fp32 buffers instead of
    returning fresh tensors (then dgamma, dbeta come back as None)."""
    _chk_s(dy, "bn_backward")
    B, H, W, C = x.shape
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    dres = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device) if want_dres else None
    if grads_into is not None:
        dgamma, dbeta = grads_into
        for t in (dgamma, dbeta):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != C or not t.is_cuda:
                raise ValueError("bn_backward: grads_into buffers must be contiguous fp32 of C elements on the GPU")
    else:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    dt = _dt(x)
    ws = _workspace(dt, C, 1, x.device)
    lib = _lib.load()
    _lib.check(lib.jspsr_bn_backward(dt, dy.data_ptr(), pitch(dy), 0, y.data_ptr() if y is not None else None,
                                     pitch(y) if y is not None else 0, 0, x.data_ptr(), pitch(x), 0, gamma.data_ptr(),
                                     beta.data_ptr() if beta is not None else None,
                                     mean.data_ptr(), invstd.data_ptr(), int(training), int(relu), float(res_scale),
                                     dx.data_ptr(), dres.data_ptr() if dres is not None else None, dgamma.data_ptr(),
                                     dbeta.data_ptr(), int(grads_into is not None), B * H * W, C, ws.data_ptr(),
                                     ext_partial.data_ptr() if ext_partial is not None else None,
                                     ext_partial.shape[0] if ext_partial is not None else 0,
                                     _stream()), "jspsr_bn_backward")
    if grads_into is not None:
        return dx, dres, None, None
    return dx, dres, dgamma, dbeta


def act_backward(dy, y, relu, want_dz=True, want_dbias=True, dz_channels=None):
    """dz = dy*[y>0] (optionally into a wider, zero-initialised channel pitch), dbias = sum dz."""
    _chk_s(dy, "act_backward")
    B, H, W, C = dy.shape
    dz = None
    if want_dz:
        cz = C if dz_channels is None else dz_channels
        dz = torch.empty((B, H, W, cz), dtype=dy.dtype, device=dy.device) if cz == C else \
            torch.zeros((B, H, W, cz), dtype=dy.dtype, device=dy.device)
    dbias = torch.empty(C, dtype=torch.float32, device=dy.device) if want_dbias else None
    dt = _dt(dy)
    ws = _workspace(dt, C, 1, dy.device)
    lib = _lib.load()
    if y is not None:
        y = nhwc(y)
    _lib.check(lib.jspsr_act_backward(dt, dy.data_ptr(), pitch(dy), 0, y.data_ptr() if y is not None else None,
                                      pitch(y) if y is not None else 0, int(relu),
                                      dz.data_ptr() if dz is not None else None, dz.shape[3] if dz is not None else 0,
                                      dbias.data_ptr() if dbias is not None else None, B * H * W, C, ws.data_ptr(),
                                      _stream()), "jspsr_act_backward")
    return dz, dbias


def gate_pool(x):
    _chk(x, "gate_pool")
    B, H, W, C = x.shape
    avg = torch.empty((B, C), dtype=torch.float32, device=x.device)
    mx = torch.empty_like(avg)
    amax = torch.empty((B, C), dtype=torch.int32, device=x.device)
    dt = _dt(x)
    ws = _workspace(dt, C, B, x.device)
    lib = _lib.load()
    _lib.check(lib.jspsr_gate_pool(dt, x.data_ptr(), B, H * W, C, avg.data_ptr(), mx.data_ptr(), amax.data_ptr(),
                                   ws.data_ptr(), _stream()), "jspsr_gate_pool")
    return avg, mx, amax


def gate_scale(x, s):
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.jspsr_gate_scale(_dt(x), x.data_ptr(), s.data_ptr(), y.data_ptr(), B, H * W, C, _stream()),
               "jspsr_gate_scale")
    return y


def gate_backward_reduce(dy, x):
    B, H, W, C = x.shape
    ds = torch.empty((B, C), dtype=torch.float32, device=x.device)
    dt = _dt(x)
    ws = _workspace(dt, C, B, x.device)
    lib = _lib.load()
    _lib.check(lib.jspsr_gate_backward_reduce(dt, dy.data_ptr(), x.data_ptr(), ds.data_ptr(), B, H * W, C,
                                              ws.data_ptr(), _stream()), "jspsr_gate_backward_reduce")
    return ds


def gate_backward_apply(dy, s, davg, dmax, amax):
    B, H, W, C = dy.shape
    dx = torch.empty_like(dy)
    lib = _lib.load()
    _lib.check(lib.jspsr_gate_backward_apply(_dt(dy), dy.data_ptr(), s.data_ptr(), davg.data_ptr(), dmax.data_ptr(),
                                             amax.data_ptr(), dx.data_ptr(), B, H * W, C, _stream()),
               "jspsr_gate_backward_apply")
    return dx


def nchw_to_nhwc(x, dtype, c_pad):
    """Boundary tensor (B,C,H,W) contiguous fp32 -> (B,H,W,c_pad) in `dtype`, channels last, zero-padded: one launch."""
    if x.dtype != torch.float32 or not x.is_contiguous() or not x.is_cuda:
        raise ValueError("nchw_to_nhwc: a contiguous fp32 CUDA tensor (B,C,H,W) is expected")
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, c_pad), dtype=dtype, device=x.device)
    lib = _lib.load()
    _lib.check(lib.jspsr_nchw_to_nhwc(BF16 if dtype == torch.bfloat16 else F32, x.data_ptr(), out.data_ptr(),
                                      B, C, H, W, c_pad, _stream()), "jspsr_nchw_to_nhwc")
    return out


def gate_mlp_forward(avg, mx, w1, w2):
    """s = sigmoid(W2 relu(W1 avg) + W2 relu(W1 mx)) on (B, C) vectors; w1 (Ch, C), w2 (C, Ch) fp32.  Returns (s, hid)."""
    B, C = avg.shape
    Ch = w1.shape[0]
    s = torch.empty((B, C), dtype=torch.float32, device=avg.device)
    hid = torch.empty((B, 2, Ch), dtype=torch.float32, device=avg.device)
    lib = _lib.load()
    _lib.check(lib.jspsr_gate_mlp_forward(avg.data_ptr(), mx.data_ptr(), w1.data_ptr(), w2.data_ptr(), B, C, Ch, s.data_ptr(),
                                          hid.data_ptr(), _stream()), "jspsr_gate_mlp_forward")
    return s, hid


def gate_mlp_backward(ds, s, hid, avg, mx, w1, w2):
    """Gradients of the gate MLP: (davg, dmax, dw1, dw2)."""
    B, C = avg.shape
    Ch = w1.shape[0]
    davg, dmax = torch.empty_like(avg), torch.empty_like(avg)
    dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
    lib = _lib.load()
    ws = torch.empty(lib.jspsr_gate_mlp_backward_workspace_bytes(B, C, Ch) // 4, dtype=torch.float32, device=avg.device)
    _lib.check(lib.jspsr_gate_mlp_backward(ds.data_ptr(), s.data_ptr(), hid.data_ptr(), avg.data_ptr(), mx.data_ptr(), w1.data_ptr(),
                                           w2.data_ptr(), B, C, Ch, davg.data_ptr(), dmax.data_ptr(), dw1.data_ptr(), dw2.data_ptr(),
                                           ws.data_ptr(), _stream()), "jspsr_gate_mlp_backward")
    return davg, dmax, dw1, dw2
