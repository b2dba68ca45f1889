"""GPU parity: K1 (fused propagation, HIP, through the C ABI) against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import jspsr_ref as R
from oracle import prop_ref as C


def _ops():
    from jspsr_amd import ops
    return ops


def _rand_case(B, H, W, sigma, seed, oc=18):
    g = torch.Generator().manual_seed(seed)
    dem = torch.rand(B, 1, H, W, generator=g)
    weight = torch.sigmoid(torch.randn(B, 9, H, W, generator=g))
    offset = sigma * torch.randn(B, 18, H, W, generator=g)
    offset[:, 8:10] = 0
    w = 1 + 0.3 * torch.randn(1, 1, 3, 3, generator=g)
    b = 0.1 * torch.randn(1, generator=g)
    gout = torch.randn(B, 1, H, W, generator=g)
    return dem, weight, offset, w, b, gout


def _run_hip(dem, weight, offset, w, b, gout, oc=18, scale=1.0):
    ops = _ops()
    d = lambda t: t.cuda()
    off = offset if oc == 18 else torch.cat((offset[:, :8], offset[:, 10:]), 1)
    wt, of, wp, bp = d(weight).requires_grad_(), d(off).requires_grad_(), d(w).requires_grad_(), d(b).requires_grad_()
    out = ops.propagate(d(dem), wt, of, wp, bp, scale)
    out.backward(d(gout))
    go = of.grad.cpu()
    if oc == 16:
        go = torch.cat((go[:, :8], torch.zeros_like(go[:, :2]), go[:, 8:]), 1)
    return out.detach().cpu(), wt.grad.cpu(), go, wp.grad.cpu(), bp.grad.cpu()


def _run_oracle64(dem, weight, offset, w, b, gout, scale=1.0):
    f = lambda t: t.double()
    out = R.propagate(f(dem), f(weight), f(offset), f(w), f(b), scale)
    gw, go, gW, gb = R.propagate_analytic_backward(f(dem), f(weight), f(offset), f(w), f(b), f(gout))
    return out, gw, go, gW, gb


def _close(a, b, rtol, atol, what):
    err = (a.double() - b.double()).abs()
    tol = atol + rtol * b.double().abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max().item():.3e} (tol {tol.min().item():.1e})"


@pytest.mark.parametrize("oc", [18, 16])
def test_golden_fixture(golden_dir, oc):
    z = np.load(os.path.join(golden_dir, "g1_postprocessor.npz"))
    t = lambda k: torch.from_numpy(z[k]).float()
    got = _run_hip(t("dem"), t("weight"), t("offset"), t("w"), t("b"), t("grad_out"), oc)
    # compare with the oracle evaluated on the same fp32-rounded inputs (fp64 arithmetic)
    exp = list(_run_oracle64(t("dem"), t("weight"), t("offset"), t("w"), t("b"), t("grad_out")))
    if oc == 16:  # the 16-channel layout has no centre-tap offset, hence no gradient for it
        exp[2] = exp[2].clone()
        exp[2][:, 8:10] = 0
    off = t("offset").double().reshape(2, 9, 2, 20, 24)
    near = (off.abs().amax((1, 2)) < 20).unsqueeze(1)  # |p| ~ 100 px: fp32 coordinate rounding, see below
    _close(got[0] * near, exp[0] * near, 1e-5, 2e-6, "out")
    _close(got[1] * near, exp[1] * near, 1e-5, 5e-6, "grad_weight")
    _close(got[2] * near, exp[2] * near, 1e-4, 5e-6, "grad_offset")
    _close(got[3], exp[3], 1e-4, 1e-4, "grad_w")
    _close(got[4], exp[4], 1e-5, 1e-4, "grad_b")
    # and directly with what the reference's own module produced (fp64 inputs), forward only
    _close(got[0] * near, torch.from_numpy(z["out"]) * near, 1e-5, 5e-6, "out vs reference fixture")


@pytest.mark.parametrize("shape,sigma,oc", [
    ((1, 16, 64), 1.5, 18),      # exactly one tile
    ((2, 37, 53), 1.5, 18),      # ragged both ways, W % 4 != 0 -> scalar path
    ((2, 40, 100), 2.5, 16),     # W % 4 == 0 but not a tile multiple
    ((3, 8, 8), 0.7, 18),        # smaller than a tile
    ((1, 129, 260), 8.0, 18),    # stress: most taps leave tile+halo -> global fallback
    ((1, 64, 64), 0.0, 16),      # zero offsets: plain 3x3 window
    ((2, 5, 3), 3.0, 16),        # tiny raster, taps mostly outside
])
def test_random_cases(shape, sigma, oc):
    case = _rand_case(*shape, sigma, seed=sum(shape) + int(sigma * 10))
    got = _run_hip(*case, oc=oc)
    exp = list(_run_oracle64(*case))
    if oc == 16:
        exp[2] = exp[2].clone()
        exp[2][:, 8:10] = 0
    # fp32 coordinates: ulp(p) ~ 4e-6 at |p| ~ 50 px times the white-noise DEM's unit slope
    tol = 2e-5 if sigma <= 3 else 1e-4
    _close(got[0], exp[0], 1e-5, tol, "out")
    _close(got[1], exp[1], 1e-5, tol, "grad_weight")
    # d/d(offset) of a bilinear sample jumps at integer coordinates: skip samples whose fp32
    # coordinate could land on the other side of the kink (|frac| < 1e-4)
    off = case[2].double()
    B, _, H, W = off.shape
    ys = torch.arange(H, dtype=torch.float64).view(1, 1, H, 1)
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, 1, W)
    pos = off.clone()
    pos[:, 0::2] += ys
    pos[:, 1::2] += xs
    frac = (pos - pos.round()).abs().reshape(B, 9, 2, H, W)
    smooth = (frac.amin(2, keepdim=True) > 1e-4).expand(B, 9, 2, H, W).reshape(B, 18, H, W)
    _close(got[2] * smooth, exp[2] * smooth, 1e-4, 2 * tol, "grad_offset")
    assert sigma == 0 or smooth.double().mean() > 0.85  # the centre tap (offset 0) always sits on a kink
    _close(got[3], exp[3], 1e-4, 5e-4, "grad_w")
    _close(got[4], exp[4], 1e-5, 5e-4, "grad_b")


def test_matches_c_oracle_fp32_bitclose():
    """Same fp32 arithmetic in plain C (oracle/prop_ref.c): agreement to a few ulp."""
    case = _rand_case(2, 48, 96, 2.0, seed=7)
    got = _run_hip(*case)
    n = [t.numpy() for t in case]
    out = C.forward(n[0], n[1], n[2], n[3], float(n[4][0]), 1.0)
    gw, go, gwk, gb = C.backward(n[5], n[0], n[1], n[2], n[3])
    assert np.abs(got[0].numpy() - out).max() < 2e-6
    assert np.abs(got[1].numpy() - gw).max() < 2e-6
    assert np.abs(got[2].numpy() - go).max() < 1e-5


def test_integer_and_edge_offsets():
    """Integer offsets = shifted gather; taps pushed past -1 / H contribute exactly 0."""
    B, H, W = 1, 32, 64
    dem = torch.rand(B, 1, H, W)
    weight = torch.rand(B, 9, H, W)
    w, b = torch.ones(1, 1, 3, 3), torch.zeros(1)
    off = torch.zeros(B, 18, H, W)
    off[:, 0] = -100.0   # tap 0 far above
    off[:, 2] = 3.0      # tap 1: dy +3
    off[:, 3] = -2.0     #        dx -2
    off[:, 16] = float("inf")
    off[:, 17] = float("nan")
    got = _run_hip(dem, weight, off, w, b, torch.ones(B, 1, H, W))
    offc = off.clone()
    offc[:, 16:18] = 1e6  # oracle: inf/nan are outside the raster -> tap contributes 0
    exp = _run_oracle64(dem, weight, offc, w, b, torch.ones(B, 1, H, W))
    _close(got[0], exp[0], 1e-5, 2e-6, "out")
    assert torch.isfinite(got[0]).all() and torch.isfinite(got[2]).all()


def test_scale_and_equal_weights():
    dem, weight, offset, w, b, gout = _rand_case(2, 24, 72, 2.0, seed=3)
    eq = torch.full_like(weight, 0.37)
    got = _run_hip(dem, eq, offset, w, b, gout, scale=0.5)
    _close(got[0], 0.5 * dem + b, 1e-6, 1e-6, "equal weights -> b + scale*dem")


def test_cpu_tensor_is_rejected():
    ops = _ops()
    dem, weight, offset, w, b, _ = _rand_case(1, 8, 8, 1.0, seed=1)
    with pytest.raises(RuntimeError):
        ops.propagate(dem, weight, offset, w, b)


def test_bad_shapes_raise():
    ops = _ops()
    dem, weight, offset, w, b, _ = (t.cuda() for t in _rand_case(1, 8, 8, 1.0, seed=1))
    with pytest.raises(ValueError):
        ops.propagate(dem, weight[:, :8], offset, w, b)
    with pytest.raises(ValueError):
        ops.propagate(dem, weight, offset[:, :17], w, b)


def test_full_size_properties():
    """BASELINE-size check (8 x 512 x 512) through size-independent properties: linearity in the
    affinities' tap weights, zero-sum of grad_weight over taps, grad_b = sum(grad_out)."""
    ops = _ops()
    B, H, W = 8, 512, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    dem = torch.rand(B, 1, H, W, device="cuda", generator=g)
    weight = torch.rand(B, 9, H, W, device="cuda", generator=g).requires_grad_()
    offset = (1.5 * torch.randn(B, 16, H, W, device="cuda", generator=g)).requires_grad_()
    w = torch.randn(1, 1, 3, 3, device="cuda", generator=g).requires_grad_()
    b = torch.zeros(1, device="cuda").requires_grad_()
    gout = torch.randn(B, 1, H, W, device="cuda", generator=g)
    o1 = ops.propagate(dem, weight, offset, w, b)
    o2 = ops.propagate(dem, weight, offset, 2 * w, b)
    assert torch.allclose((o2 - dem) , 2 * (o1 - dem), rtol=1e-4, atol=1e-5)
    o1.backward(gout)
    assert weight.grad.sum(1).abs().max().item() < 1e-4
    assert abs(b.grad.item() - gout.double().sum().item()) < 1e-2
    # random sub-block against the oracle
    sl = (slice(2, 3), slice(None), slice(100, 164), slice(200, 328))
    off18 = torch.cat((offset[:, :8], torch.zeros_like(offset[:, :2]), offset[:, 8:]), 1)
    ref = R.propagate(dem.cpu().double(), weight.detach().cpu().double(), off18.detach().cpu().double(),
                      w.detach().cpu().double(), b.detach().cpu().double())
    _close(o1.detach().cpu()[sl], ref[sl], 1e-5, 3e-5, "sub-block")  # fp32 coordinates at |p| ~ 500 px


def test_backward_split_into_stream_and_fold():
    """jspsr_prop_backward_f32 with grad_wk = grad_b0 = NULL launches only the streaming kernel; the fold entry
    point finishes the job from the workspace -- bit-identical to the single call."""
    from jspsr_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 40, 72
    dem = torch.rand(B, 1, H, W, generator=g).cuda()
    wt = torch.sigmoid(torch.randn(B, 9, H, W, generator=g)).cuda()
    off = (1.5 * torch.randn(B, 16, H, W, generator=g)).cuda()
    wk = torch.randn(1, 1, 3, 3, generator=g).cuda()
    go = torch.randn(B, 1, H, W, generator=g).cuda()
    outs = []
    for split in (False, True):
        gw_, go_ = torch.empty_like(wt), torch.empty_like(off)
        gk, gb = torch.full_like(wk, 7.0), torch.full((1,), 7.0, device="cuda")
        ws = ops.prop_backward_workspace(B, H, W, "cuda")
        if split:
            ops.prop_backward_raw(go, dem, wt, off, wk, gw_, go_, None, None, ws)
            assert (gk == 7).all() and (gb == 7).all()          # untouched until the fold
            ops.prop_backward_fold_raw(ws, B, H, W, gk, gb)
        else:
            ops.prop_backward_raw(go, dem, wt, off, wk, gw_, go_, gk, gb, ws)
        outs.append((gw_, go_, gk, gb))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
