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
    ((2, 6, 132), 1.5, 16),      # LDS-DMA path: H % 4 != 0 (a tile row with idle waves), W = 2 tiles + 4 pixels
    ((1, 3, 64), 2.0, 18),       # LDS-DMA path: fewer rows than a tile, 18-channel offsets
    ((5, 20, 72), 1.5, 16),      # LDS-DMA path: a workgroup's run crosses images
    ((1, 260, 1028), 1.5, 16),   # LDS-DMA path: more tiles than workgroups (persistent runs of several tiles, ragged both ways)
])
def test_random_cases(shape, sigma, oc):
    case = _rand_case(*shape, sigma, seed=sum(shape) + int(sigma * 10))
    got = _run_hip(*case, oc=oc)
    exp = list(_run_oracle64(*case))
    if oc == 16:
        exp[2] = exp[2].clone()
        exp[2][:, 8:10] = 0
    # fp32 coordinates: ulp(p) ~ 4e-6 at |p| ~ 50 px times the white-noise DEM's unit slope
    # (|p| ~ 1000 px: ulp(p) = 6e-5, times |grad_out| up to 4.5 and a tap weight up to 1.6 in the gradients)
    tol = (2e-5 if sigma <= 3 else 1e-4) if max(shape) <= 300 else 4e-4
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


def _census(names):
    from jspsr_amd import _lib
    lib = _lib.load()
    return {n: lib.jspsr_launch_count(n.encode()) for n in names}


def test_dma_and_general_kernels_are_selected_by_shape():
    """Rows of 16-byte-aligned operands with W % 4 == 0 take the persistent LDS-DMA kernels (prop_dma.hip, prop_head_dma.hip
    for bf16 heads); anything else the general kernels -- asserted from the library's launch census, so the parity cases
    above are known to cover both."""
    ops = _ops()
    names = ("prop_forward (dma)", "prop_backward (dma)", "prop_forward", "prop_backward",
             "prop_head_forward (dma)", "prop_head_backward (dma)", "prop_head_forward", "prop_head_backward")
    for shape, dma in (((2, 40, 100), True), ((2, 37, 53), False)):
        case = _rand_case(*shape, 1.5, seed=11)
        before = _census(names)
        _run_hip(*case, oc=16)
        d = {k: v - before[k] for k, v in _census(names).items()}
        assert (d["prop_forward (dma)"], d["prop_backward (dma)"], d["prop_forward"], d["prop_backward"]) == ((1, 1, 0, 0) if dma else (0, 0, 1, 1)), d
        B, H, W = shape
        for dtype, hd in ((torch.bfloat16, dma), (torch.float32, False)):
            head = torch.randn(B, H, W, 32).to(dtype).cuda().requires_grad_()
            before = _census(names)
            ops.propagate_head(case[0].cuda(), head, case[3].cuda(), case[4].cuda(), 1.0).sum().backward()
            d = {k: v - before[k] for k, v in _census(names).items()}
            assert (d["prop_head_forward (dma)"], d["prop_head_backward (dma)"], d["prop_head_forward"], d["prop_head_backward"]) == \
                ((1, 1, 0, 0) if hd else (0, 0, 1, 1)), (dtype, d)


@pytest.mark.parametrize("env", [{"JSPSR_PROP_SPLIT": "0", "JSPSR_PROP_HEAD_SPLIT": "0"}, {"JSPSR_PROP_SPLIT": "1", "JSPSR_PROP_NW": "8"},
                                 {"JSPSR_PROP_NTL": "0", "JSPSR_PROP_WGS": "1", "JSPSR_PROP_HEAD_WGS": "1"},
                                 {"JSPSR_PROP_DMA": "0", "JSPSR_PROP_HEAD_DMA": "0"},
                                 {"JSPSR_PROP_SPLIT": "0", "JSPSR_PROP_RP": "2"}, {"JSPSR_PROP_SPLIT": "0", "JSPSR_PROP_RP": "4"}])
def test_other_forms_of_the_dma_kernels_in_a_child_process(env):
    """The library reads its switches once per process.  The defaults run the forward with mover / compute waves and the
    backward with symmetric waves (NW = 4, non-temporal loads); the other instantiations -- both directions in either
    form, 8-row tiles, default-policy loads, one workgroup per CU, two / four rows per wave under one staged DEM tile, and the
    general kernels on DMA-eligible shapes -- must
    stay correct too: same fp64 oracle, same tolerances, in a child process per setting."""
    import subprocess
    import sys
    code = r"""
import torch
from oracle import jspsr_ref as R
from tests.test_prop_gpu import _rand_case, _run_hip, _run_oracle64, _close, _head_case
from jspsr_amd import ops
for shape, oc in (((2, 40, 100), 16), ((1, 6, 132), 18), ((3, 64, 192), 16)):
    case = _rand_case(*shape, 1.5, seed=sum(shape))
    got, exp = _run_hip(*case, oc=oc), list(_run_oracle64(*case))
    if oc == 16:
        exp[2] = exp[2].clone(); exp[2][:, 8:10] = 0
    _close(got[0], exp[0], 1e-5, 2e-5, "out"); _close(got[1], exp[1], 1e-5, 2e-5, "grad_weight")
    off = case[2].double(); B, _, H, W = off.shape
    pos = off.clone(); pos[:, 0::2] += torch.arange(H, dtype=torch.float64).view(1, 1, H, 1); pos[:, 1::2] += torch.arange(W, dtype=torch.float64).view(1, 1, 1, W)
    smooth = ((pos - pos.round()).abs().reshape(B, 9, 2, H, W).amin(2, keepdim=True) > 1e-4).expand(B, 9, 2, H, W).reshape(B, 18, H, W)
    _close(got[2] * smooth, exp[2] * smooth, 1e-4, 4e-5, "grad_offset")
    _close(got[3], exp[3], 1e-4, 5e-4, "grad_w"); _close(got[4], exp[4], 1e-5, 5e-4, "grad_b")
dem, head, weight, offset, w, b, gout = _head_case(2, 24, 136, torch.bfloat16, seed=5)
hd = head.cuda().requires_grad_()
out = ops.propagate_head(dem.float().cuda(), hd, w.float().cuda(), b.float().cuda(), 1.0)
out.backward(gout.float().cuda())
h64 = head.double().requires_grad_()
lg, o16 = ops.split_head(h64)
o16 = o16.permute(0, 3, 1, 2)
ref = R.propagate(dem, torch.sigmoid(lg).permute(0, 3, 1, 2), torch.cat((o16[:, :8], torch.zeros(2, 2, 24, 136, dtype=torch.float64), o16[:, 8:]), 1), w, b)
ref.backward(gout)
assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 5e-6
h5 = head.double().view(2, 24, 136, 8, 4)
sm = (h5[..., 1:3] != h5[..., 1:3].round()).all(-1, keepdim=True).expand(2, 24, 136, 8, 4).reshape(2, 24, 136, 32).double()
assert (((hd.grad.cpu().double() - h64.grad) * sm).norm() / h64.grad.norm()).item() < 2.0 ** -8
print("forms ok")
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "forms ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


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


# ---- K1h: the head-fed entry (jspsr_prop_head_forward / _backward) -------------------------------------------------
def _head_case(B, H, W, dtype, seed, sigma=2.0):
    """Random tap-major head tensor (stored in `dtype`) + the planar operands the reference would see for it."""
    from jspsr_amd import ops
    g = torch.Generator().manual_seed(seed)
    dem = torch.rand(B, 1, H, W, generator=g, dtype=torch.float64)
    head = torch.randn(B, H, W, 32, generator=g, dtype=torch.float64)
    h5 = head.view(B, H, W, 8, 4)
    h5[..., 1:3] *= sigma
    h5[0, : min(4, H), : min(6, W), :, 1:3] = torch.randint(-3, 4, (min(4, H), min(6, W), 8, 2), generator=g).double()   # integer taps
    h5[0, min(4, H - 1):min(7, H), : min(6, W), :, 1:3] *= 20.0                                                       # far out of the tile / raster
    head = head.to(dtype)                       # what the head convolution would have stored
    logits, off16 = ops.split_head(head.double())
    weight = torch.sigmoid(logits).permute(0, 3, 1, 2).contiguous()
    off16 = off16.permute(0, 3, 1, 2).contiguous()
    zero = torch.zeros(B, 2, H, W, dtype=torch.float64)
    offset = torch.cat((off16[:, :8], zero, off16[:, 8:]), 1)
    w = 1 + 0.3 * torch.randn(1, 1, 3, 3, generator=g, dtype=torch.float64)
    b = 0.1 * torch.randn(1, generator=g, dtype=torch.float64)
    gout = torch.randn(B, 1, H, W, generator=g, dtype=torch.float64)
    return dem, head, weight, offset, w, b, gout


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W", [(2, 24, 40), (1, 13, 70), (3, 8, 64), (1, 1, 1), (2, 40, 129), (2, 6, 132), (1, 70, 200)])
def test_head_entry_matches_oracle(B, H, W, dtype):
    """Forward and backward of the head-fed kernel against the fp64 oracle evaluated on the SAME stored head values
    (sigmoid and zero centre offset applied by the oracle as the reference's Generator does, spn.py:43,69-73).  Ragged
    sizes (W % 8 != 0, H < tile), integer taps and far-out taps included.  fp32: 5e-6 abs on the output, 2e-5 relative
    on the gradients; bf16: same bound on the output (the arithmetic is fp32 either way), gradient tensor compared
    after ITS storage rounding (2^-8 relative per element)."""
    from jspsr_amd import ops
    dem, head, weight, offset, w, b, gout = _head_case(B, H, W, dtype, seed=B * 100 + H + W)
    head_d = head.cuda().requires_grad_()
    w_d, b_d = w.float().cuda().requires_grad_(), b.float().cuda().requires_grad_()
    out = ops.propagate_head(dem.float().cuda(), head_d, w_d, b_d, 1.0)
    out.backward(gout.float().cuda())
    # oracle: autograd through sigmoid / split on the stored values, in fp64
    head64 = head.double().requires_grad_()
    logits, off16 = ops.split_head(head64)
    wt = torch.sigmoid(logits).permute(0, 3, 1, 2)
    o16 = off16.permute(0, 3, 1, 2)
    off18 = torch.cat((o16[:, :8], torch.zeros(B, 2, H, W, dtype=torch.float64), o16[:, 8:]), 1)
    w64, b64 = w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = R.propagate(dem, wt, off18, w64, b64)
    ref.backward(gout)
    # the float32 operand of the kernel is dem.float(): evaluate the tolerance against the fp64 oracle on fp64 dem
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 5e-6
    gh, gr = head_d.grad.cpu().double(), head64.grad
    # d/d(offset) jumps at integer sampling positions and where a tap crosses the raster border: compare off that set
    h5 = head.double().view(B, H, W, 8, 4)
    # (within 1e-4 of an integer: the fp32 coordinate may land on the other side of the kink)
    smooth = ((h5[..., 1:3] - h5[..., 1:3].round()).abs() > 1e-4).all(-1, keepdim=True).expand(B, H, W, 8, 4).reshape(B, H, W, 32).double()
    if dtype == torch.float32:
        assert ((gh - gr) * smooth).abs().max().item() < 2e-5 * gr.abs().max().item() + 1e-7
    else:
        assert ((gh - gr) * smooth).abs().max().item() < 2.0 ** -8 * gr.abs().max().item() + 1e-7
        assert (((gh - gr) * smooth).norm() / gr.norm()).item() < 2.0 ** -8
    pad = gh.view(B, H, W, 8, 4)[..., 1:, 3]
    assert pad.abs().max().item() == 0.0                               # unused channels carry exact zeros
    assert abs(w_d.grad.cpu().double() - w64.grad).max().item() < 1e-5 * w64.grad.abs().max().item() + 1e-6
    assert abs(b_d.grad.cpu().double() - b64.grad).max().item() < 1e-5 * abs(b64.grad).max().item() + 1e-6


def test_head_entry_equals_planar_entry_on_the_same_numbers():
    """The two ABI entries are the same operator on two layouts: fp32 head -> (sigmoid, split) -> planar kernel."""
    from jspsr_amd import ops
    dem, head, weight, offset, w, b, gout = _head_case(2, 32, 96, torch.float32, seed=77)
    out_h = ops.propagate_head(dem.float().cuda(), head.cuda(), w.float().cuda(), b.float().cuda(), 1.0)
    out_p = ops.propagate(dem.float().cuda(), weight.float().cuda(), offset.float().cuda(), w.float().cuda(), b.float().cuda(), 1.0)
    assert (out_h - out_p).abs().max().item() < 2e-6


def test_head_entry_full_size_properties():
    """BASELINE size (8 x 512 x 512), both dtypes: equal affinity logits => out = b + dem exactly up to rounding
    (zero-sum affinities); linearity in the upstream gradient; finite everywhere."""
    from jspsr_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    B, H, W = 8, 512, 512
    dem = torch.rand(B, 1, H, W, device="cuda", generator=g)
    w = (1 + 0.2 * torch.randn(1, 1, 3, 3, device="cuda", generator=g))
    b = torch.full((1,), 0.125, device="cuda")
    for dtype in (torch.float32, torch.bfloat16):
        head = torch.randn(B, H, W, 32, device="cuda", generator=g)
        h5 = head.view(B, H, W, 8, 4)
        h5[..., 0] = 0.3
        h5[..., 0, 3] = 0.3                      # all nine logits equal
        w1 = torch.ones(1, 1, 3, 3, device="cuda")
        out = ops.propagate_head(dem, head.to(dtype), w1, b, 1.0)
        assert (out - (dem + 0.125)).abs().max().item() < 2e-6
        head2 = (1.5 * torch.randn(B, H, W, 32, device="cuda", generator=g)).to(dtype).requires_grad_()
        o = ops.propagate_head(dem, head2, w, b, 1.0)
        g1 = torch.randn(B, 1, H, W, device="cuda", generator=g)
        (ga,) = torch.autograd.grad(o, head2, g1, retain_graph=True)
        (gb,) = torch.autograd.grad(o, head2, 2 * g1)
        assert torch.isfinite(o).all() and torch.isfinite(ga).all()
        tol = 1e-6 if dtype == torch.float32 else 2.0 ** -7
        assert ((gb.float() - 2 * ga.float()).abs().max() <= tol * ga.float().abs().max()).item()
