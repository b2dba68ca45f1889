"""NLSPN (models/components/nlspn.py): the N-iteration, fixed-affinity user of the propagation kernel.
CPU: the oracle (oracle/nlspn_ref.py) against fixtures made by the reference's own NLSPN class (the pin).
GPU: the product module (jspsr_amd/nlspn.py: MFMA guidance conv + HIP step kernels incl. grad wrt the raster)."""
import types

import numpy as np
import pytest
import torch

from oracle import nlspn_ref as NR
from tests import fixtures as Fx

CASES = ["g8_nlspn_tgass_conf_fix.npz", "g8_nlspn_as_plain.npz"]


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _oracle(z):
    feat = _t(z["feat"]).requires_grad_()
    conf = _t(z["confidence"]).requires_grad_()
    w, b = _t(z["conv_w"]).requires_grad_(), _t(z["conv_b"]).requires_grad_()
    sc = _t(z["scale_const"]).requires_grad_()
    raw = torch.nn.functional.conv2d(_t(z["guidance"]), w, b, 1, 1)
    offset, aff = NR.offset_affinity(raw, str(z["affinity"]), sc, conf, bool(z["conf_prop"]), bool(z["legacy"]))
    steps = NR.propagate(feat, offset, aff, int(z["prop_time"]), _t(z["feat_fix"]) if bool(z["preserve_input"]) else None)
    probes = _t(z["probes"])
    sum((f * probes[:, i:i + 1]).mean() for i, f in enumerate(steps)).backward()
    return offset, aff, steps, feat, conf, w, b, sc


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_nlspn(golden_dir, name):
    z = Fx.load(golden_dir, name)
    offset, aff, steps, feat, conf, w, b, sc = _oracle(z)
    assert torch.allclose(offset.detach(), _t(z["offset"]), rtol=0, atol=1e-12)
    assert torch.allclose(aff.detach(), _t(z["aff"]), rtol=0, atol=1e-12)
    assert torch.allclose(torch.cat(steps, 1).detach(), _t(z["steps"]), rtol=0, atol=1e-11)
    assert torch.allclose(feat.grad, _t(z["grad_feat"]), rtol=1e-8, atol=1e-13)
    # gradients that pass through d/d(offset): the grid_sample stand-in and the gather agree except on the measure-zero
    # set of exactly-integer sampling positions (none here: offsets are continuous random numbers)
    assert Fx.rel(w.grad, z["grad_conv_w"]) < 1e-8 and Fx.rel(b.grad, z["grad_conv_b"]) < 1e-8
    if bool(z["conf_prop"]):
        assert Fx.rel(conf.grad, z["grad_conf"]) < 1e-8
    if str(z["affinity"]) == "TGASS":
        assert Fx.rel(sc.grad, z["grad_scale_const"]) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_product_nlspn_matches_reference(golden_dir, name):
    """Forward: every step's raster within 2e-5 of the fp64 reference (6 chained fp32 steps); offsets / affinities
    within 1e-5.  Backward: relative L2 of each gradient tensor within 2e-4 -- no ReLU in this path, so there is no
    mask-flip floor; d/d(offset) is compared through the guidance convolution's weight gradient."""
    from jspsr_amd.nlspn import NLSPN
    z = Fx.load(golden_dir, name)
    args = types.SimpleNamespace(prop_time=int(z["prop_time"]), affinity=str(z["affinity"]), affinity_gamma=0.5,
                                 conf_prop=bool(z["conf_prop"]), preserve_input=bool(z["preserve_input"]), legacy=bool(z["legacy"]))
    m = NLSPN(args, z["guidance"].shape[1], 1, 3, 3)
    with torch.no_grad():
        m.conv_offset_aff.weight.copy_(_t(z["conv_w"]).float())
        m.conv_offset_aff.bias.copy_(_t(z["conv_b"]).float())
        m.aff_scale_const.copy_(_t(z["scale_const"]).float())
    m = m.cuda()
    feat = _t(z["feat"]).float().cuda().requires_grad_()
    conf = _t(z["confidence"]).float().cuda().requires_grad_()
    res, lst, offset, aff, sc = m(feat, _t(z["guidance"]).float().cuda(), conf if args.conf_prop else None,
                                  _t(z["feat_fix"]).float().cuda() if args.preserve_input else None)
    assert len(lst) == args.prop_time and res is lst[-1]
    assert (offset.detach().cpu().double() - _t(z["offset"])).abs().max().item() < 1e-5
    assert (aff.detach().cpu().double() - _t(z["aff"])).abs().max().item() < 1e-5
    steps = torch.cat(lst, 1).detach().cpu().double()
    assert (steps - _t(z["steps"])).abs().max().item() < 2e-5
    probes = _t(z["probes"]).float().cuda()
    sum((f * probes[:, i:i + 1]).mean() for i, f in enumerate(lst)).backward()
    assert Fx.rel(feat.grad, z["grad_feat"]) < 2e-4
    assert Fx.rel(m.conv_offset_aff.weight.grad, z["grad_conv_w"]) < 2e-4
    assert Fx.rel(m.conv_offset_aff.bias.grad, z["grad_conv_b"]) < 2e-4
    if args.conf_prop:
        assert Fx.rel(conf.grad, z["grad_conf"]) < 2e-4
    if args.affinity == "TGASS":
        assert Fx.rel(m.aff_scale_const.grad, z["grad_scale_const"]) < 2e-4


@pytest.mark.gpu
def test_step_kernel_grad_dem_is_the_transpose_of_the_gather():
    """<g, J v> == <J^T g, v> for the linear map dem -> out of one step (random offsets incl. far-out taps, both
    normalisation modes, residual scale): the scatter in the backward is the exact adjoint of the forward gather."""
    from jspsr_amd import ops
    g_ = torch.Generator().manual_seed(3)
    B, H, W = 2, 37, 150
    aff = torch.rand(B, 9, H, W, generator=g_).cuda()
    off = (3.0 * torch.randn(B, 18, H, W, generator=g_))
    off[0, :, :5, :9] *= 15.0
    off = off.cuda()
    wk = (1 + 0.3 * torch.randn(9, generator=g_)).cuda()
    b0 = torch.zeros(1).cuda()
    v = torch.randn(B, 1, H, W, generator=g_).cuda()
    g = torch.randn(B, 1, H, W, generator=g_).cuda()
    ws = ops._step_workspace(B, H, W, "cuda")
    for normalize in (0, 1):
        for scale in (0.0, 0.7):
            jv = ops._step_forward(v, aff, off, wk, b0, scale, normalize, torch.empty_like(v))
            gd = torch.zeros_like(v)
            ops._step_backward(g, v, aff, off, wk, scale, normalize, 0, torch.empty_like(aff), torch.empty_like(off), gd, ws)
            lhs, rhs = (g.double() * jv.double()).sum().item(), (gd.double() * v.double()).sum().item()
            assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), 1.0), (normalize, scale, lhs, rhs)


@pytest.mark.gpu
def test_step_entry_equals_postprocessor_entry():
    """normalize = 1 without grad_dem is the PostProcessor operator: same numbers as jspsr_prop_forward/backward_f32."""
    from jspsr_amd import ops
    g_ = torch.Generator().manual_seed(4)
    B, H, W = 2, 40, 96
    dem = torch.rand(B, 1, H, W, generator=g_).cuda()
    wt = torch.sigmoid(torch.randn(B, 9, H, W, generator=g_)).cuda()
    off = (2.0 * torch.randn(B, 16, H, W, generator=g_)).cuda()
    wk = (1 + 0.3 * torch.randn(1, 1, 3, 3, generator=g_)).cuda()
    b0 = torch.full((1,), 0.05).cuda()
    gout = torch.randn(B, 1, H, W, generator=g_).cuda()
    out_a = torch.empty_like(dem)
    ops.prop_forward_raw(dem, wt, off, wk, b0, 1.0, out_a)
    out_b = ops._step_forward(dem, wt, off, wk, b0, 1.0, 1, torch.empty_like(dem))
    assert (out_a - out_b).abs().max().item() < 1e-6
    gw_a, go_a = torch.empty_like(wt), torch.empty_like(off)
    gwk, gb = torch.empty(9, device="cuda"), torch.empty(1, device="cuda")
    ops.prop_backward_raw(gout, dem, wt, off, wk, gw_a, go_a, gwk, gb, ops.prop_backward_workspace(B, H, W, "cuda"))
    gw_b, go_b = torch.empty_like(wt), torch.empty_like(off)
    ops._step_backward(gout, dem, wt, off, wk, 1.0, 1, 0, gw_b, go_b, None, ops._step_workspace(B, H, W, "cuda"))
    assert (gw_a - gw_b).abs().max().item() < 1e-6 and (go_a - go_b).abs().max().item() < 1e-5


@pytest.mark.gpu
def test_step_kernel_grad_dem_is_bit_reproducible_without_far_taps():
    """Round 4: the raster gradient is accumulated in 64-bit fixed point in the tile's LDS window (integer adds: exact,
    order-independent) and gathered by a pixel-ordered second pass (no float atomics), so it is bit-identical from run
    to run whenever no tap leaves tile + halo (|offset| < 7 px here); it is ADDED into what the buffer holds; and it still
    equals the fp64 adjoint of the gather (autograd through the oracle's sampler)."""
    from jspsr_amd import ops
    from oracle import jspsr_ref as R
    g_ = torch.Generator().manual_seed(11)
    B, H, W = 3, 45, 200                      # ragged both ways: 6 x 4 tiles per image, last ones partly outside the raster
    aff = torch.rand(B, 9, H, W, generator=g_)
    off = (2.0 * torch.randn(B, 18, H, W, generator=g_)).clamp(-6.5, 6.5)
    off[:, 8:10] = 0
    wk = 1 + 0.3 * torch.randn(9, generator=g_)
    v = torch.randn(B, 1, H, W, generator=g_)
    g = torch.randn(B, 1, H, W, generator=g_)
    ws = ops._step_workspace(B, H, W, "cuda")
    runs = []
    for _ in range(3):
        gd = torch.full_like(v, 0.25).cuda()          # the gradient is added into the buffer's content
        ops._step_backward(g.cuda(), v.cuda(), aff.cuda(), off.cuda(), wk.cuda(), 0.7, 0, 0, torch.empty_like(aff).cuda(),
                           torch.empty_like(off).cuda(), gd, ws)
        runs.append(gd.cpu())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    vd = v.double().requires_grad_()
    S = R.sample_taps(vd, off.double())
    out = (wk.double().view(1, 9, 1, 1) * aff.double() * S).sum(1, keepdim=True) + 0.7 * vd      # normalize = 0: affinities as they are
    out.backward(g.double())
    assert (runs[0].double() - 0.25 - vd.grad).abs().max().item() < 1e-4       # fp32 coordinates and products; |gradient| up to ~10
