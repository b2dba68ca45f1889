"""CPU: the oracle (oracle/jspsr_ref.py, formulation A) against the fixtures produced by the
reference's own modules with the grid_sample stand-in (formulation B) -- oracle/gen_golden.py."""
import os

import numpy as np
import pytest
import torch

from oracle import jspsr_ref as R
from tests import fixtures as Fx


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _smooth(z):
    """1 where the sampling coordinate is not an exact integer.  At exact integers the bilinear
    sampler has a kink: torchvision (and the oracle / HIP kernel) take the floor-side one-sided
    derivative, while the stand-in's normalise/unnormalise round trip lands on either side, so the
    fixture's grad_offset is only defined off that measure-zero set."""
    off = _t(z["offset"])
    return (off != off.round()).to(off.dtype)


def test_integer_positions_follow_torchvision_rule(golden_dir):
    """On the kink itself autograd-of-the-gather and the closed form must still agree with each
    other wherever the sample is inside the raster (floor-side derivative, get_coordinate_weight)."""
    z = _load(golden_dir, "g1_postprocessor.npz")
    dem, weight, offset = _t(z["dem"]), _t(z["weight"]), _t(z["offset"]).requires_grad_()
    w, b, g = _t(z["w"]), _t(z["b"]), _t(z["grad_out"])
    R.propagate(dem, weight, offset, w, b).backward(g)
    go = R.propagate_analytic_backward(dem, weight, offset.detach(), w, b, g)[1]
    B, _, H, W = dem.shape
    off = offset.detach().reshape(B, 9, 2, H, W)
    ys = torch.arange(H, dtype=off.dtype).view(1, 1, H, 1)
    xs = torch.arange(W, dtype=off.dtype).view(1, 1, 1, W)
    ky = torch.tensor([k // 3 - 1 for k in range(9)], dtype=off.dtype).view(1, 9, 1, 1)
    kx = torch.tensor([k % 3 - 1 for k in range(9)], dtype=off.dtype).view(1, 9, 1, 1)
    py, px = ys + ky + off[:, :, 0], xs + kx + off[:, :, 1]
    inside = ((py > -1) & (py < H) & (px > -1) & (px < W)).unsqueeze(2).expand(B, 9, 2, H, W)
    d = (offset.grad - go).abs().reshape(B, 9, 2, H, W)
    assert d[inside].max() < 1e-13
    assert (~inside).any()


def test_postprocessor_forward_backward_fp64(golden_dir):
    z = _load(golden_dir, "g1_postprocessor.npz")
    dem, weight, offset = _t(z["dem"]), _t(z["weight"]).requires_grad_(), _t(z["offset"]).requires_grad_()
    w, b = _t(z["w"]).requires_grad_(), _t(z["b"]).requires_grad_()
    out = R.propagate(dem, weight, offset, w, b, 1.0)
    out.backward(_t(z["grad_out"]))
    assert torch.allclose(out.detach(), _t(z["out"]), rtol=0, atol=1e-12)
    assert torch.allclose(weight.grad, _t(z["grad_weight"]), rtol=0, atol=1e-12)
    assert ((offset.grad - _t(z["grad_offset"])).abs() * _smooth(z)).max() < 1e-11
    assert torch.allclose(w.grad, _t(z["grad_w"]), rtol=1e-12, atol=1e-11)
    assert torch.allclose(b.grad, _t(z["grad_b"]), rtol=1e-12, atol=1e-11)


def test_postprocessor_analytic_backward_matches_reference(golden_dir):
    z = _load(golden_dir, "g1_postprocessor.npz")
    gw, go, gW, gb = R.propagate_analytic_backward(
        _t(z["dem"]), _t(z["weight"]), _t(z["offset"]), _t(z["w"]), _t(z["b"]), _t(z["grad_out"]))
    assert torch.allclose(gw, _t(z["grad_weight"]), rtol=0, atol=1e-12)
    assert ((go - _t(z["grad_offset"])).abs() * _smooth(z)).max() < 1e-11
    assert torch.allclose(gW, _t(z["grad_w"]), rtol=1e-12, atol=1e-11)
    assert torch.allclose(gb, _t(z["grad_b"]), rtol=1e-12, atol=1e-11)


def test_known_answers():
    """Identities of SURVEY.md section 8c."""
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 9, 11
    dem = torch.rand(B, 1, H, W, generator=g, dtype=torch.float64)
    weight = torch.rand(B, 9, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(1, 1, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(1, generator=g, dtype=torch.float64)
    zero = torch.zeros(B, 18, H, W, dtype=torch.float64)
    # zero offsets == plain 3x3 unfold
    m = weight - weight.mean(1, keepdim=True)
    unf = torch.nn.functional.unfold(dem, 3, padding=1).reshape(B, 9, H, W)
    exp = (w.reshape(1, 9, 1, 1) * m * unf).sum(1, keepdim=True) + b + dem
    assert torch.allclose(R.propagate(dem, weight, zero, w, b), exp, atol=1e-14)
    # equal weights -> zero-sum affinities -> out = b + dem
    eq = torch.full_like(weight, 0.37)
    off = torch.randn(B, 18, H, W, generator=g, dtype=torch.float64) * 3
    assert torch.allclose(R.propagate(dem, eq, off, w, b), dem + b, atol=1e-14)
    # a tap pushed beyond the raster contributes 0
    far = zero.clone()
    far[:, 0] = -100.0
    S = R.sample_taps(dem, far)
    assert S[:, 0].abs().max() == 0
    # integer offsets == shifted gather
    io = zero.clone()
    io[:, 2 * 5] = 1.0  # tap 5 (dy 0, dx +1): +1 row
    io[:, 2 * 5 + 1] = -2.0  # -2 cols -> samples dem[y+1, x-1]
    S = R.sample_taps(dem, io)
    shifted = torch.zeros_like(dem)
    shifted[:, :, : H - 1, 1:] = dem[:, :, 1:, : W - 1]
    assert torch.allclose(S[:, 5:6], shifted, atol=0)


MODEL_CASES = [
    ("g3_img_nf32_64_train.npz", {"lr_dem": 1, "image": 3}),
    ("g3_img_nf32_64_eval.npz", {"lr_dem": 1, "image": 3}),
    ("g3_img_nf8_b2_48x80_train.npz", {"lr_dem": 1, "image": 3}),
    ("g4_msk_nf8_b2_64_train.npz", {"lr_dem": 1, "image": 3, "mask": 15}),
    ("g4_msk_nf8_b2_64_eval.npz", {"lr_dem": 1, "image": 3, "mask": 15}),
    ("g4_msk_nf32_b1_64_train.npz", {"lr_dem": 1, "image": 3, "mask": 15}),   # the benched architecture
]


def regen(z, in_channels, dtype):
    sd, inputs, gt = Fx.regen_jspsr(z, in_channels)      # fails (never skips) if the fixture does not regenerate
    cast = lambda v: v.to(dtype) if v.is_floating_point() else v
    return {k: cast(v) for k, v in sd.items()}, [cast(t) for t in inputs], cast(gt)


@pytest.mark.parametrize("name,in_channels", MODEL_CASES)
def test_model_fp64(golden_dir, name, in_channels):
    z = _load(golden_dir, name)
    sd, inputs, gt = regen(z, in_channels, torch.float64)
    training = bool(z["training"])
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    sd.update(params)
    pred = R.jspsr_forward(sd, inputs, training)
    assert torch.allclose(pred.detach(), _t(z["pred"]), rtol=0, atol=1e-11)
    if not training:
        return
    loss = (pred - gt).abs().mean() + ((pred - gt) ** 2).mean()
    assert abs(loss.item() - float(z["loss"])) < 1e-12
    (pred * R.probe_gradient(pred.shape, int(z["seed"]) + 2)).mean().backward()   # fixed linear probe
    for k, n in zip(z["grad_names"], z["grad_norms"]):
        got = params[str(k)].grad.norm().item()
        assert abs(got - n) <= 1e-8 * max(n, 1e-30) + 1e-13, (k, got, n)
    for k in z.files:
        if k.startswith("grad:"):
            assert torch.allclose(params[k[5:]].grad, _t(z[k]), rtol=1e-8, atol=1e-12), k
        if k.startswith("buf:"):
            assert torch.allclose(sd[k[4:]], _t(z[k]), rtol=0, atol=1e-12), k


def test_config1_fp32_matches_reference_fp32(golden_dir):
    """BASELINE config 1: fp32 restatement vs fp32 reference modules on the CPU.  Both are fp32 evaluations of the
    same formulae in different operation orders; the yardstick is the reference's OWN fp32 rounding error against its
    fp64 run (stored in the fixture, 1e-6 .. 6e-6 here): the restatement must sit within twice that of the fp32
    reference and within 2e-5 of the fp64 one."""
    for name in ("g3_img_nf32_64_eval.npz", "g3_img_nf32_64_train.npz"):
        z = _load(golden_dir, name)
        sd, inputs, _ = regen(z, {"lr_dem": 1, "image": 3}, torch.float32)
        with torch.no_grad():
            pred = R.jspsr_forward(sd, inputs, bool(z["training"]))
        ref_rounding = (_t(z["pred_fp32"]).double() - _t(z["pred"])).abs().max().item()
        assert 1e-7 < ref_rounding < 1e-5
        assert (pred - _t(z["pred_fp32"])).abs().max().item() < 2 * ref_rounding
        assert (pred.double() - _t(z["pred"])).abs().max().item() < 2e-5


def test_wrong_arity_raises():
    sd = R.make_state_dict(R.jspsr_param_shapes({"lr_dem": 1, "image": 3}, 8), 1)
    x = torch.zeros(1, 1, 16, 16)
    with pytest.raises(NotImplementedError):
        R.jspsr_forward(sd, [x], False)


def _regen_simple(z, shapes, with_mask=False):
    return Fx.regen(z, shapes, with_mask)


def _check_grads(z, params):
    for k, n in zip(z["grad_names"], z["grad_norms"]):
        got = params[str(k)].grad.norm().item()
        assert abs(got - n) <= 1e-8 * max(n, 1e-30) + 1e-13, (k, got, n)
    for k in z.files:
        if k.startswith("grad:"):
            assert torch.allclose(params[k[5:]].grad, _t(z[k]), rtol=1e-8, atol=1e-12), k


@pytest.mark.parametrize("name", ["g5_lrru_b1_64_train.npz", "g5_lrru_b2_32x48_eval.npz"])
def test_lrru_fp64(golden_dir, name):
    """models.LRRU.Model (4 propagation steps, LRRU.py:403-507) restated."""
    z = _load(golden_dir, name)
    sd, inputs, gt = _regen_simple(z, R.lrru_param_shapes(16))
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    sd.update(params)
    pred = R.lrru_forward(sd, inputs, bool(z["training"]))
    assert torch.allclose(pred.detach(), _t(z["pred"]), rtol=0, atol=1e-11)
    if bool(z["training"]):
        (pred * R.probe_gradient(pred.shape, int(z["seed"]) + 2)).mean().backward()
        _check_grads(z, params)


def test_edsr_fp64(golden_dir):
    """models.EDSR.EDSR(scale=1, spn=True) (EDSR.py:123-137) restated."""
    z = _load(golden_dir, "g6_edsr_b2_40x56_train.npz")
    sd, inputs, gt = _regen_simple(z, R.edsr_param_shapes(4, 4, 32))
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    sd.update(params)
    pred = R.edsr_forward(sd, torch.cat(inputs, 1), True, n_resblocks=4)
    assert torch.allclose(pred.detach(), _t(z["pred"]), rtol=0, atol=1e-11)
    (pred * R.probe_gradient(pred.shape, int(z["seed"]) + 2)).mean().backward()
    _check_grads(z, params)
