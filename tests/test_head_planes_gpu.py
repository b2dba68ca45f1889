"""GPU parity of the in-model propagation route of round 4: K1c (the generator's two 1x1 heads as one convolution that
writes planes, csrc/head.hip) and the propagation step on those planes (jspsr_prop_logits_*: the kernels of the public
PostProcessor boundary with the Sigmoid folded in), against the oracle -- through the C ABI (ctypes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import jspsr_ref as R


def _ops():
    from jspsr_amd import ops
    return ops


def _census(names):
    from jspsr_amd import _lib
    lib = _lib.load()
    return {n: lib.jspsr_launch_count(n.encode()) for n in names}


def _logits_case(B, H, W, sigma, seed):
    g = torch.Generator().manual_seed(seed)
    dem = torch.rand(B, 1, H, W, generator=g)
    head = torch.cat((1.5 * torch.randn(B, 9, H, W, generator=g), sigma * torch.randn(B, 16, H, W, generator=g)), 1)
    w = 1 + 0.3 * torch.randn(1, 1, 3, 3, generator=g)
    b = 0.1 * torch.randn(1, generator=g)
    gout = torch.randn(B, 1, H, W, generator=g)
    return dem, head, w, b, gout


def _oracle_logits(dem, head, w, b, gout):
    """fp64 oracle on the same stored values: sigmoid -> zero centre offset -> propagate; gradients by autograd."""
    f = lambda t: t.double()
    h = f(head).requires_grad_()
    wd, bd = f(w).requires_grad_(), f(b).requires_grad_()
    B, _, H, W = dem.shape
    off = torch.cat((h[:, 9:17], torch.zeros(B, 2, H, W, dtype=torch.float64), h[:, 17:]), 1)
    out = R.propagate(f(dem), torch.sigmoid(h[:, :9]), off, wd, bd, 1.0)
    out.backward(f(gout))
    return out.detach(), h.grad, wd.grad, bd.grad


def _smooth_mask(head):
    """(B,25,H,W) mask of gradient entries away from the bilinear kinks (integer coordinates) of their tap."""
    B, _, H, W = head.shape
    off = head[:, 9:].double().reshape(B, 8, 2, H, W)
    frac = (off - off.round()).abs().amin(2, keepdim=True) > 1e-4
    return torch.cat((torch.ones(B, 9, H, W, dtype=torch.bool), frac.expand(B, 8, 2, H, W).reshape(B, 16, H, W)), 1)


@pytest.mark.parametrize("shape,sigma", [
    ((1, 16, 64), 1.5),        # one tile (LDS-DMA kernel)
    ((2, 40, 100), 2.5),       # W % 4 == 0, not a tile multiple
    ((2, 6, 132), 1.5),        # rows with idle waves, ragged last column tile
    ((5, 20, 72), 1.5),        # a workgroup's run crosses images
    ((1, 260, 1028), 1.5),     # persistent runs of several tiles
    ((1, 129, 260), 8.0),      # most taps leave tile + halo
    ((2, 37, 53), 1.5),        # W % 4 != 0: the general kernel
    ((2, 5, 3), 3.0),          # tiny raster, general kernel
])
def test_prop_logits_against_the_oracle(shape, sigma):
    ops = _ops()
    dem, head, w, b, gout = _logits_case(*shape, sigma, seed=sum(shape))
    names = ("prop_logits_forward (dma)", "prop_logits_backward (dma)", "prop_logits_forward", "prop_logits_backward")
    before = _census(names)
    hd, wd, bd = head.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    out = ops.propagate_logits(dem.cuda(), hd, wd, bd, 1.0)
    out.backward(gout.cuda())
    after = _census(names)
    dma = shape[2] % 4 == 0
    assert after[names[0 if dma else 2]] == before[names[0 if dma else 2]] + 1
    assert after[names[1 if dma else 3]] == before[names[1 if dma else 3]] + 1
    eo, eh, ew, eb = _oracle_logits(dem, head, w, b, gout)
    tol = (2e-5 if sigma <= 3 else 1e-4) if max(shape) <= 300 else 4e-4
    err = lambda a, e: (a.cpu().double() - e).abs().max().item()
    assert err(out.detach(), eo) < tol + 1e-5 * eo.abs().max().item()
    m = _smooth_mask(head)
    assert ((hd.grad.cpu().double() - eh).abs() * m).max().item() < 2 * tol + 1e-4 * eh.abs().max().item()
    assert err(wd.grad, ew) < 5e-4 + 1e-4 * ew.abs().max().item()
    assert err(bd.grad, eb) < 5e-4


def test_prop_logits_equals_the_public_boundary_entry():
    """Same kernel, same numbers: propagate(sigmoid(logits), offsets) at the PostProcessor boundary vs the logits entry."""
    ops = _ops()
    dem, head, w, b, gout = _logits_case(2, 64, 128, 1.5, seed=5)
    a = ops.propagate_logits(dem.cuda(), head.cuda(), w.cuda(), b.cuda(), 1.0)
    c = ops.propagate(dem.cuda(), torch.sigmoid(head[:, :9].cuda()).contiguous(), head[:, 9:].cuda().contiguous(), w.cuda(), b.cuda(), 1.0)
    assert (a - c).abs().max().item() < 2e-6       # v_exp / v_rcp sigmoid in the kernel against torch's


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,shape,pitch", [
    (32, (2, 16, 24), 0),          # nf 8 models
    (64, (1, 8, 36), 0),           # LRRU bc 16 ... 288 pixels = 9 groups
    (128, (2, 24, 40), 0),         # the benched architecture
    (128, (1, 16, 16), 192),       # a channel slice of a wider buffer
])
def test_head_planes_forward_backward(dtype, cin, shape, pitch):
    ops = _ops()
    B, H, W = shape
    g = torch.Generator().manual_seed(cin + H)
    wide = torch.randn(B, H, W, pitch or cin, generator=g)
    x_full = wide.to(dtype).cuda()
    x = x_full[..., 16:16 + cin] if pitch else x_full
    ww, wo = 0.2 * torch.randn(9, cin, 1, 1, generator=g), 0.2 * torch.randn(16, cin, 1, 1, generator=g)
    bw, bo = 0.1 * torch.randn(9, generator=g), 0.1 * torch.randn(16, generator=g)
    gp = torch.randn(B, 25, H, W, generator=g)
    assert ops.head_planes_ok(x)
    xd = x.detach().requires_grad_()
    ps = [t.cuda().requires_grad_() for t in (ww, bw, wo, bo)]
    planes = ops.head_planes(xd, *ps)
    planes.backward(gp.cuda())
    torch.cuda.synchronize()
    # fp64 reference on the stored values (bf16: operands as the MFMA sees them -- weights and gradient rounded to bf16)
    q = (lambda t: t.to(torch.bfloat16).double()) if dtype == torch.bfloat16 else (lambda t: t.double())
    xr = x.detach().cpu().double().requires_grad_()
    w25 = torch.cat((ww, wo)).reshape(25, cin)
    ref = torch.einsum("bhwc,nc->bnhw", xr, q(w25)) + torch.cat((bw, bo)).double().view(1, 25, 1, 1)
    tol = 1e-5 if dtype == torch.float32 else 1e-5      # fp32 accumulation of exactly representable products
    assert (planes.detach().cpu().double() - ref.detach()).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    # backward: dx = g^T W (the kernel rounds g and W to the compute dtype), db = sum g, dW = sum g x (wgrad kernel: g and x in
    # the compute dtype)
    dx_ref = torch.einsum("bnhw,nc->bhwc", q(gp), q(w25))
    e_dx = (xd.grad.cpu().double() - dx_ref).abs().max().item()
    assert e_dx < (1e-5 if dtype == torch.float32 else 2.0 ** -8) * max(1.0, dx_ref.abs().max().item()), e_dx     # bf16: one storage rounding of dx
    db = torch.cat((ps[1].grad, ps[3].grad)).cpu().double()
    assert (db - gp.double().sum((0, 2, 3))).abs().max().item() < 1e-4 * (B * H * W) ** 0.5
    dW = torch.cat((ps[0].grad, ps[2].grad)).reshape(25, cin).cpu().double()
    dW_ref = torch.einsum("bnhw,bhwc->nc", q(gp), x.detach().cpu().double())
    assert (dW - dW_ref).abs().max().item() < 1e-4 * max(1.0, dW_ref.abs().max().item())


def test_models_take_the_planar_route():
    """JSPSR in training mode launches K1c and the logits entry of the boundary kernel (launch census), and
    JSPSR_HEAD_PLANES=0's route (the 32-channel NHWC head, K1h) gives the same prediction."""
    from jspsr_amd import engine as E
    from jspsr_amd.JSPSR import Model
    ic = {"lr_dem": 1, "image": 3}
    sd = R.make_state_dict(R.jspsr_param_shapes(ic, 8), seed=5)
    inputs, _ = R.synthetic_batch(1, 64, 64, False, seed=6)
    m = Model(dict(ic, COP30=1), num_feature=8)
    m.load_state_dict(sd)
    m = m.cuda().train()
    names = ("head_forward", "head_backward", "prop_logits_forward (dma)", "prop_logits_backward (dma)", "prop_head_forward")
    before = _census(names)
    x = [t.cuda() for t in inputs]
    a = m(*x)
    a.sum().backward()
    after = _census(names)
    assert all(after[n] == before[n] + 1 for n in names[:4]) and after[names[4]] == before[names[4]], (before, after)
    ga = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    E.planar_heads = False
    try:
        b = m(*x)
        b.sum().backward()
    finally:
        E.planar_heads = True
    assert _census(names)[names[4]] == after[names[4]] + 1
    assert (a - b).abs().max().item() < 1e-5
    for k, p in m.named_parameters():
        if "conv_weight" in k or "conv_offset" in k or "postprocessor" in k:
            r = (p.grad - ga[k]).norm().item() / max(ga[k].norm().item(), 1e-30)
            assert r < 1e-4, (k, r)
