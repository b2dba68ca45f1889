"""GPU parity: K4/K5 per-channel kernels (BN, ReLU/bias backward, channel gate) against torch CPU fp64."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _k():
    from jspsr_amd import kernels
    return kernels


def _nhwc(t, dtype=torch.float32):
    return t.permute(0, 2, 3, 1).contiguous().cuda().to(dtype)


def _nchw(t):
    return t.float().permute(0, 3, 1, 2).cpu().double()


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W,relu,with_res,training", [
    (2, 32, 17, 23, True, False, True),
    (1, 64, 64, 64, True, True, True),
    (3, 192, 9, 11, False, True, True),
    (2, 1536, 8, 8, True, False, True),
    (2, 128, 16, 16, True, True, False),
    (8, 64, 128, 128, True, True, True),
])
def test_bn_forward_backward(dtype, B, C, H, W, relu, with_res, training):
    K = _k()
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, C, H, W, generator=g) * 1.7 + 0.4)
    res = torch.randn(B, C, H, W, generator=g) if with_res else None
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    rm, rv = 0.1 * torch.randn(C, generator=g), 1 + 0.2 * torch.rand(C, generator=g)
    dy = torch.randn(B, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
        res = res.bfloat16().float() if res is not None else None
    # reference, fp64
    xr = x.double().requires_grad_()
    rr = res.double().requires_grad_() if res is not None else None
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    v = F.batch_norm(xr, rm_r, rv_r, gr, br, training, 0.1, 1e-5)
    if rr is not None:
        v = v * 0.5 + rr
    yr = F.relu(v) if relu else v
    yr.backward(dy.double())
    # HIP
    rm_d, rv_d = rm.cuda(), rv.cuda()
    xd = _nhwc(x, dtype)
    resd = _nhwc(res, dtype) if res is not None else None
    y, mean, invstd = K.bn_forward(xd, gamma.cuda(), beta.cuda(), rm_d, rv_d, 0.1, 1e-5, training, relu, resd, 0.5 if with_res else 1.0)
    tol = 2e-6 if dtype == torch.float32 else 5e-3
    assert _rel(_nchw(y), yr.detach()) < tol
    if training:
        assert _rel(rm_d.cpu(), rm_r) < 1e-5 and _rel(rv_d.cpu(), rv_r) < 1e-5
    mode = 0 if not relu else (1 if with_res else 2)   # 2: ReLU mask recomputed from x (no residual)
    if dtype == torch.bfloat16 and mode == 2:
        mode = 1   # the bf16-rounded y and the recomputed fp32 pre-activation disagree on a few |v| < 1e-2 entries
    dx, dres, dgamma, dbeta = K.bn_backward(_nhwc(dy, dtype), y if mode == 1 else None, xd, gamma.cuda(), mean, invstd,
                                            training, mode, 0.5 if with_res else 1.0, want_dres=with_res,
                                            beta=beta.cuda())
    gtol = 1e-5 if dtype == torch.float32 else 1.5e-2
    assert _rel(_nchw(dx), xr.grad) < gtol
    assert _rel(dgamma.cpu(), gr.grad) < gtol and _rel(dbeta.cpu(), br.grad) < gtol
    if with_res:
        assert _rel(_nchw(dres), rr.grad) < gtol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_act_backward(dtype):
    K = _k()
    g = torch.Generator().manual_seed(3)
    y = torch.relu(torch.randn(2, 64, 19, 21, generator=g))
    dy = torch.randn(2, 64, 19, 21, generator=g)
    if dtype == torch.bfloat16:
        y, dy = y.bfloat16().float(), dy.bfloat16().float()
    dz_ref = dy * (y > 0)
    dz, db = K.act_backward(_nhwc(dy, dtype), _nhwc(y, dtype), True)
    assert _rel(_nchw(dz), dz_ref) < 1e-6
    assert _rel(db.cpu(), dz_ref.double().sum((0, 2, 3))) < 1e-5
    # no relu, padded dz pitch (9 -> 16 channel head)
    dy9 = torch.randn(1, 8, 10, 10, generator=g)
    dzp, db = K.act_backward(_nhwc(dy9, dtype), None, False, dz_channels=16)
    assert dzp.shape[3] == 16 and (dzp[..., 8:] == 0).all()
    assert _rel(dzp[..., :8].float().cpu(), dy9.permute(0, 2, 3, 1).to(dtype).float()) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W", [(2, 64, 16, 24), (1, 1536, 8, 8), (8, 256, 64, 64), (3, 192, 5, 7)])
def test_channel_gate(dtype, B, C, H, W):
    """pool -> (tiny MLP in torch) -> scale, forward and backward, against the oracle's formula."""
    K = _k()
    g = torch.Generator().manual_seed(C + W)
    x = torch.relu(torch.randn(B, C, H, W, generator=g))
    dy = torch.randn(B, C, H, W, generator=g)
    w1 = torch.randn(C // 16, C, 1, 1, generator=g) / C ** 0.5
    w2 = torch.randn(C, C // 16, 1, 1, generator=g) / (C // 16) ** 0.5
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    xr = x.double().requires_grad_()
    mlp = lambda v: F.conv2d(F.relu(F.conv2d(v, w1.double())), w2.double())
    sr = torch.sigmoid(mlp(xr.mean((2, 3), keepdim=True)) + mlp(F.adaptive_max_pool2d(xr, 1)))
    (sr * xr).backward(dy.double())
    xd = _nhwc(x, dtype)
    avg, mx, amax = K.gate_pool(xd)
    assert _rel(avg.cpu(), x.double().mean((2, 3))) < 1e-5 and _rel(mx.cpu(), x.double().amax((2, 3))) < 1e-6
    avg.requires_grad_(), mx.requires_grad_()
    m2 = lambda v: F.linear(F.relu(F.linear(v, w1.cuda().flatten(1))), w2.cuda().flatten(1))
    s = torch.sigmoid(m2(avg) + m2(mx))
    y = K.gate_scale(xd, s.detach().contiguous())
    tol = 2e-6 if dtype == torch.float32 else 5e-3
    assert _rel(_nchw(y), (sr * xr).detach()) < tol
    dyd = _nhwc(dy, dtype)
    ds = K.gate_backward_reduce(dyd, xd)
    davg, dmax = torch.autograd.grad(s, (avg, mx), ds)
    dx = K.gate_backward_apply(dyd, s.detach().contiguous(), davg.contiguous(), dmax.contiguous(), amax)
    assert _rel(_nchw(dx), xr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W", [(2, 1, 24, 40), (1, 3, 17, 33), (3, 15, 16, 16), (1, 19, 8, 24)])
def test_boundary_conversion_nchw_to_nhwc(dtype, B, C, H, W):
    """jspsr_nchw_to_nhwc (what engine.from_nchw does with every model input) == cast + channels-last + zero pad."""
    from jspsr_amd import kernels as K
    x = torch.randn(B, C, H, W, device="cuda")
    e = K.epc(dtype)
    cp = (C + e - 1) // e * e
    y = K.nchw_to_nhwc(x, dtype, cp)
    ref = torch.nn.functional.pad(x.permute(0, 2, 3, 1).to(dtype), (0, cp - C)).contiguous()
    assert y.shape == ref.shape and torch.equal(y, ref)
    with pytest.raises(Exception):
        K.nchw_to_nhwc(x, dtype, C if C % e else C + 1)          # c_pad must be a whole number of 16-byte chunks


@pytest.mark.parametrize("B,C", [(8, 1536), (2, 256), (3, 64), (1, 1024)])
def test_gate_mlp_kernels_against_torch_autograd(B, C):
    """jspsr_gate_mlp_forward / _backward (resnet_cbam.py:41-53 on the pooled vectors) == the same MLP written with torch
    ops and differentiated by autograd, in fp64."""
    from jspsr_amd import kernels as K
    g = torch.Generator().manual_seed(B * 7 + C)
    Ch = C // 16
    avg, mx = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g).abs() + 0.5
    w1, w2 = torch.randn(Ch, C, generator=g) / C ** 0.5, torch.randn(C, Ch, generator=g) / Ch ** 0.5
    ds = torch.randn(B, C, generator=g)
    s, hid = K.gate_mlp_forward(avg.cuda(), mx.cuda(), w1.cuda(), w2.cuda())
    a, m, u1, u2 = (t.double().requires_grad_() for t in (avg, mx, w1, w2))
    h = lambda v: F.linear(F.relu(F.linear(v, u1)), u2)
    ref = torch.sigmoid(h(a) + h(m))
    assert _rel(s.cpu(), ref.detach()) < 2e-6
    gr = torch.autograd.grad(ref, (a, m, u1, u2), ds.double())
    out = K.gate_mlp_backward(ds.cuda(), s, hid, avg.cuda(), mx.cuda(), w1.cuda(), w2.cuda())
    for o, r in zip(out, gr):
        assert _rel(o.cpu(), r) < 5e-6
