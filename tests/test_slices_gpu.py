"""GPU: channel-slice addressing.  The reference's torch.cat sites (Guide cat_only, basics.py:134; decoder skips,
JSPSR.py:354-368; generator, spn.py:63) are served by producers writing their channel slice of one wide NHWC
buffer.  These tests pin that path against the dense + torch.cat formulation of the very same operators:
identical kernels, identical arithmetic, so the comparison is exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _mods():
    from jspsr_amd import kernels, ops
    return kernels, ops


def _rand(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).cuda().to(dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k,stride", [(3, 1), (3, 2), (1, 2), (5, 1)])
def test_conv_reads_and_writes_slices(dtype, k, stride):
    K, ops = _mods()
    B, H, W, C, O = 2, 20, 28, 32, 48
    wide = _rand((B, H, W, 3 * C), dtype, 1)
    w = (_rand((O, C, k, k), torch.float32, 2) * 0.1).requires_grad_()
    xs = wide.narrow(3, C, C)
    assert K.is_slice(xs) and not xs.is_contiguous()
    xa = xs.detach().requires_grad_()
    xb = xs.contiguous().requires_grad_()
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    buf = ops.SliceBuffer(B, OH, OW, 2 * O + 16, dtype, "cuda")
    buf.buf.fill_(7.0)
    ya = ops.conv2d(xa, w, None, stride, k // 2, False, dest=(buf, 16))
    yb = ops.conv2d(xb, w, None, stride, k // 2, False)
    assert ya.data_ptr() == buf.buf.data_ptr() + 16 * buf.buf.element_size()
    assert torch.equal(ya, yb)
    # neighbours of the slice untouched
    assert (buf.buf[..., :16] == 7).all() and (buf.buf[..., 16 + O:] == 7).all()
    dy = _rand((B, OH, OW, 2 * O), dtype, 3)
    (ga_x, ga_w) = torch.autograd.grad(ya, (xa, w), dy.narrow(3, O, O))            # sliced upstream gradient
    (gb_x, gb_w) = torch.autograd.grad(yb, (xb, w), dy.narrow(3, O, O).contiguous())
    assert torch.equal(ga_x, gb_x) and torch.equal(ga_w, gb_w)


@pytest.mark.parametrize("dtype", DT)
def test_conv_bias_relu_into_slice(dtype):
    """BN-free Basic2d (bias + ReLU epilogue): the backward reads the saved output through its pitch."""
    K, ops = _mods()
    B, H, W, C, O = 2, 16, 24, 32, 64
    x = _rand((B, H, W, C), dtype, 4)
    w = (_rand((O, C, 3, 3), torch.float32, 5) * 0.1).requires_grad_()
    b = (_rand((O,), torch.float32, 6) * 0.1).requires_grad_()
    buf = ops.SliceBuffer(B, H, W, 2 * O, dtype, "cuda")
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya = ops.conv2d(xa, w, b, 1, 1, True, dest=(buf, O))
    yb = ops.conv2d(xb, w, b, 1, 1, True)
    assert torch.equal(ya, yb)
    dy = _rand((B, H, W, O), dtype, 7)
    ga = torch.autograd.grad(ya, (xa, w, b), dy)
    gb = torch.autograd.grad(yb, (xb, w, b), dy)
    for a, c in zip(ga, gb):
        assert torch.equal(a, c)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("relu,with_res", [(True, True), (True, False), (False, True)])
def test_batch_norm_slices(dtype, relu, with_res):
    K, ops = _mods()
    B, H, W, C = 2, 18, 22, 64
    wide = _rand((B, H, W, 3 * C), dtype, 8)
    gamma = (1 + 0.2 * _rand((C,), torch.float32, 9)).requires_grad_()
    beta = (0.1 * _rand((C,), torch.float32, 10)).requires_grad_()
    xs, rs = wide.narrow(3, 0, C), wide.narrow(3, 2 * C, C)

    def run(x, res, dest, dy):
        x = x.detach().requires_grad_()
        res = res.detach().requires_grad_() if with_res else None
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        y = ops.batch_norm(x, gamma, beta, rm, rv, 0.1, 1e-5, True, relu, res, 0.5, dest=dest)
        ins = (x, gamma, beta) + ((res,) if with_res else ())
        return y, torch.autograd.grad(y, ins, dy), rm, rv

    buf = ops.SliceBuffer(B, H, W, 2 * C, dtype, "cuda")
    dyw = _rand((B, H, W, 2 * C), dtype, 11)
    ya, ga, rma, rva = run(xs, rs, (buf, C), dyw.narrow(3, C, C))
    yb, gb, rmb, rvb = run(xs.contiguous(), rs.contiguous(), None, dyw.narrow(3, C, C).contiguous())
    assert not ya.is_contiguous() and torch.equal(ya, yb)
    assert torch.equal(rma, rmb) and torch.equal(rva, rvb)
    for a, c in zip(ga, gb):
        assert torch.equal(a, c)


@pytest.mark.parametrize("dtype", DT)
def test_join_matches_cat(dtype):
    """Two producers + join + a consumer conv == the same producers + torch.cat + the consumer (values, all grads);
    a second join over a sub-range (the encoder's fused view beside the decoder's full view) accumulates correctly."""
    K, ops = _mods()
    B, H, W, C = 2, 16, 20, 32
    x1, x2 = _rand((B, H, W, C), dtype, 12), _rand((B, H, W, C), dtype, 13)
    ws = [(_rand((C, C, 3, 3), torch.float32, 14 + i) * 0.1).requires_grad_() for i in range(2)]
    wc = (_rand((16, 2 * C, 3, 3), torch.float32, 20) * 0.1).requires_grad_()
    wd = (_rand((16, C, 1, 1), torch.float32, 21) * 0.1).requires_grad_()
    dy1, dy2 = _rand((B, H, W, 16), dtype, 22), _rand((B, H, W, 16), dtype, 23)

    def tail(full, second):
        return ops.conv2d(full, wc, None, 1, 1), ops.conv2d(second, wd, None, 1, 0)

    buf = ops.SliceBuffer(B, H, W, 2 * C, dtype, "cuda")
    a1 = ops.conv2d(x1, ws[0], None, 1, 1, dest=(buf, 0))
    a2 = ops.conv2d(x2, ws[1], None, 1, 1, dest=(buf, C))
    full = buf.join((a1, a2))
    assert full.is_contiguous() and full.data_ptr() == buf.buf.data_ptr()
    za, zb = tail(full, buf.join((a2,), C))
    ga = torch.autograd.grad((za, zb), ws + [wc, wd], (dy1, dy2))

    b1 = ops.conv2d(x1, ws[0], None, 1, 1)
    b2 = ops.conv2d(x2, ws[1], None, 1, 1)
    zc, zd = tail(torch.cat((b1, b2), 3), b2)
    gb = torch.autograd.grad((zc, zd), ws + [wc, wd], (dy1, dy2))
    assert torch.equal(za, zc) and torch.equal(zb, zd)
    for a, c in zip(ga, gb):
        assert torch.equal(a, c)


def test_slice_buffer_rejects_bad_requests():
    K, ops = _mods()
    buf = ops.SliceBuffer(1, 8, 8, 40, torch.bfloat16, "cuda")
    with pytest.raises(ValueError):
        buf.slice(32, 16)                      # past the end
    with pytest.raises(ValueError):
        buf.slice(4, 8)                        # 8-byte start: not a 16-byte row
    with pytest.raises(ValueError):
        buf.slice(8, 8, (1, 4, 4))             # producer of another size
    other = ops.SliceBuffer(1, 8, 8, 40, torch.bfloat16, "cuda")
    with pytest.raises(RuntimeError):
        buf.join((other.slice(0, 8),))
    x = torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(8, 16, 3, 3, device="cuda")
    with pytest.raises(ValueError):
        ops.conv2d(x, w, None, 2, 1, False, dest=(buf, 0))   # stride-2 output does not fit the 8x8 buffer
