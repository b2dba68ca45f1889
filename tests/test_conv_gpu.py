"""GPU parity: K2 implicit-GEMM convolutions (HIP/MFMA, through the C ABI) against torch CPU fp64."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _k():
    from jspsr_amd import kernels
    return kernels


def _nhwc(t):  # (B,C,H,W) cpu -> (B,H,W,C) cuda contiguous
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


def _relerr(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 24, 32, 64, 3, 1, 1),
    (1, 40, 36, 64, 128, 3, 1, 1),
    (2, 33, 47, 32, 32, 3, 1, 1),       # ragged M
    (1, 32, 32, 192, 128, 3, 2, 1),     # stride-2 first block
    (2, 16, 16, 64, 128, 1, 2, 0),      # 1x1 stride-2 shortcut
    (1, 24, 24, 128, 9, 1, 1, 0),       # head, tiny N
    (1, 24, 24, 128, 16, 1, 1, 0),
    (2, 20, 20, 4, 32, 5, 1, 2),        # stem, padded 1/3 -> 4 channels
    (1, 20, 28, 16, 32, 5, 1, 2),       # stem, 15 -> 16
    (1, 8, 8, 1536, 256, 3, 1, 1),      # deep K
    (1, 17, 19, 8, 200, 3, 1, 1),       # N not a tile multiple
    (2, 12, 20, 96, 32, 3, 1, 1),       # several channel chunks, raster smaller than / ragged against the 8x16 tile
    (2, 6, 10, 256, 64, 3, 1, 1),
    (2, 24, 40, 64, 32, 3, 1, 1),
    (2, 48, 80, 192, 16, 3, 1, 1),
    (1, 6, 10, 128, 128, 3, 2, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", CASES)
def test_conv_forward(dtype, B, H, W, Cin, Cout, k, stride, pad):
    K = _k()
    if Cin % K.epc(dtype):
        pytest.skip("channel granularity")
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()  # same rounded operands on both sides
    ref = F.relu(F.conv2d(x.double(), w.double(), bias.double(), stride, pad))
    wp = K.pack_weight(w.cuda(), 0, Cin, dtype)
    y = K.conv2d_forward(_nhwc(x).to(dtype), wp, bias.cuda(), stride, pad, relu=True)
    tol = 2e-6 if dtype == torch.float32 else 6e-3
    assert _relerr(_nchw(y.float()), ref) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", CASES)
def test_conv_dgrad(dtype, B, H, W, Cin, Cout, k, stride, pad):
    """dX of the conv == conv_transpose2d(gout, W)."""
    K = _k()
    e = K.epc(dtype)
    Cg = (Cout + e - 1) // e * e  # gathered tensor (gout) is channel-padded to the chunk size
    g = torch.Generator().manual_seed(B * 77 + H + Cin + Cout)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    go = torch.randn(B, Cout, OH, OW, generator=g)
    if dtype == torch.bfloat16:
        go, w = go.bfloat16().float(), w.bfloat16().float()
    x = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w.double(), None, stride, pad).backward(go.double())
    wpt = K.pack_weight(w.cuda(), 1, Cg, dtype)
    gop = F.pad(go, (0, 0, 0, 0, 0, Cg - Cout))
    dx = K.conv2d_dgrad(_nhwc(gop).to(dtype), wpt, (H, W), stride, pad)
    tol = 2e-6 if dtype == torch.float32 else 6e-3
    assert _relerr(_nchw(dx.float()), x.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", CASES + [(3, 40, 56, 4, 32, 5, 1, 2), (2, 64, 64, 64, 64, 3, 1, 1)])
def test_conv_wgrad(dtype, B, H, W, Cin, Cout, k, stride, pad):
    K = _k()
    e = K.epc(dtype)
    if Cin % e:
        pytest.skip("channel granularity")
    Cg = (Cout + e - 1) // e * e
    g = torch.Generator().manual_seed(B * 31 + H + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    go = torch.randn(B, Cout, OH, OW, generator=g)
    if dtype == torch.bfloat16:
        x, go = x.bfloat16().float(), go.bfloat16().float()
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, stride, pad).backward(go.double())
    gop = F.pad(go, (0, 0, 0, 0, 0, Cg - Cout))
    dW = K.conv2d_wgrad(_nhwc(gop).to(dtype), _nhwc(x).to(dtype), Cout, Cin, k, k, stride, pad)
    assert dW.shape == (Cout, Cin, k, k) and dW.dtype == torch.float32
    tol = 3e-6 if dtype == torch.float32 else 1e-5  # same (rounded) operands, fp32 accumulation both ways
    assert _relerr(dW.cpu(), w.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 256, 256, 64, 64), (5, 232, 250, 128, 40), (2, 256, 256, 64, 128), (3, 200, 216, 64, 200)])
def test_conv_16x16_tile_path(dtype, B, H, W, Cin, Cout):
    """Layers with 33..64 output channels on rasters of >= 1024 16x16-pixel tiles take the tall-tile
    instantiation of the patch kernel (conv.hip: launch<T>): forward + BN partial statistics, ragged bottom/right
    tiles and an odd count of 8-row tile bands (statistics rows are numbered by 8x16 tiles), and the data
    gradient of a conv whose INPUT has that many channels."""
    K = _k()
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    wp = K.pack_weight(w.cuda(), 0, Cin, dtype)
    y, st = K.conv2d_forward(_nhwc(x).to(dtype), wp, None, 1, 1, stats=True)
    tol = 2e-6 if dtype == torch.float32 else 6e-3
    assert _relerr(_nchw(y.float()), ref) < tol
    # statistics come from the fp32 accumulators (before rounding to the storage type)
    tot = st.double().sum(0).cpu()
    assert _relerr(tot[0], ref.sum((0, 2, 3))) < 1e-4 and _relerr(tot[1], (ref * ref).sum((0, 2, 3))) < 1e-5
    # data gradient: gathered tensor has Cout channels, written tensor Cin -> swap roles so the WRITTEN side is <= 64
    e = K.epc(dtype)
    Cg = (Cin + e - 1) // e * e
    w2 = torch.randn(Cin, Cout, 3, 3, generator=g) / (Cin * 9) ** 0.5   # conv Cout -> Cin; its dgrad writes Cout channels
    go = torch.randn(B, Cin, H, W, generator=g)
    if dtype == torch.bfloat16:
        go, w2 = go.bfloat16().float(), w2.bfloat16().float()
    xx = torch.zeros(B, Cout, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xx, w2.double(), None, 1, 1).backward(go.double())
    dx = K.conv2d_dgrad(_nhwc(F.pad(go, (0, 0, 0, 0, 0, Cg - Cin))).to(dtype), K.pack_weight(w2.cuda(), 1, Cg, dtype), (H, W), 1, 1)
    assert _relerr(_nchw(dx.float()), xx.grad) < tol


@pytest.mark.parametrize("B,H,W", [(4, 256, 256), (5, 232, 250), (1, 512, 520), (2, 256, 256), (3, 200, 216)])
def test_conv_resident_weights_path(B, H, W):
    """3x3 64->64 bf16 convs on >= 512 16x16 tiles run K2r (conv64.hip: weights in registers, patches by LDS-DMA,
    zero padding from the descriptor's range check): forward with the BatchNorm partial statistics on ragged rasters,
    every epilogue form (bias + ReLU; scale + addend + ReLU), channel slices on both sides, and the data gradient
    (reversed tap walk) with an addend."""
    K = _k()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(B + H + W)
    x = torch.randn(B, 64, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    wp = K.pack_weight(w.cuda(), 0, 64, dtype)
    xd = _nhwc(x).to(dtype)
    y, st = K.conv2d_forward(xd, wp, None, 1, 1, stats=True)
    assert _relerr(_nchw(y.float()), ref) < 6e-3
    # border pixels alone (zero padding) and interior alone
    edge = torch.zeros(H, W, dtype=torch.bool)
    edge[0] = edge[-1] = True
    edge[:, 0] = edge[:, -1] = True
    assert _relerr(_nchw(y.float())[..., edge], ref[..., edge]) < 6e-3
    tot = st.double().sum(0).cpu()
    assert _relerr(tot[0], ref.sum((0, 2, 3))) < 1e-4 and _relerr(tot[1], (ref * ref).sum((0, 2, 3))) < 1e-5
    # statistics rows are numbered by 8x16 tiles: each row holds exactly its own pixels
    rows = st.double().cpu().reshape(B, (H + 7) // 8, (W + 15) // 16, 2, 64)
    r8 = F.pad(ref, (0, -W % 16, 0, -H % 8)).reshape(B, 64, (H + 7) // 8, 8, (W + 15) // 16, 16).sum((3, 5)).permute(0, 2, 3, 1)
    assert (rows[..., 0, :] - r8).abs().max() < 1e-3 * r8.abs().max()
    # bias + ReLU
    bias = torch.randn(64, generator=g)
    y2 = K.conv2d_forward(xd, wp, bias.cuda(), 1, 1, relu=True)
    assert _relerr(_nchw(y2.float()), F.relu(ref + bias.double().view(1, -1, 1, 1))) < 6e-3
    # scale + bias + addend + ReLU, reading a channel slice of a wider tensor and writing into a slice of a wider one
    scale = torch.rand(64, generator=g) + 0.5
    res = torch.randn(B, 64, H, W, generator=g).bfloat16().float()
    wide = torch.randn(B, H, W, 160, generator=g).bfloat16().cuda()
    wide[..., 32:96] = xd
    out = torch.full((B, H, W, 128), 7.0, dtype=dtype, device="cuda")
    K.conv2d_forward(wide, wp, bias.cuda(), 1, 1, relu=True, out=out, out_coff=64, cin=64, in_coff=32, scale=scale.cuda(),
                     addend=_nhwc(res).to(dtype))
    ref3 = F.relu((ref * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)).bfloat16().double() + res.double())
    assert _relerr(_nchw(out[..., 64:].float()), ref3) < 6e-3
    assert (out[..., :64] == 7.0).all()
    # data gradient with an addend
    go = torch.randn(B, 64, H, W, generator=g).bfloat16().float()
    xx = torch.zeros(B, 64, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xx, w.double(), None, 1, 1).backward(go.double())
    dx = K.conv2d_dgrad(_nhwc(go).to(dtype), K.pack_weight(w.cuda(), 1, 64, dtype), (H, W), 1, 1, addend=_nhwc(res).to(dtype))
    assert _relerr(_nchw(dx.float()), xx.grad.bfloat16().double() + res.double()) < 6e-3


@pytest.mark.parametrize("B,H,W", [(2, 256, 512), (4, 250, 512), (1, 512, 520)])
def test_conv_resident_128_path(B, H, W):
    """3x3 128->128 bf16 convs on >= 2048 tiles of 8x16 pixels run K2q (conv128.hip: the whole weight matrix in one CU's
    registers, one patch-fragment read per four MFMAs, swizzled LDS-DMA patches, the two K-halves of a channel group meeting
    through LDS): training forward with the BatchNorm partial statistics on ragged rasters, plain forward with ReLU, channel
    slices on both sides, and the data gradient (reversed tap walk) with an addend and with ReLU."""
    K = _k()
    from jspsr_amd import _lib
    lib = _lib.load()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(B + H + W)
    x = torch.randn(B, 128, H, W, generator=g).bfloat16().float()
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    wp = K.pack_weight(w.cuda(), 0, 128, dtype)
    xd = _nhwc(x).to(dtype)
    n0 = lib.jspsr_launch_count(b"conv128_resident")
    y, st = K.conv2d_forward(xd, wp, None, 1, 1, stats=True)
    assert lib.jspsr_launch_count(b"conv128_resident") == n0 + 1
    assert _relerr(_nchw(y.float()), ref) < 6e-3
    edge = torch.zeros(H, W, dtype=torch.bool)
    edge[0] = edge[-1] = True
    edge[:, 0] = edge[:, -1] = True
    assert _relerr(_nchw(y.float())[..., edge], ref[..., edge]) < 6e-3
    tot = st.double().sum(0).cpu()
    assert _relerr(tot[0], ref.sum((0, 2, 3))) < 1e-4 and _relerr(tot[1], (ref * ref).sum((0, 2, 3))) < 1e-5
    # statistics rows are numbered by 8x16 tiles: each row holds exactly its own pixels
    rows = st.double().cpu().reshape(B, (H + 7) // 8, (W + 15) // 16, 2, 128)
    r8 = F.pad(ref, (0, -W % 16, 0, -H % 8)).reshape(B, 128, (H + 7) // 8, 8, (W + 15) // 16, 16).sum((3, 5)).permute(0, 2, 3, 1)
    assert (rows[..., 0, :] - r8).abs().max() < 1e-3 * r8.abs().max()
    q8 = F.pad(ref * ref, (0, -W % 16, 0, -H % 8)).reshape(B, 128, (H + 7) // 8, 8, (W + 15) // 16, 16).sum((3, 5)).permute(0, 2, 3, 1)
    assert (rows[..., 1, :] - q8).abs().max() < 1e-3 * q8.abs().max()
    # ReLU, reading a channel slice of a wider tensor and writing into a slice of a wider one
    wide = torch.randn(B, H, W, 192, generator=g).bfloat16().cuda()
    wide[..., 32:160] = xd
    out = torch.full((B, H, W, 256), 7.0, dtype=dtype, device="cuda")
    K.conv2d_forward(wide, wp, None, 1, 1, relu=True, out=out, out_coff=64, cin=128, in_coff=32)
    assert lib.jspsr_launch_count(b"conv128_resident") == n0 + 2
    assert _relerr(_nchw(out[..., 64:192].float()), F.relu(ref)) < 6e-3
    assert (out[..., :64] == 7.0).all() and (out[..., 192:] == 7.0).all()
    # scale + bias + addend + ReLU (the inference epilogue: BatchNorm folded into the conv), through the same slices
    scale = torch.rand(128, generator=g) + 0.5
    bias = torch.randn(128, generator=g)
    res0 = torch.randn(B, 128, H, W, generator=g).bfloat16().float()
    out.fill_(7.0)
    K.conv2d_forward(wide, wp, bias.cuda(), 1, 1, relu=True, out=out, out_coff=64, cin=128, in_coff=32, scale=scale.cuda(),
                     addend=_nhwc(res0).to(dtype))
    assert lib.jspsr_launch_count(b"conv128_resident") == n0 + 3
    ref3 = F.relu((ref * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)).bfloat16().double() + res0.double())
    assert _relerr(_nchw(out[..., 64:192].float()), ref3) < 6e-3
    assert (out[..., :64] == 7.0).all() and (out[..., 192:] == 7.0).all()
    del wide, out, res0
    # data gradient, plain and with an addend + ReLU
    res = torch.randn(B, 128, H, W, generator=g).bfloat16().float()
    go = torch.randn(B, 128, H, W, generator=g).bfloat16().float()
    xx = torch.zeros(B, 128, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xx, w.double(), None, 1, 1).backward(go.double())
    wpt = K.pack_weight(w.cuda(), 1, 128, dtype)
    dx = K.conv2d_dgrad(_nhwc(go).to(dtype), wpt, (H, W), 1, 1)
    assert _relerr(_nchw(dx.float()), xx.grad) < 6e-3
    dx = K.conv2d_dgrad(_nhwc(go).to(dtype), wpt, (H, W), 1, 1, addend=_nhwc(res).to(dtype), relu=True)
    assert lib.jspsr_launch_count(b"conv128_resident") == n0 + 5
    assert _relerr(_nchw(dx.float()), F.relu(xx.grad.bfloat16().double() + res.double())) < 6e-3


def test_conv_resident_128_random_geometries():
    """K2q on 30 random geometries in a child process (JSPSR_CONV_RESIDENT128_MIN=1: every size takes the kernel): rasters from
    less than one tile to 256 x 512, ragged both ways, batch 1..5, channel slices of wider tensors on both sides, addend, ReLU,
    the tile queue on and off -- forward, statistics (totals and per-tile rows), slice writes (nothing outside the slice is
    touched) and data gradient against torch's fp32 convolution (tools/lab/k2q_fuzz.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "lab", "k2q_fuzz.py"), "30", "7"], capture_output=True, text=True,
                       timeout=600, cwd=root, env=dict(os.environ, JSPSR_CONV_RESIDENT128_MIN="1"))
    assert r.returncode == 0 and "FUZZ OK: 30 cases" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_conv_dynamic_queue_switch_gives_the_same_bits():
    """jspsr_conv_dynamic_queue(1) -- what GradReducer sets for world sizes > 1 -- makes K2r and K2q draw their tiles from a
    global ticket instead of the static stride walk: a scheduling change only, every output bit and every statistics row
    stays what it was."""
    K = _k()
    from jspsr_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    outs = {}
    for C, shape in ((64, (4, 512, 512)), (128, (4, 256, 512))):      # 4096 tiles each: where the queue starts
        x = torch.randn(*shape, C, generator=g).cuda().bfloat16()
        w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda()
        wp, wpt = K.pack_weight(w, 0, C, torch.bfloat16), K.pack_weight(w, 1, C, torch.bfloat16)
        for mode in (0, 1, 1, 0):
            prev = lib.jspsr_conv_dynamic_queue(mode)
            try:
                y, st = K.conv2d_forward(x, wp, None, 1, 1, stats=True)
                dx = K.conv2d_dgrad(x, wpt, shape[1:], 1, 1, addend=x)
                torch.cuda.synchronize()
            finally:
                lib.jspsr_conv_dynamic_queue(-1)
            assert prev == -1
            if (C, 0) not in outs:
                outs[(C, 0)] = (y, st, dx)
            else:
                for a, b in zip((y, st, dx), outs[(C, 0)]):
                    assert torch.equal(a, b), (C, mode)


def test_conv_resident_kernels_overlap_on_two_streams():
    """K2r is a persistent kernel (one workgroup per CU, most of the CU's LDS and registers) with no device-global state:
    launches of its three modes queued on two streams at once -- as the step's branch streams do -- must give the bits of
    the same launches run one after the other."""
    K = _k()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(2, 256, 256, 64, generator=g).cuda().to(dtype) for _ in range(2)]
    ws = [(torch.randn(64, 64, 3, 3, generator=g) / 24.0).cuda() for _ in range(2)]
    bias = torch.randn(64, generator=g).cuda()
    packs = [(K.pack_weight(w, 0, 64, dtype), K.pack_weight(w, 1, 64, dtype)) for w in ws]

    def work(i):
        y, st = K.conv2d_forward(xs[i], packs[i][0], None, 1, 1, stats=True)
        z = K.conv2d_forward(y, packs[i][0], bias, 1, 1, relu=True)
        return y, st, z, K.conv2d_dgrad(z, packs[i][1], (256, 256), 1, 1, addend=xs[i])

    ref = [work(0), work(1)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    out = [None, None]
    for rep in range(3):
        for i, st in enumerate(streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                out[i] = work(i)
        torch.cuda.synchronize()
        for i in range(2):
            for a, b in zip(out[i], ref[i]):
                assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (2, 64, 64, 64, 64),      # one tile pair
    (1, 23, 128, 128, 64),    # odd row count, two X chunks
    (3, 9, 64, 96, 160),      # channel tails on both sides (96 = 64 + 32, 160 = 2*64 + 32)
    (1, 5, 192, 40, 72),      # fewer rows than a row block, ragged channels
    (2, 128, 128, 64, 128),   # many row blocks per strip
])
def test_conv_wgrad_3x3_all_taps_path(dtype, B, H, W, Cin, Cout):
    """The 3x3 / stride-1 layers take the nine-taps-per-staging kernel (wgrad.hip: wgrad_patch_kernel) when the
    width is a multiple of its strip (64 px bf16, 32 px fp32); same oracle and tolerance as the generic kernel,
    plus exact agreement with the generic kernel's slab layout through the shared ordered reduce."""
    K = _k()
    g = torch.Generator().manual_seed(B * 7 + H + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    go = torch.randn(B, Cout, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, go = x.bfloat16().float(), go.bfloat16().float()
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(go.double())
    dW = K.conv2d_wgrad(_nhwc(go).to(dtype), _nhwc(x).to(dtype), Cout, Cin, 3, 3, 1, 1)
    tol = 3e-6 if dtype == torch.float32 else 1e-5
    assert _relerr(dW.cpu(), w.grad) < tol
    # border taps in isolation: the corner weight gradient sees the zero padding on two sides
    assert _relerr(dW.cpu()[:, :, 0, 0], w.grad[:, :, 0, 0]) < 10 * tol
    assert _relerr(dW.cpu()[:, :, 2, 2], w.grad[:, :, 2, 2]) < 10 * tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_wgrad_3x3_channel_slices(dtype):
    """Operands that are channel slices of wider buffers (pitch != channels), as the slice buffers hand them over."""
    K = _k()
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 16, 64
    xw = torch.randn(B, 192, H, W, generator=g)
    gw = torch.randn(B, 160, H, W, generator=g)
    if dtype == torch.bfloat16:
        xw, gw = xw.bfloat16().float(), gw.bfloat16().float()
    w = torch.zeros(64, 64, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xw[:, 64:128].double(), w, None, 1, 1).backward(gw[:, 32:96].double())
    Xd, Gd = _nhwc(xw).to(dtype), _nhwc(gw).to(dtype)
    dW = K.conv2d_wgrad(Gd.narrow(3, 32, 64), Xd.narrow(3, 64, 64), 64, 64, 3, 3, 1, 1)
    assert _relerr(dW.cpu(), w.grad) < (3e-6 if dtype == torch.float32 else 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", [
    (2, 24, 40, 64, 64, 3, 1, 1),     # patch kernel, 64-wide written tensor
    (1, 40, 36, 128, 64, 3, 1, 1),    # patch kernel, 128-wide written tensor
    (1, 32, 32, 192, 128, 3, 2, 1),   # stride 2: four phase launches, each written pixel gets the addend once
    (2, 16, 16, 64, 128, 1, 2, 0),    # 1x1 stride 2: three of the four phases have no tap at all
    (1, 17, 19, 200, 8, 3, 1, 1),     # ragged written channels -> scalar epilogue
    (4, 256, 256, 64, 64, 3, 1, 1),   # 16x16-pixel tile instantiation
])
def test_conv_dgrad_addend(dtype, B, H, W, Cin, Cout, k, stride, pad):
    """gin = dgrad(gout) + addend in one launch: the meeting point of a BasicBlock's two gradient paths
    (basics.py:113-122).  Checked against the two-step formulation with the same roundings."""
    K = _k()
    e = K.epc(dtype)
    if Cin % e:
        pytest.skip("channel granularity")
    Cg = (Cout + e - 1) // e * e
    g = torch.Generator().manual_seed(B * 13 + H + Cin + Cout)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    go = F.pad(torch.randn(B, Cout, OH, OW, generator=g), (0, 0, 0, 0, 0, Cg - Cout))
    wide = torch.randn(B, H, W, Cin + 2 * e, generator=g).cuda().to(dtype)
    addend = wide.narrow(3, e, Cin)                  # a channel slice: pitch != channels
    wpt = K.pack_weight(w.cuda(), 1, Cg, dtype)
    god = _nhwc(go).to(dtype)
    plain = K.conv2d_dgrad(god, wpt, (H, W), stride, pad)
    fused = K.conv2d_dgrad(god, wpt, (H, W), stride, pad, addend=addend)
    ref = (plain.float() + addend.float()).to(dtype)  # same two roundings as the kernel's epilogue
    assert torch.equal(fused, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,with_res,relu", [
    (2, 24, 40, 64, 64, 3, 1, 1, True, True),      # BasicBlock tail: conv2 -> bn2 -> + shortcut -> ReLU
    (1, 40, 36, 64, 128, 3, 1, 1, False, True),    # conv1 -> bn1 -> ReLU
    (2, 16, 16, 64, 128, 1, 2, 0, False, False),   # projection shortcut: conv1x1 s2 -> bn
    (1, 17, 19, 8, 200, 3, 1, 1, True, True),      # partly filled N tile
    (1, 12, 12, 32, 9, 1, 1, 0, True, True),       # scalar epilogue (9 channels)
    (4, 256, 256, 64, 64, 3, 1, 1, True, True),    # 16x16-pixel tile instantiation
])
def test_conv_inference_epilogue(dtype, B, H, W, Cin, Cout, k, stride, pad, with_res, relu):
    """out = [relu](conv(x) * scale + shift (+ addend)) with (scale, shift) = eval-mode BatchNorm folded by
    jspsr_bn_fold: the reference's conv -> BatchNorm2d.eval() (-> + residual) (-> ReLU), basics.py:49-53,111-123."""
    K = _k()
    if Cin % K.epc(dtype):
        pytest.skip("channel granularity")
    g = torch.Generator().manual_seed(B * 5 + H + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    gamma, beta = 1 + 0.3 * torch.randn(Cout, generator=g), 0.2 * torch.randn(Cout, generator=g)
    rm, rv = 0.3 * torch.randn(Cout, generator=g), 0.5 + torch.rand(Cout, generator=g)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, OH, OW, generator=g) if with_res else None
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
        res = res.bfloat16().float() if res is not None else None
    rs = 0.5 if with_res else 1.0
    ref = F.batch_norm(F.conv2d(x.double(), w.double(), None, stride, pad), rm.double(), rv.double(), gamma.double(),
                       beta.double(), False, 0.1, 1e-5)
    if res is not None:
        ref = ref * rs + res.double()
    if relu:
        ref = F.relu(ref)
    sc, sh = K.bn_fold(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5, rs)
    resd = _nhwc(res).to(dtype) if res is not None else None
    y = K.conv2d_forward(_nhwc(x).to(dtype), K.pack_weight(w.cuda(), 0, Cin, dtype), sh, stride, pad, relu,
                         scale=sc, addend=resd)
    tol = 3e-6 if dtype == torch.float32 else 8e-3
    assert _relerr(_nchw(y.float()), ref) < tol
    if relu:
        assert (y >= 0).all()


def test_conv_transpose_wgrad():
    """ConvTranspose2d weight (I,O,kh,kw): G = its input, X = grad of its output."""
    K = _k()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 64, 12, 10, generator=g)
    gy = torch.randn(2, 32, 24, 20, generator=g)
    w = torch.zeros(64, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(x.double(), w, None, 2, 1, 1).backward(gy.double())
    dW = K.conv2d_wgrad(_nhwc(x), _nhwc(gy), 64, 32, 3, 3, 2, 1)
    assert _relerr(dW.cpu(), w.grad) < 3e-6


@pytest.mark.parametrize("C", [64, 256])
def test_conv_transpose_forward(C):
    """ConvTranspose2d k3 s2 p1 op1 (basics.py:69-77) == dgrad launch with the (I,O,KH,KW) weight."""
    K = _k()
    g = torch.Generator().manual_seed(C)
    x = torch.randn(2, C, 12, 10, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5  # ConvTranspose2d weight: (in, out, kh, kw)
    ref = F.conv_transpose2d(x.double(), w.double(), None, 2, 1, 1)
    wpt = K.pack_weight(w.cuda(), 1, C, torch.float32)  # [out_T][ky][kx][in_T]
    y = K.conv2d_dgrad(_nhwc(x), wpt, (24, 20), 2, 1)
    assert _relerr(_nchw(y), ref) < 2e-6


def test_channel_slices():
    """A conv can read a channel slice and write into a slice of a wider (concat) buffer."""
    K = _k()
    g = torch.Generator().manual_seed(9)
    xw = torch.randn(1, 96, 16, 16, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    ref = F.conv2d(xw[:, 32:64].double(), w.double(), None, 1, 1)
    wp = K.pack_weight(w.cuda(), 0, 32, torch.float32)
    out = torch.full((1, 16, 16, 160), 7.0, device="cuda")
    K.conv2d_forward(_nhwc(xw), wp, None, 1, 1, out=out, out_coff=64, in_coff=32)
    assert _relerr(_nchw(out[..., 64:128].contiguous()), ref) < 2e-6
    assert (out[..., :64] == 7).all() and (out[..., 128:] == 7).all()


def test_bad_arguments_raise():
    K = _k()
    from jspsr_amd._lib import JspsrHipError
    x = torch.zeros(1, 8, 8, 6, device="cuda")  # 6 channels: not a multiple of 4
    wp = torch.zeros(8, 3, 3, 6, device="cuda")
    with pytest.raises(JspsrHipError):
        K.conv2d_forward(x, wp, None, 1, 1)
    with pytest.raises(RuntimeError):
        K.conv2d_forward(torch.zeros(1, 8, 8, 8), wp, None, 1, 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", [(2, 33, 47, 32, 32, 3, 1, 1), (1, 40, 36, 64, 128, 3, 1, 1),
                                                         (2, 16, 16, 64, 128, 1, 2, 0), (1, 24, 24, 16, 200, 3, 2, 1)])
def test_conv_epilogue_statistics(dtype, B, H, W, Cin, Cout, k, stride, pad):
    """BatchNorm partial sums written by the conv epilogue == sums of the fp32 result."""
    K = _k()
    if Cin % K.epc(dtype):
        pytest.skip("channel granularity")
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    y, st = K.conv2d_forward(_nhwc(x).to(dtype), K.pack_weight(w.cuda(), 0, Cin, dtype), None, stride, pad, stats=True)
    s = st.double().sum(0).cpu()
    assert _relerr(s[0], ref.sum((0, 2, 3))) < 1e-4
    assert _relerr(s[1], (ref * ref).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("env", [{"JSPSR_CONV_TALL": "2"}, {"JSPSR_CONV_NOPATCH": "1"}, {"JSPSR_WGRAD_NOPATCH": "1"},
                                 {"JSPSR_CONV_TALL": "0"}, {"JSPSR_CONV_RESIDENT": "0"}, {"JSPSR_CONV_DYNQ": "1"},
                                 {"JSPSR_CONV_RESIDENT128": "0"}, {"JSPSR_CONV_DYNQ128": "1"}])
def test_opt_in_kernel_variants_in_a_child_process(env):
    """The library reads its lab switches once per process: the opt-in / fallback instantiations (8-wave 256x128 tile,
    generic implicit-GEMM kernel only, generic weight-gradient kernel only, no 16x16 tile, no register-resident 64-channel kernel, K2r's dynamic tile queue, no register-resident 128-channel kernel, K2q's dynamic tile queue) are exercised in a child
    process against the same fp64 reference, so that they stay correct while they are not the default."""
    import subprocess
    import sys
    code = r"""
import torch, torch.nn.functional as F
from jspsr_amd import kernels as K
g = torch.Generator().manual_seed(0)
for (B, H, W, Ci, Co) in [(2, 256, 256, 64, 128), (4, 256, 256, 64, 64), (8, 512, 512, 64, 64), (2, 256, 512, 128, 128), (4, 256, 512, 128, 128)]:
    if B == 8 and not __import__("os").environ.get("JSPSR_CONV_DYNQ"):      # the dynamic tile queue's size
        continue
    if Ci == 128 and not (B == 2 and __import__("os").environ.get("JSPSR_CONV_RESIDENT128") or B == 4 and __import__("os").environ.get("JSPSR_CONV_DYNQ128")):
        continue      # K2q's size on the patch kernel; K2q's dynamic tile queue at the 4096 tiles it starts at
    x = torch.randn(B, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).bfloat16().float()
    go = torch.randn(B, Co, H, W, generator=g).bfloat16().float()
    xr = x.double().requires_grad_(); wr = w.double().requires_grad_()
    y = F.conv2d(xr, wr, None, 1, 1); y.backward(go.double())
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
    rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
    yd = K.conv2d_forward(nh(x), K.pack_weight(w.cuda(), 0, Ci, torch.bfloat16), None, 1, 1)
    dx = K.conv2d_dgrad(nh(go), K.pack_weight(w.cuda(), 1, Co, torch.bfloat16), (H, W), 1, 1)
    dw = K.conv2d_wgrad(nh(go), nh(x), Co, Ci, 3, 3, 1, 1)
    assert rel(yd.float().permute(0, 3, 1, 2), y.detach()) < 6e-3
    assert rel(dx.float().permute(0, 3, 1, 2), xr.grad) < 6e-3
    assert rel(dw, wr.grad) < 1e-5
print("variants ok")
"""
    import os
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "variants ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_packed_weight_cache_and_single_launch_repack():
    """ops._packed keeps the re-laid copy of a Parameter until its values can have changed (torch's version counter, or
    the module's weights epoch for writers that bypass it); ops.repack_all() re-makes every cached copy in ONE launch
    (jspsr_pack_weights_multi) -- bit-identical to per-tensor jspsr_pack_weight, both layouts, both dtypes."""
    from jspsr_amd import ops
    K = _k()
    g_ = torch.Generator().manual_seed(21)
    params = [torch.nn.Parameter(torch.randn(s, generator=g_).cuda()) for s in ((24, 16, 3, 3), (8, 40, 1, 1), (32, 8, 5, 5))]
    cases = []
    for p in params:
        for mode, dt in ((0, torch.bfloat16), (1, torch.bfloat16), (0, torch.float32), (1, torch.float32)):
            cpad = (p.shape[1] if mode == 0 else p.shape[0])
            cpad = (cpad + 7) // 8 * 8
            a = ops._packed(p, p.detach(), mode, cpad, dt)
            assert ops._packed(p, p.detach(), mode, cpad, dt) is a           # cached
            assert torch.equal(a, K.pack_weight(p.detach(), mode, cpad, dt))
            cases.append((p, mode, cpad, dt, a))
    with torch.no_grad():
        for p in params:
            p.data.mul_(1.5)                       # through .data: the version counter does not move
    ops.invalidate_packed_weights()
    n = ops.repack_all()                       # what FlatAdamW.step() does when JSPSR_REPACK_ALL=1
    assert n >= len(cases)
    for p, mode, cpad, dt, a in cases:
        b = ops._packed(p, p.detach(), mode, cpad, dt)
        assert b is a, "repack_all refreshes the cached tensors in place"
        assert torch.equal(b, K.pack_weight(p.detach(), mode, cpad, dt))
    with torch.no_grad():
        params[0].add_(1.0)                        # an ordinary in-place update: seen through the version counter
    c = ops._packed(params[0], params[0].detach(), 0, 16, torch.bfloat16)
    assert torch.equal(c, K.pack_weight(params[0].detach(), 0, 16, torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,O", [(2, 24, 64, 64, 64), (1, 19, 128, 128, 64), (2, 16, 64, 64, 128)])   # W: a multiple of the nine-tap wgrad kernel's 64-pixel strips
def test_input_affine_relu_in_conv_and_wgrad_staging(B, H, W, C, O, dtype):
    """jspsr_conv2d_forward(in_affine) / jspsr_conv2d_wgrad(x_affine): the conv reads relu(x * scale + shift) formed
    between the patch registers and LDS.  Against the same kernels fed with the materialised tensor (what a separate
    BatchNorm pass would have stored): equal up to the rounding of that tensor -- the kernel forms x * scale + shift
    with one fused multiply-add, torch with two roundings, and in bf16 the stored copy can land on the neighbouring
    value -- i.e. relative L2 < 1e-6 (fp32) / 2e-3 (bf16).  Zero padding must stay zero (a shift > 0 would otherwise leak
    relu(shift) into the border: checked separately on a constant-zero input, exactly)."""
    K = _k()
    g_ = torch.Generator().manual_seed(B * 7 + H + C + O)
    x = torch.randn(B, H, W, C, generator=g_).cuda().to(dtype)
    aff = torch.stack((1 + 0.3 * torch.randn(C, generator=g_), 0.2 + 0.3 * torch.randn(C, generator=g_))).cuda().contiguous()
    w = (torch.randn(O, C, 3, 3, generator=g_) / (C * 9) ** 0.5).cuda()
    go = torch.randn(B, H, W, O, generator=g_).cuda().to(dtype)
    assert K.fused_input_ok(dtype, B, H, W, C, O, 3, 3, 1, 1)
    tol = 1e-6 if dtype == torch.float32 else 2e-3
    rel = lambda p_, q_: ((p_.double() - q_.double()).norm() / q_.double().norm()).item()
    y_mat = torch.relu(x.float() * aff[0] + aff[1]).to(dtype)
    wp = K.pack_weight(w, 0, C, dtype)
    a = K.conv2d_forward(x, wp, None, 1, 1, in_affine=aff, in_relu=True)
    b = K.conv2d_forward(y_mat, wp, None, 1, 1)
    assert rel(a, b) < tol
    a_s, st_a = K.conv2d_forward(x, wp, None, 1, 1, stats=True, in_affine=aff, in_relu=True)
    assert torch.equal(a_s, a) and rel(st_a, K.conv2d_forward(y_mat, wp, None, 1, 1, stats=True)[1]) < 10 * tol
    da = K.conv2d_wgrad(go, x, O, C, 3, 3, 1, 1, x_affine=aff, x_relu=True)
    db = K.conv2d_wgrad(go, y_mat, O, C, 3, 3, 1, 1)
    assert rel(da, db) < tol
    y_lin = (x.float() * aff[0] + aff[1]).to(dtype)                          # without the ReLU
    assert rel(K.conv2d_forward(x, wp, None, 1, 1, in_affine=aff, in_relu=False), K.conv2d_forward(y_lin, wp, None, 1, 1)) < tol
    # padding: x = 0, scale = 1, shift = 0.5 => the transformed input is 0.5 inside the raster and 0 (not 0.5) outside:
    # the conv of ones-weights counts the in-raster taps, exactly
    zero = torch.zeros(1, 8, 64, C, device="cuda", dtype=dtype)
    half = torch.stack((torch.ones(C), torch.full((C,), 0.5))).cuda().contiguous()
    ones_w = K.pack_weight(torch.ones(O, C, 3, 3, device="cuda"), 0, C, dtype)
    cnt = K.conv2d_forward(zero, ones_w, None, 1, 1, in_affine=half, in_relu=True).float()
    taps = torch.nn.functional.conv2d(torch.ones(1, 1, 8, 64), torch.ones(1, 1, 3, 3), padding=1)[0, 0].cuda()
    assert torch.equal(cnt[0, :, :, 0], taps * 0.5 * C)
    dcnt = K.conv2d_wgrad(torch.ones(1, 8, 64, O, device="cuda", dtype=dtype), zero, O, C, 3, 3, 1, 1, x_affine=half, x_relu=True)
    want = torch.tensor([[7 * 63, 7 * 64, 7 * 63], [8 * 63, 8 * 64, 8 * 63], [7 * 63, 7 * 64, 7 * 63]], dtype=torch.float32).cuda() * 0.5
    assert torch.equal(dcnt[0, 0], want)


def test_input_affine_is_refused_where_no_kernel_applies_it():
    K = _k()
    from jspsr_amd._lib import JspsrHipError
    x = torch.randn(1, 8, 8, 16, device="cuda")
    aff = torch.ones(2, 16, device="cuda")
    assert not K.fused_input_ok(torch.float32, 1, 8, 8, 16, 16, 3, 3, 1, 1)
    with pytest.raises(JspsrHipError, match="in_affine"):
        K.conv2d_forward(x, K.pack_weight(torch.randn(16, 16, 3, 3, device="cuda"), 0, 16, torch.float32), None, 1, 1, in_affine=aff)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C", [
    (2, 40, 56, 128),        # 8x16 tiles (< 1024 tiles of 16x16), ragged in both directions
    (1, 64, 64, 256),
    (2, 256, 256, 128),      # the 16x16-pixel x 64-channel tiles (>= 1024), two N tiles per pixel tile
    (1, 100, 36, 64),        # 64 channels below the K2r threshold (bf16) / fp32: patch kernel
])
def test_bn_backward_reduce_fused_into_the_data_gradient_epilogue(dtype, B, H, W, C):
    """the reduce pass of bn1's backward (sum dz, sum dz xhat; conv -> BN -> ReLU -> conv of BasicBlock,
    basics.py:111-117) taken in the epilogue of conv2's data gradient (jspsr_conv2d_dgrad: red_*) instead of by a pass of its
    own: same dz1, dgamma, dbeta as the separate pass up to the order of the fp32 sums."""
    from jspsr_amd import kernels as K
    assert K.dgrad_reduce_ok(dtype, B, H, W, C, C, 3, 3, 1, 1)
    g_ = torch.Generator().manual_seed(C + H)
    dz2 = torch.randn(B, H, W, C, generator=g_).to(dtype).cuda()
    z1 = (1.5 * torch.randn(B, H, W, C, generator=g_) + 0.3).to(dtype).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g_) / (3 * C ** 0.5)).cuda()
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g_)).cuda(), (0.2 * torch.randn(C, generator=g_)).cuda()
    zf = z1.float()
    mean = zf.mean((0, 1, 2)).contiguous()
    invstd = (zf.var((0, 1, 2), unbiased=False) + 1e-5).rsqrt().contiguous()
    wpt = K.pack_weight(w, 1, C, dtype)
    names = ("bn_bwd_reduce", "bn_reduce_params")
    from jspsr_amd import _lib
    cnt = lambda: {n: _lib.load().jspsr_launch_count(n.encode()) for n in names}
    # separate passes
    dy_a = K.conv2d_dgrad(dz2, wpt, (H, W), 1, 1)
    c0 = cnt()
    dz_a, _, dg_a, db_a = K.bn_backward(dy_a, None, z1, gamma, mean, invstd, True, 2, 1.0, beta=beta)
    assert cnt()["bn_bwd_reduce"] == c0["bn_bwd_reduce"] + 1
    # fused
    par = K.bn_reduce_params(gamma, beta, mean, invstd)
    dy_b, part = K.conv2d_dgrad(dz2, wpt, (H, W), 1, 1, red=(z1, par))
    c1 = cnt()
    dz_b, _, dg_b, db_b = K.bn_backward(dy_b, None, z1, gamma, mean, invstd, True, 2, 1.0, beta=beta, ext_partial=part)
    assert cnt()["bn_bwd_reduce"] == c1["bn_bwd_reduce"]            # no reduce pass this time
    assert torch.equal(dy_a, dy_b)                                    # the data gradient itself is untouched
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert rel(dg_b, dg_a) < 2e-5 and rel(db_b, db_a) < 2e-5, (rel(dg_b, dg_a), rel(db_b, db_a))
    # fp64 reference of the two sums from the stored dy and z1
    dyd, zd = dy_a.double().cpu(), z1.double().cpu()
    xhat = (zd - mean.double().cpu()) * invstd.double().cpu()
    dzr = torch.where(gamma.double().cpu() * xhat + beta.double().cpu() > 0, dyd, torch.zeros((), dtype=torch.float64))
    assert rel(db_b.cpu(), dzr.sum((0, 1, 2))) < 2e-5 and rel(dg_b.cpu(), (dzr * xhat).sum((0, 1, 2))) < 2e-5
    if dtype == torch.float32:
        assert rel(dz_b, dz_a) < 1e-5
    else:
        assert rel(dz_b, dz_a) < 2e-3 and (dz_b != dz_a).float().mean().item() < 0.02      # one-ulp flips where the coefficients moved


def test_fused_reduce_is_refused_where_k2r_runs():
    from jspsr_amd import kernels as K
    assert not K.dgrad_reduce_ok(torch.bfloat16, 8, 512, 512, 64, 64, 3, 3, 1, 1)     # K2r's layers keep the separate pass
    assert K.dgrad_reduce_ok(torch.float32, 8, 512, 512, 64, 64, 3, 3, 1, 1)
    assert not K.dgrad_reduce_ok(torch.bfloat16, 8, 512, 512, 128, 128, 3, 3, 2, 1)   # stride 1 only
