"""TEST INFRASTRUCTURE ONLY -- functional CPU restatement of the JSPSR hot path.

Written against a flat ``state_dict`` (the reference's key names) so it
shares no module code with either the reference or the product package.  Every function
cites the reference lines it restates (paths relative to /root/reference).

Pinned by tests/golden/*.npz, which were produced by the reference's own modules
(oracle/gen_golden.py).  Works in fp32 or fp64; differentiable through torch autograd, and
``propagate_analytic_backward`` gives the closed-form gradients independently of autograd.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------------------
# propagation step (models/components/spn.py:99-118 + torchvision 0.16 deform_conv2d)
# --------------------------------------------------------------------------------------


def _corner(dem: torch.Tensor, yy: torch.Tensor, xx: torch.Tensor) -> torch.Tensor:
    """dem (B,1,H,W) read at integer (yy,xx) of shape (B,K,H,W); 0 outside the raster."""
    B, _, H, W = dem.shape
    ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
    idx = yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)
    flat = dem.reshape(B, 1, H * W).expand(B, idx.shape[1], H * W)
    val = torch.gather(flat, 2, idx.reshape(B, idx.shape[1], H * W)).reshape(idx.shape)
    return torch.where(ok, val, torch.zeros((), dtype=dem.dtype))


def sample_taps(dem: torch.Tensor, offset: torch.Tensor) -> torch.Tensor:
    """Formulation A (explicit 4-corner gather) of the deformable 3x3 sampler.

    torchvision semantics: offset channel 2k = dy, 2k+1 = dx of tap k
    (row-major over the window); position p = (y-1+k//3+dy, x-1+k%3+dx); bilinear with
    out-of-raster corners contributing 0 and the whole sample 0 when p_y<=-1, p_y>=H,
    p_x<=-1 or p_x>=W.  Returns S of shape (B,9,H,W).
    """
    B, _, H, W = dem.shape
    dt = dem.dtype
    ys = torch.arange(H, dtype=dt).view(1, 1, H, 1)
    xs = torch.arange(W, dtype=dt).view(1, 1, 1, W)
    ky = torch.tensor([k // 3 - 1 for k in range(9)], dtype=dt).view(1, 9, 1, 1)
    kx = torch.tensor([k % 3 - 1 for k in range(9)], dtype=dt).view(1, 9, 1, 1)
    off = offset.reshape(B, 9, 2, H, W)
    py = ys + ky + off[:, :, 0]
    px = xs + kx + off[:, :, 1]
    y0f = torch.floor(py)
    x0f = torch.floor(px)
    ly = py - y0f
    lx = px - x0f
    hy = 1 - ly
    hx = 1 - lx
    y0 = y0f.long()
    x0 = x0f.long()
    v00 = _corner(dem, y0, x0)
    v01 = _corner(dem, y0, x0 + 1)
    v10 = _corner(dem, y0 + 1, x0)
    v11 = _corner(dem, y0 + 1, x0 + 1)
    val = hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11
    inside = (py > -1) & (py < H) & (px > -1) & (px < W)
    return torch.where(inside, val, torch.zeros((), dtype=dt))


def propagate(
    dem: torch.Tensor,
    weight: torch.Tensor,
    offset: torch.Tensor,
    w: torch.Tensor,
    b: torch.Tensor,
    scale: float = 1.0,
    residual: bool = True,
) -> torch.Tensor:
    """PostProcessor.forward, models/components/spn.py:99-118.

    dem (B,1,H,W), weight (B,9,H,W), offset (B,18,H,W), w (1,1,3,3), b (1,) -> (B,1,H,W).
    """
    if residual:
        m = weight - weight.mean(1, keepdim=True)  # spn.py:100-101
    else:
        m = weight / weight.sum(1, keepdim=True)  # spn.py:103
    S = sample_taps(dem, offset)
    out = (w.reshape(1, 9, 1, 1) * m * S).sum(1, keepdim=True) + b.reshape(1, 1, 1, 1)
    if residual:
        out = out + scale * dem  # spn.py:116-117
    return out


def propagate_analytic_backward(dem, weight, offset, w, b, grad_out):
    """Closed-form gradients of ``propagate`` (residual=True) -- SURVEY.md section 8a row a10.

    Returns (grad_weight, grad_offset, grad_w, grad_b).  Coordinate derivative follows
    torchvision's get_coordinate_weight: per-corner validity only (no 'inside' gate).
    """
    B, _, H, W = dem.shape
    dt = dem.dtype
    ys = torch.arange(H, dtype=dt).view(1, 1, H, 1)
    xs = torch.arange(W, dtype=dt).view(1, 1, 1, W)
    ky = torch.tensor([k // 3 - 1 for k in range(9)], dtype=dt).view(1, 9, 1, 1)
    kx = torch.tensor([k % 3 - 1 for k in range(9)], dtype=dt).view(1, 9, 1, 1)
    off = offset.reshape(B, 9, 2, H, W)
    py = ys + ky + off[:, :, 0]
    px = xs + kx + off[:, :, 1]
    y0f, x0f = torch.floor(py), torch.floor(px)
    ly, lx = py - y0f, px - x0f
    hy, hx = 1 - ly, 1 - lx
    y0, x0 = y0f.long(), x0f.long()
    v00 = _corner(dem, y0, x0)
    v01 = _corner(dem, y0, x0 + 1)
    v10 = _corner(dem, y0 + 1, x0)
    v11 = _corner(dem, y0 + 1, x0 + 1)
    inside = ((py > -1) & (py < H) & (px > -1) & (px < W)).to(dt)
    S = inside * (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11)
    dSdy = hx * (v10 - v00) + lx * (v11 - v01)
    dSdx = hy * (v01 - v00) + ly * (v11 - v10)
    m = weight - weight.mean(1, keepdim=True)
    wk = w.reshape(1, 9, 1, 1)
    g = grad_out
    gm = g * wk * S
    grad_weight = gm - gm.mean(1, keepdim=True)
    coef = g * wk * m
    grad_offset = torch.stack((coef * dSdy, coef * dSdx), 2).reshape(B, 18, H, W)
    grad_w = (g * m * S).sum((0, 2, 3)).reshape(1, 1, 3, 3)
    grad_b = g.sum().reshape(1)
    return grad_weight, grad_offset, grad_w, grad_b


# --------------------------------------------------------------------------------------
# building blocks (models/components/basics.py, resnet_cbam.py:36-53)
# --------------------------------------------------------------------------------------


class Ctx:
    """Carries the flat state dict and the train/eval switch through the functional graph."""

    def __init__(self, sd: SD, training: bool, momentum: float = 0.1, eps: float = 1e-5):
        self.sd = sd
        self.training = training
        self.momentum = momentum
        self.eps = eps

    def __getitem__(self, k):
        return self.sd[k]

    def get(self, k):
        return self.sd.get(k)


def batch_norm(c: Ctx, x, p):
    """nn.BatchNorm2d (basics.py:49,81,105,108); updates running stats in train mode."""
    nbt = c.get(p + ".num_batches_tracked")
    if c.training and nbt is not None:
        nbt += 1
    return F.batch_norm(
        x,
        c[p + ".running_mean"],
        c[p + ".running_var"],
        c[p + ".weight"],
        c[p + ".bias"],
        c.training,
        c.momentum,
        c.eps,
    )


def channel_attention(c: Ctx, x, p):
    """ChannelAttention.forward, resnet_cbam.py:49-53 (shared bias-free 1x1 MLP)."""
    w1, w2 = c[p + ".fc.0.weight"], c[p + ".fc.2.weight"]

    def mlp(v):
        return F.conv2d(F.relu(F.conv2d(v, w1)), w2)

    avg = x.mean((2, 3), keepdim=True)
    mx = x.amax((2, 3), keepdim=True)
    return torch.sigmoid(mlp(avg) + mlp(mx))


def basic2d(c: Ctx, x, p, k=3, bn=True, relu=True, camb=False):
    """Basic2d.forward, basics.py:55-60 (conv has bias iff bn is False, :36)."""
    if camb:
        x = channel_attention(c, x, p + ".camb") * x
    y = F.conv2d(x, c[p + ".conv.0.weight"], c.get(p + ".conv.0.bias"), 1, k // 2)
    if bn:
        y = batch_norm(c, y, p + ".conv.bn")
    return F.relu(y) if relu else y


def basic2d_trans(c: Ctx, x, p):
    """Basic2dTrans.forward, basics.py:63-85: Basic2d(camb) -> ConvT k3 s2 p1 op1 -> BN -> ReLU."""
    y = basic2d(c, x, p + ".dconv.0", 3, bn=True, relu=True, camb=True)
    y = F.conv_transpose2d(y, c[p + ".dconv.1.weight"], None, 2, 1, 1)
    return F.relu(batch_norm(c, y, p + ".dconv.bn"))


def basic_block(c: Ctx, x, p, stride=1, act=True, scale=1.0):
    """BasicBlock.forward, basics.py:111-123."""
    y = F.conv2d(x, c[p + ".conv1.weight"], None, stride, 1)
    y = F.relu(batch_norm(c, y, p + ".bn1"))
    y = F.conv2d(y, c[p + ".conv2.weight"], None, 1, 1)
    y = batch_norm(c, y, p + ".bn2")
    if (p + ".downsample.0.weight") in c.sd:
        r = F.conv2d(x, c[p + ".downsample.0.weight"], None, stride, 0)
        r = batch_norm(c, r, p + ".downsample.1")
    else:
        r = x
    y = y * scale + r
    return F.relu(y) if act else y


def layer(c: Ctx, x, p, n_blocks, stride):
    """nn.Sequential of BasicBlocks built by _make_layer, JSPSR.py:382-492."""
    for i in range(n_blocks):
        x = basic_block(c, x, f"{p}.{i}", stride if i == 0 else 1)
    return x


def generator(c: Ctx, dem, ctx_feat, p="generator"):
    """Generator.forward, spn.py:54-75 -> weight (B,9,H,W) in (0,1), offset (B,18,H,W)."""
    B, _, H, W = dem.shape
    d = basic2d(c, basic2d(c, dem, p + ".convd1", bn=False), p + ".convd2", bn=False)
    f = basic2d(c, basic2d(c, ctx_feat, p + ".convf1", bn=False), p + ".convf2", bn=False)
    x = basic2d(c, torch.cat((d, f), 1), p + ".conv", bn=False)
    x = basic_block(c, x, p + ".block")
    weight = torch.sigmoid(
        F.conv2d(x, c[p + ".conv_weight.0.weight"], c[p + ".conv_weight.0.bias"])
    )
    off16 = F.conv2d(x, c[p + ".conv_offset.conv.0.weight"], c[p + ".conv_offset.conv.0.bias"])
    zero = torch.zeros(B, 2, H, W, dtype=off16.dtype)
    offset = torch.cat((off16[:, :8], zero, off16[:, 8:]), 1)  # centre tap: spn.py:70-73
    return weight, offset


# --------------------------------------------------------------------------------------
# the model (models/JSPSR.py:208-380)
# --------------------------------------------------------------------------------------


def jspsr_forward(
    sd: SD,
    inputs: Sequence[torch.Tensor],
    training: bool,
    layers: Sequence[int] = (2, 2, 2, 2),
    spn_scale: float = 1.0,
    return_aux: bool = False,
):
    """JSPSR Model.forward for inputs [dem, img] or [dem, img, msk] (JSPSR.py:208-380).

    Branch set is inferred from the state dict (conv_img / conv_aux present or not).
    """
    c = Ctx(sd, training)
    has_img = "conv_img.conv.0.weight" in sd
    has_aux = "conv_aux.conv.0.weight" in sd
    if len(inputs) != 1 + int(has_img) + int(has_aux):
        raise NotImplementedError  # parse_input, JSPSR.py:519-550
    dem = inputs[0]
    feats = {"dem": basic2d(c, dem, "conv_dem", 5, bn=False)}  # JSPSR.py:66-68,220
    if has_img:
        feats["img"] = basic2d(c, inputs[1], "conv_img", 5, bn=True)  # :70,221
    if has_aux:
        feats["aux"] = basic2d(c, inputs[-1], "conv_aux", 5, bn=False)  # :75-77,222
    order = [k for k in ("dem", "img", "aux") if k in feats]
    fuse = []
    for stage in range(4):  # JSPSR.py:230-352
        stride = 1 if stage == 0 else 2
        nxt = {}
        for br in order:
            src = feats[br]
            if br == "dem" and stage > 0:
                src = fuse[-1]  # dem branch consumes the fused tensor, :261,292,323
            nxt[br] = layer(c, src, f"layer{stage + 1}_{br}", layers[stage], stride)
        feats = nxt
        fuse.append(torch.cat([feats[b] for b in order], 1))  # Guide(cat_only), basics.py:134
    x = fuse[3]
    for name, skip in (("layer3d", fuse[2]), ("layer2d", fuse[1]), ("layer1d", fuse[0])):
        x = torch.cat((basic2d_trans(c, x, name), skip), 1)  # :354-368
    c0 = basic2d(c, x, "conv0", 3, bn=True, relu=True, camb=True)  # :369
    dem_d = dem.detach()  # :372
    weight, offset = generator(c, dem_d, c0)
    out = propagate(dem_d, weight, offset, sd["postprocessor.w"], sd["postprocessor.b"], spn_scale)
    if return_aux:
        return out, {"c0": c0, "weight": weight, "offset": offset}
    return out


# --------------------------------------------------------------------------------------
# parameter sets (JSPSR.py:10-206 shapes, :494-517 init)
# --------------------------------------------------------------------------------------


def jspsr_param_shapes(in_channels: dict, num_feature=32, layers=(2, 2, 2, 2)) -> Dict[str, tuple]:
    """Shapes of every state-dict entry of models.JSPSR.Model, in the reference's order."""
    nf = num_feature
    has_img = "image" in in_channels
    aux_key = next((k for k in ("mask", "canopy", "coord") if k in in_channels), None)
    nb = 1 + int(has_img) + int(aux_key is not None)
    out: Dict[str, tuple] = {}

    def conv(name, co, ci, k, bias):
        out[name + ".weight"] = (co, ci, k, k)
        if bias:
            out[name + ".bias"] = (co,)

    def bn(name, ch):
        out[name + ".weight"] = (ch,)
        out[name + ".bias"] = (ch,)
        out[name + ".running_mean"] = (ch,)
        out[name + ".running_var"] = (ch,)
        out[name + ".num_batches_tracked"] = ()

    def b2d(name, ci, co, k, use_bn, camb=False):
        if camb:
            out[name + ".camb.fc.0.weight"] = (ci // 16, ci, 1, 1)
            out[name + ".camb.fc.2.weight"] = (ci, ci // 16, 1, 1)
        conv(name + ".conv.0", co, ci, k, not use_bn)
        if use_bn:
            bn(name + ".conv.bn", co)

    def block(name, ci, co, down):
        conv(name + ".conv1", co, ci, 3, False)
        bn(name + ".bn1", co)
        conv(name + ".conv2", co, co, 3, False)
        bn(name + ".bn2", co)
        if down:
            conv(name + ".downsample.0", co, ci, 1, False)
            bn(name + ".downsample.1", co)

    branches = ["dem"] + (["img"] if has_img else []) + (["aux"] if aux_key else [])
    b2d("conv_dem", in_channels["lr_dem"], nf, 5, False)
    if has_img:
        b2d("conv_img", in_channels["image"], nf, 5, True)
    if aux_key:
        b2d("conv_aux", in_channels[aux_key], nf, 5, False)
    inpl = nf
    for s in range(4):
        planes = nf * 2 * (2**s)
        for br in branches:
            ci = inpl * (nb if (br == "dem" and s > 0) else 1)
            for i in range(layers[s]):
                block(f"layer{s + 1}_{br}.{i}", ci if i == 0 else planes, planes, i == 0)
        inpl = planes
    for name, ci, co in (
        ("layer3d", nf * 16 * nb, nf * 8),
        ("layer2d", nf * 8 + nf * 8 * nb, nf * 4),
        ("layer1d", nf * 4 + nf * 4 * nb, nf * 2),
    ):
        b2d(name + ".dconv.0", ci, co, 3, True, camb=True)
        out[name + ".dconv.1.weight"] = (co, co, 3, 3)
        bn(name + ".dconv.bn", co)
    b2d("conv0", nf * 2 + nf * 2 * nb, nf * 2, 3, True, camb=True)
    bc = nf
    g = "generator"
    b2d(g + ".convd1", 1, bc * 2, 3, False)
    b2d(g + ".convd2", bc * 2, bc * 2, 3, False)
    b2d(g + ".convf1", nf * 2, bc * 2, 3, False)
    b2d(g + ".convf2", bc * 2, bc * 2, 3, False)
    b2d(g + ".conv", bc * 4, bc * 4, 3, False)
    block(g + ".block", bc * 4, bc * 4, False)
    conv(g + ".conv_weight.0", 9, bc * 4, 1, True)
    conv(g + ".conv_offset.conv.0", 16, bc * 4, 1, True)
    out["postprocessor.w"] = (1, 1, 3, 3)
    out["postprocessor.b"] = (1,)
    return out


def _trunc_normal(rs, shape, std: float) -> torch.Tensor:
    """Truncated normal on [-2 std, 2 std] by inverse CDF of uniform draws of `rs` (fp64)."""
    u = torch.from_numpy(rs.random_sample(int(np.prod(shape)))).reshape(shape)
    lo = 0.5 * (1.0 + math.erf(-2.0 / math.sqrt(2.0)))
    hi = 0.5 * (1.0 + math.erf(2.0 / math.sqrt(2.0)))
    p = (lo + u * (hi - lo)).clamp(1e-12, 1 - 1e-12)
    return (std * math.sqrt(2.0)) * torch.erfinv(2.0 * p - 1.0)


def _randn(rs, shape) -> torch.Tensor:
    return torch.from_numpy(rs.standard_normal(int(np.prod(shape)))).reshape(shape)


def make_state_dict(shapes: Dict[str, tuple], seed: int, dtype=torch.float32) -> SD:
    """Deterministic parameter set with the reference's init *distribution*
    (JSPSR.py:494-517: truncated normal +-2 sigma, sigma = sqrt(2.6/(k*k*C_in)); bias 0; BN 1/0).
    Drawn from numpy's legacy ``RandomState(seed)``, whose stream is frozen by NumPy's compatibility
    policy (NEP 19), so the fixtures can be regenerated on any box and any torch build; the fixtures store
    a checksum of these values and the tests FAIL (not skip) on a mismatch.
    BN affine/running stats and conv biases are perturbed a little so parity tests see them.
    """
    rs = np.random.RandomState(seed)
    sd: SD = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif len(shp) == 4 and k != "postprocessor.w":
            n = shp[1] * shp[2] * shp[3]
            if ".dconv.1." in k:  # ConvTranspose2d: in_channels is dim 0
                n = shp[0] * shp[2] * shp[3]
            sd[k] = _trunc_normal(rs, shp, math.sqrt(2.6 / n)).to(dtype)
        elif k == "postprocessor.w":
            sd[k] = (1 + 0.2 * _randn(rs, shp)).to(dtype)
        elif k == "postprocessor.b":
            sd[k] = (0.01 * _randn(rs, shp)).to(dtype)
        elif k.endswith("running_var"):
            sd[k] = (1 + 0.2 * torch.from_numpy(rs.random_sample(int(np.prod(shp)))).reshape(shp)).to(dtype)
        elif k.endswith("running_mean"):
            sd[k] = (0.1 * _randn(rs, shp)).to(dtype)
        elif ".bn" in k or ".downsample.1." in k:
            base = 1.0 if k.endswith("weight") else 0.0
            sd[k] = (base + 0.1 * _randn(rs, shp)).to(dtype)
        else:  # conv bias
            sd[k] = (0.05 * _randn(rs, shp)).to(dtype)
    return sd


def checksum(tensors) -> float:
    """Order-sensitive fp64 checksum of a sequence of tensors (fixtures store it; tests compare it)."""
    s = 0.0
    for i, t in enumerate(tensors):
        t = t.double().reshape(-1)
        s += float((t * torch.cos(torch.arange(t.numel(), dtype=torch.float64) * 0.37 + i)).sum()) + float(t.abs().sum())
    return s


def synthetic_batch(B, H, W, with_mask: bool, seed=0, dtype=torch.float32):
    """Synthetic inputs of SURVEY.md section 8d: smooth DEM in the log-min-max range
    (data/data_utils.py:289-312), uint8 image /255 (:225-227), block one-hot mask with channel
    i scaled by (i+1)/16 (:262-265).  Returns (inputs list, target).  Random draws: numpy's frozen legacy
    RandomState stream (see make_state_dict)."""
    rs = np.random.RandomState(seed)
    z = torch.zeros(B, 1, H, W, dtype=torch.float64)
    for o in range(4):
        n = max(2, min(H, W) // (32 >> o) if (32 >> o) > 0 else 2)
        noise = _randn(rs, (B, 1, n, n))
        z = z + F.interpolate(noise, size=(H, W), mode="bilinear", align_corners=True) / (2**o)
    z = (z - z.amin((2, 3), keepdim=True)) / (z.amax((2, 3), keepdim=True) - z.amin((2, 3), keepdim=True) + 1e-12)
    z = z * 120.0
    lr = torch.log(z + 80.0) / math.log(1009.0)
    hr = torch.log((z + _randn(rs, z.shape)).clamp_min(-79.0) + 80.0) / math.log(1009.0)
    img = torch.from_numpy(rs.randint(0, 256, (B, 3, H, W))).to(torch.float64) / 255.0
    inputs = [lr.to(dtype), img.to(dtype)]
    if with_mask:
        bs = 32
        cls = torch.from_numpy(rs.randint(0, 15, (B, (H + bs - 1) // bs, (W + bs - 1) // bs)))
        cls = cls.repeat_interleave(bs, 1).repeat_interleave(bs, 2)[:, :H, :W]
        msk = F.one_hot(cls, 15).permute(0, 3, 1, 2).to(torch.float64)
        msk = msk * ((torch.arange(15, dtype=torch.float64) + 1) / 16).view(1, 15, 1, 1)
        inputs.append(msk.to(dtype))
    return inputs, hr.to(dtype)


# --------------------------------------------------------------------------------------
# loss (losses/loss_schemes.py:55-72 with configs/*.yml:67-70: L1 1, L2 1, Grad 0.1)
# --------------------------------------------------------------------------------------


def sobel_gradient(x: torch.Tensor) -> torch.Tensor:
    """kornia.filters.spatial_gradient(mode='sobel', order=1, normalized=True) restated:
    replicate padding, kernels [[-1,0,1],[-2,0,2],[-1,0,1]]/8 and its transpose (SURVEY 8c:
    recalled from kornia's public source; unpinned).  (B,C,H,W) -> (B,C,2,H,W)."""
    B, C, H, W = x.shape
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype) / 8.0
    k = torch.stack((kx, kx.t())).unsqueeze(1)
    xp = F.pad(x.reshape(B * C, 1, H, W), (1, 1, 1, 1), mode="replicate")
    return F.conv2d(xp, k).reshape(B, C, 2, H, W)


def multi_loss(pred, gt, w_l1=1.0, w_l2=1.0, w_grad=0.1):
    """MultiLoss(L1, L2, Grad): loss_schemes.py:61-72, loss_functions.py:171-185."""
    l1 = (pred - gt).abs().mean()
    l2 = ((pred - gt) ** 2).mean()
    ge = (sobel_gradient(pred) - sobel_gradient(gt)).abs().mean()
    return {"L1": l1, "L2": l2, "Grad": ge, "Total": w_l1 * l1 + w_l2 * l2 + w_grad * ge}


# --------------------------------------------------------------------------------------
# LRRU baseline re-targeted to DEMs (models/LRRU.py:301-507): 4 propagation steps
# --------------------------------------------------------------------------------------


def lrru_param_shapes(bc: int = 16, layers=(2, 2, 2, 2, 2)) -> Dict[str, tuple]:
    """state_dict shapes of models.LRRU.Model(args) with args.bc = bc, in the reference's order."""
    out: Dict[str, tuple] = {}
    c = bc * 2

    def conv(name, co, ci, k, bias):
        out[name + ".weight"] = (co, ci, k, k)
        if bias:
            out[name + ".bias"] = (co,)

    def bn(name, ch):
        for s, shp in (("weight", (ch,)), ("bias", (ch,)), ("running_mean", (ch,)), ("running_var", (ch,)),
                       ("num_batches_tracked", ())):
            out[f"{name}.{s}"] = shp

    def b2d(name, ci, co, k, norm):
        conv(name + ".conv.0", co, ci, k, not norm)
        if norm:
            bn(name + ".conv.bn", co)

    def block(name, ci, co, down):
        conv(name + ".conv1", co, ci, 3, False)
        bn(name + ".bn1", co)
        conv(name + ".conv2", co, co, 3, False)
        bn(name + ".bn2", co)
        if down:
            conv(name + ".downsample.0", co, ci, 1, False)
            bn(name + ".downsample.1", co)

    def trans(name, ci, co):
        out[name + ".conv.weight"] = (ci, co, 3, 3)
        bn(name + ".bn", co)

    def enc(name):
        b2d(name + ".convd1", 1, c, 3, False)
        b2d(name + ".convd2", c, c, 3, False)
        b2d(name + ".convf1", c, c, 3, False)
        b2d(name + ".convf2", c, c, 3, False)
        b2d(name + ".conv", 2 * c, 2 * c, 3, False)
        block(name + ".ref", 2 * c, 2 * c, False)
        conv(name + ".conv_weight", 9, 2 * c, 1, True)
        conv(name + ".conv_offset", 16, 2 * c, 1, True)

    b2d("conv_img", 3, c, 5, True)
    b2d("conv_lidar", 1, c, 5, False)
    planes = [2 * c, 4 * c, 8 * c, 8 * c, 8 * c]
    inpl = c
    for s in range(5):
        stride = 1 if s == 0 else 2
        for br in ("img", "lidar"):
            for i in range(layers[s]):
                ci = inpl if i == 0 else planes[s]
                block(f"layer{s + 1}_{br}.{i}", ci, planes[s], i == 0 and (stride != 1 or inpl != planes[s]))
        if s < 4:
            b2d(f"guide{s + 1}.conv", planes[s] * 2, planes[s], 3, True)
        inpl = planes[s]
    trans("layer4d", 8 * c, 8 * c)
    for i, (ci, co) in enumerate(((8 * c, 4 * c), (4 * c, 2 * c), (2 * c, c))):
        trans(f"upproj0.{i}", ci, co)
    enc("weight_offset0")
    trans("layer3d", 8 * c, 8 * c)
    for i, (ci, co) in enumerate(((8 * c, 4 * c), (4 * c, c))):
        trans(f"upproj1.{i}", ci, co)
    enc("weight_offset1")
    trans("layer2d", 8 * c, 4 * c)
    trans("upproj2.0", 4 * c, c)
    enc("weight_offset2")
    trans("layer1d", 4 * c, 2 * c)
    b2d("conv", 2 * c, c, 3, True)
    enc("weight_offset3")
    out["Post_process.w"] = (1, 1, 3, 3)
    out["Post_process.b"] = (1,)
    return out


def lrru_forward(sd: SD, inputs: Sequence[torch.Tensor], training: bool, layers=(2, 2, 2, 2, 2)):
    """models.LRRU.Model.forward (LRRU.py:403-507) with prob = 1 (deterministic StoDepth blocks)."""
    c = Ctx(sd, training)
    if len(inputs) != 2:
        raise NotImplementedError
    depth, img = inputs

    def b2d(x, p, k=3, norm=False):
        return basic2d(c, x, p, k, bn=norm, relu=True)

    def trans(x, p):  # Basic2dTrans, LRRU.py:67-88
        y = F.conv_transpose2d(x, c[p + ".conv.weight"], None, 2, 1, 1)
        return F.relu(batch_norm(c, y, p + ".bn"))

    def guide(feat, weight, p):  # LRRU.py:188-200
        return b2d(torch.cat((feat, weight), 1), p + ".conv", 3, True)

    def encoder(depth_, ctx_, p):  # BasicDepthEncoder.forward, LRRU.py:226-247
        B, _, H, W = depth_.shape
        d = b2d(b2d(depth_, p + ".convd1"), p + ".convd2")
        f = b2d(b2d(ctx_, p + ".convf1"), p + ".convf2")
        x = b2d(torch.cat((d, f), 1), p + ".conv")
        x = basic_block(c, x, p + ".ref", act=False)
        weight = torch.sigmoid(F.conv2d(x, c[p + ".conv_weight.weight"], c[p + ".conv_weight.bias"]))
        off16 = F.conv2d(x, c[p + ".conv_offset.weight"], c[p + ".conv_offset.bias"])
        zero = torch.zeros(B, 2, H, W, dtype=off16.dtype)
        return weight, torch.cat((off16[:, :8], zero, off16[:, 8:]), 1)

    def post(d, w, o):  # Post_process_deconv.forward, LRRU.py:267-298
        return propagate(d, w, o, sd["Post_process.w"], sd["Post_process.b"], 1.0)

    def keep_input(out):  # preserve_input blend, LRRU.py:447-450 etc.
        mask = ((depth > 0.0).sum(1, keepdim=True) > 0.0).to(depth.dtype)
        return (1.0 - mask) * out + mask * depth

    c0_img = b2d(img, "conv_img", 5, True)
    c0_lidar = b2d(depth, "conv_lidar", 5, False)
    fi, fl = c0_img, c0_lidar
    dyn = []
    for s in range(5):
        stride = 1 if s == 0 else 2
        fi = layer(c, fi, f"layer{s + 1}_img", layers[s], stride)
        fl = layer(c, fl, f"layer{s + 1}_lidar", layers[s], stride)
        if s < 4:
            fl = guide(fl, fi, f"guide{s + 1}")
            dyn.append(fl)
    c5 = fi + fl
    c4 = trans(c5, "layer4d") + dyn[3]
    up = c4
    for i in range(3):
        up = trans(up, f"upproj0.{i}")
    lidar = keep_input(depth).detach()
    out = post(lidar, *encoder(lidar, up, "weight_offset0"))
    c3 = trans(c4, "layer3d") + dyn[2]
    up = trans(trans(c3, "upproj1.0"), "upproj1.1")
    out = keep_input(out).detach()
    out = post(out, *encoder(out, up, "weight_offset1"))
    c2 = trans(c3, "layer2d") + dyn[1]
    up = trans(c2, "upproj2.0")
    out = keep_input(out).detach()
    out = post(out, *encoder(out, up, "weight_offset2"))
    c1 = trans(c2, "layer1d") + dyn[0]
    c0 = b2d(c1, "conv", 3, True) + c0_lidar
    out = keep_input(out).detach()
    return post(out, *encoder(out, c0, "weight_offset3"))


# --------------------------------------------------------------------------------------
# EDSR trunk + the same generator / propagation head (models/EDSR.py:66-137, spn=True)
# --------------------------------------------------------------------------------------


def edsr_param_shapes(in_channels: int, n_resblocks=16, n_features=64) -> Dict[str, tuple]:
    out: Dict[str, tuple] = {}
    F_ = n_features

    def conv(name, co, ci, k):
        out[name + ".weight"] = (co, ci, k, k)
        out[name + ".bias"] = (co,)

    conv("entry", F_, in_channels, 3)
    for i in range(n_resblocks):
        conv(f"encoder.{i}.body.0", F_, F_, 3)
        conv(f"encoder.{i}.body.2", F_, F_, 3)
    conv(f"encoder.{n_resblocks}", F_, F_, 3)
    g, bc = "generator", F_ // 2
    for name, ci, co in ((".convd1", 1, bc * 2), (".convd2", bc * 2, bc * 2), (".convf1", F_, bc * 2),
                         (".convf2", bc * 2, bc * 2), (".conv", bc * 4, bc * 4)):
        conv(g + name + ".conv.0", co, ci, 3)
    for nm in ("conv1", "conv2"):
        out[f"{g}.block.{nm}.weight"] = (bc * 4, bc * 4, 3, 3)
        for s, shp in (("weight", (bc * 4,)), ("bias", (bc * 4,)), ("running_mean", (bc * 4,)),
                       ("running_var", (bc * 4,)), ("num_batches_tracked", ())):
            out[f"{g}.block.bn{nm[-1]}.{s}"] = shp
    # reference order inside BasicBlock: conv1, bn1, conv2, bn2
    ordered = {}
    for k in list(out):
        if k.startswith(g + ".block."):
            ordered[k] = out.pop(k)
    for nm in ("1", "2"):
        ordered_keys = [f"{g}.block.conv{nm}.weight"] + [f"{g}.block.bn{nm}.{s}" for s in
                        ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
        for k in ordered_keys:
            out[k] = ordered[k]
    out[g + ".conv_weight.0.weight"] = (9, bc * 4, 1, 1)
    out[g + ".conv_weight.0.bias"] = (9,)
    out[g + ".conv_offset.conv.0.weight"] = (16, bc * 4, 1, 1)
    out[g + ".conv_offset.conv.0.bias"] = (16,)
    out["post_layer.w"] = (1, 1, 3, 3)
    out["post_layer.b"] = (1,)
    return out


def edsr_forward(sd: SD, x: torch.Tensor, training: bool, n_resblocks=16, res_scale=0.1):
    """EDSR.forward with scale=1, spn=True (EDSR.py:123-137); x = cat(dem, guides) (B,C,H,W)."""
    c = Ctx(sd, training)
    dem = x[:, 0:1].detach()
    xs = F.conv2d(x, sd["entry.weight"], sd["entry.bias"], 1, 1)
    h = xs
    for i in range(n_resblocks):  # ResBlock.forward, EDSR.py:40-44
        r = F.conv2d(h, sd[f"encoder.{i}.body.0.weight"], sd[f"encoder.{i}.body.0.bias"], 1, 1)
        r = F.conv2d(F.relu(r), sd[f"encoder.{i}.body.2.weight"], sd[f"encoder.{i}.body.2.bias"], 1, 1)
        h = r * res_scale + h
    h = F.conv2d(h, sd[f"encoder.{n_resblocks}.weight"], sd[f"encoder.{n_resblocks}.bias"], 1, 1)
    h = h + res_scale * xs
    weight, offset = generator(c, dem, h)
    return propagate(dem, weight, offset, sd["post_layer.w"], sd["post_layer.b"], 1.0)


def probe_gradient(shape, seed: int, dtype=torch.float64) -> torch.Tensor:
    """Fixed upstream gradient for backward-pass fixtures: d(loss)/d(pred) = G / numel with a seeded random
    G.  A data loss would make d(loss)/d(pred) = f(pred - gt); with |pred - gt| ~ 1e-3 an fp32 forward error of
    5e-6 turns into a 0.5 % relative error of every gradient, which says nothing about the backward kernels."""
    return _randn(np.random.RandomState(seed), tuple(shape)).to(dtype)
