"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz from the reference's own modules.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python oracle/gen_golden.py

What it does: puts /root/reference on sys.path (bytecode writing disabled -- the tree is
read-only), registers a stand-in for the one third-party op the reference needs
(``torchvision.ops.deform_conv2d``, torchvision 0.16, neither vendored nor installed here),
imports ``models.JSPSR`` / ``models.components.spn`` unmodified, runs them in fp64 on seeded
inputs and parameters, and stores inputs (or the seed that regenerates them) and expected
outputs / gradients.

The stand-in is formulation B of SURVEY.md section 8c: ``F.grid_sample(bilinear, zeros,
align_corners=True)`` per tap -- deliberately NOT the explicit-gather formulation A used by
oracle/jspsr_ref.py and by the HIP kernel, so the fixtures judge both independently.
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import jspsr_ref as R  # noqa: E402


def deform_conv2d_standin(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0),
                          dilation=(1, 1), mask=None):
    """grid_sample formulation of torchvision.ops.deform_conv2d (1 offset group, stride 1)."""
    B, C, H, W = input.shape
    Co, Ci, kh, kw = weight.shape
    assert Ci == C and tuple(stride) == (1, 1)
    dt = input.dtype
    x64 = input.double()
    ys = torch.arange(H, dtype=torch.float64).view(1, H, 1)
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, W)
    out = torch.zeros(B, Co, H, W, dtype=torch.float64)
    for k in range(kh * kw):
        i, j = divmod(k, kw)
        py = ys - padding[0] + i * dilation[0] + offset[:, 2 * k].double()
        px = xs - padding[1] + j * dilation[1] + offset[:, 2 * k + 1].double()
        grid = torch.stack((2 * px / (W - 1) - 1, 2 * py / (H - 1) - 1), -1)
        s = F.grid_sample(x64, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        if mask is not None:
            s = s * mask[:, k : k + 1].double()
        out = out + torch.einsum("bchw,oc->bohw", s, weight[:, :, i, j].double())
    if bias is not None:
        out = out + bias.double().view(1, -1, 1, 1)
    return out.to(dt)


def import_reference():
    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    ops.deform_conv2d = deform_conv2d_standin
    tv.ops = ops
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = ops
    sys.path.insert(0, REF)
    import models.JSPSR as ref_jspsr  # noqa
    import models.components.spn as ref_spn  # noqa

    return ref_jspsr, ref_spn


def prop_inputs(seed=1234, B=2, H=20, W=24):
    """Seeded PostProcessor operands incl. integer offsets and far out-of-raster taps."""
    g = torch.Generator().manual_seed(seed)
    dem = torch.rand(B, 1, H, W, generator=g, dtype=torch.float64)
    weight = torch.sigmoid(torch.randn(B, 9, H, W, generator=g, dtype=torch.float64))
    offset = 2.5 * torch.randn(B, 18, H, W, generator=g, dtype=torch.float64)
    offset[:, :, :4, :6] = torch.randint(-3, 4, (B, 18, 4, 6), generator=g).double()  # integer taps
    offset[:, :, 4:7, :6] = 40.0 * torch.randn(B, 18, 3, 6, generator=g, dtype=torch.float64)  # far out
    offset[:, :, 7, 0:4] = torch.tensor([-1.0, 0.0, 1.0, -2.0], dtype=torch.float64)  # exact edges
    offset[:, 8:10] = 0.0  # centre tap as Generator emits it
    w = 1 + 0.3 * torch.randn(1, 1, 3, 3, generator=g, dtype=torch.float64)
    b = 0.1 * torch.randn(1, generator=g, dtype=torch.float64)
    gout = torch.randn(B, 1, H, W, generator=g, dtype=torch.float64)
    return dem, weight, offset, w, b, gout


def gen_prop(ref_spn, path):
    dem, weight, offset, w, b, gout = prop_inputs()
    pp = ref_spn.PostProcessor(kernel_size=3, residual=True, scale=1.0).double()
    with torch.no_grad():
        pp.w.copy_(w)
        pp.b.copy_(b)
    weight.requires_grad_(True)
    offset.requires_grad_(True)
    out = pp(dem, weight, offset)
    out.backward(gout)
    np.savez(
        path,
        dem=dem.numpy(), weight=weight.detach().numpy(), offset=offset.detach().numpy(),
        w=w.numpy(), b=b.numpy(), grad_out=gout.numpy(), out=out.detach().numpy(),
        grad_weight=weight.grad.numpy(), grad_offset=offset.grad.numpy(),
        grad_w=pp.w.grad.numpy(), grad_b=pp.b.grad.numpy(),
    )
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


SAVE_GRADS = (
    "conv_dem.conv.0.weight",
    "conv_img.conv.bn.weight",
    "layer1_dem.0.downsample.0.weight",
    "layer4_dem.1.bn2.bias",
    "layer2d.dconv.1.weight",
    "conv0.camb.fc.0.weight",
    "generator.convd1.conv.0.weight",
    "generator.conv_offset.conv.0.weight",
    "generator.conv_offset.conv.0.bias",
    "generator.conv_weight.0.bias",
    "postprocessor.w",
    "postprocessor.b",
)


def gen_model(ref_jspsr, path, in_channels, nf, B, H, W, seed, training):
    shapes = R.jspsr_param_shapes(in_channels, nf)
    sd = R.make_state_dict(shapes, seed, torch.float64)
    np.random.seed(0)
    model = ref_jspsr.Model(in_channels=dict(in_channels, COP30=1), out_channels=1, num_feature=nf,
                            layers=(2, 2, 2, 2), spn=True, spn_scale=1.0)
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert list(ref_shapes.items()) == list(shapes.items()), "param table != reference state_dict"
    model = model.double()
    model.load_state_dict(sd, strict=True)
    model.train(training)
    inputs, gt = R.synthetic_batch(B, H, W, "mask" in in_channels, seed=seed + 1, dtype=torch.float64)
    pred = model(*inputs)
    loss = (pred - gt).abs().mean() + ((pred - gt) ** 2).mean()  # L1 + L2 (torch built-ins): value only
    # gradients are taken through a fixed linear probe (R.probe_gradient): see its docstring
    loss_smooth = (pred * R.probe_gradient(pred.shape, seed + 2)).mean()
    store = {
        "pred": pred.detach().numpy(), "loss": np.float64(loss.item()),
        "seed": np.int64(seed), "nf": np.int64(nf), "BHW": np.array([B, H, W]),
        "training": np.bool_(training),
        "param_abs_sum": np.float64(sum(v.double().abs().sum().item() for k, v in sd.items())),
        "input_abs_sum": np.float64(sum(t.abs().sum().item() for t in inputs) + gt.abs().sum().item()),
    }
    if training:
        loss_smooth.backward()
        names, norms = [], []
        for k, p in model.named_parameters():
            names.append(k)
            norms.append(p.grad.norm().item())
            if k in SAVE_GRADS:
                store["grad:" + k] = p.grad.numpy()
        store["grad_names"] = np.array(names)
        store["grad_norms"] = np.array(norms)
        new_sd = model.state_dict()
        for k in ("conv_img.conv.bn.running_mean", "layer3_dem.0.bn1.running_var",
                  "generator.block.bn2.running_mean"):
            store["buf:" + k] = new_sd[k].numpy()
    # fp32 run of the reference modules as well (BASELINE config 1: 1e-6 abs on output)
    model32 = ref_jspsr.Model(in_channels=dict(in_channels, COP30=1), num_feature=nf).float()
    model32.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()})
    model32.train(training)
    with torch.no_grad():
        store["pred_fp32"] = model32(*[t.float() for t in inputs]).numpy()
    np.savez(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def _store_grads(model, store, save):
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(p.grad.norm().item())
        if k in save:
            store["grad:" + k] = p.grad.numpy()
    store["grad_names"] = np.array(names)
    store["grad_norms"] = np.array(norms)


def gen_lrru(path, B, H, W, seed, training):
    """models.LRRU.Model (4 propagation steps), fp64, deterministic StoDepth (prob = 1)."""
    import models.LRRU as ref_lrru
    shapes = R.lrru_param_shapes(16)
    sd = R.make_state_dict(shapes, seed, torch.float64)
    args = types.SimpleNamespace(input_channels={"lr_dem": 1, "image": 3}, output_channels=1, kernel_size=3,
                                 bc=16, prob=1.0, dkn_residual=True)
    np.random.seed(0)
    model = ref_lrru.Model(args)
    assert [(k, tuple(v.shape)) for k, v in model.state_dict().items()] == list(shapes.items())
    model = model.double()
    model.load_state_dict(sd, strict=True)
    model.train(training)
    inputs, gt = R.synthetic_batch(B, H, W, False, seed=seed + 1, dtype=torch.float64)
    pred = model(*inputs)
    loss = ((pred - gt) ** 2).mean()
    store = {"pred": pred.detach().numpy(), "loss": np.float64(loss.item()), "seed": np.int64(seed),
             "BHW": np.array([B, H, W]), "training": np.bool_(training),
             "param_abs_sum": np.float64(sum(v.double().abs().sum().item() for v in sd.values()))}
    if training:
        (pred * R.probe_gradient(pred.shape, seed + 2)).mean().backward()
        _store_grads(model, store, ("weight_offset3.conv_weight.weight", "Post_process.w", "weight_offset3.convf1.conv.0.weight"))
    np.savez(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_edsr(path, B, H, W, seed, training):
    """models.EDSR.EDSR(scale=1, spn=True) on cat(dem, image), fp64."""
    import models.EDSR as ref_edsr
    shapes = R.edsr_param_shapes(4, 4, 32)
    sd = R.make_state_dict(shapes, seed, torch.float64)
    model = ref_edsr.EDSR(in_channels=4, out_channels=1, n_resblocks=4, n_features=32, scale=1, spn=True)
    assert [(k, tuple(v.shape)) for k, v in model.state_dict().items()] == list(shapes.items())
    model = model.double()
    model.load_state_dict(sd, strict=True)
    model.train(training)
    inputs, gt = R.synthetic_batch(B, H, W, False, seed=seed + 1, dtype=torch.float64)
    x = torch.cat(inputs, 1)
    pred = model(x)
    loss = ((pred - gt) ** 2).mean()
    store = {"pred": pred.detach().numpy(), "loss": np.float64(loss.item()), "seed": np.int64(seed),
             "BHW": np.array([B, H, W]), "training": np.bool_(training),
             "param_abs_sum": np.float64(sum(v.double().abs().sum().item() for v in sd.values()))}
    if training:
        (pred * R.probe_gradient(pred.shape, seed + 2)).mean().backward()
        _store_grads(model, store, ("entry.weight", "generator.conv_weight.0.weight", "post_layer.w"))
    np.savez(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    ref_jspsr, ref_spn = import_reference()
    out = os.path.join(REPO, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    gen_prop(ref_spn, os.path.join(out, "g1_postprocessor.npz"))
    img = {"lr_dem": 1, "image": 3}
    msk = {"lr_dem": 1, "image": 3, "mask": 15}
    gen_model(ref_jspsr, os.path.join(out, "g3_img_nf32_64_train.npz"), img, 32, 1, 64, 64, 11, True)
    gen_model(ref_jspsr, os.path.join(out, "g3_img_nf32_64_eval.npz"), img, 32, 1, 64, 64, 11, False)
    gen_model(ref_jspsr, os.path.join(out, "g3_img_nf8_b2_48x80_train.npz"), img, 8, 2, 48, 80, 12, True)
    gen_model(ref_jspsr, os.path.join(out, "g4_msk_nf8_b2_64_train.npz"), msk, 8, 2, 64, 64, 13, True)
    gen_model(ref_jspsr, os.path.join(out, "g4_msk_nf8_b2_64_eval.npz"), msk, 8, 2, 64, 64, 13, False)
    gen_lrru(os.path.join(out, "g5_lrru_b1_64_train.npz"), 1, 64, 64, 21, True)
    gen_lrru(os.path.join(out, "g5_lrru_b2_32x48_eval.npz"), 2, 32, 48, 22, False)
    gen_edsr(os.path.join(out, "g6_edsr_b2_40x56_train.npz"), 2, 40, 56, 31, True)


if __name__ == "__main__":
    main()
