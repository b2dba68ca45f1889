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
        "param_checksum": np.float64(R.checksum(sd.values())),
        "input_checksum": np.float64(R.checksum(list(inputs) + [gt])),
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
             "param_checksum": np.float64(R.checksum(sd.values())),
             "input_checksum": np.float64(R.checksum(list(inputs) + [gt]))}
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
             "param_checksum": np.float64(R.checksum(sd.values())),
             "input_checksum": np.float64(R.checksum(list(inputs) + [gt]))}
    if training:
        (pred * R.probe_gradient(pred.shape, seed + 2)).mean().backward()
        _store_grads(model, store, ("entry.weight", "generator.conv_weight.0.weight", "post_layer.w"))
    np.savez(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_init_stream(ref_jspsr, path, in_channels, nf, seed):
    """a11: the reference's `_initialize_weights` (models/JSPSR.py:494-517) under np.random.seed(seed).  The product
    module must draw the same stream (same module order, same scipy sampler): compared here bit for bit against the
    imported reference, and summarised (per-tensor sum / abs-sum / first and last values) for the box without it."""
    from jspsr_amd.JSPSR import Model
    np.random.seed(seed)
    ref = ref_jspsr.Model(in_channels=dict(in_channels, COP30=1), out_channels=1, num_feature=nf,
                          layers=(2, 2, 2, 2), spn=True, spn_scale=1.0).state_dict()
    np.random.seed(seed)
    mine = Model(dict(in_channels, COP30=1), num_feature=nf).state_dict()
    assert list(ref) == list(mine), "state_dict key order differs"
    for k in ref:
        assert ref[k].dtype == mine[k].dtype and torch.equal(ref[k], mine[k]), f"init stream differs at {k}"
    names = list(ref)
    summ = np.zeros((len(names), 4))
    for i, k in enumerate(names):
        t = ref[k].double().reshape(-1)
        summ[i] = (t.sum().item(), t.abs().sum().item(), t[0].item(), t[-1].item())
    np.savez(path, names=np.array(names), summary=summ, seed=np.int64(seed), nf=np.int64(nf),
             with_mask=np.bool_("mask" in in_channels))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB (reference init == product init, bit for bit,", len(names), "tensors)")


def gen_host_side(path):
    """g7: the rows either side of the hot path, made by the reference's own code where that code
    runs here: evaluation.metrics Meter{RMSE,Median,NMAD,LE95} (package "local": torch only; metrics.py:338-590 incl.
    MeterBase._prepare :147-199), data.data_utils ToTensor.scale_data / ToDEM.descale_data (:289-312,:441-457) and
    TileCrop (:87-194), losses.loss_schemes.MultiLoss over torch's L1Loss/MSELoss (:6-12,:55-72).
    Their files import packages this image lacks (piq, skimage, kornia, richdem, hide_warnings, affine); those names are
    bound to EMPTY placeholder modules so the import statement succeeds -- none of the functions exercised below
    touches them (a call into a placeholder would raise AttributeError).  What does need them stays unpinned and is
    NOT generated here: MeterPSNR (piq.psnr), EdgeLoss (kornia spatial_gradient), MeterSlope (richdem)."""
    for name in ("piq", "skimage", "skimage.metrics", "kornia", "kornia.filters", "richdem", "hide_warnings", "affine"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].metrics = sys.modules["skimage.metrics"]
    sys.modules["kornia"].filters = sys.modules["kornia.filters"]
    sys.modules["kornia.filters"].spatial_gradient = None        # `from kornia.filters import spatial_gradient`
    sys.modules["hide_warnings"].hide_warnings = lambda f=None, **k: (f if f is not None else (lambda g: g))
    sys.modules["affine"].Affine = None
    sys.modules["piq"].ssim = None
    import data.data_utils as du
    import evaluation.metrics as em
    import losses.loss_schemes as ls

    rs = np.random.RandomState(77)
    vmin, vmax = -80.0, 929.0
    store = {"vmin": vmin, "vmax": vmax}
    # --- scaling (torch and numpy branches of the reference) ---
    z = torch.from_numpy(rs.uniform(-60.0, 850.0, (1, 1, 100, 120))).float()
    for lg in (False, True):
        tag = "log" if lg else "lin"
        sc = du.ToTensor.scale_data(z.clone(), torch.tensor(vmin) if lg else vmin, torch.tensor(vmax) if lg else vmax, lg)
        store[f"scale_{tag}"] = sc.numpy()
        store[f"scale_np_{tag}"] = du.ToTensor.scale_data(z.numpy().copy(), vmin, vmax, lg, base_elev=3.5)
        store[f"descale_{tag}"] = du.ToDEM.descale_data(sc.clone(), vmin, vmax, lg).numpy()
    store["z"] = z.numpy()
    # --- meters: one tile per update (the reference evaluates with batch size 1), three tiles ---
    preds, gts = [], []
    for i in range(3):
        zz = torch.from_numpy(rs.uniform(-40.0, 600.0, (1, 1, 100, 120))).float()
        gt = du.ToTensor.scale_data(zz, torch.tensor(vmin), torch.tensor(vmax), True)
        noisy = (zz + torch.from_numpy(rs.standard_normal(tuple(zz.shape))).float() * (1.0 + i)).clamp_min(-70.0)
        pred = du.ToTensor.scale_data(noisy, torch.tensor(vmin), torch.tensor(vmax), True)
        pred[0, 0, 50, 60 + i] = 1.7       # exercised by _prepare's clamp
        pred[0, 0, 2, 3] = -0.2            # inside the cropped border
        preds.append(pred)
        gts.append(gt)
    store["meter_pred"] = torch.cat(preds).numpy()
    store["meter_gt"] = torch.cat(gts).numpy()
    meta = [{"subset": "train_a", "id": "x-y-12-34"}]
    for border in (0.05, 0.0):
        for lg in (True, False):
            meters = {"RMSE": em.MeterRMSE("local", border=border, value_min=vmin, value_max=vmax, verbose=False),
                      "Median": em.MeterMedian("local", border=border, value_min=vmin, value_max=vmax, verbose=False),
                      "NMAD": em.MeterNMAD("local", border=border, value_min=vmin, value_max=vmax, verbose=False),
                      "LE95": em.MeterLE95("local", border=border, value_min=vmin, value_max=vmax, verbose=False)}
            for pred, gt in zip(preds, gts):
                for m in meters.values():
                    m.update(pred, gt, meta=meta, base_elev=0, elev_log=lg)
            for k, m in meters.items():
                store[f"score_{k}_b{int(border * 100)}_{'log' if lg else 'lin'}"] = np.float64(m.get_score())
    # --- TileCrop: window walk over a 334-px sample (9 tiles, stride 103) and a 192-px one (4 tiles) ---
    for full, k, n in ((334, 128, 9), (192, 128, 4), (70, 32, 9)):
        # pixel-index images: every value identifies its source pixel and channel (and the fixture compresses)
        idx = (np.arange(full * full, dtype=np.int64).reshape(full, full, 1) * 4)
        img = (idx + np.arange(3)).astype(np.float32)
        dem = (idx + 3).astype(np.float32)
        tc = du.TileCrop(crop_size=k, n_tile=n)
        tiles_i, tiles_d = [], []
        for _ in range(n):
            out = tc({"image": img.copy(), "lr_dem": dem.copy()})
            tiles_i.append(out["image"])
            tiles_d.append(out["lr_dem"])
        ti, td = np.stack(tiles_i).astype(np.int32), np.stack(tiles_d).astype(np.int32)
        if full > 100:      # the big covers: the four corner pixels of every tile identify its window
            ti, td = ti[:, ::k - 1, ::k - 1], td[:, ::k - 1, ::k - 1]
        store[f"tiles_img_{full}"], store[f"tiles_dem_{full}"] = ti, td
        store[f"tile_params_{full}"] = np.array(du.TileCrop.get_tile(full, k, n))
    store["tile_params_322_116"] = np.array(du.TileCrop.get_tile(322, 116))
    store["tile_params_256_128"] = np.array(du.TileCrop.get_tile(256, 128))
    # --- MultiLoss bookkeeping with the two torch-only terms (the Sobel term needs kornia: unpinned) ---
    crit = ls.MultiLoss(L1={"loss_fn": ls.get_loss("l1"), "weight": 1.0}, L2={"loss_fn": ls.get_loss("l2"), "weight": 1.0})
    p64, g64 = torch.cat(preds).double().requires_grad_(), torch.cat(gts).double()
    out = crit(p64, g64)
    out["Total"].backward()
    store["loss_L1"], store["loss_L2"], store["loss_Total_L1L2"] = (np.float64(out[k].item()) for k in ("L1", "L2", "Total"))
    store["loss_grad_L1L2"] = p64.grad.numpy()
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_nlspn(path, conf_prop, preserve_input, affinity, prop_time=6, B=2, H=24, W=40, ch_g=8, seed=41, legacy=False):
    """g8: the reference's own NLSPN class (models/components/nlspn.py), fp64, random guidance convolution (the
    reference zero-initialises it, which would make every offset and affinity vanish), prop_time steps; stores all
    inputs, every step's output and the gradients of mean(sum_i feat_i * probe_i) with respect to feat_init, the
    guidance convolution and the confidence."""
    import models.components.nlspn as ref_nlspn
    args = types.SimpleNamespace(prop_time=prop_time, affinity=affinity, affinity_gamma=0.5, conf_prop=conf_prop,
                                 preserve_input=preserve_input, legacy=legacy)
    rs = np.random.RandomState(seed)
    t = lambda *shape, s=1.0: torch.from_numpy(rs.standard_normal(shape) * s)
    m = ref_nlspn.NLSPN(args, ch_g, 1, 3, 3).double()
    with torch.no_grad():
        m.conv_offset_aff.weight.copy_(t(24, ch_g, 3, 3, s=0.25))
        m.conv_offset_aff.bias.copy_(t(24, s=0.5))
        # the affinity rows: large enough that tanh(x / 100) spreads over (-1, 1) and the abs-sum clamp has both cases
        m.conv_offset_aff.weight[16:] *= 60.0
        m.conv_offset_aff.bias[16:] *= 60.0
    feat = torch.from_numpy(rs.uniform(0.2, 1.0, (B, 1, H, W))).requires_grad_()
    guid = t(B, ch_g, H, W)
    conf = torch.from_numpy(rs.uniform(0.0, 1.0, (B, 1, H, W))).requires_grad_()
    fix = torch.from_numpy(rs.uniform(0.2, 1.0, (B, 1, H, W)) * (rs.uniform(0, 1, (B, 1, H, W)) < 0.1))
    res, lst, offset, aff, sc = m(feat, guid, conf if conf_prop else None, fix if preserve_input else None)
    probes = [t(B, 1, H, W) for _ in lst]
    sum((f * p).mean() for f, p in zip(lst, probes)).backward()
    store = dict(feat=feat.detach().numpy(), guidance=guid.numpy(), confidence=conf.detach().numpy(), feat_fix=fix.numpy(),
                 conv_w=m.conv_offset_aff.weight.detach().numpy(), conv_b=m.conv_offset_aff.bias.detach().numpy(),
                 scale_const=sc.numpy(), offset=offset.detach().numpy(), aff=aff.detach().numpy(),
                 steps=torch.cat(lst, 1).detach().numpy(), probes=torch.cat(probes, 1).numpy(),
                 grad_feat=feat.grad.numpy(), grad_conv_w=m.conv_offset_aff.weight.grad.numpy(),
                 grad_conv_b=m.conv_offset_aff.bias.grad.numpy(),
                 grad_conf=(conf.grad if conf.grad is not None else torch.zeros_like(conf)).numpy(),
                 grad_scale_const=(m.aff_scale_const.grad if m.aff_scale_const.grad is not None else torch.zeros(1)).numpy(),
                 conf_prop=np.bool_(conf_prop), preserve_input=np.bool_(preserve_input), affinity=np.array(affinity),
                 prop_time=np.int64(prop_time), legacy=np.bool_(legacy))
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    ref_jspsr, ref_spn = import_reference()
    out = os.path.join(REPO, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    only = set(sys.argv[1:])

    def want(name):
        return not only or any(o in name for o in only)

    if want("g0_init"):
        gen_init_stream(ref_jspsr, os.path.join(out, "g0_init_stream_msk_nf8.npz"), {"lr_dem": 1, "image": 3, "mask": 15}, 8, 7)
        gen_init_stream(ref_jspsr, os.path.join(out, "g0_init_stream_img_nf32.npz"), {"lr_dem": 1, "image": 3}, 32, 8)
    if want("g8"):
        gen_nlspn(os.path.join(out, "g8_nlspn_tgass_conf_fix.npz"), True, True, "TGASS")
        gen_nlspn(os.path.join(out, "g8_nlspn_as_plain.npz"), False, False, "AS", prop_time=3, seed=42)
    if want("g7"):
        gen_host_side(os.path.join(out, "g7_host_side.npz"))
    if only and not any(want(g) for g in ("g1", "g3", "g4", "g5", "g6")):
        return
    gen_prop(ref_spn, os.path.join(out, "g1_postprocessor.npz"))
    img = {"lr_dem": 1, "image": 3}
    msk = {"lr_dem": 1, "image": 3, "mask": 15}
    gen_model(ref_jspsr, os.path.join(out, "g3_img_nf32_64_train.npz"), img, 32, 1, 64, 64, 11, True)
    gen_model(ref_jspsr, os.path.join(out, "g3_img_nf32_64_eval.npz"), img, 32, 1, 64, 64, 11, False)
    gen_model(ref_jspsr, os.path.join(out, "g3_img_nf8_b2_48x80_train.npz"), img, 8, 2, 48, 80, 12, True)
    gen_model(ref_jspsr, os.path.join(out, "g4_msk_nf8_b2_64_train.npz"), msk, 8, 2, 64, 64, 13, True)
    gen_model(ref_jspsr, os.path.join(out, "g4_msk_nf8_b2_64_eval.npz"), msk, 8, 2, 64, 64, 13, False)
    # the benched architecture (image+mask, num_feature 32: 1536 / 1024-channel decoder layers), training mode
    gen_model(ref_jspsr, os.path.join(out, "g4_msk_nf32_b1_64_train.npz"), msk, 32, 1, 64, 64, 14, True)
    gen_lrru(os.path.join(out, "g5_lrru_b1_64_train.npz"), 1, 64, 64, 21, True)
    gen_lrru(os.path.join(out, "g5_lrru_b2_32x48_eval.npz"), 2, 32, 48, 22, False)
    gen_edsr(os.path.join(out, "g6_edsr_b2_40x56_train.npz"), 2, 40, 56, 31, True)


if __name__ == "__main__":
    main()
