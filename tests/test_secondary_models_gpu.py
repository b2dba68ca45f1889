"""GPU parity: the secondary users of the hot-path kernels -- models.LRRU (4 propagation steps) and
models.EDSR(spn=True) -- against fixtures made by the reference's own modules."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import jspsr_ref as R


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _load(golden_dir, name, shapes):
    z = np.load(os.path.join(golden_dir, name))
    sd = R.make_state_dict(shapes, int(z["seed"]), torch.float64)
    s1 = sum(v.double().abs().sum().item() for v in sd.values())
    if abs(s1 - float(z["param_abs_sum"])) > 1e-9 * s1:
        pytest.skip("torch CPU generator stream differs from the fixture's")
    B, H, W = (int(v) for v in z["BHW"])
    inputs, gt = R.synthetic_batch(B, H, W, False, seed=int(z["seed"]) + 1, dtype=torch.float64)
    sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    return z, sd32, [t.float().cuda() for t in inputs], gt.float().cuda()


def _check(z, model, pred, gt):
    ref = torch.from_numpy(z["pred"])
    assert (pred.detach().cpu().double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
    if not bool(z["training"]):
        return
    (pred * R.probe_gradient(pred.shape, int(z["seed"]) + 2, torch.float32).cuda()).mean().backward()
    grads = dict(model.named_parameters())
    for k in z.files:
        if k.startswith("grad:"):
            assert _rel(grads[k[5:]].grad, z[k]) < 5e-2, k   # ReLU-mask noise floor: see test_model_gpu.py


@pytest.mark.parametrize("name", ["g5_lrru_b1_64_train.npz", "g5_lrru_b2_32x48_eval.npz"])
def test_lrru(golden_dir, name):
    from jspsr_amd.LRRU import Model
    z, sd, inputs, gt = _load(golden_dir, name, R.lrru_param_shapes(16))
    args = types.SimpleNamespace(input_channels={"lr_dem": 1, "image": 3}, output_channels=1, kernel_size=3,
                                 bc=16, prob=1.0, dkn_residual=True)
    m = Model(args)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == list(R.lrru_param_shapes(16).items())
    m.load_state_dict(sd)
    m = m.cuda().train(bool(z["training"]))
    _check(z, m, m(*inputs), gt)
    if not bool(z["training"]):
        # inference path (BatchNorm folded into the conv epilogues, ops.conv_bn_infer): same bound
        with torch.no_grad():
            _check(z, m, m(*inputs), gt)


def test_edsr(golden_dir):
    from jspsr_amd.EDSR import EDSR
    z, sd, inputs, gt = _load(golden_dir, "g6_edsr_b2_40x56_train.npz", R.edsr_param_shapes(4, 4, 32))
    m = EDSR(in_channels=4, out_channels=1, n_resblocks=4, n_features=32, scale=1, spn=True)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == list(R.edsr_param_shapes(4, 4, 32).items())
    m.load_state_dict(sd)
    m = m.cuda().train()
    _check(z, m, m(torch.cat(inputs, 1)), gt)
