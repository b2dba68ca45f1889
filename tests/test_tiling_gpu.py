"""GPU: whole-scene inference cut into strips (+halo, +scene-wide gate statistics) equals the monolithic
forward (BASELINE config 5's parity check, on one GPU through the single-process emulation)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import jspsr_ref as R


def test_strip_sharding_matches_monolithic():
    from jspsr_amd.JSPSR import Model
    from jspsr_amd import tiling
    ic = {"lr_dem": 1, "image": 3, "mask": 15}
    sd = R.make_state_dict(R.jspsr_param_shapes(ic, 8), seed=3)
    m = Model(dict(ic, COP30=1), num_feature=8)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    inputs, _ = R.synthetic_batch(1, 1024, 256, True, seed=4)
    inputs = [t.cuda() for t in inputs]
    with torch.no_grad():
        mono = m(*inputs)
    for world in (2, 4):
        out = tiling.emulate_sharded_forward(m, inputs, world, halo=128)
        assert out.shape == mono.shape
        err = (out - mono).abs().max().item()
        assert err < 2e-5, (world, err)
    # without the scene-wide gate statistics the strips disagree: the sync is doing real work
    strips = tiling.plan_strips(1024, 4, 128)
    with torch.no_grad():
        naive = torch.cat([m(*[t[:, :, s.ty0:s.ty1].contiguous() for t in inputs])[:, :, s.y0 - s.ty0:s.y1 - s.ty0]
                           for s in strips], 2)
    assert (naive - mono).abs().max().item() > 10 * err


def test_training_mode_is_rejected():
    from jspsr_amd.JSPSR import Model
    from jspsr_amd import tiling
    m = Model({"lr_dem": 1, "image": 3}, num_feature=8).cuda().train()
    x = [torch.rand(1, 1, 512, 64, device="cuda"), torch.rand(1, 3, 512, 64, device="cuda")]
    with pytest.raises(RuntimeError, match="eval"):
        tiling.emulate_sharded_forward(m, x, 2)
