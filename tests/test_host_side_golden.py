"""The rows either side of the hot path against fixtures made by the reference's OWN code
(tests/golden/g7_host_side.npz, g0_init_stream_*.npz; oracle/gen_golden.py::gen_host_side, gen_init_stream):

  * the numpy oracles (oracle/metrics_ref.py, oracle/tiles_ref.py) against those fixtures -- the pin;
  * the product's host-side functions (jspsr_amd.metrics, jspsr_amd.tiles, jspsr_amd.losses bookkeeping) against
    the same fixtures, on CPU tensors here and on device tensors under -m gpu.
"""
import numpy as np
import pytest
import torch

from jspsr_amd import metrics as M
from jspsr_amd import tiles as T
from oracle import metrics_ref as MR
from oracle import tiles_ref as TR
from tests import fixtures as Fx

NAMES = ("RMSE", "Median", "NMAD", "LE95")
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


@pytest.fixture(scope="module")
def g7(golden_dir):
    return Fx.load(golden_dir, "g7_host_side.npz")


# ---- the pin: numpy oracles vs the reference-made numbers ---------------------------------------------------------
def test_oracle_scores_match_reference_meters(g7):
    vmin, vmax = float(g7["vmin"]), float(g7["vmax"])
    p, g = g7["meter_pred"], g7["meter_gt"]
    for border in (0.05, 0.0):
        for lg in (True, False):
            s = MR.mean_scores(p, g, vmin, vmax, border, lg)
            for k in NAMES:
                ref = float(g7[f"score_{k}_b{int(border * 100)}_{'log' if lg else 'lin'}"])
                # the reference computes in fp32 (exp of ~6.9 * v: one fp32 ulp of a 900 m elevation is 6e-5 m)
                assert abs(s[k] - ref) < 2e-3 + 1e-5 * abs(ref), (k, border, lg, s[k], ref)


def test_oracle_scaling_matches_reference(g7):
    vmin, vmax, z = float(g7["vmin"]), float(g7["vmax"]), g7["z"]
    for lg, tag in ((False, "lin"), (True, "log")):
        assert np.allclose(MR.scale_data(z, vmin, vmax, lg), g7[f"scale_{tag}"], rtol=0, atol=2e-7)
        assert np.allclose(MR.scale_data(z, vmin, vmax, lg, base_elev=3.5), g7[f"scale_np_{tag}"], rtol=0, atol=2e-7)
        assert np.allclose(MR.descale_data(g7[f"scale_{tag}"], vmin, vmax, lg), g7[f"descale_{tag}"], rtol=2e-6, atol=1e-4)


def test_oracle_tile_windows_match_reference_tilecrop(g7):
    for full, k, n in ((334, 128, 9), (192, 128, 4), (70, 32, 9)):
        assert TR.get_tile(full, k, n) == tuple(int(v) for v in g7[f"tile_params_{full}"])
        idx = np.arange(full * full, dtype=np.int64).reshape(full, full, 1) * 4
        img, dem = (idx + np.arange(3)).astype(np.float32), (idx + 3).astype(np.float32)
        for src, key in ((img, "tiles_img"), (dem, "tiles_dem")):
            got = np.stack(TR.crop_tiles(src, k, n)).astype(np.int32)
            if full > 100:
                got = got[:, ::k - 1, ::k - 1]
            assert np.array_equal(got, g7[f"{key}_{full}"])
    assert TR.get_tile(322, 116) == tuple(int(v) for v in g7["tile_params_322_116"])
    assert TR.get_tile(256, 128) == tuple(int(v) for v in g7["tile_params_256_128"])


def test_oracle_l1_l2_match_reference_multiloss(g7):
    l1, l2, grad = MR.l1_l2(g7["meter_pred"], g7["meter_gt"])
    assert abs(l1 - float(g7["loss_L1"])) < 1e-12 and abs(l2 - float(g7["loss_L2"])) < 1e-12
    assert abs(l1 + l2 - float(g7["loss_Total_L1L2"])) < 1e-12
    assert np.allclose(grad, g7["loss_grad_L1L2"], rtol=0, atol=1e-15)


# ---- the product's host-side functions vs the same reference-made numbers ------------------------------------------
@pytest.mark.parametrize("device", DEVICES)
def test_product_meter_matches_reference_meters(g7, device):
    vmin, vmax = float(g7["vmin"]), float(g7["vmax"])
    p = torch.from_numpy(g7["meter_pred"]).to(device)
    g = torch.from_numpy(g7["meter_gt"]).to(device)
    for border in (0.05, 0.0):
        for lg in (True, False):
            m = M.Meter(vmin, vmax, border=border, elev_log=lg)
            for i in range(p.shape[0]):
                m.update(p[i:i + 1], g[i:i + 1])
            s = m.scores()
            for k in NAMES:
                ref = float(g7[f"score_{k}_b{int(border * 100)}_{'log' if lg else 'lin'}"])
                assert abs(s[k] - ref) < 2e-3 + 1e-5 * abs(ref), (k, border, lg, s[k], ref)
            # PSNR has no reference-made number (piq absent): against the oracle's formula
            assert abs(s["PSNR"] - MR.mean_scores(g7["meter_pred"], g7["meter_gt"], vmin, vmax, border, lg)["PSNR"]) < 1e-3


@pytest.mark.parametrize("device", DEVICES)
def test_product_scaling_matches_reference(g7, device):
    vmin, vmax = float(g7["vmin"]), float(g7["vmax"])
    z = torch.from_numpy(g7["z"]).to(device)
    for lg, tag in ((False, "lin"), (True, "log")):
        v = M.scale_data(z, vmin, vmax, lg)
        assert np.allclose(v.cpu().numpy(), g7[f"scale_{tag}"], rtol=0, atol=2e-7)
        assert np.allclose(M.scale_data(z, vmin, vmax, lg, base_elev=3.5).cpu().numpy(), g7[f"scale_np_{tag}"], rtol=0, atol=2e-7)
        back = M.descale_data(torch.from_numpy(g7[f"scale_{tag}"]).to(device), vmin, vmax, lg)
        assert np.allclose(back.cpu().numpy(), g7[f"descale_{tag}"], rtol=2e-6, atol=1e-4)


@pytest.mark.parametrize("device", DEVICES)
def test_product_tile_windows_match_reference_tilecrop(g7, device):
    for full, k, n in ((334, 128, 9), (192, 128, 4), (70, 32, 9)):
        assert T.get_tile(full, k, n) == tuple(int(v) for v in g7[f"tile_params_{full}"])
        idx = torch.arange(full * full, dtype=torch.int64).reshape(1, full, full) * 4
        img = (idx + torch.arange(3).view(3, 1, 1)).float().to(device)       # (C,H,W) on the device
        dem = (idx + 3).float().to(device)
        for src, key in ((img, "tiles_img"), (dem, "tiles_dem")):
            got = T.crop_tiles(src, k, n).permute(0, 2, 3, 1).cpu().numpy().astype(np.int32)   # -> (n,k,k,C) like the reference
            if full > 100:
                got = got[:, ::k - 1, ::k - 1]
            assert np.array_equal(got, g7[f"{key}_{full}"])
    assert T.get_tile(322, 116) == tuple(int(v) for v in g7["tile_params_322_116"])
    assert T.get_tile(256, 128) == tuple(int(v) for v in g7["tile_params_256_128"])


@pytest.mark.parametrize("device", DEVICES)
def test_product_merge_and_padding_on_device_match_oracle(device):
    """utils/utils.py:802-967,1501-1553 need rioxarray/rasterio objects and cannot run here: the numpy restatement
    (oracle/tiles_ref.py, pinned by the cited lines only) is the checker for these two."""
    rng = np.random.default_rng(1)
    for full, k, n, border in ((334, 128, 9, 0.05), (192, 128, 4, 0.0)):
        tiles = rng.standard_normal((n, k, k))
        ref = TR.merge_tiles(list(tiles), full, border)
        got = T.merge_tiles(torch.from_numpy(tiles)[:, None].to(device), full, border)
        assert np.allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-12)
    img = rng.standard_normal((20, 20, 2)).astype(np.float32)
    got = T.add_padding(torch.from_numpy(img).permute(2, 0, 1).to(device), 6)
    assert np.array_equal(got.permute(1, 2, 0).cpu().numpy(), TR.add_padding(img, 6))
    assert torch.equal(T.remove_padding(got, 6).cpu(), torch.from_numpy(img).permute(2, 0, 1))


@pytest.mark.gpu
def test_fused_loss_matches_reference_multiloss_l1_l2(g7):
    """jspsr_amd.losses.MultiLoss (one fused HIP pass each way) with the Sobel weight at 0 against the reference's
    MultiLoss(L1, L2) numbers and gradient."""
    from jspsr_amd.losses import MultiLoss
    p = torch.from_numpy(g7["meter_pred"]).cuda().requires_grad_()
    g = torch.from_numpy(g7["meter_gt"]).cuda()
    crit = MultiLoss(1.0, 1.0, 0.0)
    out = crit(p, g)
    out["Total"].backward()
    assert abs(out["L1"].item() - float(g7["loss_L1"])) < 1e-6 * float(g7["loss_L1"]) + 1e-9
    assert abs(out["L2"].item() - float(g7["loss_L2"])) < 1e-5 * float(g7["loss_L2"]) + 1e-9
    assert abs(out["Total"].item() - float(g7["loss_Total_L1L2"])) < 1e-6
    assert np.allclose(p.grad.cpu().numpy(), g7["loss_grad_L1L2"], rtol=1e-5, atol=1e-9)
    crit.reset()        # the reference's loop calls it every iteration (train/train_utils.py:206)


# ---- a11: initialisation stream ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g0_init_stream_msk_nf8.npz", "g0_init_stream_img_nf32.npz"])
def test_init_stream_equals_reference(golden_dir, name):
    """Model._initialize_weights under np.random.seed(s) draws the reference's stream (models/JSPSR.py:494-517):
    gen_golden.py compared the two state_dicts bit for bit with the reference imported; here the per-tensor
    summaries it stored must be reproduced exactly."""
    from jspsr_amd.JSPSR import Model
    z = Fx.load(golden_dir, name)
    ic = dict(Fx.MSK if bool(z["with_mask"]) else Fx.IMG, COP30=1)
    np.random.seed(int(z["seed"]))
    sd = Model(ic, num_feature=int(z["nf"])).state_dict()
    assert list(sd) == [str(k) for k in z["names"]]
    for i, k in enumerate(sd):
        t = sd[k].double().reshape(-1)
        got = np.array([t.sum().item(), t.abs().sum().item(), t[0].item(), t[-1].item()])
        # first / last element exactly; the two sums to the last bits a double sum's ORDER can move (torch's parallel
        # reduction splits the tensor by thread count: 8 threads where the fixture was made, 16 on a GPU box)
        assert np.array_equal(got[2:], z["summary"][i][2:]), k
        assert np.allclose(got[:2], z["summary"][i][:2], rtol=0, atol=1e-12 * got[1]), k


@pytest.mark.gpu
def test_fused_tile_scores_match_the_torch_composition_and_the_reference(g7):
    """jspsr_metrics_forward (one C-ABI call per tile: crop / clamp / de-scale / reductions + radix select) against
    (a) the reference-made meter numbers (through Meter, above), (b) the same formulas as torch operators on the same
    device tensors: the three order statistics must be BIT-identical (an exact select returns an element of the
    array), PSNR / RMSE within fp32 summation error; sizes incl. odd ones and a 1024 x 768 tile."""
    vmin, vmax = float(g7["vmin"]), float(g7["vmax"])
    g_ = torch.Generator().manual_seed(9)
    cases = [(torch.from_numpy(g7["meter_pred"][i:i + 1]), torch.from_numpy(g7["meter_gt"][i:i + 1])) for i in range(3)]
    for H, W in ((37, 53), (128, 128), (1024, 768)):
        z = 300 * torch.rand(1, 1, H, W, generator=g_)
        gt = M.scale_data(z, vmin, vmax, True)
        pred = M.scale_data((z + 2 * torch.randn(z.shape, generator=g_)).clamp_min(-70), vmin, vmax, True)
        pred[0, 0, H // 2, W // 2] = 1.3
        cases.append((pred, gt))
    for pred, gt in cases:
        for border in (0.05, 0.0):
            for lg in (True, False):
                p, g = pred.cuda(), gt.cuda()
                got = M.tile_scores(p, g, vmin, vmax, border, lg).cpu()
                pp, gg = M.prepare(p.float(), g.float(), border)
                dh = M.descale_data(pp, vmin, vmax, lg) - M.descale_data(gg, vmin, vmax, lg)
                want = torch.stack((M.psnr(pp, gg), M.rmse(dh), M.median(dh), M.nmad(dh), M.le95(dh))).cpu()
                assert abs(got[0] - want[0]) < 1e-3 and abs(got[1] - want[1]) < 1e-4 * abs(want[1]) + 1e-6
                # the device exp/log of the fused kernel and torch's agree to 1 ulp of a ~900 m elevation (6e-5 m); the
                # select itself is exact: compare through a tolerance of 2 ulp of the de-scaled magnitude
                tol = 2.5e-4 if lg else 0.0
                for i in (2, 3, 4):
                    assert abs(got[i] - want[i]) <= tol + 1e-6 * abs(want[i]), (i, border, lg, got[i].item(), want[i].item())


@pytest.mark.gpu
def test_radix_select_is_exact_on_given_differences():
    """Linear de-scaling with value_min 0, value_max 2 makes dh = 2 (pred - gt) exactly: the order statistics of the
    fused path must equal torch.median / kthvalue of that array BIT for bit (ties, negatives, zeros included)."""
    g_ = torch.Generator().manual_seed(10)
    for n_side in (16, 100, 513):
        gt = torch.rand(1, 1, n_side, n_side, generator=g_) * 0.5
        d = torch.randn(1, 1, n_side, n_side, generator=g_) * 0.01
        d[0, 0, :3] = 0.0                                       # ties
        pred = (gt + d).clamp(0, 1)
        got = M.tile_scores(pred.cuda(), gt.cuda(), 0.0, 2.0, 0.0, False).cpu()
        dh = (pred.cuda() * 2.0 + 0.0) - (gt.cuda() * 2.0 + 0.0)
        assert got[2].item() == torch.median(dh).item()
        assert got[3].item() == (1.4826 * torch.median((dh - torch.median(dh)).abs())).item()
        k = 1 + round(0.95 * (dh.numel() - 1))
        assert got[4].item() == torch.kthvalue(dh.abs().flatten(), k).values.item()
