"""GPU: K8 (csrc/tiles.hip) -- tile cover, feather merge, mirror padding and elevation scaling as HIP kernels on fp32 device
rasters, against the same functions run on the host tensors (torch indexing: index for index the reference's numpy code,
tests/test_tiles_cpu.py) and against the numbers the reference's own classes produced (tests/golden/g7_host_side.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from jspsr_amd import metrics as M
from jspsr_amd import tiles as T
from oracle import tiles_ref as TR


def _count(names):
    from jspsr_amd import _lib
    lib = _lib.load()
    return {n: lib.jspsr_launch_count(n.encode()) for n in names}


NAMES = ("tiles_crop", "tiles_merge", "mirror_pad", "elev_scale")


@pytest.mark.parametrize("full,k,n,C", [(334, 128, 9, 3), (192, 128, 4, 1), (70, 32, 9, 15)])
def test_crop_is_the_reference_cover(full, k, n, C):
    g = torch.Generator().manual_seed(full)
    x = torch.randn(C, full, full, generator=g)
    before = _count(NAMES)
    got = T.crop_tiles(x.cuda(), k, n)
    assert _count(NAMES)["tiles_crop"] == before["tiles_crop"] + 1
    assert torch.equal(got.cpu(), T.crop_tiles(x, k, n))
    ref = TR.crop_tiles(x.permute(1, 2, 0).numpy(), k, n)           # the numpy restatement of TileCrop, (k,k,C) tiles
    assert all(np.array_equal(got[i].permute(1, 2, 0).cpu().numpy(), t) for i, t in enumerate(ref))


@pytest.mark.parametrize("full,k,n,border", [(334, 128, 9, 0.05), (334, 128, 9, 0.0), (192, 128, 4, 0.05), (192, 128, 4, 0.0)])
def test_merge_is_bit_equal_to_the_tile_by_tile_accumulation(full, k, n, border):
    g = torch.Generator().manual_seed(n)
    tiles = torch.randn(n, 1, k, k, generator=g)
    before = _count(NAMES)
    got = T.merge_tiles(tiles.cuda(), full, border)
    assert _count(NAMES)["tiles_merge"] == before["tiles_merge"] + 1
    host = T.merge_tiles(tiles, full, border)                          # fp32, the reference's order of additions
    assert got.shape == host.shape and torch.equal(got.cpu(), host)
    ref = TR.merge_tiles([t[0].double().numpy() for t in tiles], full, border)     # fp64 restatement of utils.py:897-967
    assert np.allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("C,H,W,n", [(2, 20, 20, 6), (1, 334, 334, 89), (3, 33, 47, 32)])
def test_mirror_padding_index_for_index(C, H, W, n):
    g = torch.Generator().manual_seed(H)
    x = torch.randn(C, H, W, generator=g)
    before = _count(NAMES)
    got = T.add_padding(x.cuda(), n)
    assert _count(NAMES)["mirror_pad"] == before["mirror_pad"] + 1
    assert torch.equal(got.cpu(), T.add_padding(x, n))
    assert np.array_equal(got.permute(1, 2, 0).cpu().numpy(), TR.add_padding(x.permute(1, 2, 0).numpy(), n))
    assert torch.equal(T.remove_padding(got, n).cpu(), x)


def test_elevation_scaling_against_the_reference_made_numbers(golden_dir):
    g7 = np.load(os.path.join(golden_dir, "g7_host_side.npz"))
    vmin, vmax = float(g7["vmin"]), float(g7["vmax"])
    z = torch.from_numpy(g7["z"]).float().cuda()
    before = _count(NAMES)
    for lg, tag in ((False, "lin"), (True, "log")):
        v = M.scale_data(z, vmin, vmax, lg)
        assert np.allclose(v.cpu().numpy(), g7[f"scale_{tag}"], rtol=0, atol=3e-7)
        assert np.allclose(M.scale_data(z, vmin, vmax, lg, base_elev=3.5).cpu().numpy(), g7[f"scale_np_{tag}"], rtol=0, atol=3e-7)
        back = M.descale_data(torch.from_numpy(g7[f"scale_{tag}"]).float().cuda(), vmin, vmax, lg)
        assert np.allclose(back.cpu().numpy(), g7[f"descale_{tag}"], rtol=2e-6, atol=1e-4)
        # and against the host path of the same functions (torch CPU): same formula, fp32
        assert np.allclose(v.cpu().numpy(), M.scale_data(z.cpu(), vmin, vmax, lg).numpy(), rtol=0, atol=2e-7)
    assert _count(NAMES)["elev_scale"] == before["elev_scale"] + 6
