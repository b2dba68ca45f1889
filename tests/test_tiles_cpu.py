"""Host logic: tile cover / feather merge / mirror padding against the numpy restatement of the reference
(oracle/tiles_ref.py; utils/utils.py:802-967,1501-1553, data/data_utils.py:87-194)."""
import numpy as np
import pytest
import torch

from jspsr_amd import tiles as T
from oracle import tiles_ref as R


def test_cover_parameters_of_the_validation_protocol():
    # 334-px sample, 128-px tiles, 9 tiles (configs: patches_per_image 9) -> stride 103; cropped 322/116 -> same
    assert T.get_tile(334, 128, 9) == R.get_tile(334, 128, 9) == (103, 9)
    assert T.get_tile(322, 116) == R.get_tile(322, 116) == (103, 9)
    assert T.get_tile(256, 128) == R.get_tile(256, 128) == (64, 9)   # the formula adds a tile when k divides w
    with pytest.raises(AssertionError):
        T.get_tile(301, 128, 9)   # (301-128)/2 is not an integer stride


@pytest.mark.parametrize("full,k,n", [(334, 128, 9), (192, 128, 4)])
def test_crop_order_and_windows(full, k, n):
    rng = np.random.default_rng(0)
    img = rng.standard_normal((full, full, 3)).astype(np.float32)
    ref = R.crop_tiles(img, k, n)
    got = T.crop_tiles(torch.from_numpy(img).permute(2, 0, 1), k, n)
    assert got.shape == (n, 3, k, k)
    for i, t in enumerate(ref):
        assert np.array_equal(got[i].permute(1, 2, 0).numpy(), t)


@pytest.mark.parametrize("full,k,n,border", [(334, 128, 9, 0.05), (334, 128, 9, 0.0), (192, 128, 4, 0.05)])
def test_merge_matches_reference_ramps(full, k, n, border):
    rng = np.random.default_rng(1)
    tiles = rng.standard_normal((n, k, k))
    ref = R.merge_tiles(list(tiles), full, border)
    got = T.merge_tiles(torch.from_numpy(tiles)[:, None], full, border)
    assert got.shape == ref.shape
    assert np.allclose(got.numpy(), ref, rtol=0, atol=1e-14)


def test_merge_of_a_cover_is_the_identity():
    """The ramps of overlapping tiles sum to one: cropping a scene into its cover and merging gives the scene back."""
    g = torch.Generator().manual_seed(2)
    scene = torch.randn(1, 334, 334, generator=g, dtype=torch.float64)
    tiles = T.crop_tiles(scene, 128, 9)
    b = int(128 * 0.05)
    merged = T.merge_tiles(tiles, 334, 0.05)
    assert torch.allclose(merged, scene[0, b:334 - b, b:334 - b], atol=1e-13)


@pytest.mark.parametrize("h,n", [(20, 6), (334, 89), (7, 3)])
def test_mirror_padding(h, n):
    rng = np.random.default_rng(3)
    img = rng.standard_normal((h, h, 2)).astype(np.float32)
    ref = R.add_padding(img, n)
    got = T.add_padding(torch.from_numpy(img).permute(2, 0, 1), n)
    assert np.array_equal(got.permute(1, 2, 0).numpy(), ref)
    assert torch.equal(T.remove_padding(got, n), torch.from_numpy(img).permute(2, 0, 1))


def test_cal_pad():
    assert T.cal_pad(334, 334) == 89 and T.cal_pad(512, 512) == 0 and T.cal_pad(100, 100) == 14


def test_input_scaling_formulas():
    """data/data_utils.py:217-283,289-312 on a synthetic sample, against the formulas written out in numpy."""
    from jspsr_amd.metrics import scale_data, descale_data
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (3, 8, 8), dtype=np.uint8)
    assert np.allclose(T.scale_image(torch.from_numpy(img)).numpy(), img.astype(np.float32) / 255.0)
    assert np.allclose(T.scale_image(torch.from_numpy(img), "[-1, 1]").numpy(), 2 * (img.astype(np.float32) / 255.0) - 1)
    m = (rng.integers(0, 15, (8, 8))[None] == np.arange(15)[:, None, None]).astype(np.float32)
    ref = np.stack([m[i] * (i + 1) / 16 for i in range(15)])
    assert np.allclose(T.scale_mask(torch.from_numpy(m)).numpy(), ref)
    assert T.scale_mask(torch.from_numpy(m)).max() <= 15 / 16
    z = rng.uniform(0, 120, (1, 8, 8)).astype(np.float32) + 300.0
    v = scale_data(torch.from_numpy(z), -80.0, 933.0, True, base_elev=300.0)
    ref = np.log((z - 300.0) + 80.0) / np.log(933.0 + 80.0) + 1e-8
    assert np.allclose(v.numpy(), ref, rtol=1e-6) and 0 <= v.min() and v.max() <= 1
    back = descale_data(v, -80.0, 933.0, True) + 300.0
    assert np.allclose(back.numpy(), z, rtol=0, atol=2e-3)
    assert np.allclose(T.scale_canopy(torch.full((1, 2, 2), 34.0)).numpy(), 0.5)
