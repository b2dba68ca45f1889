"""Tile cover and feather merge around the model call, on device tensors.

The reference feeds the network 128-px crops of a 334-px sample in a 3x3 cover (stride 103) and stitches the
predictions back with 13-px linear ramps (`TileCrop`, data/data_utils.py:87-194; `gen_weight_row/col` +
`merge_dem(..., method=copyto_add)`, utils/utils.py:802-967,1272).  It does so through GeoTIFF files; here the same
arithmetic runs on NCHW tensors already on the GPU: `crop_tiles` builds the batch, `merge_tiles` folds the batch of
predictions into the border-cropped mosaic the metrics are computed on.  Mirror padding for whole-scene inference
(`add_padding` / `remove_padding` / `cal_pad`, utils/utils.py:1501-1553) is restated with its exact index rule.
"""
from __future__ import annotations

from math import ceil

import torch

from . import _lib


def _hip(t: torch.Tensor) -> bool:
    """Device rasters in fp32 go through the HIP kernels (csrc/tiles.hip: jspsr_tiles_crop_f32 / _merge_f32,
    jspsr_mirror_pad_f32); host tensors (the reference does all of this in numpy on the CPU) and other dtypes through the
    same arithmetic as torch indexing -- index-for-index and, for the merge, bit-for-bit equal (tests/test_tiles_gpu.py)."""
    return t.is_cuda and t.dtype == torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def get_tile(w: int, k: int, n: int | None = None):
    """(stride, number of tiles) of the square cover of a w-px side by k-px tiles; data_utils.py:170-194."""
    n_x = (w - w % k) / k + 1 if n is None else ceil(n ** 0.5)
    assert n_x % 1 == 0, "cannot divide the image into n_tile tiles, check the input."
    stride = (w - k) / (n_x - 1)
    assert stride % 1 == 0, "no padding for cropping to tile evenly, check the input."
    return int(stride), int(n_x ** 2)


def crop_tiles(x: torch.Tensor, k: int, n: int | None = None) -> torch.Tensor:
    """x (C,H,W) -> (n,C,k,k), row-major over the cover, as TileCrop walks it (data_utils.py:98-168)."""
    C, H, W = x.shape
    if k > H or k > W or (k == H == W):
        return x[None]
    stride, n = get_tile(W, k, n)
    n_x = int(round(n ** 0.5))
    if _hip(x):
        x = x.contiguous()
        out = torch.empty((n, C, k, k), dtype=x.dtype, device=x.device)
        _lib.check(_lib.load().jspsr_tiles_crop_f32(x.data_ptr(), out.data_ptr(), C, H, W, k, stride, n_x, _stream()), "jspsr_tiles_crop_f32")
        return out
    return torch.stack([x[:, stride * r: stride * r + k, stride * c: stride * c + k]
                        for r in range(n_x) for c in range(n_x)])


def _weight_1d(w_l_c: int, s: int, n_x: int, pos: int, device, dtype):
    p = w_l_c - s
    ramp = torch.linspace(1, 0, p + 2, dtype=torch.float64)[1:-1]      # ends (1 and 0) removed, utils.py:815
    w = torch.ones(w_l_c, dtype=torch.float64)
    if n_x not in (2, 3):
        raise NotImplementedError(f"n {n_x * n_x} is not 9 or 4")
    if pos > 0:
        w[:p] = ramp.flip(0)
    if pos < n_x - 1:
        w[-p:] = ramp
    return w.to(device=device, dtype=dtype)


def merge_tiles(tiles: torch.Tensor, full: int, border: float = 0.0) -> torch.Tensor:
    """tiles (n,1,k,k) predictions on the cover of a (full, full) sample -> (full-2b', full-2b') mosaic, where
    each tile first loses int(k*border) px per side, then is weighted by the row and column ramps and summed
    (utils/utils.py:897-967 with method=copyto_add)."""
    n, one, k, _ = tiles.shape
    n_x = int(round(n ** 0.5))
    assert one == 1 and n_x * n_x == n, f"n {n} is not a square number."
    b = int(k * border)
    w_l_c = k - 2 * b
    w_h_c = full - (k - w_l_c)
    s, n2 = get_tile(w_h_c, w_l_c)
    assert n2 == n
    if _hip(tiles):
        if n_x not in (2, 3):
            raise NotImplementedError(f"n {n} is not 9 or 4")
        p = w_l_c - s
        ramp = torch.linspace(1, 0, p + 2, dtype=torch.float64)[1:-1].to(device=tiles.device, dtype=torch.float32).contiguous()
        tiles = tiles.contiguous()
        out = torch.empty((w_h_c, w_h_c), dtype=tiles.dtype, device=tiles.device)
        _lib.check(_lib.load().jspsr_tiles_merge_f32(tiles.data_ptr(), ramp.data_ptr() if p > 0 else None, out.data_ptr(), n_x, k, b, s,
                                                     _stream()), "jspsr_tiles_merge_f32")
        return out
    out = torch.zeros((w_h_c, w_h_c), dtype=tiles.dtype, device=tiles.device)
    for i in range(n):
        r, c = divmod(i, n_x)
        wx = _weight_1d(w_l_c, s, n_x, c, tiles.device, tiles.dtype)
        wy = _weight_1d(w_l_c, s, n_x, r, tiles.device, tiles.dtype)
        t = tiles[i, 0, b: k - b, b: k - b]
        out[s * r: s * r + w_l_c, s * c: s * c + w_l_c] += t * wx[None, :] * wy[:, None]
    return out


def cal_pad(h: int, w: int) -> int:
    """Border that brings a side up to the next power of two (utils/utils.py:1534-1553)."""
    if bin(h).count("1") == 1 and bin(w).count("1") == 1:
        return 0
    h_pad = w_pad = 0
    for i in range(1, 10):
        if 2 ** i > h:
            h_pad, w_pad = (2 ** i - h) // 2, (2 ** i - w) // 2
            break
    assert h_pad == w_pad
    return h_pad


def add_padding(x: torch.Tensor, n: int) -> torch.Tensor:
    """x (C,H,W) -> (C,H+2n,W+2n) mirrored border, index for index as utils/utils.py:1501-1520 (left/right mirror
    the image columns; top mirrors the first n padded rows; the bottom strip mirrors rows [-2n-1, -n-1) of the
    padded image, i.e. one row above a true mirror)."""
    if n == 0:
        return x
    C, H, W = x.shape
    if _hip(x) and n < H and n <= W:
        x = x.contiguous()
        o = torch.empty((C, H + 2 * n, W + 2 * n), dtype=x.dtype, device=x.device)
        _lib.check(_lib.load().jspsr_mirror_pad_f32(x.data_ptr(), o.data_ptr(), C, H, W, n, _stream()), "jspsr_mirror_pad_f32")
        return o
    o = torch.empty((C, H + 2 * n, W + 2 * n), dtype=x.dtype, device=x.device)
    o[:, n:n + H, n:n + W] = x
    o[:, n:n + H, :n] = x[:, :, :n].flip(2)
    o[:, n:n + H, W + n:] = x[:, :, W - n:].flip(2)
    o[:, :n] = o[:, n:2 * n].flip(1)
    o[:, H + n:] = o[:, H - 1: H + n - 1].flip(1)
    return o


def remove_padding(x: torch.Tensor, pad: int) -> torch.Tensor:
    return x[..., pad: x.shape[-2] - pad, pad: x.shape[-1] - pad]


# ---- input scaling: what `ToTensor.__call__` does to each raster before the model sees it (data_utils.py:217-283)
def scale_image(img_u8: torch.Tensor, image_range: str | None = None) -> torch.Tensor:
    """uint8 (C,H,W) -> float32 in [0,1] (torchvision ToTensor = /255); "[-1, 1]" and "[0, 255]" as :234-238."""
    x = img_u8.to(torch.float32) / 255.0
    if image_range == "[-1, 1]":
        x = 2.0 * x - 1.0
    elif image_range == "[0, 255]":
        x = x / 255.0
    return x


def scale_mask(mask: torch.Tensor, n_channels: int | None = None) -> torch.Tensor:
    """One-hot land-use mask (C,H,W): channel i times (i+1)/(len(mask_channel)+1), data_utils.py:262-265."""
    C = mask.shape[0]
    n = C if n_channels is None else n_channels
    f = torch.arange(1, C + 1, device=mask.device, dtype=torch.float32) / float(n + 1)
    return mask.to(torch.float32) * f[:, None, None]


def scale_canopy(canopy: torch.Tensor) -> torch.Tensor:
    """Canopy height / 68 m (data_utils.py:266-267)."""
    return canopy.to(torch.float32) / 68.0
