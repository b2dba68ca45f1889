"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's tile-cover / feather-merge arithmetic.
Only tests/ may import this module.

Restates, loop by loop, what the reference does through GeoTIFF files and rioxarray:
  * TileCrop.get_tile                      data/data_utils.py:170-194
  * TileCrop.__call__ window walk          data/data_utils.py:98-168   (row-major, stride*row / stride*col)
  * gen_weight_row / gen_weight_col        utils/utils.py:802-894      (linear ramps over the overlap, ends removed)
  * merge_dem with method=copyto_add       utils/utils.py:897-967,1272 (border crop, weight, sum where tiles overlap)
  * add_padding / remove_padding           utils/utils.py:1501-1531    (mirror border; the bottom strip is taken one
                                                                         row higher than a true mirror -- kept)
The georeferenced merge (rioxarray.merge_arrays on tile bounds) places tile (r, c) at pixel offset stride*(r, c) of
the border-cropped mosaic; that placement is what is restated here.  rasterio/rioxarray are not installed, so this
restatement is pinned by the cited lines only (parity unpinned for this row, like the loss/metric helpers).
"""
from math import ceil

import numpy as np


def get_tile(w, k, n=None):
    n_x = (w - w % k) / k + 1 if n is None else ceil(n ** 0.5)
    assert n_x % 1 == 0
    stride = (w - k) / (n_x - 1)
    assert stride % 1 == 0
    return int(stride), int(n_x ** 2)


def crop_tiles(img, k, n=None):
    """img (H, W, C) -> list of (k, k, C) tiles in the order TileCrop emits them."""
    h, w, _ = img.shape
    stride, n = get_tile(w, k, n)
    n_x = int(round(n ** 0.5))
    out = []
    for row in range(n_x):
        for col in range(n_x):
            out.append(img[stride * row: stride * row + k, stride * col: stride * col + k, :])
    return out


def _ramps(w_l_c, s):
    p = w_l_c - s
    weight = np.linspace(1, 0, p + 2)[1:-1]
    one = np.ones(w_l_c)
    one[-p:] = weight
    two = np.ones(w_l_c)
    two[:p] = np.flip(weight)
    two[-p:] = weight
    return one, two


def weight_1d(w_l_c, s, n_x, pos):
    """Ramp along one axis for the tile at position `pos` (0..n_x-1) of that axis."""
    one, two = _ramps(w_l_c, s)
    if n_x == 3:
        return (one, two, np.flip(one))[pos]
    if n_x == 2:
        return (one, np.flip(one))[pos]
    raise NotImplementedError


def merge_tiles(tiles, full, border=0.0):
    """tiles: n arrays (k, k) predicted on the TileCrop cover of a (full, full) sample.  Returns the
    ((full - 2*int(k*border)),)*2 mosaic the reference evaluates (utils/utils.py:1272-1290)."""
    n = len(tiles)
    n_x = int(round(n ** 0.5))
    k = tiles[0].shape[0]
    b = int(k * border)
    w_l_c = k - 2 * b
    w_h_c = full - (k - w_l_c)
    s, n2 = get_tile(w_h_c, w_l_c)
    assert n2 == n
    out = np.zeros((w_h_c, w_h_c), np.float64)
    for i, t in enumerate(tiles):
        r, c = divmod(i, n_x)
        t = np.asarray(t, np.float64)[b: k - b, b: k - b]
        wr = weight_1d(w_l_c, s, n_x, c)          # gen_weight_row: varies along the row (x), by column index
        wc = weight_1d(w_l_c, s, n_x, r)          # gen_weight_col: varies along the column (y), by row index
        out[s * r: s * r + w_l_c, s * c: s * c + w_l_c] += t * wr[None, :] * wc[:, None]
    return out


def add_padding(img, n_pixels):
    h, w, c = img.shape
    o = np.empty((h + 2 * n_pixels, w + 2 * n_pixels, c), np.float32)
    o[n_pixels: n_pixels + h, n_pixels: n_pixels + w, :] = img
    o[n_pixels: n_pixels + h, 0:n_pixels, :] = img[:, 0:n_pixels, :][:, ::-1, :]
    o[n_pixels: n_pixels + h, -n_pixels:, :] = img[:, -n_pixels:, :][:, ::-1, :]
    top = o[n_pixels: 2 * n_pixels, :, :]
    btm = o[-2 * n_pixels - 1: -n_pixels - 1, :, :]
    o[0:n_pixels, :, :] = top[::-1, :, :]
    o[-n_pixels:, :, :] = btm[::-1, :, :]
    return o
