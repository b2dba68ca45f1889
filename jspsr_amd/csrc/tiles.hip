// K8 -- the steps either side of the model call, on device rasters: the tile cover of the validation
// protocol (TileCrop.get_tile, data/data_utils.py:87-194), the feather merge of the predictions (gen_weight_row / _col +
// merge_dem(method = copyto_add), utils/utils.py:802-967), the mirror padding of whole-scene inference (add_padding,
// utils/utils.py:1501-1520) and the elevation scaling / de-scaling (ToTensor.scale_data / ToDEM.descale_data,
// data/data_utils.py:289-312, 441-457).  HBM-bound one-pass kernels; the merge is a GATHER (every mosaic pixel sums the
// <= 4 tiles that cover it, in tile order -- the reference's own order of additions -- instead of n read-modify-write
// passes over the mosaic).
#include "common.h"

#include <cmath>

namespace {

// out[i][c][y][x] = x[c][stride r + y][stride col + x], i = r n_x + col
__global__ __launch_bounds__(256) void tiles_crop_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W,
                                                        int k, int stride, int n_x) {
  const long long total = (long long)n_x * n_x * C * k * k;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int xx = (int)(i % k);
    long long r = i / k;
    const int yy = (int)(r % k);
    r /= k;
    const int c = (int)(r % C), t = (int)(r / C);
    const int tr = t / n_x, tc = t % n_x;
    out[i] = x[((size_t)c * H + stride * tr + yy) * W + stride * tc + xx];
  }
}

// weight of position j (0 .. w_l_c-1) of a tile at cover position pos (0 .. n_x-1): 1 inside, the linear ramp over the
// p = w_l_c - s pixels it shares with a neighbour (utils.py:802-895: linspace(1, 0, p + 2) without its ends)
__device__ __forceinline__ float ramp_weight(const float* __restrict__ ramp, int p, int w_l_c, int n_x, int pos, int j) {
  float w = 1.f;
  if (pos > 0 && j < p) w = ramp[p - 1 - j];
  if (pos < n_x - 1 && j >= w_l_c - p) w = ramp[j - (w_l_c - p)];
  return w;
}

// mosaic[Y][X] = sum over the tiles (r, c) covering it, in tile order, of tile[r n_x + c][b + Y - s r][b + X - s c] * wx * wy
__global__ __launch_bounds__(256) void tiles_merge_kernel(const float* __restrict__ tiles, const float* __restrict__ ramp,
                                                         float* __restrict__ out, int n_x, int k, int b, int s, int w_l_c,
                                                         int w_h_c) {
  // (hipcc contracts a * b + c into a fused multiply-add by default, __fmul_rn / __fadd_rn included: this file is built with
  // -ffp-contract=off, csrc/Makefile)
  const int p = w_l_c - s;
  const long long total = (long long)w_h_c * w_h_c;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int X = (int)(i % w_h_c), Y = (int)(i / w_h_c);
    float acc = 0.f;
    for (int r = 0; r < n_x; ++r) {
      const int jy = Y - s * r;
      if (jy < 0 || jy >= w_l_c) continue;
      const float wy = ramp_weight(ramp, p, w_l_c, n_x, r, jy);
      for (int c = 0; c < n_x; ++c) {
        const int jx = X - s * c;
        if (jx < 0 || jx >= w_l_c) continue;
        const float wx = ramp_weight(ramp, p, w_l_c, n_x, c, jx);
        const float t = tiles[((size_t)(r * n_x + c) * k + b + jy) * k + b + jx];
        acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(t, wx), wy));      // (t * wx) * wy, then the add: no contraction -- the bits of the tile-by-tile composition
      }
    }
    out[i] = acc;
  }
}

// utils/utils.py:1501-1520, index for index: left / right mirror the image columns; the top n rows mirror the first n padded
// rows; the bottom strip mirrors rows [-2n-1, -n-1) of the padded image (one row above a true mirror)
__global__ __launch_bounds__(256) void mirror_pad_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W, int n) {
  const int HP = H + 2 * n, WP = W + 2 * n;
  const long long total = (long long)C * HP * WP;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int X = (int)(i % WP);
    long long r = i / WP;
    int Y = (int)(r % HP);
    const int c = (int)(r / HP);
    if (Y < n) Y = 2 * n - 1 - Y;                          // o[:, :n] = o[:, n:2n].flip(1)
    else if (Y >= H + n) Y = 2 * H + 2 * n - 2 - Y;        // o[:, H+n:] = o[:, H-1:H+n-1].flip(1)
    const int y = Y - n;
    int xs = X - n;
    if (X < n) xs = n - 1 - X;                             // o[.., :n] = x[.., :n].flip(2)
    else if (X >= W + n) xs = 2 * W + n - 1 - X;           // o[.., W+n:] = x[.., W-n:].flip(2)
    out[i] = x[((size_t)c * H + y) * W + xs];
  }
}

// mode 0: (z - base - lo) / (hi - lo); mode 1: log(z - base - lo) / log(hi - lo) + 1e-8        (scale_data)
// mode 2: v (hi - lo) + lo;            mode 3: exp(v log(hi - lo)) + lo                        (descale_data)
__global__ __launch_bounds__(256) void elev_scale_kernel(const float* __restrict__ in, float* __restrict__ out, long long n, int mode,
                                                        float lo, float span, float log_span, float base) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = in[i];
    float o;
    // every operation rounded on its own, as the reference's element-wise tensor expressions are (no fused multiply-add: the
    // scores are differences of de-scaled elevations of ~500 m, where a contraction moves the last bit and the median with it)
    if (mode == 0) o = __fdiv_rn(__fsub_rn(__fsub_rn(v, base), lo), span);
    else if (mode == 1) o = __fadd_rn(__fdiv_rn(logf(__fsub_rn(__fsub_rn(v, base), lo)), log_span), 1e-8f);
    else if (mode == 2) o = __fadd_rn(__fmul_rn(v, span), lo);
    else o = __fadd_rn(expf(__fmul_rn(v, log_span)), lo);
    out[i] = o;
  }
}

int blocks_for(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int jspsr_tiles_crop_f32(const float* x, float* out, int C, int H, int W, int k, int stride, int n_x,
                                    jspsr_stream_t stream) {
  if (!x || !out || C <= 0 || H <= 0 || W <= 0 || k <= 0 || stride < 0 || n_x <= 0) return jspsr::fail(JSPSR_EINVAL, "tiles_crop: bad arguments");
  if (stride * (n_x - 1) + k > H || stride * (n_x - 1) + k > W) return jspsr::fail(JSPSR_EINVAL, "tiles_crop: the cover leaves the raster");
  const long long total = (long long)n_x * n_x * C * k * k;
  hipLaunchKernelGGL(tiles_crop_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, C, H, W, k, stride, n_x);
  return jspsr::check_launch("tiles_crop");
}

extern "C" int jspsr_tiles_merge_f32(const float* tiles, const float* ramp, float* out, int n_x, int k, int border_px, int stride,
                                     jspsr_stream_t stream) {
  const int w_l_c = k - 2 * border_px, w_h_c = stride * (n_x - 1) + w_l_c, p = w_l_c - stride;
  if (!tiles || !out || n_x <= 0 || k <= 0 || border_px < 0 || w_l_c <= 0 || stride <= 0 || p < 0 || (p > 0 && !ramp) || p > w_l_c)
    return jspsr::fail(JSPSR_EINVAL, "tiles_merge: bad arguments");
  hipLaunchKernelGGL(tiles_merge_kernel, dim3(blocks_for((long long)w_h_c * w_h_c)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     tiles, ramp, out, n_x, k, border_px, stride, w_l_c, w_h_c);
  return jspsr::check_launch("tiles_merge");
}

extern "C" int jspsr_mirror_pad_f32(const float* x, float* out, int C, int H, int W, int n, jspsr_stream_t stream) {
  if (!x || !out || C <= 0 || H <= 0 || W <= 0 || n <= 0 || n >= H || n > W) return jspsr::fail(JSPSR_EINVAL, "mirror_pad: bad arguments (0 < n < H, n <= W)");
  hipLaunchKernelGGL(mirror_pad_kernel, dim3(blocks_for((long long)C * (H + 2 * n) * (W + 2 * n))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, out, C, H, W, n);
  return jspsr::check_launch("mirror_pad");
}

extern "C" int jspsr_elev_scale_f32(const float* in, float* out, long long n, int descale, int elev_log, double elev_min, double elev_max,
                                    double base_elev, jspsr_stream_t stream) {
  if (!in || !out || n <= 0 || !(elev_max > elev_min)) return jspsr::fail(JSPSR_EINVAL, "elev_scale: bad arguments");
  // the constants as the reference's Python forms them (doubles), rounded once to the tensors' fp32
  const float span = (float)(elev_max - elev_min);
  const float log_span = (float)log(elev_max - elev_min);
  hipLaunchKernelGGL(elev_scale_kernel, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, n,
                     (descale ? 2 : 0) + (elev_log ? 1 : 0), (float)elev_min, span, log_span, descale ? 0.f : (float)base_elev);
  return jspsr::check_launch("elev_scale");
}
