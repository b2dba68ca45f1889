// Shared device-side pieces of the propagation kernels (prop.hip: planar operands; prop_head.hip: operands straight
// from the generator head's NHWC output): tile geometry, DEM tile staging in LDS, the bilinear corner fetch with
// torchvision's border rule, the 10-value parameter-gradient fold.
#pragma once
#include "common.h"

namespace {

constexpr int HALO = 8;   // LDS halo on every side
constexpr int NT = 256;   // threads per workgroup
constexpr int NRED = 10;  // grad_wk[9] + grad_b0

struct Geom {
  int B, H, W, tiles_x, tiles_y, nblk, th, tw;
  int dem_vec4;  // DEM rows may be staged with 16-byte loads (W % 4 == 0 and a 16-byte aligned base)
};

// Stage the DEM tile + halo of image `img` into LDS; zero outside the raster.
template <int LH, int LW>
__device__ __forceinline__ void stage_dem(float* __restrict__ lds, const float* __restrict__ img,
                                          int ty0, int tx0, int H, int W, const bool VEC) {
  for (int i = threadIdx.x; i < LH * (LW / 4); i += NT) {
    const int r = i / (LW / 4), c = (i % (LW / 4)) * 4;
    const int gy = ty0 - HALO + r, gx = tx0 - HALO + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < H) {
      if (VEC) {
        if (gx >= 0 && gx < W) v = *reinterpret_cast<const float4*>(img + (size_t)gy * W + gx);
      } else {
        const float* row = img + (size_t)gy * W;
        if (gx + 0 >= 0 && gx + 0 < W) v.x = row[gx + 0];
        if (gx + 1 >= 0 && gx + 1 < W) v.y = row[gx + 1];
        if (gx + 2 >= 0 && gx + 2 < W) v.z = row[gx + 2];
        if (gx + 3 >= 0 && gx + 3 < W) v.w = row[gx + 3];
      }
    }
    *reinterpret_cast<float4*>(lds + r * LW + c) = v;
  }
}

struct Corners {
  float v00, v01, v10, v11, ly, lx;
};

// The four bilinear corners of position (py,px); out-of-raster corners are 0
// (torchvision bilinear_interpolate / get_coordinate_weight corner rule).
template <int LH, int LW>
__device__ __forceinline__ Corners corners(const float* __restrict__ lds,
                                           const float* __restrict__ img, int H, int W,
                                           int ly0, int lx0, float py, float px) {
  Corners c;
#ifdef JSPSR_LAB_NOCOMPUTE   // lab build only (tools/lab): no gather, to read the streaming ceiling of this load pattern
  c.v00 = py; c.v01 = px; c.v10 = c.v11 = 0.f; c.ly = c.lx = 0.5f;
  return c;
#endif
  const float fy = floorf(py), fx = floorf(px);
  c.ly = py - fy;
  c.lx = px - fx;
  c.v00 = c.v01 = c.v10 = c.v11 = 0.f;
  // NaN-safe "can any corner be inside the raster" test; also keeps the int casts defined.
  const bool near = (py > -2.f) && (py < (float)(H + 1)) && (px > -2.f) && (px < (float)(W + 1));
  if (near) {
    const int y0 = (int)fy, x0 = (int)fx;
    const int ry = y0 - ly0, rx = x0 - lx0;
    if ((unsigned)ry < (unsigned)(LH - 1) && (unsigned)rx < (unsigned)(LW - 1)) {
      const float* p = lds + ry * LW + rx;
      c.v00 = p[0];
      c.v01 = p[1];
      c.v10 = p[LW];
      c.v11 = p[LW + 1];
    } else {
      const bool y0ok = (unsigned)y0 < (unsigned)H, y1ok = (unsigned)(y0 + 1) < (unsigned)H;
      const bool x0ok = (unsigned)x0 < (unsigned)W, x1ok = (unsigned)(x0 + 1) < (unsigned)W;
      const float* p = img + (ptrdiff_t)y0 * W + x0;
      if (y0ok && x0ok) c.v00 = p[0];
      if (y0ok && x1ok) c.v01 = p[1];
      if (y1ok && x0ok) c.v10 = p[W];
      if (y1ok && x1ok) c.v11 = p[W + 1];
    }
  } else {
    c.ly = c.lx = 0.f;  // keep inf/nan coordinates out of the arithmetic: the tap contributes 0
  }
  return c;
}

// Same result as corners(), arranged for the common case: the LDS reads are issued unconditionally (index clamped to a
// safe slot, result zeroed when the sample is outside tile + halo), so a wave takes no divergent branch unless one of
// its lanes really needs the global fallback (a tap more than HALO pixels outside the tile but still near the raster).
template <int LH, int LW>
__device__ __forceinline__ Corners corners_fast(const float* __restrict__ lds, const float* __restrict__ img, int H, int W,
                                                int ly0, int lx0, float py, float px) {
  Corners c;
  const float fy = floorf(py), fx = floorf(px);
  c.ly = py - fy;
  c.lx = px - fx;
  // fmaxf / fminf return the non-NaN operand: NaN and +-inf coordinates become huge finite ones (out of every range)
  const int y0 = (int)fminf(fmaxf(fy, -1.0e9f), 1.0e9f), x0 = (int)fminf(fmaxf(fx, -1.0e9f), 1.0e9f);
  const int ry = y0 - ly0, rx = x0 - lx0;
  const bool inl = (unsigned)ry < (unsigned)(LH - 1) && (unsigned)rx < (unsigned)(LW - 1);
  const bool near = (py > -2.f) && (py < (float)(H + 1)) && (px > -2.f) && (px < (float)(W + 1));   // false for NaN
  const float* p = lds + (inl ? ry * LW + rx : 0);
  const float t00 = p[0], t01 = p[1], t10 = p[LW], t11 = p[LW + 1];
  c.v00 = inl ? t00 : 0.f;
  c.v01 = inl ? t01 : 0.f;
  c.v10 = inl ? t10 : 0.f;
  c.v11 = inl ? t11 : 0.f;
  const bool fb = near && !inl;
  if (__builtin_amdgcn_ballot_w64(fb) != 0) {
    if (fb) {
      const bool y0ok = (unsigned)y0 < (unsigned)H, y1ok = (unsigned)(y0 + 1) < (unsigned)H;
      const bool x0ok = (unsigned)x0 < (unsigned)W, x1ok = (unsigned)(x0 + 1) < (unsigned)W;
      const float* q = img + (ptrdiff_t)y0 * W + x0;
      if (y0ok && x0ok) c.v00 = q[0];
      if (y0ok && x1ok) c.v01 = q[1];
      if (y1ok && x0ok) c.v10 = q[W];
      if (y1ok && x1ok) c.v11 = q[W + 1];
    }
  }
  if (!near && !inl) c.ly = c.lx = 0.f;   // keep inf/nan coordinates out of the arithmetic: the tap contributes 0
  return c;
}

__device__ __forceinline__ void tile_coords(const Geom& g, int& b, int& ty0, int& tx0) {
  const int t = jspsr::xcd_contiguous(blockIdx.x, g.nblk);
  const int per_img = g.tiles_x * g.tiles_y;
  b = t / per_img;
  const int r = t - b * per_img;
  ty0 = (r / g.tiles_x) * g.th;
  tx0 = (r % g.tiles_x) * g.tw;
}

// One workgroup per parameter gradient (9 tap weights + bias): 256 lanes stride over the partial
// rows in fp64, then a fixed-order tree -> bit-reproducible run to run.  nblk < 0: the planar entry's workspace, whose
// first 16 bytes hold the row count the streaming kernel wrote (its grid depends on the path taken), rows behind it.
__global__ __launch_bounds__(256) void prop_bwd_finalize(const float* __restrict__ partial,
                                                        int nblk, float* __restrict__ gwk,
                                                        float* __restrict__ gb0) {
  __shared__ double red[4];
  if (nblk < 0) {
    nblk = *reinterpret_cast<const int*>(partial);
    partial += 4;
  }
  const int col = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partial[(size_t)i * NRED + col];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    if (col < 9) gwk[col] = (float)t; else gb0[0] = (float)t;
  }
}

}  // namespace
