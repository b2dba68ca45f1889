// K1 -- fused spatial propagation for gfx950 (MI355X), forward and backward.
//
// Restates, in one pass over HBM, PostProcessor.forward of the reference
// (models/components/spn.py:99-118): zero-sum affinities, 3x3 deformable bilinear gather of a
// one-channel DEM (torchvision deform_conv2d semantics), learnable 3x3
// tap weights + bias, residual add.  HBM-bound: 116 B/pixel forward, 224 B/pixel backward
// (108 / 208 with the 16-channel offset layout that drops the all-zero centre pair).
//
// Data movement
//   * a workgroup (256 threads = 4 waves) owns a TH x TW = 16 x 64 pixel tile of one image;
//     a lane owns 4 consecutive pixels of one row, so every operand plane is read with one
//     coalesced 16-byte load per lane (a wave instruction covers 4 rows x 256 B);
//   * the DEM tile plus an 8-pixel halo is staged once in LDS (zero outside the raster, which
//     IS the sampler's border rule); the 9 x 4 corner reads per pixel are LDS reads;
//   * a tap that lands outside tile+halo (|offset| > ~8 px) falls back to bounds-checked
//     global reads of the DEM (1 channel: L2 resident);
//   * tiles are numbered so that an XCD's L2 sees a contiguous run of tiles (halo reuse).
#include "common.h"

#include <cstdlib>
#include <initializer_list>

namespace {

constexpr int TW = 64;    // tile width  (pixels)
constexpr int HALO = 8;   // LDS halo on every side
constexpr int LW = TW + 2 * HALO;  // 80
constexpr int NT = 256;   // threads per workgroup
constexpr int NRED = 10;  // grad_wk[9] + grad_b0

struct Geom {
  int B, H, W, tiles_x, tiles_y, nblk, th;
  int dem_vec4;  // DEM rows may be staged with 16-byte loads (W % 4 == 0 and a 16-byte aligned base)
};

// PX consecutive pixels of one plane per lane: PX*4-byte loads when the row pitch and the
// pointers allow it (VEC), otherwise predicated scalar accesses.
template <int PX>
struct Vec {
  float v[PX];
};

template <int PX, bool VEC>
__device__ __forceinline__ Vec<PX> ldv(const float* __restrict__ p, int x, int W) {
  Vec<PX> r;
  if (VEC) {
    if (PX == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      r.v[0] = t.x; r.v[1 % PX] = t.y; r.v[2 % PX] = t.z; r.v[3 % PX] = t.w;
    } else if (PX == 2) {
      const float2 t = *reinterpret_cast<const float2*>(p);
      r.v[0] = t.x; r.v[1 % PX] = t.y;
    } else {
      r.v[0] = p[0];
    }
  } else {
#pragma unroll
    for (int j = 0; j < PX; ++j) r.v[j] = (x + j < W) ? p[j] : 0.f;
  }
  return r;
}

// NT_STORE: streaming (non-temporal) store.  The backward writes 25 planes once and never reads them back: letting them
// bypass the cache hierarchy measured -4.5 % on the backward kernel (95.7 -> 91.4 us at 8 x 512 x 512); the forward's
// single output plane measured no better (+2 %) and keeps ordinary stores.
template <int PX, bool VEC, bool NT_STORE = false>
__device__ __forceinline__ void stv(float* __restrict__ p, const Vec<PX>& r, int x, int W) {
  if (VEC) {
    if (PX == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % PX], r.v[2 % PX], r.v[3 % PX]);
    else if (PX == 2) *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1 % PX]);
    else if (NT_STORE) __builtin_nontemporal_store(r.v[0], p);
    else p[0] = r.v[0];
  } else {
#pragma unroll
    for (int j = 0; j < PX; ++j)
      if (x + j < W) p[j] = r.v[j];
  }
}

// Stage the DEM tile + halo of image `img` into LDS; zero outside the raster.
template <int LH>
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
template <int LH>
__device__ __forceinline__ Corners corners(const float* __restrict__ lds,
                                           const float* __restrict__ img, int H, int W,
                                           int ly0, int lx0, float py, float px) {
  Corners c;
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

__device__ __forceinline__ void tile_coords(const Geom& g, int& b, int& ty0, int& tx0) {
  const int t = jspsr::xcd_contiguous(blockIdx.x, g.nblk);
  const int per_img = g.tiles_x * g.tiles_y;
  b = t / per_img;
  const int r = t - b * per_img;
  ty0 = (r / g.tiles_x) * g.th;
  tx0 = (r % g.tiles_x) * TW;
}

// offset channel of (tap k, component c) in the OC-channel layout
template <int OC>
__device__ __forceinline__ constexpr int och(int k, int c) {
  return OC == 18 ? 2 * k + c : 2 * (k < 4 ? k : k - 1) + c;
}

// Lane -> pixel map: a row of the tile is TW/PX lanes wide; a workgroup pass covers
// RPP = NT*PX/TW rows and the tile's TH rows take TH/RPP passes (not unrolled: it bounds the
// live registers to one pass; occupancy, not unrolling, hides the HBM latency).
template <int OC, int PX, bool VEC, int TH>
__global__ __launch_bounds__(NT) void prop_fwd_kernel(const float* __restrict__ dem,
                                                     const float* __restrict__ weight,
                                                     const float* __restrict__ offset,
                                                     const float* __restrict__ wk,
                                                     const float* __restrict__ b0, float scale,
                                                     float* __restrict__ out, Geom g) {
  constexpr int LH = TH + 2 * HALO;
  __shared__ __attribute__((aligned(16))) float lds[LH * LW];
  constexpr int LPR = TW / PX;       // lanes per tile row
  constexpr int RPP = NT / LPR;      // rows per pass
  int b, ty0, tx0;
  tile_coords(g, b, ty0, tx0);
  const int H = g.H, W = g.W;
  const size_t P = (size_t)H * W;
  const float* img = dem + (size_t)b * P;
  stage_dem<LH>(lds, img, ty0, tx0, H, W, g.dem_vec4 != 0);
  float wreg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wreg[k] = wk[k];
  const float bias = b0[0];
  __syncthreads();
  const int x = tx0 + (threadIdx.x % LPR) * PX;
  const int ly0 = ty0 - HALO, lx0 = tx0 - HALO;
  if (x >= W) return;
#pragma unroll 1
  for (int y = ty0 + threadIdx.x / LPR; y < min(ty0 + TH, H); y += RPP) {
    const size_t pix = (size_t)y * W + x;
    const float* wp = weight + (size_t)b * 9 * P + pix;
    const float* op = offset + (size_t)b * OC * P + pix;
    Vec<PX> a[9], oy[9], ox[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = ldv<PX, VEC>(wp + k * P, x, W);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (OC == 18 || k != 4) {
        oy[k] = ldv<PX, VEC>(op + (size_t)och<OC>(k, 0) * P, x, W);
        ox[k] = ldv<PX, VEC>(op + (size_t)och<OC>(k, 1) * P, x, W);
      } else {
#pragma unroll
        for (int j = 0; j < PX; ++j) oy[k].v[j] = ox[k].v[j] = 0.f;
      }
    }
    const Vec<PX> dc = ldv<PX, VEC>(img + pix, x, W);
    Vec<PX> o;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) s += a[k].v[j];
      const float mean = s / 9.f;
      float acc = bias;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float py = (float)(y - 1 + k / 3) + oy[k].v[j];
        const float px = (float)(x + j - 1 + k % 3) + ox[k].v[j];
        const Corners c = corners<LH>(lds, img, H, W, ly0, lx0, py, px);
        const float hy = 1.f - c.ly, hx = 1.f - c.lx;
        const float S = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
        acc += wreg[k] * (a[k].v[j] - mean) * S;
      }
      o.v[j] = acc + scale * dc.v[j];
    }
    stv<PX, VEC>(out + (size_t)b * P + pix, o, x, W);
  }
}

template <int OC, int PX, bool VEC, int TH>
__global__ __launch_bounds__(NT) void prop_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ dem,
    const float* __restrict__ weight, const float* __restrict__ offset,
    const float* __restrict__ wk, float* __restrict__ gweight, float* __restrict__ goffset,
    float* __restrict__ partial, Geom g) {
  constexpr int LH = TH + 2 * HALO;
  __shared__ __attribute__((aligned(16))) float lds[LH * LW];
  __shared__ float red[NT / 64][NRED];
  constexpr int LPR = TW / PX;
  constexpr int RPP = NT / LPR;
  int b, ty0, tx0;
  tile_coords(g, b, ty0, tx0);
  const int H = g.H, W = g.W;
  const size_t P = (size_t)H * W;
  const float* img = dem + (size_t)b * P;
  stage_dem<LH>(lds, img, ty0, tx0, H, W, g.dem_vec4 != 0);
  float wreg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wreg[k] = wk[k];
  __syncthreads();
  const int x = tx0 + (threadIdx.x % LPR) * PX;
  const int ly0 = ty0 - HALO, lx0 = tx0 - HALO;

  float sums[NRED];
#pragma unroll
  for (int i = 0; i < NRED; ++i) sums[i] = 0.f;

  if (x < W) {
#pragma unroll 1
    for (int y = ty0 + threadIdx.x / LPR; y < min(ty0 + TH, H); y += RPP) {
      const size_t pix = (size_t)y * W + x;
      const float* wp = weight + (size_t)b * 9 * P + pix;
      const float* op = offset + (size_t)b * OC * P + pix;
      float* gop = goffset + (size_t)b * OC * P + pix;
      float* gwp = gweight + (size_t)b * 9 * P + pix;
      Vec<PX> a[9], oy[9], ox[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) a[k] = ldv<PX, VEC>(wp + k * P, x, W);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (OC == 18 || k != 4) {
          oy[k] = ldv<PX, VEC>(op + (size_t)och<OC>(k, 0) * P, x, W);
          ox[k] = ldv<PX, VEC>(op + (size_t)och<OC>(k, 1) * P, x, W);
        } else {
#pragma unroll
          for (int j = 0; j < PX; ++j) oy[k].v[j] = ox[k].v[j] = 0.f;
        }
      }
      const Vec<PX> go = ldv<PX, VEC>(gout + (size_t)b * P + pix, x, W);
      Vec<PX> gm[9], gy[9], gx[9];
#pragma unroll
      for (int j = 0; j < PX; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += a[k].v[j];
        const float mean = s / 9.f;
        const float gj = go.v[j];  // 0 for x+j >= W on the scalar path (ldv pads with 0)
        float gsum = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float py = (float)(y - 1 + k / 3) + oy[k].v[j];
          const float px = (float)(x + j - 1 + k % 3) + ox[k].v[j];
          const Corners c = corners<LH>(lds, img, H, W, ly0, lx0, py, px);
          const float hy = 1.f - c.ly, hx = 1.f - c.lx;
          const float S = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
          const float dSdy = hx * (c.v10 - c.v00) + c.lx * (c.v11 - c.v01);
          const float dSdx = hy * (c.v01 - c.v00) + c.ly * (c.v11 - c.v10);
          const float m = a[k].v[j] - mean;
          const float coef = gj * wreg[k] * m;
          gy[k].v[j] = coef * dSdy;
          gx[k].v[j] = coef * dSdx;
          const float gmk = gj * wreg[k] * S;
          gm[k].v[j] = gmk;
          gsum += gmk;
          sums[k] += gj * m * S;
        }
        gsum /= 9.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) gm[k].v[j] -= gsum;
        sums[9] += gj;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        stv<PX, VEC, true>(gwp + k * P, gm[k], x, W);
        if (OC == 18 || k != 4) {
          stv<PX, VEC, true>(gop + (size_t)och<OC>(k, 0) * P, gy[k], x, W);
          stv<PX, VEC, true>(gop + (size_t)och<OC>(k, 1) * P, gx[k], x, W);
        }
      }
    }
  }

  // workgroup reduction of the 10 parameter-gradient partial sums
#pragma unroll
  for (int i = 0; i < NRED; ++i) {
    float v = sums[i];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NRED) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) v += red[w][threadIdx.x];
    partial[(size_t)blockIdx.x * NRED + threadIdx.x] = v;
  }
}

// One workgroup per parameter gradient (9 tap weights + bias): 256 lanes stride over the partial
// rows in fp64, then a fixed-order tree -> bit-reproducible run to run.
__global__ __launch_bounds__(256) void prop_bwd_finalize(const float* __restrict__ partial,
                                                        int nblk, float* __restrict__ gwk,
                                                        float* __restrict__ gb0) {
  __shared__ double red[4];
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

// Tile height.  Tunable through JSPSR_PROP_TH (4, 8 or 16); 8 measured best on MI355X.
int prop_th() {
  static const int th = [] {
    const char* e = getenv("JSPSR_PROP_TH");
    const int v = e ? atoi(e) : 8;
    return (v == 4 || v == 8 || v == 16) ? v : 8;
  }();
  return th;
}

int make_geom(int B, int H, int W, Geom& g) {
  if (B <= 0 || H <= 0 || W <= 0) return jspsr::fail(JSPSR_EINVAL, "prop: bad shape B=%d H=%d W=%d", B, H, W);
  g.B = B;
  g.H = H;
  g.W = W;
  g.tiles_x = (W + TW - 1) / TW;
  g.th = prop_th();
  g.tiles_y = (H + g.th - 1) / g.th;
  const long long n = (long long)B * g.tiles_x * g.tiles_y;
  if (n > 0x7fffffffLL || (long long)B * 18 * H * W > (1LL << 40))
    return jspsr::fail(JSPSR_EINVAL, "prop: problem too large");
  g.nblk = (int)n;
  return JSPSR_OK;
}

// Pixels per lane.  Tunable for experiments through JSPSR_PROP_PX (1, 2 or 4); the default is
// what measured fastest on MI355X.
int prop_px() {
  static const int px = [] {
    const char* e = getenv("JSPSR_PROP_PX");
    const int v = e ? atoi(e) : 1;
    return (v == 1 || v == 2 || v == 4) ? v : 1;
  }();
  return px;
}

bool can_vec(int W, int px, std::initializer_list<const void*> ptrs) {
  if (W % px) return false;
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(4 * px - 1)) return false;
  return true;
}

}  // namespace

extern "C" int jspsr_prop_forward_f32(const float* dem, const float* weight, const float* offset,
                                      int offset_channels, const float* wk, const float* b0,
                                      float scale, float* out, int B, int H, int W,
                                      jspsr_stream_t stream) {
  if (!dem || !weight || !offset || !wk || !b0 || !out) return jspsr::fail(JSPSR_EINVAL, "prop_forward: null pointer");
  if (offset_channels != 16 && offset_channels != 18)
    return jspsr::fail(JSPSR_EINVAL, "prop_forward: offset_channels must be 16 or 18, got %d", offset_channels);
  Geom g;
  if (int e = make_geom(B, H, W, g)) return e;
  for (const void* p : {(const void*)dem, (const void*)weight, (const void*)offset, (const void*)out})
    if (!jspsr::aligned4(p)) return jspsr::fail(JSPSR_EALIGN, "prop_forward: pointer not 4-byte aligned");
  const int px = prop_px();
  const bool vec = can_vec(W, px, {dem, weight, offset, out});
  g.dem_vec4 = can_vec(W, 4, {dem});
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(g.nblk), block(NT);
#define LAUNCH_TH(OC, PX, V, T) hipLaunchKernelGGL((prop_fwd_kernel<OC, PX, V, T>), grid, block, 0, s, dem, weight, offset, wk, b0, scale, out, g)
#define LAUNCH(OC, PX, V) do { if (g.th == 16) LAUNCH_TH(OC, PX, V, 16); else if (g.th == 8) LAUNCH_TH(OC, PX, V, 8); else LAUNCH_TH(OC, PX, V, 4); } while (0)
#define BY_VEC(OC, PX) do { if (vec) LAUNCH(OC, PX, true); else LAUNCH(OC, PX, false); } while (0)
#define BY_PX(OC) do { if (px == 1) LAUNCH(OC, 1, true); else if (px == 2) BY_VEC(OC, 2); else BY_VEC(OC, 4); } while (0)
  if (offset_channels == 18) BY_PX(18); else BY_PX(16);
#undef LAUNCH
#undef LAUNCH_TH
  return jspsr::check_launch("prop_forward");
}

extern "C" size_t jspsr_prop_backward_workspace_bytes(int B, int H, int W) {
  Geom g;
  if (make_geom(B, H, W, g)) return 0;
  return ((size_t)g.nblk * NRED * sizeof(float) + 15) & ~(size_t)15;
}

extern "C" int jspsr_prop_backward_f32(const float* grad_out, const float* dem, const float* weight,
                                       const float* offset, int offset_channels, const float* wk,
                                       float* grad_weight, float* grad_offset, float* grad_wk,
                                       float* grad_b0, void* workspace, int B, int H, int W,
                                       jspsr_stream_t stream) {
  if (!grad_out || !dem || !weight || !offset || !wk || !grad_weight || !grad_offset || !workspace || (!grad_wk != !grad_b0))
    return jspsr::fail(JSPSR_EINVAL, "prop_backward: null pointer");
  if (offset_channels != 16 && offset_channels != 18)
    return jspsr::fail(JSPSR_EINVAL, "prop_backward: offset_channels must be 16 or 18, got %d", offset_channels);
  Geom g;
  if (int e = make_geom(B, H, W, g)) return e;
  for (const void* p : {(const void*)grad_out, (const void*)dem, (const void*)weight, (const void*)offset,
                        (const void*)grad_weight, (const void*)grad_offset})
    if (!jspsr::aligned4(p)) return jspsr::fail(JSPSR_EALIGN, "prop_backward: pointer not 4-byte aligned");
  if (!jspsr::aligned16(workspace)) return jspsr::fail(JSPSR_EALIGN, "prop_backward: workspace not 16-byte aligned");
  const int px = prop_px();
  const bool vec = can_vec(W, px, {grad_out, dem, weight, offset, grad_weight, grad_offset});
  g.dem_vec4 = can_vec(W, 4, {dem});
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  dim3 grid(g.nblk), block(NT);
#define LAUNCH_TH(OC, PX, V, T) hipLaunchKernelGGL((prop_bwd_kernel<OC, PX, V, T>), grid, block, 0, s, grad_out, dem, weight, offset, wk, grad_weight, grad_offset, partial, g)
#define LAUNCH(OC, PX, V) do { if (g.th == 16) LAUNCH_TH(OC, PX, V, 16); else if (g.th == 8) LAUNCH_TH(OC, PX, V, 8); else LAUNCH_TH(OC, PX, V, 4); } while (0)
  if (offset_channels == 18) BY_PX(18); else BY_PX(16);
#undef LAUNCH
#undef LAUNCH_TH
#undef BY_VEC
#undef BY_PX
  if (int e = jspsr::check_launch("prop_backward")) return e;
  if (!grad_wk) return JSPSR_OK;   // partial rows only: the caller folds them later (jspsr_prop_backward_fold_f32)
  hipLaunchKernelGGL(prop_bwd_finalize, dim3(NRED), dim3(256), 0, s, partial, g.nblk, grad_wk, grad_b0);
  return jspsr::check_launch("prop_backward_finalize");
}

extern "C" int jspsr_prop_backward_fold_f32(const void* workspace, int B, int H, int W, float* grad_wk, float* grad_b0,
                                            jspsr_stream_t stream) {
  if (!workspace || !grad_wk || !grad_b0) return jspsr::fail(JSPSR_EINVAL, "prop_backward_fold: null pointer");
  Geom g;
  if (int e = make_geom(B, H, W, g)) return e;
  hipLaunchKernelGGL(prop_bwd_finalize, dim3(NRED), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(workspace), g.nblk, grad_wk, grad_b0);
  return jspsr::check_launch("prop_backward_fold");
}
