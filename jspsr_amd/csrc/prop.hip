// K1 -- fused spatial propagation for gfx950 (MI355X), forward and backward: the C-ABI entry points, and the GENERAL
// kernels (any width, any alignment).  Rows of 16-byte-aligned operands with W % 4 == 0 -- every shape the models and
// the benchmark produce -- take the persistent LDS-DMA kernels of prop_dma.hip instead (same arithmetic).
//
// Restates, in one pass over HBM, PostProcessor.forward of the reference
// (models/components/spn.py:99-118): zero-sum affinities, 3x3 deformable bilinear gather of a
// one-channel DEM (torchvision deform_conv2d semantics), learnable 3x3
// tap weights + bias, residual add.  HBM-bound; algorithmic bytes per pixel: 108 forward / 208 backward
// with the 16-channel offset layout the models use (the all-zero centre pair is not stored), 116 / 224 with
// torchvision's 18 channels.
//
// Data movement of the general kernels
//   * a workgroup (256 threads = 4 waves) owns a TH x TW pixel tile of one image (default 8 x 64; JSPSR_PROP_TH /
//     JSPSR_PROP_TW); a lane owns PX consecutive pixels of one row per pass (default 1: a wave instruction is one
//     256-byte row segment of one operand plane; 2 / 4 = 8- / 16-byte loads, measured slower: DESIGN.md);
//   * the DEM tile plus an 8-pixel halo is staged once in LDS (zero outside the raster, which
//     IS the sampler's border rule); the 9 x 4 corner reads per pixel are LDS reads;
//   * a tap that lands outside tile+halo (|offset| > ~8 px) falls back to bounds-checked
//     global reads of the DEM (1 channel: L2 resident);
//   * tiles are numbered so that an XCD's L2 sees a contiguous run of tiles (halo reuse).
#include "prop_tile.h"

#include <cstdlib>
#include <initializer_list>
#include <type_traits>

namespace jspsr {   // prop_dma.hip: the persistent LDS-DMA form of the same kernels (16-byte rows, aligned operands)
int prop_dma_max_rows();
bool prop_dma_ok(int B, int H, int W, int oc, std::initializer_list<const void*> ptrs);
int prop_dma_forward(const float* dem, const float* weight, const float* offset, int oc, const float* wk, const float* b0,
                     float scale, float* out, int B, int H, int W, hipStream_t s);
int prop_dma_logits_forward(const float* dem, const float* head, const float* wk, const float* b0, float scale, float* out,
                            int B, int H, int W, hipStream_t s);
int prop_dma_logits_backward(const float* gout, const float* dem, const float* head, const float* wk, float* ghead, float* partial,
                             int B, int H, int W, hipStream_t s);
int prop_dma_backward(const float* gout, const float* dem, const float* weight, const float* offset, int oc, const float* wk,
                      float* gweight, float* goffset, float* partial, int B, int H, int W, hipStream_t s);
}  // namespace jspsr

namespace {

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

// offset channel of (tap k, component c) in the OC-channel layout
template <int OC>
__device__ __forceinline__ constexpr int och(int k, int c) {
  return OC == 18 ? 2 * k + c : 2 * (k < 4 ? k : k - 1) + c;
}

// Lane -> pixel map: a row of the tile is TW/PX lanes wide; a workgroup pass covers
// RPP = NT*PX/TW rows and the tile's TH rows take TH/RPP passes (not unrolled: it bounds the
// live registers to one pass; occupancy, not unrolling, hides the HBM latency).
// SIG: the affinity planes hold logits (sigmoid applied here; the backward writes d/d(logit)) -- jspsr_prop_logits_*.
// wbs / obs: elements between consecutive images of weight / offset (9 P / OC P at the public boundary).
template <int OC, int PX, bool VEC, int TH, int TW, bool SIG = false>
__global__ __launch_bounds__(NT) void prop_fwd_kernel(const float* __restrict__ dem,
                                                     const float* __restrict__ weight,
                                                     const float* __restrict__ offset,
                                                     const float* __restrict__ wk,
                                                     const float* __restrict__ b0, float scale,
                                                     float* __restrict__ out, Geom g, size_t wbs, size_t obs) {
  constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  __shared__ __attribute__((aligned(16))) float lds[LH * LW];
  constexpr int LPR = TW / PX;       // lanes per tile row
  constexpr int RPP = NT / LPR;      // rows per pass
  int b, ty0, tx0;
  tile_coords(g, b, ty0, tx0);
  const int H = g.H, W = g.W;
  const size_t P = (size_t)H * W;
  const float* img = dem + (size_t)b * P;
  stage_dem<LH, LW>(lds, img, ty0, tx0, H, W, g.dem_vec4 != 0);
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
    const float* wp = weight + (size_t)b * wbs + pix;
    const float* op = offset + (size_t)b * obs + pix;
    Vec<PX> a[9], oy[9], ox[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      a[k] = ldv<PX, VEC>(wp + k * P, x, W);
      if (SIG) {
#pragma unroll
        for (int j = 0; j < PX; ++j) a[k].v[j] = __builtin_amdgcn_rcpf(1.f + __expf(-a[k].v[j]));
      }
    }
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
        const Corners c = corners<LH, LW>(lds, img, H, W, ly0, lx0, py, px);
        const float hy = 1.f - c.ly, hx = 1.f - c.lx;
        const float S = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
        acc += wreg[k] * (a[k].v[j] - mean) * S;
      }
      o.v[j] = acc + scale * lds[(y - ly0) * LW + (x + j - lx0)];   // the residual term reads dem[y][x] from the staged tile
    }
    stv<PX, VEC>(out + (size_t)b * P + pix, o, x, W);
  }
}

template <int OC, int PX, bool VEC, int TH, int TW, bool SIG = false>
__global__ __launch_bounds__(NT) void prop_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ dem,
    const float* __restrict__ weight, const float* __restrict__ offset,
    const float* __restrict__ wk, float* __restrict__ gweight, float* __restrict__ goffset,
    float* __restrict__ partial, Geom g, size_t wbs, size_t obs) {      // partial: 16-byte header (row count) + one row per workgroup
  constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  __shared__ __attribute__((aligned(16))) float lds[LH * LW];
  __shared__ float red[NT / 64][NRED];
  constexpr int LPR = TW / PX;
  constexpr int RPP = NT / LPR;
  int b, ty0, tx0;
  tile_coords(g, b, ty0, tx0);
  const int H = g.H, W = g.W;
  const size_t P = (size_t)H * W;
  const float* img = dem + (size_t)b * P;
  stage_dem<LH, LW>(lds, img, ty0, tx0, H, W, g.dem_vec4 != 0);
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
      const float* wp = weight + (size_t)b * wbs + pix;
      const float* op = offset + (size_t)b * obs + pix;
      float* gop = goffset + (size_t)b * obs + pix;
      float* gwp = gweight + (size_t)b * wbs + pix;
      Vec<PX> a[9], oy[9], ox[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        a[k] = ldv<PX, VEC>(wp + k * P, x, W);
        if (SIG) {
#pragma unroll
          for (int j = 0; j < PX; ++j) a[k].v[j] = __builtin_amdgcn_rcpf(1.f + __expf(-a[k].v[j]));
        }
      }
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
          const Corners c = corners_fast<LH, LW>(lds, img, H, W, ly0, lx0, py, px);
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
        for (int k = 0; k < 9; ++k) gm[k].v[j] = SIG ? (gm[k].v[j] - gsum) * a[k].v[j] * (1.f - a[k].v[j]) : gm[k].v[j] - gsum;
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
    partial[4 + (size_t)blockIdx.x * NRED + threadIdx.x] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int*>(partial)[0] = (int)gridDim.x;
}

// Tile shape.  Defaults are what measured best on MI355X; JSPSR_PROP_TH x JSPSR_PROP_TW select one of the
// built shapes for A/B measurements: 64-wide tiles 4 / 8 / 16 rows tall, 128 x {4, 8}, 256 x {2, 4, 8}.
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

void tile_shape(int& th, int& tw) {
  static const int tw_ = [] { const int v = env_int("JSPSR_PROP_TW", 64); return (v == 64 || v == 128 || v == 256) ? v : 64; }();
  static const int th_ = [] {
    const int v = env_int("JSPSR_PROP_TH", 8);
    if (tw_ == 64) return (v == 4 || v == 8 || v == 16) ? v : 8;
    if (tw_ == 128) return (v == 4 || v == 8) ? v : 8;
    return (v == 2 || v == 4 || v == 8) ? v : 4;
  }();
  th = th_;
  tw = tw_;
}

int make_geom(int B, int H, int W, Geom& g) {
  if (B <= 0 || H <= 0 || W <= 0) return jspsr::fail(JSPSR_EINVAL, "prop: bad shape B=%d H=%d W=%d", B, H, W);
  g.B = B;
  g.H = H;
  g.W = W;
  tile_shape(g.th, g.tw);
  g.tiles_x = (W + g.tw - 1) / g.tw;
  g.tiles_y = (H + g.th - 1) / g.th;
  const long long n = (long long)B * g.tiles_x * g.tiles_y;
  if (n > 0x7fffffffLL || (long long)B * 18 * H * W > (1LL << 40))
    return jspsr::fail(JSPSR_EINVAL, "prop: problem too large");
  g.nblk = (int)n;
  return JSPSR_OK;
}

// Pixels per lane.  Tunable for experiments through JSPSR_PROP_PX (1, 2 or 4; 64-wide tiles only); the default is
// what measured fastest on MI355X.
int prop_px() {
  static const int px = [] {
    const int v = env_int("JSPSR_PROP_PX", 1);
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

// Calls f(std::integral_constant...) for the built (OC, PX, VEC, TH, TW) combination matching the run-time choice.
template <int V> using IC = std::integral_constant<int, V>;

template <typename F>
void by_shape(int oc, int px, bool vec, int th, int tw, F&& f) {
  auto with_tile = [&](auto OCc, auto PXc, auto VECc) {
    if (tw == 64) {
      if (th == 16) f(OCc, PXc, VECc, IC<16>{}, IC<64>{});
      else if (th == 4) f(OCc, PXc, VECc, IC<4>{}, IC<64>{});
      else f(OCc, PXc, VECc, IC<8>{}, IC<64>{});
    }
  };
  auto wide = [&](auto OCc) {          // wide tiles: one pixel per lane only
    if (tw == 128) {
      if (th == 4) f(OCc, IC<1>{}, IC<1>{}, IC<4>{}, IC<128>{});
      else f(OCc, IC<1>{}, IC<1>{}, IC<8>{}, IC<128>{});
    } else {
      if (th == 2) f(OCc, IC<1>{}, IC<1>{}, IC<2>{}, IC<256>{});
      else if (th == 8) f(OCc, IC<1>{}, IC<1>{}, IC<8>{}, IC<256>{});
      else f(OCc, IC<1>{}, IC<1>{}, IC<4>{}, IC<256>{});
    }
  };
  auto with_px = [&](auto OCc) {
    if (tw != 64) { wide(OCc); return; }
    if (px == 1) with_tile(OCc, IC<1>{}, IC<1>{});
    else if (px == 2) { if (vec) with_tile(OCc, IC<2>{}, IC<1>{}); else with_tile(OCc, IC<2>{}, IC<0>{}); }
    else { if (vec) with_tile(OCc, IC<4>{}, IC<1>{}); else with_tile(OCc, IC<4>{}, IC<0>{}); }
  };
  if (oc == 18) with_px(IC<18>{}); else with_px(IC<16>{});
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (jspsr::prop_dma_ok(B, H, W, offset_channels, {dem, weight, offset, out}))
    return jspsr::prop_dma_forward(dem, weight, offset, offset_channels, wk, b0, scale, out, B, H, W, s);
  const int px = g.tw == 64 ? prop_px() : 1;
  const bool vec = can_vec(W, px, {dem, weight, offset, out});
  g.dem_vec4 = can_vec(W, 4, {dem});
  dim3 grid(g.nblk), block(NT);
  by_shape(offset_channels, px, vec, g.th, g.tw, [&](auto OCc, auto PXc, auto VECc, auto THc, auto TWc) {
    hipLaunchKernelGGL((prop_fwd_kernel<decltype(OCc)::value, decltype(PXc)::value, decltype(VECc)::value != 0,
                                        decltype(THc)::value, decltype(TWc)::value>),
                       grid, block, 0, s, dem, weight, offset, wk, b0, scale, out, g, (size_t)9 * H * W, (size_t)offset_channels * H * W);
  });
  return jspsr::check_launch("prop_forward");
}

extern "C" size_t jspsr_prop_backward_workspace_bytes(int B, int H, int W) {
  Geom g;
  if (make_geom(B, H, W, g)) return 0;
  size_t rows = (size_t)g.nblk > (size_t)jspsr::prop_dma_max_rows() ? (size_t)g.nblk : (size_t)jspsr::prop_dma_max_rows();
  const size_t rows_8x64 = (size_t)B * ((W + 63) / 64) * ((H + 7) / 8);      // jspsr_prop_logits_backward_f32's general path
  if (rows_8x64 > rows) rows = rows_8x64;
#ifdef K1D_STAMPS
  return 16 + 4096 * NRED * sizeof(float) + (1 << 20);                  // lab build: cycle stamps behind the rows
#endif
  return 16 + ((rows * NRED * sizeof(float) + 15) & ~(size_t)15);      // header (row count) + rows, whichever path runs
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  if (jspsr::prop_dma_ok(B, H, W, offset_channels, {grad_out, dem, weight, offset, grad_weight, grad_offset})) {
    if (int e = jspsr::prop_dma_backward(grad_out, dem, weight, offset, offset_channels, wk, grad_weight, grad_offset, partial, B, H, W, s))
      return e;
  } else {
    const int px = g.tw == 64 ? prop_px() : 1;
    const bool vec = can_vec(W, px, {grad_out, dem, weight, offset, grad_weight, grad_offset});
    g.dem_vec4 = can_vec(W, 4, {dem});
    dim3 grid(g.nblk), block(NT);
    by_shape(offset_channels, px, vec, g.th, g.tw, [&](auto OCc, auto PXc, auto VECc, auto THc, auto TWc) {
      hipLaunchKernelGGL((prop_bwd_kernel<decltype(OCc)::value, decltype(PXc)::value, decltype(VECc)::value != 0,
                                          decltype(THc)::value, decltype(TWc)::value>),
                         grid, block, 0, s, grad_out, dem, weight, offset, wk, grad_weight, grad_offset, partial, g,
                         (size_t)9 * H * W, (size_t)offset_channels * H * W);
    });
    if (int e = jspsr::check_launch("prop_backward")) return e;
  }
  if (!grad_wk) return JSPSR_OK;   // partial rows only: the caller folds them later (jspsr_prop_backward_fold_f32)
  hipLaunchKernelGGL(prop_bwd_finalize, dim3(NRED), dim3(256), 0, s, partial, -1, grad_wk, grad_b0);
  return jspsr::check_launch("prop_backward_finalize");
}

extern "C" int jspsr_prop_backward_fold_f32(const void* workspace, int B, int H, int W, float* grad_wk, float* grad_b0,
                                            jspsr_stream_t stream) {
  if (!workspace || !grad_wk || !grad_b0) return jspsr::fail(JSPSR_EINVAL, "prop_backward_fold: null pointer");
  Geom g;
  if (int e = make_geom(B, H, W, g)) return e;
  hipLaunchKernelGGL(prop_bwd_finalize, dim3(NRED), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(workspace), -1, grad_wk, grad_b0);
  return jspsr::check_launch("prop_backward_fold");
}

// ---- the in-model form: logits + offsets as planes of ONE (B,25,H,W) tensor (include/jspsr_hip.h, jspsr_prop_logits_*) ----
extern "C" int jspsr_prop_logits_forward_f32(const float* dem, const float* head, const float* wk, const float* b0, float scale,
                                             float* out, int B, int H, int W, jspsr_stream_t stream) {
  if (!dem || !head || !wk || !b0 || !out) return jspsr::fail(JSPSR_EINVAL, "prop_logits_forward: null pointer");
  Geom g;
  if (int e = make_geom(B, H, W, g)) return e;
  for (const void* p : {(const void*)dem, (const void*)head, (const void*)out})
    if (!jspsr::aligned4(p)) return jspsr::fail(JSPSR_EALIGN, "prop_logits_forward: pointer not 4-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (jspsr::prop_dma_ok(B, H, W, 16, {dem, head, out}))
    return jspsr::prop_dma_logits_forward(dem, head, wk, b0, scale, out, B, H, W, s);
  // general shapes (W % 4 != 0, unaligned operands): the one-pixel-per-lane kernel on 64 x 8 tiles
  g.th = 8; g.tw = 64;
  g.tiles_x = (W + 63) / 64; g.tiles_y = (H + 7) / 8;
  g.nblk = B * g.tiles_x * g.tiles_y;
  g.dem_vec4 = can_vec(W, 4, {dem});
  const size_t P = (size_t)H * W;
  hipLaunchKernelGGL((prop_fwd_kernel<16, 1, true, 8, 64, true>), dim3(g.nblk), dim3(NT), 0, s, dem, head, head + 9 * P, wk, b0, scale,
                     out, g, 25 * P, 25 * P);
  return jspsr::check_launch("prop_logits_forward");
}

extern "C" int jspsr_prop_logits_backward_f32(const float* grad_out, const float* dem, const float* head, const float* wk,
                                              float* grad_head, float* grad_wk, float* grad_b0, void* workspace, int B, int H,
                                              int W, jspsr_stream_t stream) {
  if (!grad_out || !dem || !head || !wk || !grad_head || !workspace || (!grad_wk != !grad_b0))
    return jspsr::fail(JSPSR_EINVAL, "prop_logits_backward: null pointer");
  Geom g;
  if (int e = make_geom(B, H, W, g)) return e;
  for (const void* p : {(const void*)grad_out, (const void*)dem, (const void*)head, (const void*)grad_head})
    if (!jspsr::aligned4(p)) return jspsr::fail(JSPSR_EALIGN, "prop_logits_backward: pointer not 4-byte aligned");
  if (!jspsr::aligned16(workspace)) return jspsr::fail(JSPSR_EALIGN, "prop_logits_backward: workspace not 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  if (jspsr::prop_dma_ok(B, H, W, 16, {grad_out, dem, head, grad_head})) {
    if (int e = jspsr::prop_dma_logits_backward(grad_out, dem, head, wk, grad_head, partial, B, H, W, s)) return e;
  } else {
    g.th = 8; g.tw = 64;
    g.tiles_x = (W + 63) / 64; g.tiles_y = (H + 7) / 8;
    g.nblk = B * g.tiles_x * g.tiles_y;
    g.dem_vec4 = can_vec(W, 4, {dem});
    const size_t P = (size_t)H * W;
    hipLaunchKernelGGL((prop_bwd_kernel<16, 1, true, 8, 64, true>), dim3(g.nblk), dim3(NT), 0, s, grad_out, dem, head, head + 9 * P, wk,
                       grad_head, grad_head + 9 * P, partial, g, 25 * P, 25 * P);
    if (int e = jspsr::check_launch("prop_logits_backward")) return e;
  }
  if (!grad_wk) return JSPSR_OK;
  hipLaunchKernelGGL(prop_bwd_finalize, dim3(NRED), dim3(256), 0, s, partial, -1, grad_wk, grad_b0);
  return jspsr::check_launch("prop_backward_finalize");
}
