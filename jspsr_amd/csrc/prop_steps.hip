// K1s -- one propagation step in its general form, for chains of steps (gfx950).
//
// prop.hip is PostProcessor.forward exactly as JSPSR / LRRU / EDSR call it: zero-sum affinities, no gradient with
// respect to the DEM (every caller detaches it).  The N-iteration users of the same sampler need two more things
// (models/components/nlspn.py:177-233, `_propagate_once` inside the `prop_time` loop):
//   * affinities taken as they are (NLSPN normalises them once, outside the loop: nlspn.py:158-173) -> NORM = false:
//       out = b0 + sum_k wk[k] a_k S_k + scale * dem
//   * the gradient with respect to the propagated raster itself, because iteration i+1 reads iteration i's output:
//       grad_dem[q] += sum_{p,k} g_p wk[k] m_k(p) dS_k(p)/d dem[q]  (+ scale * g_q)
//     a bilinear SCATTER of every tap into its four corners.  Round 4: no float atomics on the common path.  Corners inside
//     tile + halo are accumulated in the tile's LDS window as 64-bit FIXED-POINT integers (ds_add_u64: exact, order-
//     independent; scale chosen per tile), the window (24 x 80) is written back as floats to a compact buffer in the
//     workspace, and a second, pixel-ordered pass (prop_step_gdem_gather_kernel) lets every pixel GATHER the 3..6 windows
//     that cover it, in fixed order, and adds the sum into grad_dem.  Rounds 2-3 used ds_add_f32 (~170 cycles per wave
//     instruction on gfx950: 320 of the kernel's 450 us) and flushed every window with device-scope float atomics.
//     Only taps BEYOND tile + halo (|offset| > 8 px) still add to grad_dem with global float atomics, so grad_dem is
//     bit-reproducible whenever no such tap exists (everything else always is).
//   * gradients of the fixed affinities / offsets SUMMED over the iterations -> ACC: add into grad_weight / grad_offset.
// Same tile (8 x 64, one pixel per lane), same LDS staging and border rule as prop.hip.
#include "prop_tile.h"

namespace {

constexpr int STW = 64, STH = 8, SLW = STW + 2 * HALO, SLH = STH + 2 * HALO, SRPP = NT / STW;

struct SGeom {
  int B, H, W, tiles_x, tiles_y, nblk;
  int dem_vec4;
};

__device__ __forceinline__ void stile_coords(const SGeom& g, int& b, int& ty0, int& tx0) {
  const int t = jspsr::xcd_contiguous(blockIdx.x, g.nblk);
  const int per_img = g.tiles_x * g.tiles_y;
  b = t / per_img;
  const int r = t - b * per_img;
  ty0 = (r / g.tiles_x) * STH;
  tx0 = (r % g.tiles_x) * STW;
}

template <int OC>
__device__ __forceinline__ constexpr int soch(int k, int c) {
  return OC == 18 ? 2 * k + c : 2 * (k < 4 ? k : k - 1) + c;
}

// Bilinear scatter of `v` into the four corners of (py, px): the transpose of corners_fast().
// Corners inside tile + halo go to the tile's LDS window as 64-BIT FIXED-POINT integers (ds_add_u64): on gfx950 a
// ds_add_f32 costs ~170 cycles per wave instruction (measured: 36 of them per pixel were 320 of the kernel's 450 us,
// profiles/r04_k1s_backward.txt) while the integer atomics run at the LDS's normal rate -- and integer addition is
// associative, so the window's content does not depend on the order the waves arrive in (bit-reproducible without
// per-wave windows).  `fscale` = 2^e, chosen per tile so that the largest |contribution| sits at 2^45: 24 significant bits
// of every fp32 product survive exactly, and 2^17 such terms cannot overflow.
__device__ __forceinline__ unsigned long long to_fixed(float c) { return (unsigned long long)(long long)c; }

template <int LH, int LW>
__device__ __forceinline__ void scatter_corners(unsigned long long* __restrict__ ldsacc, float* __restrict__ gimg, int H, int W,
                                                int ly0, int lx0, float py, float px, float v, float fscale) {
  const float fy = floorf(py), fx = floorf(px);
  const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
  const int y0 = (int)fminf(fmaxf(fy, -1.0e9f), 1.0e9f), x0 = (int)fminf(fmaxf(fx, -1.0e9f), 1.0e9f);
  const int ry = y0 - ly0, rx = x0 - lx0;
  const bool inl = (unsigned)ry < (unsigned)(LH - 1) && (unsigned)rx < (unsigned)(LW - 1);
  const bool near = (py > -2.f) && (py < (float)(H + 1)) && (px > -2.f) && (px < (float)(W + 1));
  if (inl) {
    unsigned long long* p = ldsacc + ry * LW + rx;      // slots outside the raster collect values that are never gathered
    const float vs = v * fscale;                        // exact: a power of two
#ifdef K1S_LAB_F32ATOMICS                               // lab: rounds 2-3's ds_add_f32 on the low words (timing only, WRONG results)
    float* q_ = reinterpret_cast<float*>(p);
    __hip_atomic_fetch_add(q_, vs * hy * hx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(q_ + 2, vs * hy * lx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(q_ + 2 * LW, vs * ly * hx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(q_ + 2 * LW + 2, vs * ly * lx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    atomicAdd(p, to_fixed(vs * hy * hx));
    atomicAdd(p + 1, to_fixed(vs * hy * lx));
    atomicAdd(p + LW, to_fixed(vs * ly * hx));
    atomicAdd(p + LW + 1, to_fixed(vs * ly * lx));
#endif
  } else if (near) {
    // a tap beyond tile + halo (|offset| > 8 px): global float atomics -- the one place where grad_dem's last bits depend
    // on the execution order
    const bool y0ok = (unsigned)y0 < (unsigned)H, y1ok = (unsigned)(y0 + 1) < (unsigned)H;
    const bool x0ok = (unsigned)x0 < (unsigned)W, x1ok = (unsigned)(x0 + 1) < (unsigned)W;
    float* q = gimg + (ptrdiff_t)y0 * W + x0;
    if (y0ok && x0ok) __hip_atomic_fetch_add(q, v * hy * hx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (y0ok && x1ok) __hip_atomic_fetch_add(q + 1, v * hy * lx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (y1ok && x0ok) __hip_atomic_fetch_add(q + W, v * ly * hx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (y1ok && x1ok) __hip_atomic_fetch_add(q + W + 1, v * ly * lx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int OC, bool NORM>
__global__ __launch_bounds__(NT) void prop_step_fwd_kernel(const float* __restrict__ dem, const float* __restrict__ weight,
                                                          const float* __restrict__ offset, const float* __restrict__ wk,
                                                          const float* __restrict__ b0, float scale,
                                                          float* __restrict__ out, SGeom g) {
  __shared__ __attribute__((aligned(16))) float lds[SLH * SLW];
  int b, ty0, tx0;
  stile_coords(g, b, ty0, tx0);
  const int H = g.H, W = g.W;
  const size_t P = (size_t)H * W;
  const float* img = dem + (size_t)b * P;
  stage_dem<SLH, SLW>(lds, img, ty0, tx0, H, W, g.dem_vec4 != 0);
  float wreg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wreg[k] = wk[k];
  const float bias = b0[0];
  __syncthreads();
  const int x = tx0 + (int)(threadIdx.x % STW);
  const int ly0 = ty0 - HALO, lx0 = tx0 - HALO;
  if (x >= W) return;
#pragma unroll 1
  for (int y = ty0 + (int)(threadIdx.x / STW); y < min(ty0 + STH, H); y += SRPP) {
    const size_t pix = (size_t)y * W + x;
    const float* wp = weight + (size_t)b * 9 * P + pix;
    const float* op = offset + (size_t)b * OC * P + pix;
    float a[9], oy[9], ox[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = wp[k * P];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (OC == 18 || k != 4) {
        oy[k] = op[(size_t)soch<OC>(k, 0) * P];
        ox[k] = op[(size_t)soch<OC>(k, 1) * P];
      } else {
        oy[k] = ox[k] = 0.f;
      }
    }
    const float dc = img[pix];
    float mean = 0.f;
    if (NORM) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) s += a[k];
      mean = s / 9.f;
    }
    float acc = bias;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float py = (float)(y - 1 + k / 3) + oy[k];
      const float px = (float)(x - 1 + k % 3) + ox[k];
      const Corners c = corners_fast<SLH, SLW>(lds, img, H, W, ly0, lx0, py, px);
      const float hy = 1.f - c.ly, hx = 1.f - c.lx;
      const float S = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
      acc += wreg[k] * (a[k] - mean) * S;
    }
    out[(size_t)b * P + pix] = acc + scale * dc;
  }
}

template <int OC, bool NORM, bool GDEM, bool ACC>
__global__ __launch_bounds__(NT) void prop_step_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ dem,
                                                          const float* __restrict__ weight, const float* __restrict__ offset,
                                                          const float* __restrict__ wk, float scale,
                                                          float* __restrict__ gweight, float* __restrict__ goffset,
                                                          float* __restrict__ gdem, float* __restrict__ partial, float* __restrict__ windows,
                                                          SGeom g) {
  __shared__ __attribute__((aligned(16))) float lds[SLH * SLW];
  __shared__ __attribute__((aligned(16))) unsigned long long ldsacc[GDEM ? SLH * SLW : 2];     // the tile's window, 64-bit fixed point
  __shared__ float tmax[NT / 64];
  __shared__ float red[NT / 64][NRED];
  int b, ty0, tx0;
  stile_coords(g, b, ty0, tx0);
  const int H = g.H, W = g.W;
  const size_t P = (size_t)H * W;
  const float* img = dem + (size_t)b * P;
  float* gimg = GDEM ? gdem + (size_t)b * P : nullptr;
  stage_dem<SLH, SLW>(lds, img, ty0, tx0, H, W, g.dem_vec4 != 0);
  if (GDEM)
    for (int i = threadIdx.x; i < SLH * SLW / 2; i += NT) reinterpret_cast<uint4*>(ldsacc)[i] = make_uint4(0u, 0u, 0u, 0u);
  float wreg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wreg[k] = wk[k];
  const int x = tx0 + (int)(threadIdx.x % STW);
  const int ly0 = ty0 - HALO, lx0 = tx0 - HALO;
  if (GDEM) {
    // the tile's largest |contribution| to the raster gradient (g w_k m_k before the bilinear weights, and scale * g): sets
    // the fixed-point scale.  gout and the nine affinities are read here and again in the main loop (from L2 / L1 then).
    float cmax = 0.f;
    if (x < W) {
      for (int y = ty0 + (int)(threadIdx.x / STW); y < min(ty0 + STH, H); y += SRPP) {
        const size_t pix = (size_t)y * W + x;
        const float* wp = weight + (size_t)b * 9 * P + pix;
        const float gj = gout[(size_t)b * P + pix];
        float a[9], mean = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = wp[k * P];
        if (NORM) {
          float s_ = 0.f;
#pragma unroll
          for (int k = 0; k < 9; ++k) s_ += a[k];
          mean = s_ / 9.f;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float c = fabsf(gj * wreg[k] * (a[k] - mean));
          if (c <= 3.0e38f) cmax = fmaxf(cmax, c);        // (false for NaN / inf: such a tap is dropped below)
        }
        const float r_ = fabsf(scale * gj);
        if (r_ <= 3.0e38f) cmax = fmaxf(cmax, r_);
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, d, 64));
    if ((threadIdx.x & 63) == 0) tmax[threadIdx.x >> 6] = cmax;
  }
  __syncthreads();
  float fscale = 1.f;
  int fexp = 0;
  if (GDEM) {
    float t_ = tmax[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) t_ = fmaxf(t_, tmax[w]);
    fexp = t_ > 0.f ? min(100, 45 - ilogbf(t_)) : 0;       // the largest term lands in [2^45, 2^46)
    fscale = ldexpf(1.f, fexp);
  }
  float sums[NRED];
#pragma unroll
  for (int i = 0; i < NRED; ++i) sums[i] = 0.f;
  if (x < W) {
#pragma unroll 1
    for (int y = ty0 + (int)(threadIdx.x / STW); y < min(ty0 + STH, H); y += SRPP) {
      const size_t pix = (size_t)y * W + x;
      const float* wp = weight + (size_t)b * 9 * P + pix;
      const float* op = offset + (size_t)b * OC * P + pix;
      float* gop = goffset + (size_t)b * OC * P + pix;
      float* gwp = gweight + (size_t)b * 9 * P + pix;
      float a[9], oy[9], ox[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) a[k] = wp[k * P];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (OC == 18 || k != 4) {
          oy[k] = op[(size_t)soch<OC>(k, 0) * P];
          ox[k] = op[(size_t)soch<OC>(k, 1) * P];
        } else {
          oy[k] = ox[k] = 0.f;
        }
      }
      const float gj = gout[(size_t)b * P + pix];
      float mean = 0.f;
      if (NORM) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += a[k];
        mean = s / 9.f;
      }
      float gm[9], gy[9], gx[9], gsum = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float py = (float)(y - 1 + k / 3) + oy[k];
        const float px = (float)(x - 1 + k % 3) + ox[k];
        const Corners c = corners_fast<SLH, SLW>(lds, img, H, W, ly0, lx0, py, px);
        const float hy = 1.f - c.ly, hx = 1.f - c.lx;
        const float S = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
        const float dSdy = hx * (c.v10 - c.v00) + c.lx * (c.v11 - c.v01);
        const float dSdx = hy * (c.v01 - c.v00) + c.ly * (c.v11 - c.v10);
        const float m = a[k] - mean;
        const float coef = gj * wreg[k] * m;
        gy[k] = coef * dSdy;
        gx[k] = coef * dSdx;
        gm[k] = gj * wreg[k] * S;
        gsum += gm[k];
        sums[k] += gj * m * S;
        if (GDEM) scatter_corners<SLH, SLW>(ldsacc, gimg, H, W, ly0, lx0, py, px, fabsf(coef) <= 3.0e38f ? coef : 0.f, fscale);
      }
      sums[9] += gj;
      if (GDEM && scale != 0.f && fabsf(scale * gj) <= 3.0e38f)
        atomicAdd(ldsacc + (y - ly0) * SLW + (x - lx0), to_fixed(scale * gj * fscale));
      if (NORM) {
        gsum /= 9.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) gm[k] -= gsum;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (ACC) gm[k] += gwp[k * P];
        __builtin_nontemporal_store(gm[k], gwp + k * P);
        if (OC == 18 || k != 4) {
          float* py_ = gop + (size_t)soch<OC>(k, 0) * P;
          float* px_ = gop + (size_t)soch<OC>(k, 1) * P;
          if (ACC) {
            gy[k] += *py_;
            gx[k] += *px_;
          }
          __builtin_nontemporal_store(gy[k], py_);
          __builtin_nontemporal_store(gx[k], px_);
        }
      }
    }
  }
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
  if (GDEM) {
    // the tile's window, back from fixed point (barrier above: every ds_add has landed), written whole -- 24 x 80 floats, plain
    // 16-byte stores -- to this tile's slot of the window buffer; the gather pass does the rest
    const int t = jspsr::xcd_contiguous(blockIdx.x, g.nblk);
    float4* dst = reinterpret_cast<float4*>(windows + (size_t)t * (SLH * SLW));
    const double inv = ldexp(1.0, -fexp);
    for (int i = threadIdx.x; i < SLH * SLW / 4; i += NT) {
      const long long* a_ = reinterpret_cast<const long long*>(ldsacc) + 4 * i;
      dst[i] = make_float4((float)((double)a_[0] * inv), (float)((double)a_[1] * inv), (float)((double)a_[2] * inv), (float)((double)a_[3] * inv));
    }
  }
}

// Second pass of the raster gradient: pixel (b, y, x) of tile (ty, tx) is covered by the windows of tiles (ty + dy, tx + dx),
// dy in {-1, 0, 1} (a window is 24 rows tall: 8 of halo either side of its 8 rows) and dx = -1 / +1 only within 8 pixels of
// the tile's left / right edge.  Four pixels per lane, fixed order of the (up to 6) terms, sum ADDED into grad_dem.
__global__ __launch_bounds__(256) void prop_step_gdem_gather_kernel(const float* __restrict__ windows, float* __restrict__ gdem, SGeom g) {
  const int W4 = (g.W + 3) / 4;
  const long long n = (long long)g.B * g.H * W4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W4) * 4;
    const long long r = i / W4;
    const int y = (int)(r % g.H), b = (int)(r / g.H);
    const int ty = y / STH, tx = x / STW, xr = x - tx * STW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int tyy = ty + dy;
      if (tyy < 0 || tyy >= g.tiles_y) continue;
      const int wy = y - (tyy * STH - HALO);                  // row of this pixel inside that tile's window: 0 .. 23 by construction
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int txx = tx + dx;
        if (txx < 0 || txx >= g.tiles_x) continue;
        if ((dx < 0 && xr >= HALO) || (dx > 0 && xr + 4 <= STW - HALO)) continue;      // (x % 4 == 0 and HALO % 4 == 0: all four pixels or none)
        const int wx = x - (txx * STW - HALO);
        const float* wrow = windows + ((size_t)((size_t)b * g.tiles_y + tyy) * g.tiles_x + txx) * (SLH * SLW) + wy * SLW + wx;
        const float4 v = *reinterpret_cast<const float4*>(wrow);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      }
    }
    float* q = gdem + ((size_t)b * g.H + y) * g.W + x;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x + j < g.W) q[j] += acc[j];
  }
}

int make_sgeom(int B, int H, int W, SGeom& g) {
  if (B <= 0 || H <= 0 || W <= 0) return jspsr::fail(JSPSR_EINVAL, "prop_step: bad shape B=%d H=%d W=%d", B, H, W);
  g.B = B;
  g.H = H;
  g.W = W;
  g.tiles_x = (W + STW - 1) / STW;
  g.tiles_y = (H + STH - 1) / STH;
  const long long n = (long long)B * g.tiles_x * g.tiles_y;
  if (n > 0x7fffffffLL || (long long)B * 18 * H * W > (1LL << 40)) return jspsr::fail(JSPSR_EINVAL, "prop_step: problem too large");
  g.nblk = (int)n;
  return JSPSR_OK;
}

}  // namespace

extern "C" int jspsr_prop_step_forward_f32(const float* dem, const float* weight, const float* offset, int offset_channels,
                                           const float* wk, const float* b0, float scale, int normalize, float* out,
                                           int B, int H, int W, jspsr_stream_t stream) {
  if (!dem || !weight || !offset || !wk || !b0 || !out) return jspsr::fail(JSPSR_EINVAL, "prop_step_forward: null pointer");
  if (offset_channels != 16 && offset_channels != 18)
    return jspsr::fail(JSPSR_EINVAL, "prop_step_forward: offset_channels must be 16 or 18, got %d", offset_channels);
  if (dem == out) return jspsr::fail(JSPSR_EINVAL, "prop_step_forward: in-place propagation is not possible (neighbours are read)");
  SGeom g;
  if (int e = make_sgeom(B, H, W, g)) return e;
  for (const void* p : {(const void*)dem, (const void*)weight, (const void*)offset, (const void*)out})
    if (!jspsr::aligned4(p)) return jspsr::fail(JSPSR_EALIGN, "prop_step_forward: pointer not 4-byte aligned");
  g.dem_vec4 = (W % 4 == 0) && jspsr::aligned16(dem);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define GO(OC, NORM) hipLaunchKernelGGL((prop_step_fwd_kernel<OC, NORM>), dim3(g.nblk), dim3(NT), 0, s, dem, weight, offset, wk, b0, scale, out, g)
  if (offset_channels == 18) { if (normalize) GO(18, true); else GO(18, false); }
  else { if (normalize) GO(16, true); else GO(16, false); }
#undef GO
  return jspsr::check_launch("prop_step_forward");
}

extern "C" size_t jspsr_prop_step_backward_workspace_bytes(int B, int H, int W) {
  SGeom g;
  if (make_sgeom(B, H, W, g)) return 0;
  // partial rows of the parameter gradients, then one 24 x 80 window per tile for the raster gradient's gather pass
  return (((size_t)g.nblk * NRED * sizeof(float) + 15) & ~(size_t)15) + (size_t)g.nblk * SLH * SLW * sizeof(float);
}

extern "C" int jspsr_prop_step_backward_f32(const float* grad_out, const float* dem, const float* weight, const float* offset,
                                            int offset_channels, const float* wk, float scale, int normalize, int accumulate,
                                            float* grad_weight, float* grad_offset, float* grad_dem, float* grad_wk,
                                            float* grad_b0, void* workspace, int B, int H, int W, jspsr_stream_t stream) {
  if (!grad_out || !dem || !weight || !offset || !wk || !grad_weight || !grad_offset || !workspace || (!grad_wk != !grad_b0))
    return jspsr::fail(JSPSR_EINVAL, "prop_step_backward: null pointer");
  if (offset_channels != 16 && offset_channels != 18)
    return jspsr::fail(JSPSR_EINVAL, "prop_step_backward: offset_channels must be 16 or 18, got %d", offset_channels);
  if (grad_dem && (grad_dem == dem || grad_dem == grad_out))
    return jspsr::fail(JSPSR_EINVAL, "prop_step_backward: grad_dem must not alias dem / grad_out");
  SGeom g;
  if (int e = make_sgeom(B, H, W, g)) return e;
  for (const void* p : {(const void*)grad_out, (const void*)dem, (const void*)weight, (const void*)offset,
                        (const void*)grad_weight, (const void*)grad_offset, (const void*)grad_dem})
    if (p && !jspsr::aligned4(p)) return jspsr::fail(JSPSR_EALIGN, "prop_step_backward: pointer not 4-byte aligned");
  if (!jspsr::aligned16(workspace)) return jspsr::fail(JSPSR_EALIGN, "prop_step_backward: workspace not 16-byte aligned");
  g.dem_vec4 = (W % 4 == 0) && jspsr::aligned16(dem);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  float* windows = reinterpret_cast<float*>(static_cast<char*>(workspace) + (((size_t)g.nblk * NRED * sizeof(float) + 15) & ~(size_t)15));
#define GO(OC, NORM, GD, AC) hipLaunchKernelGGL((prop_step_bwd_kernel<OC, NORM, GD, AC>), dim3(g.nblk), dim3(NT), 0, s, grad_out, dem, weight, offset, wk, scale, grad_weight, grad_offset, grad_dem, partial, windows, g)
#define BY_AC(OC, NORM, GD) do { if (accumulate) GO(OC, NORM, GD, true); else GO(OC, NORM, GD, false); } while (0)
#define BY_GD(OC, NORM) do { if (grad_dem) BY_AC(OC, NORM, true); else BY_AC(OC, NORM, false); } while (0)
#define BY_NORM(OC) do { if (normalize) BY_GD(OC, true); else BY_GD(OC, false); } while (0)
  if (offset_channels == 18) BY_NORM(18); else BY_NORM(16);
#undef BY_NORM
#undef BY_GD
#undef BY_AC
#undef GO
  if (int e = jspsr::check_launch("prop_step_backward")) return e;
  if (grad_dem) {
    const long long n4 = (long long)B * H * ((W + 3) / 4);
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(prop_step_gdem_gather_kernel, dim3(blocks), dim3(256), 0, s, windows, grad_dem, g);
    if (int e = jspsr::check_launch("prop_step_gdem_gather")) return e;
  }
  if (!grad_wk) return JSPSR_OK;
  hipLaunchKernelGGL(prop_bwd_finalize, dim3(NRED), dim3(256), 0, s, partial, g.nblk, grad_wk, grad_b0);
  return jspsr::check_launch("prop_step_backward_finalize");
}
