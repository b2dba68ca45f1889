// K1h -- the propagation step fed straight from the generator head's NHWC output (gfx950).
//
// Same arithmetic as prop.hip (PostProcessor.forward, models/components/spn.py:99-118, on the two 1x1 heads of
// Generator.forward, spn.py:66-73), other operand layout: inside the models the affinity logits and the learned
// offsets are ONE 32-channel NHWC tensor written by the merged 1x1 head convolution.  Reading it where it lies
// removes, per step, the sigmoid pass, two NHWC -> planar fp32 transposes and their three backward passes, and the
// sigmoid is evaluated in fp32 from the logits (a bf16 affinity rounded at ~0.5 loses 3 % of `a - mean a`).
//
// Channel order ("tap-major"): c = 4 t + j for the 8 learned taps t (k = t < 4 ? t : t + 1, row-major over the 3x3
// window), j = 0 affinity logit of tap k, j = 1 dy_k, j = 2 dx_k, j = 3: the centre tap's affinity logit for t == 0,
// unused (zero weights, zero gradient) otherwise.  16 bytes of a pixel = one tap (fp32) or two (bf16).
//
// Data movement: LPP = 8 (fp32) / 4 (bf16) neighbouring lanes share a pixel, one 16-byte load each, so a wave
// instruction reads 1 KiB contiguous; a lane evaluates its tap(s), the pixel's sums are formed by DPP butterflies
// inside the lane group; the backward writes the head's gradient (sigmoid backward included) in the same layout and
// dtype, 16 bytes per lane, ready for the head convolution's data / weight gradient.  DEM tile + halo in LDS as in
// prop.hip.
#include "prop_tile.h"

#include <initializer_list>

namespace jspsr {   // prop_head_dma.hip: the persistent LDS-DMA form for bf16 heads (what the benchmarked step runs)
bool prop_head_dma_ok(int B, int H, int W, std::initializer_list<const void*> ptrs);
int prop_head_dma_forward(const float* dem, const void* head, const float* wk, const float* b0, float scale, float* out, int B,
                          int H, int W, hipStream_t s);
int prop_head_dma_backward(const float* gout, const float* dem, const void* head, const float* wk, void* ghead, float* partial,
                           int B, int H, int W, hipStream_t s);
}  // namespace jspsr

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HTW = 64, HTH = 8;           // tile of a workgroup: 4 waves x 2 rows x 64 pixels
constexpr int HLW = HTW + 2 * HALO, HLH = HTH + 2 * HALO;
constexpr int HROWS = HTH / 4;             // rows per wave

template <typename T> struct HeadLane;      // how 16 bytes of head channels split into taps
template <> struct HeadLane<float> { static constexpr int TPL = 1; };
template <> struct HeadLane<__bf16> { static constexpr int TPL = 2; };

template <typename T>
__device__ __forceinline__ void unpack16(const uint4& raw, float (&f)[HeadLane<T>::TPL][4]) {
  if constexpr (sizeof(T) == 4) {
    f[0][0] = __uint_as_float(raw.x); f[0][1] = __uint_as_float(raw.y);
    f[0][2] = __uint_as_float(raw.z); f[0][3] = __uint_as_float(raw.w);
  } else {
    f[0][0] = __uint_as_float(raw.x << 16); f[0][1] = __uint_as_float(raw.x & 0xffff0000u);
    f[0][2] = __uint_as_float(raw.y << 16); f[0][3] = __uint_as_float(raw.y & 0xffff0000u);
    f[1][0] = __uint_as_float(raw.z << 16); f[1][1] = __uint_as_float(raw.z & 0xffff0000u);
    f[1][2] = __uint_as_float(raw.w << 16); f[1][3] = __uint_as_float(raw.w & 0xffff0000u);
  }
}

__device__ __forceinline__ unsigned bf16_bits(float x) {  // round-to-nearest-even, NaN preserved
  const __bf16 h = (__bf16)x;
  return (unsigned)__builtin_bit_cast(unsigned short, h);
}

template <typename T>
__device__ __forceinline__ uint4 pack16(const float (&f)[HeadLane<T>::TPL][4]) {
  uint4 raw;
  if constexpr (sizeof(T) == 4) {
    raw.x = __float_as_uint(f[0][0]); raw.y = __float_as_uint(f[0][1]);
    raw.z = __float_as_uint(f[0][2]); raw.w = __float_as_uint(f[0][3]);
  } else {
    raw.x = bf16_bits(f[0][0]) | (bf16_bits(f[0][1]) << 16);
    raw.y = bf16_bits(f[0][2]) | (bf16_bits(f[0][3]) << 16);
    raw.z = bf16_bits(f[1][0]) | (bf16_bits(f[1][1]) << 16);
    raw.w = bf16_bits(f[1][2]) | (bf16_bits(f[1][3]) << 16);
  }
  return raw;
}

// Sum over the LPP lanes of a pixel (every lane ends up with the total): DPP quad_perm xor 1, xor 2, then the
// half-row mirror (lane i <-> 7 - i of each 8) for the 8-lane groups.
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  if constexpr (LPP == 8)
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): |error| < 2e-7 on a value in (0, 1)
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

struct HGeom {
  int B, H, W, tiles_x, tiles_y, nblk;
  int dem_vec4;
};

__device__ __forceinline__ void htile_coords(const HGeom& g, int& b, int& ty0, int& tx0) {
  const int t = jspsr::xcd_contiguous(blockIdx.x, g.nblk);
  const int per_img = g.tiles_x * g.tiles_y;
  b = t / per_img;
  const int r = t - b * per_img;
  ty0 = (r / g.tiles_x) * HTH;
  tx0 = (r % g.tiles_x) * HTW;
}

// BWD = false: out = b0 + sum_k wk (a_k - mean a) S_k + scale * dem.
// BWD = true : ghead (same layout and dtype as head) = d/d(head) incl. the sigmoid's derivative, and one row of 10
//              partial sums (grad_wk[9], grad_b0) per workgroup.
template <typename T, bool BWD>
__global__ __launch_bounds__(NT) void prop_head_kernel(const float* __restrict__ dem, const T* __restrict__ head,
                                                      const float* __restrict__ wk, const float* __restrict__ b0,
                                                      float scale, float* __restrict__ out,
                                                      const float* __restrict__ gout, T* __restrict__ ghead,
                                                      float* __restrict__ partial, HGeom g) {
  constexpr int TPL = HeadLane<T>::TPL, LPP = 8 / TPL, PPW = 64 / LPP, RS = HTW / PPW;   // RS wave-steps per row
  __shared__ __attribute__((aligned(16))) float lds[HLH * HLW];
  __shared__ float red[NT / 64][NRED];
  int b, ty0, tx0;
  htile_coords(g, b, ty0, tx0);
  const int H = g.H, W = g.W;
  const size_t P = (size_t)H * W;
  const float* img = dem + (size_t)b * P;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPP, pxl = lane / LPP;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

  stage_dem<HLH, HLW>(lds, img, ty0, tx0, H, W, g.dem_vec4 != 0);
  // this lane's tap(s): window position and tap weight
  float wt[TPL];
  int ky[TPL], kx[TPL];
#pragma unroll
  for (int i = 0; i < TPL; ++i) {
    const int t = sub * TPL + i, k = t < 4 ? t : t + 1;
    wt[i] = wk[k];
    ky[i] = k / 3 - 1;
    kx[i] = k % 3 - 1;
  }
  const float w4 = wk[4];
  const float bias = BWD ? 0.f : b0[0];
  __syncthreads();
  const int ly0 = ty0 - HALO, lx0 = tx0 - HALO;
  // parameter-gradient partial sums in fp64: sum g and sum g m S cancel heavily (|sum| ~ 1e-2 of sum |.|)
  double acc_t[TPL], acc_c = 0.0, acc_g = 0.0;
#pragma unroll
  for (int i = 0; i < TPL; ++i) acc_t[i] = 0.0;

#pragma unroll 1
  for (int r = 0; r < HROWS; ++r) {
    const int y = ty0 + wave + 4 * r;
    // one row of this wave: RS contiguous 1-KiB loads in flight, then the arithmetic
    uint4 raw[RS];
    float gj[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const int x = tx0 + s * PPW + pxl;
      const bool ok = y < H && x < W;
      const size_t pix = (size_t)b * P + (size_t)(ok ? y : 0) * W + (ok ? x : 0);
      raw[s] = zero4;
      if (ok) raw[s] = *reinterpret_cast<const uint4*>(head + pix * 32 + sub * (4 * TPL));
      if (BWD) gj[s] = ok ? gout[pix] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const int x = tx0 + s * PPW + pxl;
      const bool ok = y < H && x < W;         // lanes of one pixel agree; inactive pixels run on zeros
      float f[TPL][4];
      unpack16<T>(raw[s], f);
      const float dc = lds[(y - ly0) * HLW + (x - lx0)];     // dem[y][x]: the centre tap's sample, and the residual
      float a[TPL], S[TPL], dSy[TPL], dSx[TPL];
      float asum = 0.f;
#pragma unroll
      for (int i = 0; i < TPL; ++i) {
        a[i] = sigmoidf(f[i][0]);
        asum += a[i];
        const float py = (float)(y + ky[i]) + f[i][1];
        const float px = (float)(x + kx[i]) + f[i][2];
        const Corners c = corners_fast<HLH, HLW>(lds, img, H, W, ly0, lx0, py, px);
        const float hy = 1.f - c.ly, hx = 1.f - c.lx;
        S[i] = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
        if (BWD) {
          dSy[i] = hx * (c.v10 - c.v00) + c.lx * (c.v11 - c.v01);
          dSx[i] = hy * (c.v01 - c.v00) + c.ly * (c.v11 - c.v10);
        }
      }
      const float s4 = sigmoidf(f[0][3]);
      const float a4 = sub == 0 ? s4 : 0.f;                  // centre tap: zero offset, S = dem[y][x]
      const float mean = group_sum<LPP>(asum + a4) * (1.f / 9.f);
      if (!BWD) {
        float part = sub == 0 ? w4 * (a4 - mean) * dc : 0.f;
#pragma unroll
        for (int i = 0; i < TPL; ++i) part += wt[i] * (a[i] - mean) * S[i];
        const float tot = group_sum<LPP>(part);
        if (ok && sub == 0) out[(size_t)b * P + (size_t)y * W + x] = bias + tot + scale * dc;
      } else {
        const float gv = gj[s];
        float gm[TPL], gmsum = 0.f;
#pragma unroll
        for (int i = 0; i < TPL; ++i) {
          gm[i] = gv * wt[i] * S[i];
          gmsum += gm[i];
        }
        const float gm4 = sub == 0 ? gv * w4 * dc : 0.f;
        const float gs = group_sum<LPP>(gmsum + gm4) * (1.f / 9.f);
        float o[TPL][4];
#pragma unroll
        for (int i = 0; i < TPL; ++i) {
          const float m = a[i] - mean;
          const float coef = gv * wt[i] * m;
          o[i][0] = (gm[i] - gs) * a[i] * (1.f - a[i]);
          o[i][1] = coef * dSy[i];
          o[i][2] = coef * dSx[i];
          o[i][3] = 0.f;
          acc_t[i] += gv * m * S[i];
        }
        if (sub == 0) {
          o[0][3] = (gm4 - gs) * a4 * (1.f - a4);
          acc_c += gv * (a4 - mean) * dc;
          acc_g += gv;
        }
        if (ok) {
          const size_t pix = (size_t)b * P + (size_t)y * W + x;
          const uint4 packed = pack16<T>(o);
          // written once, never read back by this kernel: one 16-byte streaming store per lane
          u32x4 pv;
          pv.x = packed.x; pv.y = packed.y; pv.z = packed.z; pv.w = packed.w;
          __builtin_nontemporal_store(pv, reinterpret_cast<u32x4*>(ghead + pix * 32 + sub * (4 * TPL)));
        }
      }
    }
  }
  if (!BWD) return;

  // ---- parameter gradients: lanes with the same `sub` hold the same tap(s) ----
#pragma unroll
  for (int i = 0; i < TPL; ++i)
    for (int d = LPP; d < 64; d <<= 1) acc_t[i] += __shfl_xor(acc_t[i], d, 64);
  for (int d = LPP; d < 64; d <<= 1) {
    acc_c += __shfl_xor(acc_c, d, 64);
    acc_g += __shfl_xor(acc_g, d, 64);
  }
  if (lane < LPP) {
#pragma unroll
    for (int i = 0; i < TPL; ++i) {
      const int t = lane * TPL + i;
      red[wave][t < 4 ? t : t + 1] = (float)acc_t[i];
    }
    if (lane == 0) {
      red[wave][4] = (float)acc_c;
      red[wave][9] = (float)acc_g;
    }
  }
  __syncthreads();
  if (threadIdx.x < NRED) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) v += red[w][threadIdx.x];
    partial[4 + (size_t)blockIdx.x * NRED + threadIdx.x] = v;      // behind the 16-byte header (row count)
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int*>(partial)[0] = (int)gridDim.x;
}

int make_hgeom(int B, int H, int W, HGeom& g) {
  if (B <= 0 || H <= 0 || W <= 0) return jspsr::fail(JSPSR_EINVAL, "prop_head: bad shape B=%d H=%d W=%d", B, H, W);
  g.B = B;
  g.H = H;
  g.W = W;
  g.tiles_x = (W + HTW - 1) / HTW;
  g.tiles_y = (H + HTH - 1) / HTH;
  const long long n = (long long)B * g.tiles_x * g.tiles_y;
  if (n > 0x7fffffffLL || (long long)B * 32 * H * W > (1LL << 40)) return jspsr::fail(JSPSR_EINVAL, "prop_head: problem too large");
  g.nblk = (int)n;
  return JSPSR_OK;
}

}  // namespace

extern "C" int jspsr_prop_head_forward(int dtype, const float* dem, const void* head, const float* wk, const float* b0,
                                       float scale, float* out, int B, int H, int W, jspsr_stream_t stream) {
  if (!dem || !head || !wk || !b0 || !out) return jspsr::fail(JSPSR_EINVAL, "prop_head_forward: null pointer");
  if (dtype != JSPSR_F32 && dtype != JSPSR_BF16) return jspsr::fail(JSPSR_EINVAL, "prop_head_forward: bad dtype %d", dtype);
  HGeom g;
  if (int e = make_hgeom(B, H, W, g)) return e;
  if (!jspsr::aligned16(head)) return jspsr::fail(JSPSR_EALIGN, "prop_head_forward: head must be 16-byte aligned");
  if (!jspsr::aligned4(dem) || !jspsr::aligned4(out)) return jspsr::fail(JSPSR_EALIGN, "prop_head_forward: pointer not 4-byte aligned");
  g.dem_vec4 = (W % 4 == 0) && jspsr::aligned16(dem);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == JSPSR_BF16 && jspsr::prop_head_dma_ok(B, H, W, {dem, head, out}))
    return jspsr::prop_head_dma_forward(dem, head, wk, b0, scale, out, B, H, W, s);
  if (dtype == JSPSR_F32)
    hipLaunchKernelGGL((prop_head_kernel<float, false>), dim3(g.nblk), dim3(NT), 0, s, dem, static_cast<const float*>(head), wk,
                       b0, scale, out, nullptr, nullptr, nullptr, g);
  else
    hipLaunchKernelGGL((prop_head_kernel<__bf16, false>), dim3(g.nblk), dim3(NT), 0, s, dem, static_cast<const __bf16*>(head),
                       wk, b0, scale, out, nullptr, nullptr, nullptr, g);
  return jspsr::check_launch("prop_head_forward");
}

extern "C" size_t jspsr_prop_head_backward_workspace_bytes(int B, int H, int W) {
  HGeom g;
  if (make_hgeom(B, H, W, g)) return 0;
  const size_t rows = g.nblk > 4096 ? (size_t)g.nblk : 4096;          // the DMA form's grid is capped at 4096 workgroups
  return 16 + ((rows * NRED * sizeof(float) + 15) & ~(size_t)15);      // header (row count) + rows, whichever kernel runs
}

extern "C" int jspsr_prop_head_backward(int dtype, const float* grad_out, const float* dem, const void* head,
                                        const float* wk, void* grad_head, float* grad_wk, float* grad_b0,
                                        void* workspace, int B, int H, int W, jspsr_stream_t stream) {
  if (!grad_out || !dem || !head || !wk || !grad_head || !workspace || (!grad_wk != !grad_b0))
    return jspsr::fail(JSPSR_EINVAL, "prop_head_backward: null pointer");
  if (dtype != JSPSR_F32 && dtype != JSPSR_BF16) return jspsr::fail(JSPSR_EINVAL, "prop_head_backward: bad dtype %d", dtype);
  HGeom g;
  if (int e = make_hgeom(B, H, W, g)) return e;
  if (!jspsr::aligned16(head) || !jspsr::aligned16(grad_head) || !jspsr::aligned16(workspace))
    return jspsr::fail(JSPSR_EALIGN, "prop_head_backward: head / grad_head / workspace must be 16-byte aligned");
  if (!jspsr::aligned4(dem) || !jspsr::aligned4(grad_out)) return jspsr::fail(JSPSR_EALIGN, "prop_head_backward: pointer not 4-byte aligned");
  g.dem_vec4 = (W % 4 == 0) && jspsr::aligned16(dem);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  if (dtype == JSPSR_BF16 && jspsr::prop_head_dma_ok(B, H, W, {grad_out, dem, head, grad_head})) {
    if (int e = jspsr::prop_head_dma_backward(grad_out, dem, head, wk, grad_head, partial, B, H, W, s)) return e;
    if (!grad_wk) return JSPSR_OK;
    hipLaunchKernelGGL(prop_bwd_finalize, dim3(NRED), dim3(256), 0, s, partial, -1, grad_wk, grad_b0);
    return jspsr::check_launch("prop_head_backward_finalize");
  }
  if (dtype == JSPSR_F32)
    hipLaunchKernelGGL((prop_head_kernel<float, true>), dim3(g.nblk), dim3(NT), 0, s, dem, static_cast<const float*>(head), wk,
                       nullptr, 0.f, nullptr, grad_out, static_cast<float*>(grad_head), partial, g);
  else
    hipLaunchKernelGGL((prop_head_kernel<__bf16, true>), dim3(g.nblk), dim3(NT), 0, s, dem, static_cast<const __bf16*>(head), wk,
                       nullptr, 0.f, nullptr, grad_out, static_cast<__bf16*>(grad_head), partial, g);
  if (int e = jspsr::check_launch("prop_head_backward")) return e;
  if (!grad_wk) return JSPSR_OK;
  hipLaunchKernelGGL(prop_bwd_finalize, dim3(NRED), dim3(256), 0, s, partial, -1, grad_wk, grad_b0);
  return jspsr::check_launch("prop_head_backward_finalize");
}
