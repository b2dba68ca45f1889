// K4/K5 -- the HBM-bound per-channel operators around the convolutions, NHWC, fp32 or bf16 storage
// with fp32 arithmetic:
//   * BatchNorm2d training/eval forward + backward, fused with ReLU and the residual add
//     (reference: models/components/basics.py:49-53,81-85,105-123)
//   * bias-less ReLU backward + bias gradient for the BN-free convs (basics.py:36-53)
//   * ChannelAttention pooling / scaling and their backward (resnet_cbam.py:36-53, basics.py:57-58)
//
// All reductions over pixels use one scheme: a (64 x 4)-thread workgroup owns a chunk of pixels
// and 64 channel groups (16 bytes of channels per lane -> every wave instruction reads whole
// 1 KiB / 512 B runs of the NHWC rows), accumulates in registers, folds its 4 pixel lanes through
// LDS and writes one partial row; a finalize kernel sums the partial rows in a fixed order (fp64)
// -> bit-reproducible statistics, no atomics.
#include "common.h"

#include <cstdlib>

namespace {

using namespace jspsr;

constexpr int TX = 64, TY = 4;

template <typename T> struct V;   // VEC channels per lane = 16 bytes
template <> struct V<float> { static constexpr int N = 4; };
template <> struct V<__bf16> { static constexpr int N = 8; };

template <typename T, int N>
struct Pack { T v[N]; };

// 16 bytes of T (4 fp32 / 8 bf16 channels) in two halves -- the raw bytes now, the fp32 values later -- so that the pixel
// walks below can keep several pixels in flight
template <typename T>
__device__ __forceinline__ uint4 load_raw(const T* p) { return *reinterpret_cast<const uint4*>(p); }
template <typename T>
__device__ __forceinline__ void unpack(const uint4& raw, float (&f)[V<T>::N]) {
  if constexpr (sizeof(T) == 4) {
    f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y); f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
  } else {
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
}

__device__ __forceinline__ unsigned bf16_bits(float x) {  // round-to-nearest-even, NaN preserved
  const __bf16 h = (__bf16)x;
  return (unsigned)__builtin_bit_cast(unsigned short, h);
}

template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float (&f)[V<T>::N]) {
  uint4 raw;
  if constexpr (sizeof(T) == 4) {
    raw.x = __float_as_uint(f[0]); raw.y = __float_as_uint(f[1]); raw.z = __float_as_uint(f[2]); raw.w = __float_as_uint(f[3]);
  } else {
    raw.x = bf16_bits(f[0]) | (bf16_bits(f[1]) << 16);
    raw.y = bf16_bits(f[2]) | (bf16_bits(f[3]) << 16);
    raw.z = bf16_bits(f[4]) | (bf16_bits(f[5]) << 16);
    raw.w = bf16_bits(f[6]) | (bf16_bits(f[7]) << 16);
  }
  *reinterpret_cast<uint4*>(p) = raw;
}

struct Slice {  // a channel slice of an NHWC tensor
  const void* p;
  int cs, coff;
};

struct RedGeom {
  long long npix;        // pixels reduced per segment
  int nseg;              // independent segments (1 for BN, B for per-image pooling)
  int C;                 // channels
  int chunks;            // pixel chunks per segment
  long long chunk_pix;   // pixels per chunk
  int lpp;               // lanes (16-byte channel groups) per pixel inside a 64-lane row: min(64, C/N)
  int pl;                // pixels handled side by side by one 64-lane row: 64 / lpp
};

// lane -> (channel group, pixel sub-lane).  Narrow tensors (C/N < 64) put several pixels in one
// 64-lane row so every lane works and a wave instruction still covers contiguous memory.
struct LaneMap {
  int c;      // first channel of this lane's group (>= C: idle lane)
  int sub;    // pixel sub-lane inside the row
};
template <int N>
__device__ __forceinline__ LaneMap lane_map(const RedGeom& g) {
  LaneMap m;
  const int tx = threadIdx.x;
  m.sub = tx / g.lpp;
  const int cg = blockIdx.y * TX + (tx - m.sub * g.lpp);
  m.c = (m.sub < g.pl) ? cg * N : g.C;
  return m;
}

RedGeom make_red(long long npix, int nseg, int C, int vec) {
  RedGeom g;
  g.npix = npix; g.nseg = nseg; g.C = C;
  const int gy = (C / vec + TX - 1) / TX;
  static const int target = [] { const char* e = getenv("JSPSR_RED_BLOCKS"); return e ? atoi(e) : 1024; }();   // 768-1024 measured best on MI355X (256 CUs x 3-4)
  long long want = target / ((long long)gy * nseg);
  if (want < 1) want = 1;
  long long chunks = (npix + 63) / 64;  // >= 64 pixels per chunk
  if (chunks > want) chunks = want;
  if (chunks < 1) chunks = 1;
  g.chunk_pix = (npix + chunks - 1) / chunks;
  g.chunks = (int)((npix + g.chunk_pix - 1) / g.chunk_pix);
  const int groups = C / vec;
  g.lpp = groups < TX ? groups : TX;
  g.pl = TX / g.lpp;
  return g;
}
dim3 red_grid(const RedGeom& g, int vec) { return dim3(g.nseg * g.chunks, (g.C / vec + TX - 1) / TX); }

// The two pixel walks of this file: a per-channel reduction (K accumulators per channel, partial layout
// [seg][chunk][K][C]) and the same geometry without one (a lane owns one 16-byte channel group, per-channel parameters in
// registers, and walks the pixels of its chunk) -- both with FOUR pixels in flight per lane since round 3 (one before).
// `load(pix, c, raw)` fetches the NTEN 16-byte pieces
// of one pixel, `use(pix, c, raw[, acc])` consumes them; a trip issues the loads of four pixels back to back and only then
// works through them, in pixel order (so sums keep their order and their bits).  One pixel per trip left 2-3 x 16 bytes
// per lane in flight -- about 32 KB per CU at 16 waves, under half of what hides an HBM miss; the BatchNorm backward
// pair ran at 4.5 TB/s effective where a plain stream reaches 5.5 (tools/bench_bn.py).
#ifndef K4_PIX_FLY
#define K4_PIX_FLY 4      // (lab builds: -DK4_PIX_FLY=2|8)
#endif
constexpr int PIX_FLY = K4_PIX_FLY;
template <typename T, int K, int NTEN, typename L, typename F>
__device__ __forceinline__ void reduce_pixels_mlp(const RedGeom& g, float* __restrict__ partial, L&& load, F&& use) {
  constexpr int N = V<T>::N;
  __shared__ float red[TY][K][TX * N];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const LaneMap lm = lane_map<N>(g);
  const int c = lm.c;
  const int seg = blockIdx.x / g.chunks, chunk = blockIdx.x % g.chunks;
  float acc[K][N];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < N; ++i) acc[k][i] = 0.f;
  const long long p0 = chunk * g.chunk_pix;
  const long long p1 = (p0 + g.chunk_pix < g.npix) ? p0 + g.chunk_pix : g.npix;
  if (c < g.C) {
    const long long step = TY * g.pl, base = (long long)seg * g.npix;
    long long p = p0 + ty * g.pl + lm.sub;
    for (; p + (PIX_FLY - 1) * step < p1; p += PIX_FLY * step) {
      uint4 raw[PIX_FLY][NTEN];
#pragma unroll
      for (int u = 0; u < PIX_FLY; ++u) load(base + p + u * step, c, raw[u]);
#pragma unroll
      for (int u = 0; u < PIX_FLY; ++u) use(base + p + u * step, c, raw[u], acc);
    }
    for (; p < p1; p += step) {
      uint4 raw[NTEN];
      load(base + p, c, raw);
      use(base + p, c, raw, acc);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < N; ++i) red[ty][k][tx * N + i] = acc[k][i];
  __syncthreads();
  if (ty == 0 && lm.sub == 0 && c < g.C) {
    float* dst = partial + ((size_t)blockIdx.x * K) * g.C + c;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int i = 0; i < N; ++i) {
        float s = 0.f;
        for (int j = 0; j < g.pl; ++j)
#pragma unroll
          for (int y = 0; y < TY; ++y) s += red[y][k][(tx + j * g.lpp) * N + i];
        dst[(size_t)k * g.C + i] = s;
      }
  }
}
template <typename T, int NTEN, typename L, typename F>
__device__ __forceinline__ void for_pixels_mlp(const RedGeom& g, L&& load, F&& use) {
  constexpr int N = V<T>::N;
  const LaneMap lm = lane_map<N>(g);
  const int c = lm.c;
  if (c >= g.C) return;
  const int seg = blockIdx.x / g.chunks, chunk = blockIdx.x % g.chunks;
  const long long p0 = chunk * g.chunk_pix;
  const long long p1 = (p0 + g.chunk_pix < g.npix) ? p0 + g.chunk_pix : g.npix;
  const long long step = TY * g.pl, base = (long long)seg * g.npix;
  long long p = p0 + threadIdx.y * g.pl + lm.sub;
  for (; p + (PIX_FLY - 1) * step < p1; p += PIX_FLY * step) {
    uint4 raw[PIX_FLY][NTEN];
#pragma unroll
    for (int u = 0; u < PIX_FLY; ++u) load(base + p + u * step, c, raw[u]);
#pragma unroll
    for (int u = 0; u < PIX_FLY; ++u) use(p + u * step, base + p + u * step, c, raw[u]);
  }
  for (; p < p1; p += step) {
    uint4 raw[NTEN];
    load(base + p, c, raw);
    use(p, base + p, c, raw);
  }
}

template <int N>
__device__ __forceinline__ void load_param(const float* __restrict__ p, float (&f)[N]) {
#pragma unroll
  for (int i = 0; i < N; i += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + i);
    f[i] = v.x; f[i + 1] = v.y; f[i + 2] = v.z; f[i + 3] = v.w;
  }
}

// One wave sums column `off` of the partial rows (row pitch `pitch` floats) in fp64, fixed order.
__device__ __forceinline__ double wave_sum_rows(const float* __restrict__ partial, int rows, size_t pitch, size_t off) {
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int r = lane; r < rows; r += 64) s += (double)partial[(size_t)r * pitch + off];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  return s;
}

// Two columns at once (same per-column order as wave_sum_rows, so the same bits): the finalize kernels are one
// dependent latency chain per column otherwise -- 13 us for 1024 rows, on the step's critical path 70 times.
__device__ __forceinline__ void wave_sum_rows2(const float* __restrict__ partial, int rows, size_t pitch, size_t off0, size_t off1,
                                               double& s0, double& s1) {
  const int lane = threadIdx.x & 63;
  double a = 0.0, b = 0.0;
  int r = lane;
  for (; r + 192 < rows; r += 256) {      // four rows of each column in flight
    const float x0 = partial[(size_t)r * pitch + off0], x1 = partial[(size_t)(r + 64) * pitch + off0];
    const float x2 = partial[(size_t)(r + 128) * pitch + off0], x3 = partial[(size_t)(r + 192) * pitch + off0];
    const float y0 = partial[(size_t)r * pitch + off1], y1 = partial[(size_t)(r + 64) * pitch + off1];
    const float y2 = partial[(size_t)(r + 128) * pitch + off1], y3 = partial[(size_t)(r + 192) * pitch + off1];
    a += (double)x0; a += (double)x1; a += (double)x2; a += (double)x3;
    b += (double)y0; b += (double)y1; b += (double)y2; b += (double)y3;
  }
  for (; r < rows; r += 64) {
    a += (double)partial[(size_t)r * pitch + off0];
    b += (double)partial[(size_t)r * pitch + off1];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    a += __shfl_xor(a, d, 64);
    b += __shfl_xor(b, d, 64);
  }
  s0 = a;
  s1 = b;
}

// ---------------------------------------------------------------- BatchNorm forward
template <typename T>
__global__ __launch_bounds__(TX * TY) void bn_stats_kernel(const T* __restrict__ x, int cs, int coff, RedGeom g,
                                                          float* __restrict__ partial) {
  constexpr int N = V<T>::N;
  reduce_pixels_mlp<T, 2, 1>(
      g, partial, [&](long long pix, int c, uint4 (&raw)[1]) { raw[0] = load_raw<T>(x + (size_t)pix * cs + coff + c); },
      [&](long long, int, const uint4 (&raw)[1], float (&acc)[2][N]) {
        float f[N];
        unpack<T>(raw[0], f);
#pragma unroll
        for (int i = 0; i < N; ++i) { acc[0][i] += f[i]; acc[1][i] += f[i] * f[i]; }
      });
}

// out[r][j] = sum of in[i][j] over rows i == r (mod nout): folds many partial rows into nout rows,
// fixed order, coalesced along j
__global__ void fold_rows_kernel(const float* __restrict__ in, int nin, int width, int nout, float* __restrict__ out) {
  const int r = blockIdx.x, j = blockIdx.y * blockDim.x + threadIdx.x;
  if (j >= width) return;
  float s = 0.f;
  int i = r;
  for (; i + 7 * nout < nin; i += 8 * nout) {      // eight rows in flight, added in row order
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[(size_t)(i + k * nout) * width + j];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; i < nin; i += nout) s += in[(size_t)i * width + j];
  out[(size_t)r * width + j] = s;
}

// mean / biased var from the partial rows; running stats (momentum, unbiased var); scale/shift.
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int rows, int C, long long count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, int training, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd, float* __restrict__ scale,
                                   float* __restrict__ shift) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per channel
  if (c >= C) return;
  float mean, var;
  double s = 0.0, ss = 0.0;
  if (training) {
    wave_sum_rows2(partial, rows, (size_t)2 * C, c, (size_t)C + c, s, ss);
  }
  if ((threadIdx.x & 63) != 0) return;
  if (training) {
    const double m = s / (double)count;
    double v = ss / (double)count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (running_mean) {
      const double unb = count > 1 ? v * (double)count / (double)(count - 1) : v;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  save_mean[c] = mean;
  save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
}

// inference: BatchNorm as a per-channel affine, y = x * scale + shift (optionally times res_scale), for folding into a
// convolution's epilogue
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps, float res_scale,
                               int C, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * (1.0f / sqrtf(var[c] + eps));
  scale[c] = sc * res_scale;
  shift[c] = (beta[c] - mean[c] * sc) * res_scale;
}

// y = [relu]( (x*scale + shift) * res_scale + res )
template <typename T, bool RES>
__global__ __launch_bounds__(TX * TY) void bn_apply_kernel(const T* __restrict__ x, int x_cs, int x_coff,
                                                          const T* __restrict__ res, int r_cs, int r_coff,
                                                          T* __restrict__ y, int y_cs, int y_coff,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ raff, float res_scale, int relu, RedGeom g) {
  constexpr int N = V<T>::N;
  float sc[N], sh[N], rsc[N], rsh[N];
  const int c0 = lane_map<N>(g).c;
  if (c0 < g.C) { load_param<N>(scale + c0, sc); load_param<N>(shift + c0, sh); }
  // raff: the residual operand is itself the pre-normalisation output of a BatchNorm (the 1x1 projection of a
  // BasicBlock's shortcut): its per-channel (scale | shift) is applied here, in fp32, instead of in a pass of its own
  if (raff && c0 < g.C) { load_param<N>(raff + c0, rsc); load_param<N>(raff + g.C + c0, rsh); }
  constexpr int NTEN = RES ? 2 : 1;
  for_pixels_mlp<T, NTEN>(
      g,
      [&](long long pix, int c, uint4 (&raw)[NTEN]) {
        raw[0] = load_raw<T>(x + (size_t)pix * x_cs + x_coff + c);
        if constexpr (RES) raw[1] = load_raw<T>(res + (size_t)pix * r_cs + r_coff + c);
      },
      [&](long long, long long pix, int c, const uint4 (&raw)[NTEN]) {
        float f[N], r[N];
        unpack<T>(raw[0], f);
        if constexpr (RES) unpack<T>(raw[1], r);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          float v = f[k] * sc[k] + sh[k];
          if (RES) v = v * res_scale + (raff ? r[k] * rsc[k] + rsh[k] : r[k]);
          if (relu) v = fmaxf(v, 0.f);
          f[k] = v;
        }
        store_vec<T>(y + (size_t)pix * y_cs + y_coff + c, f);
      });
}

// ---------------------------------------------------------------- BatchNorm backward
// dz = dy * (y > 0 if relu);  partial sums of dz and dz * xhat
// RELU (compile time, so that the pixel loop has no branch between its loads): 0 none; 1 mask from the saved output y;
// 2 (no residual) mask recomputed from x, [gamma*xhat + beta > 0], which saves reading y.
template <typename T, int RELU>
__global__ __launch_bounds__(TX * TY) void bn_bwd_reduce_kernel(const T* __restrict__ dy, int dy_cs, int dy_coff,
                                                               const T* __restrict__ y, int y_cs, int y_coff,
                                                               const T* __restrict__ x, int x_cs, int x_coff,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, RedGeom g,
                                                               float* __restrict__ partial) {
  constexpr int N = V<T>::N, NTEN = RELU == 1 ? 3 : 2;
  float mu[N], is[N], ga[N], be[N];       // per-channel parameters live in registers
  const int c0 = lane_map<N>(g).c;
  if (c0 < g.C) {
    load_param<N>(mean + c0, mu); load_param<N>(invstd + c0, is);
    if (RELU == 2) { load_param<N>(gamma + c0, ga); load_param<N>(beta + c0, be); }
  }
  reduce_pixels_mlp<T, 2, NTEN>(
      g, partial,
      [&](long long pix, int c, uint4 (&raw)[NTEN]) {
        raw[0] = load_raw<T>(dy + (size_t)pix * dy_cs + dy_coff + c);
        raw[1] = load_raw<T>(x + (size_t)pix * x_cs + x_coff + c);
        if constexpr (RELU == 1) raw[2] = load_raw<T>(y + (size_t)pix * y_cs + y_coff + c);
      },
      [&](long long, int, const uint4 (&raw)[NTEN], float (&acc)[2][N]) {
        float d[N], xv[N], yv[N];
        unpack<T>(raw[0], d);
        unpack<T>(raw[1], xv);
        if constexpr (RELU == 1) unpack<T>(raw[2], yv);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const float xhat = (xv[i] - mu[i]) * is[i];
          const bool off = RELU == 1 ? !(yv[i] > 0.f) : (RELU == 2 ? !(ga[i] * xhat + be[i] > 0.f) : false);
          const float dz = off ? 0.f : d[i];
          acc[0][i] += dz;
          acc[1][i] += dz * xhat;
        }
      });
}

// dbeta = sum dz ; dgamma = sum dz*xhat ; coefficients for the apply pass
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int rows, int C, long long count,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       int training, float res_scale, int accumulate, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ coef) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per channel
  if (c >= C) return;
  double s, sx;
  wave_sum_rows2(partial, rows, (size_t)2 * C, c, (size_t)C + c, s, sx);
  if ((threadIdx.x & 63) != 0) return;
  // the BN branch sees dz * res_scale
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)(s * res_scale);
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)(sx * res_scale);
  const float k = gamma[c] * invstd[c] * res_scale;
  coef[c] = k;                                                     // dx = k * (dz - a - xhat * b)
  coef[C + c] = training ? (float)(s / (double)count) : 0.f;        // a = mean(dz)
  coef[2 * C + c] = training ? (float)(sx / (double)count) : 0.f;   // b = mean(dz * xhat)
}

template <typename T, int RELU>
__global__ __launch_bounds__(TX * TY) void bn_bwd_apply_kernel(const T* __restrict__ dy, int dy_cs, int dy_coff,
                                                              const T* __restrict__ y, int y_cs, int y_coff,
                                                              const T* __restrict__ x, int x_cs, int x_coff,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ coef,
                                                              T* __restrict__ dx, T* __restrict__ dres, RedGeom g) {
  constexpr int N = V<T>::N, NTEN = RELU == 1 ? 3 : 2;
  const int C = g.C;
  float mu[N], is[N], k0[N], ka[N], kb[N], ga[N], be[N];
  const int c0 = lane_map<N>(g).c;
  if (c0 < C) {
    load_param<N>(mean + c0, mu); load_param<N>(invstd + c0, is);
    load_param<N>(coef + c0, k0); load_param<N>(coef + C + c0, ka); load_param<N>(coef + 2 * C + c0, kb);
    if (RELU == 2) { load_param<N>(gamma + c0, ga); load_param<N>(beta + c0, be); }
  }
  for_pixels_mlp<T, NTEN>(
      g,
      [&](long long pix, int c, uint4 (&raw)[NTEN]) {
        raw[0] = load_raw<T>(dy + (size_t)pix * dy_cs + dy_coff + c);
        raw[1] = load_raw<T>(x + (size_t)pix * x_cs + x_coff + c);
        if constexpr (RELU == 1) raw[2] = load_raw<T>(y + (size_t)pix * y_cs + y_coff + c);
      },
      [&](long long, long long pix, int c, const uint4 (&raw)[NTEN]) {
        float d[N], xv[N], yv[N], o[N];
        unpack<T>(raw[0], d);
        unpack<T>(raw[1], xv);
        if constexpr (RELU == 1) unpack<T>(raw[2], yv);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const float xhat = (xv[k] - mu[k]) * is[k];
          const bool off = RELU == 1 ? !(yv[k] > 0.f) : (RELU == 2 ? !(ga[k] * xhat + be[k] > 0.f) : false);
          const float dz = off ? 0.f : d[k];
          d[k] = dz;
          o[k] = k0[k] * (dz - ka[k] - xhat * kb[k]);
        }
        store_vec<T>(dx + (size_t)pix * C + c, o);
        if (dres) store_vec<T>(dres + (size_t)pix * C + c, d);
      });
}

// ---------------------------------------------------------------- ReLU backward + bias gradient
template <typename T, bool RELU>
__global__ __launch_bounds__(TX * TY) void act_bwd_kernel(const T* __restrict__ dy, int dy_cs, int dy_coff,
                                                         const T* __restrict__ y, int y_cs, int relu,
                                                         T* __restrict__ dz, int dz_cs, RedGeom g,
                                                         float* __restrict__ partial) {
  constexpr int N = V<T>::N;
  constexpr int NTEN = RELU ? 2 : 1;
  (void)relu;
  reduce_pixels_mlp<T, 1, NTEN>(
      g, partial,
      [&](long long pix, int c, uint4 (&raw)[NTEN]) {
        raw[0] = load_raw<T>(dy + (size_t)pix * dy_cs + dy_coff + c);
        if constexpr (RELU) raw[1] = load_raw<T>(y + (size_t)pix * y_cs + c);
      },
      [&](long long pix, int c, const uint4 (&raw)[NTEN], float (&acc)[1][N]) {
        float d[N], yv[N];
        unpack<T>(raw[0], d);
        if constexpr (RELU) unpack<T>(raw[1], yv);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          if (RELU && !(yv[i] > 0.f)) d[i] = 0.f;
          acc[0][i] += d[i];
        }
        if (dz) store_vec<T>(dz + (size_t)pix * dz_cs + c, d);
      });
}

__global__ void sum_rows_kernel(const float* __restrict__ partial, int rows, int K, int k, int C, float* __restrict__ out) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per channel
  if (c >= C) return;
  const double s = wave_sum_rows(partial, rows, (size_t)K * C, (size_t)k * C + c);
  if ((threadIdx.x & 63) == 0) out[c] = (float)s;
}

// ---------------------------------------------------------------- ChannelAttention pieces
// per image and channel: sum and max (+ smallest pixel index of the max)
template <typename T>
__global__ __launch_bounds__(TX * TY) void gate_pool_kernel(const T* __restrict__ x, RedGeom g,
                                                           float* __restrict__ psum, float* __restrict__ pmax,
                                                           int* __restrict__ pidx) {
  constexpr int N = V<T>::N;
  __shared__ float rs[TY][TX * N], rm[TY][TX * N];
  __shared__ int ri[TY][TX * N];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const LaneMap lm = lane_map<N>(g);
  const int c = lm.c;
  const int seg = blockIdx.x / g.chunks, chunk = blockIdx.x % g.chunks;
  float s[N], m[N];
  int idx[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { s[i] = 0.f; m[i] = -INFINITY; idx[i] = 0x7fffffff; }
  const long long p0 = chunk * g.chunk_pix;
  const long long p1 = (p0 + g.chunk_pix < g.npix) ? p0 + g.chunk_pix : g.npix;
  if (c < g.C) {
    const long long step = TY * g.pl;
    long long p = p0 + ty * g.pl + lm.sub;
    auto take = [&](const uint4& raw, long long pp) {
      float f[N];
      unpack<T>(raw, f);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        s[i] += f[i];
        if (f[i] > m[i]) { m[i] = f[i]; idx[i] = (int)pp; }
      }
    };
    for (; p + (PIX_FLY - 1) * step < p1; p += PIX_FLY * step) {      // four pixels in flight (see reduce_pixels_mlp)
      uint4 raw[PIX_FLY];
#pragma unroll
      for (int u = 0; u < PIX_FLY; ++u) raw[u] = load_raw<T>(x + ((size_t)seg * g.npix + p + u * step) * g.C + c);
#pragma unroll
      for (int u = 0; u < PIX_FLY; ++u) take(raw[u], p + u * step);
    }
    for (; p < p1; p += step) take(load_raw<T>(x + ((size_t)seg * g.npix + p) * g.C + c), p);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { rs[ty][tx * N + i] = s[i]; rm[ty][tx * N + i] = m[i]; ri[ty][tx * N + i] = idx[i]; }
  __syncthreads();
  if (ty == 0 && lm.sub == 0 && c < g.C) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float ss = 0.f, mm = -INFINITY;
      int ii = 0x7fffffff;
      for (int q = 0; q < g.pl; ++q)
#pragma unroll
      for (int y = 0; y < TY; ++y) {
        const int col = (tx + q * g.lpp) * N + i;
        ss += rs[y][col];
        const float v = rm[y][col];
        const int j = ri[y][col];
        if (v > mm || (v == mm && j < ii)) { mm = v; ii = j; }
      }
      const size_t o = (size_t)blockIdx.x * g.C + c + i;
      psum[o] = ss; pmax[o] = mm; pidx[o] = ii;
    }
  }
}

__global__ void gate_pool_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ pmax,
                                          const int* __restrict__ pidx, int chunks, int C, long long npix,
                                          float* __restrict__ avg, float* __restrict__ mx, int* __restrict__ amax) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  double s = 0.0;
  float m = -INFINITY;
  int ii = 0x7fffffff;
  int r = 0;
  for (; r + 8 <= chunks; r += 8) {      // eight chunks' loads in flight; combined in chunk order
    float ps[8], pm[8];
    int pi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t o = ((size_t)b * chunks + r + k) * C + c;
      ps[k] = psum[o]; pm[k] = pmax[o]; pi[k] = pidx[o];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s += (double)ps[k];
      if (pm[k] > m || (pm[k] == m && pi[k] < ii)) { m = pm[k]; ii = pi[k]; }
    }
  }
  for (; r < chunks; ++r) {
    const size_t o = ((size_t)b * chunks + r) * C + c;
    s += (double)psum[o];
    if (pmax[o] > m || (pmax[o] == m && pidx[o] < ii)) { m = pmax[o]; ii = pidx[o]; }
  }
  avg[(size_t)b * C + c] = (float)(s / (double)npix);
  mx[(size_t)b * C + c] = m;
  amax[(size_t)b * C + c] = ii;
}

// y[b,p,c] = x[b,p,c] * s[b,c]
template <typename T>
__global__ __launch_bounds__(TX * TY) void gate_scale_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                            T* __restrict__ y, RedGeom g) {
  constexpr int N = V<T>::N;
  float sv[N];
  const int c0 = lane_map<N>(g).c;
  if (c0 < g.C) load_param<N>(s + (size_t)(blockIdx.x / g.chunks) * g.C + c0, sv);
  for_pixels_mlp<T, 1>(
      g, [&](long long pix, int c, uint4 (&raw)[1]) { raw[0] = load_raw<T>(x + (size_t)pix * g.C + c); },
      [&](long long, long long pix, int c, const uint4 (&raw)[1]) {
        float f[N];
        unpack<T>(raw[0], f);
#pragma unroll
        for (int k = 0; k < N; ++k) f[k] *= sv[k];
        store_vec<T>(y + (size_t)pix * g.C + c, f);
      });
}

// ds[b,c] = sum_p dy*x
template <typename T>
__global__ __launch_bounds__(TX * TY) void gate_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                 RedGeom g, float* __restrict__ partial) {
  constexpr int N = V<T>::N;
  reduce_pixels_mlp<T, 1, 2>(
      g, partial,
      [&](long long pix, int c, uint4 (&raw)[2]) {
        raw[0] = load_raw<T>(dy + (size_t)pix * g.C + c);
        raw[1] = load_raw<T>(x + (size_t)pix * g.C + c);
      },
      [&](long long, int, const uint4 (&raw)[2], float (&acc)[1][N]) {
        float d[N], xv[N];
        unpack<T>(raw[0], d);
        unpack<T>(raw[1], xv);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[0][i] += d[i] * xv[i];
      });
}

__global__ void gate_bwd_finalize_kernel(const float* __restrict__ partial, int chunks, int C, float* __restrict__ ds) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  double s = 0.0;
  int r = 0;
  for (; r + 8 <= chunks; r += 8) {      // eight loads in flight, added in chunk order
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = partial[((size_t)b * chunks + r + k) * C + c];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (double)v[k];
  }
  for (; r < chunks; ++r) s += (double)partial[((size_t)b * chunks + r) * C + c];
  ds[(size_t)b * C + c] = (float)s;
}

// dx = dy*s + davg/npix + [p == argmax] dmax
template <typename T>
__global__ __launch_bounds__(TX * TY) void gate_bwd_apply_kernel(const T* __restrict__ dy, const float* __restrict__ s,
                                                                const float* __restrict__ davg, const float* __restrict__ dmax,
                                                                const int* __restrict__ amax, T* __restrict__ dx, RedGeom g) {
  constexpr int N = V<T>::N;
  float sv[N], da[N], dm[N];
  int am[N];
  const int c0 = lane_map<N>(g).c;
  const float inv = 1.f / (float)g.npix;
  if (c0 < g.C) {
    const size_t o = (size_t)(blockIdx.x / g.chunks) * g.C + c0;
    load_param<N>(s + o, sv); load_param<N>(davg + o, da); load_param<N>(dmax + o, dm);
#pragma unroll
    for (int k = 0; k < N; ++k) { am[k] = amax[o + k]; da[k] *= inv; }
  }
  for_pixels_mlp<T, 1>(
      g, [&](long long pix, int c, uint4 (&raw)[1]) { raw[0] = load_raw<T>(dy + (size_t)pix * g.C + c); },
      [&](long long p, long long pix, int c, const uint4 (&raw)[1]) {
        float f[N];
        unpack<T>(raw[0], f);
#pragma unroll
        for (int k = 0; k < N; ++k) f[k] = f[k] * sv[k] + da[k] + (am[k] == (int)p ? dm[k] : 0.f);
        store_vec<T>(dx + (size_t)pix * g.C + c, f);
      });
}

int ew_blocks(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int check_c(int dtype, int C, const char* what) {
  if (dtype != JSPSR_F32 && dtype != JSPSR_BF16) return fail(JSPSR_EINVAL, "%s: bad dtype", what);
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  if (C <= 0 || C % vec) return fail(JSPSR_EINVAL, "%s: C=%d must be a multiple of %d", what, C, vec);
  return JSPSR_OK;
}

}  // namespace

#define DISPATCH(dtype, CALL)                    \
  do {                                           \
    if ((dtype) == JSPSR_F32) { using T = float; CALL; } else { using T = __bf16; CALL; } \
  } while (0)

extern "C" size_t jspsr_reduce_workspace_bytes(int dtype, int C, int nseg) {
  if (C <= 0 || nseg <= 0) return 0;
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  const RedGeom g = make_red(1LL << 40, nseg, C, vec);  // the most chunks any pixel count can get
  const size_t rows = (size_t)nseg * g.chunks;
  return (rows * 3 + 4) * (size_t)C * sizeof(float) + 64;
}

extern "C" int jspsr_bn_forward(int dtype, const void* x, int x_cs, int x_coff, const void* res, int r_cs, int r_coff,
                                void* y, int y_cs, int y_coff, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, int training,
                                int relu, float res_scale, float* save_mean, float* save_invstd, long long npix, int C,
                                const float* ext_partial, int ext_rows, const float* res_affine, float* affine_out,
                                void* workspace, jspsr_stream_t stream) {
  if (int e = check_c(dtype, C, "bn_forward")) return e;
  if (!x || (!y && !affine_out) || !gamma || !beta || !save_mean || !save_invstd || !workspace || npix <= 0)
    return fail(JSPSR_EINVAL, "bn_forward: null pointer or empty tensor");
  if (res_affine && !res) return fail(JSPSR_EINVAL, "bn_forward: res_affine without a residual operand");
  if (!training && (!running_mean || !running_var)) return fail(JSPSR_EINVAL, "bn_forward: eval mode needs running stats");
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  if (x_cs % vec || x_coff % vec || y_cs % vec || y_coff % vec || (res && (r_cs % vec || r_coff % vec)))
    return fail(JSPSR_EINVAL, "bn_forward: channel pitches/offsets must be multiples of %d", vec);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  float* scale = affine_out ? affine_out : ws;
  float* shift = affine_out ? affine_out + C : ws + C;
  float* partial = ws + 2 * C;
  int rows = 0;
  if (training && ext_partial) {
    // statistics were accumulated by the producing convolution's epilogue (one row per M-tile):
    // fold them to <= 256 rows first so the per-channel finalize stays short
    if (ext_rows <= 0) return fail(JSPSR_EINVAL, "bn_forward: ext_rows must be positive");
    rows = ext_rows < 256 ? ext_rows : 256;
    if (ext_rows > 256) {
      hipLaunchKernelGGL(fold_rows_kernel, dim3(rows, (2 * C + 255) / 256), dim3(256), 0, s, ext_partial, ext_rows, 2 * C, rows, partial);
      if (int e = check_launch("bn_fold_rows")) return e;
    } else {
      (void)hipMemcpyAsync(partial, ext_partial, (size_t)ext_rows * 2 * C * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
  } else if (training) {
    const RedGeom g = make_red(npix, 1, C, vec);
    rows = g.chunks;
    DISPATCH(dtype, hipLaunchKernelGGL(bn_stats_kernel<T>, red_grid(g, vec), dim3(TX, TY), 0, s,
                                       static_cast<const T*>(x), x_cs, x_coff, g, partial));
    if (int e = check_launch("bn_stats")) return e;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, rows, C, npix, gamma, beta,
                     running_mean, running_var, momentum, eps, training, save_mean, save_invstd, scale, shift);
  if (int e = check_launch("bn_finalize")) return e;
  if (!y) return JSPSR_OK;      // statistics + (scale | shift) only: the consumer applies them (res_affine)
  const RedGeom ga = make_red(npix, 1, C, vec);
#define BN_APPLY(R) DISPATCH(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, R>), red_grid(ga, vec), dim3(TX, TY), 0, s,             \
                                     static_cast<const T*>(x), x_cs, x_coff, static_cast<const T*>(res), r_cs, r_coff,        \
                                     static_cast<T*>(y), y_cs, y_coff, scale, shift, res_affine, res_scale, relu, ga))
  if (res) { BN_APPLY(true); } else { BN_APPLY(false); }
#undef BN_APPLY
  return check_launch("bn_apply");
}

// par [4][C] = a | b | is | mis for the reduce fused into a data-gradient epilogue (ConvGeom::red_par): the ReLU behind the
// BatchNorm was open where a z + b > 0 (a = gamma invstd, b = beta - gamma invstd mean), xhat = z is + mis
__global__ void bn_reduce_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                        const float* __restrict__ invstd, int C, float* __restrict__ par) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float a = gamma[c] * invstd[c];
  par[c] = a;
  par[C + c] = beta[c] - a * mean[c];
  par[2 * C + c] = invstd[c];
  par[3 * C + c] = -mean[c] * invstd[c];
}

extern "C" int jspsr_bn_reduce_params(const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int C,
                                      float* par, jspsr_stream_t stream) {
  if (!gamma || !beta || !save_mean || !save_invstd || !par || C <= 0) return fail(JSPSR_EINVAL, "bn_reduce_params: bad arguments");
  hipLaunchKernelGGL(bn_reduce_params_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), gamma, beta,
                     save_mean, save_invstd, C, par);
  return check_launch("bn_reduce_params");
}

extern "C" int jspsr_bn_backward(int dtype, const void* dy, int dy_cs, int dy_coff, const void* y, int y_cs, int y_coff,
                                 const void* x, int x_cs, int x_coff, const float* gamma, const float* beta, const float* save_mean,
                                 const float* save_invstd, int training, int relu, float res_scale, void* dx, void* dres,
                                 float* dgamma, float* dbeta, int accumulate, long long npix, int C, void* workspace,
                                 const float* ext_partial, int ext_rows, jspsr_stream_t stream) {
  if (int e = check_c(dtype, C, "bn_backward")) return e;
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace || npix <= 0 ||
      (relu == 1 && !y) || (relu == 2 && !beta) || relu < 0 || relu > 2)
    return fail(JSPSR_EINVAL, "bn_backward: null pointer, empty tensor or bad relu mode");
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  if (dy_cs % vec || dy_coff % vec || x_cs % vec || x_coff % vec || (relu == 1 && (y_cs % vec || y_coff % vec)))
    return fail(JSPSR_EINVAL, "bn_backward: channel pitches/offsets must be multiples of %d", vec);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  float* coef = ws;
  float* partial = ws + 3 * C;
  const RedGeom g = make_red(npix, 1, C, vec);
#define BN_BWD_REDUCE(R) DISPATCH(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, R>), red_grid(g, vec), dim3(TX, TY), 0, s, \
                                     static_cast<const T*>(dy), dy_cs, dy_coff, static_cast<const T*>(y), y_cs, y_coff,     \
                                     static_cast<const T*>(x), x_cs, x_coff, save_mean, save_invstd, gamma, beta, g, partial))
  int rows = g.chunks;
  if (ext_partial) {
    // the two sums per channel came out of the producing data gradient's epilogue (jspsr_conv2d_dgrad: red_out), one row
    // per 8x16-pixel tile: no reduce pass here; many rows are folded to <= 1024 first (fixed order), as the forward does
    if (ext_rows <= 0) return fail(JSPSR_EINVAL, "bn_backward: ext_rows must be positive");
    if (ext_rows > 1024) {
      rows = g.chunks < 1024 ? (g.chunks > 0 ? g.chunks : 1) : 1024;
      hipLaunchKernelGGL(fold_rows_kernel, dim3(rows, (2 * C + 255) / 256), dim3(256), 0, s, ext_partial, ext_rows, 2 * C, rows, partial);
      if (int e = check_launch("bn_fold_rows")) return e;
    } else {
      partial = const_cast<float*>(ext_partial);
      rows = ext_rows;
    }
  } else {
    if (relu == 0) { BN_BWD_REDUCE(0); } else if (relu == 1) { BN_BWD_REDUCE(1); } else { BN_BWD_REDUCE(2); }
    if (int e = check_launch("bn_bwd_reduce")) return e;
  }
#undef BN_BWD_REDUCE
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, rows, C, npix, gamma,
                     save_invstd, training, res_scale, accumulate, dgamma, dbeta, coef);
  if (int e = check_launch("bn_bwd_finalize")) return e;
#define BN_BWD_APPLY(R) DISPATCH(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R>), red_grid(g, vec), dim3(TX, TY), 0, s,  \
                                     static_cast<const T*>(dy), dy_cs, dy_coff, static_cast<const T*>(y), y_cs, y_coff,     \
                                     static_cast<const T*>(x), x_cs, x_coff, save_mean, save_invstd, gamma, beta, coef,     \
                                     static_cast<T*>(dx), static_cast<T*>(dres), g))
  if (relu == 0) { BN_BWD_APPLY(0); } else if (relu == 1) { BN_BWD_APPLY(1); } else { BN_BWD_APPLY(2); }
#undef BN_BWD_APPLY
  return check_launch("bn_bwd_apply");
}

extern "C" int jspsr_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                             float eps, float res_scale, int C, float* scale, float* shift, jspsr_stream_t stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0)
    return fail(JSPSR_EINVAL, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), gamma, beta,
                     running_mean, running_var, eps, res_scale, C, scale, shift);
  return check_launch("bn_fold");
}

extern "C" int jspsr_act_backward(int dtype, const void* dy, int dy_cs, int dy_coff, const void* y, int y_cs, int relu,
                                  void* dz, int dz_cs, float* dbias, long long npix, int C, void* workspace,
                                  jspsr_stream_t stream) {
  if (int e = check_c(dtype, C, "act_backward")) return e;
  if (!dy || (relu && !y) || !workspace || npix <= 0) return fail(JSPSR_EINVAL, "act_backward: null pointer or empty tensor");
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  if (dy_cs % vec || dy_coff % vec || (dz && dz_cs % vec) || (relu && (y_cs % vec || y_cs < C)))
    return fail(JSPSR_EINVAL, "act_backward: bad pitch");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  const RedGeom g = make_red(npix, 1, C, vec);
#define ACT_BWD(R) DISPATCH(dtype, hipLaunchKernelGGL((act_bwd_kernel<T, R>), red_grid(g, vec), dim3(TX, TY), 0, s, static_cast<const T*>(dy), \
                                     dy_cs, dy_coff, static_cast<const T*>(y), y_cs, relu, static_cast<T*>(dz), dz_cs, g, partial))
  if (relu) { ACT_BWD(true); } else { ACT_BWD(false); }
#undef ACT_BWD
  if (int e = check_launch("act_backward")) return e;
  if (dbias) {
    hipLaunchKernelGGL(sum_rows_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, g.chunks, 1, 0, C, dbias);
    return check_launch("act_backward_bias");
  }
  return JSPSR_OK;
}

extern "C" int jspsr_gate_pool(int dtype, const void* x, int B, long long npix, int C, float* avg, float* mx, int* amax,
                               void* workspace, jspsr_stream_t stream) {
  if (int e = check_c(dtype, C, "gate_pool")) return e;
  if (!x || !avg || !mx || !amax || !workspace || B <= 0 || npix <= 0 || npix > 0x7fffffffLL)
    return fail(JSPSR_EINVAL, "gate_pool: bad arguments");
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RedGeom g = make_red(npix, B, C, vec);
  const size_t rows = (size_t)B * g.chunks;
  float* psum = static_cast<float*>(workspace);
  float* pmax = psum + rows * C;
  int* pidx = reinterpret_cast<int*>(pmax + rows * C);
  DISPATCH(dtype, hipLaunchKernelGGL(gate_pool_kernel<T>, red_grid(g, vec), dim3(TX, TY), 0, s, static_cast<const T*>(x), g,
                                     psum, pmax, pidx));
  if (int e = check_launch("gate_pool")) return e;
  hipLaunchKernelGGL(gate_pool_finalize_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, psum, pmax, pidx, g.chunks, C,
                     npix, avg, mx, amax);
  return check_launch("gate_pool_finalize");
}

extern "C" int jspsr_gate_scale(int dtype, const void* x, const float* s_, void* y, int B, long long npix, int C,
                                jspsr_stream_t stream) {
  if (int e = check_c(dtype, C, "gate_scale")) return e;
  if (!x || !s_ || !y || B <= 0 || npix <= 0) return fail(JSPSR_EINVAL, "gate_scale: bad arguments");
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RedGeom g = make_red(npix, B, C, vec);
  DISPATCH(dtype, hipLaunchKernelGGL(gate_scale_kernel<T>, red_grid(g, vec), dim3(TX, TY), 0, s,
                                     static_cast<const T*>(x), s_, static_cast<T*>(y), g));
  return check_launch("gate_scale");
}

extern "C" int jspsr_gate_backward_reduce(int dtype, const void* dy, const void* x, float* ds, int B, long long npix, int C,
                                          void* workspace, jspsr_stream_t stream) {
  if (int e = check_c(dtype, C, "gate_backward_reduce")) return e;
  if (!dy || !x || !ds || !workspace || B <= 0 || npix <= 0) return fail(JSPSR_EINVAL, "gate_backward_reduce: bad arguments");
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RedGeom g = make_red(npix, B, C, vec);
  float* partial = static_cast<float*>(workspace);
  DISPATCH(dtype, hipLaunchKernelGGL(gate_bwd_reduce_kernel<T>, red_grid(g, vec), dim3(TX, TY), 0, s,
                                     static_cast<const T*>(dy), static_cast<const T*>(x), g, partial));
  if (int e = check_launch("gate_backward_reduce")) return e;
  hipLaunchKernelGGL(gate_bwd_finalize_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, partial, g.chunks, C, ds);
  return check_launch("gate_backward_finalize");
}

extern "C" int jspsr_gate_backward_apply(int dtype, const void* dy, const float* s_, const float* davg, const float* dmax,
                                         const int* amax, void* dx, int B, long long npix, int C, jspsr_stream_t stream) {
  if (int e = check_c(dtype, C, "gate_backward_apply")) return e;
  if (!dy || !s_ || !davg || !dmax || !amax || !dx || B <= 0 || npix <= 0)
    return fail(JSPSR_EINVAL, "gate_backward_apply: bad arguments");
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RedGeom g = make_red(npix, B, C, vec);
  DISPATCH(dtype, hipLaunchKernelGGL(gate_bwd_apply_kernel<T>, red_grid(g, vec), dim3(TX, TY), 0, s,
                                     static_cast<const T*>(dy), s_, davg, dmax, amax, static_cast<T*>(dx), g));
  return check_launch("gate_backward_apply");
}

// ---------------------------------------------------------------- boundary: planar fp32 -> NHWC activations
// What the models do first with every input (utils/utils.py:156-179 hands them planar fp32 tensors): cast to the compute
// dtype, put the channels last, zero-pad them to whole 16-byte chunks -- one pass instead of torch's cast + pad + copy.
// A thread owns a pixel: its C plane reads are coalesced across the wave, its Cpad channels leave as 16-byte stores.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, long long HW,
                                                          long long npix, int Cpad) {
  constexpr int N = V<T>::N;
  for (long long p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    const long long b = p / HW, q = p - b * HW;
    const float* s = src + (size_t)b * C * HW + q;
    T* d = dst + (size_t)p * Cpad;
    for (int c0 = 0; c0 < Cpad; c0 += N) {
      float f[N];
#pragma unroll
      for (int i = 0; i < N; ++i) f[i] = c0 + i < C ? s[(size_t)(c0 + i) * HW] : 0.f;
      store_vec<T>(d + c0, f);
    }
  }
}

extern "C" int jspsr_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int c_pad,
                                  jspsr_stream_t stream) {
  if (dtype != JSPSR_F32 && dtype != JSPSR_BF16) return fail(JSPSR_EINVAL, "nchw_to_nhwc: dtype must be JSPSR_F32 or JSPSR_BF16");
  const int vec = dtype == JSPSR_F32 ? 4 : 8;
  if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || c_pad < C || c_pad % vec)
    return fail(JSPSR_EINVAL, "nchw_to_nhwc: bad arguments (c_pad must be a multiple of %d and >= C)", vec);
  if (!aligned4(src) || !aligned16(dst)) return fail(JSPSR_EALIGN, "nchw_to_nhwc: src must be 4-byte, dst 16-byte aligned");
  const long long HW = (long long)H * W, npix = HW * B;
  const long long want = (npix + 255) / 256;
  const int blocks = (int)(want > 8192 ? 8192 : want);
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(blocks), dim3(256), 0, s, src, static_cast<T*>(dst), C, HW, npix, c_pad));
  return check_launch("nchw_to_nhwc");
}

// ---------------------------------------------------------------- ChannelAttention: the C -> C/16 -> C MLP on (B, C) vectors
// s[b] = sigmoid(W2 relu(W1 avg[b]) + W2 relu(W1 mx[b])) (resnet_cbam.py:41-53; both 1x1 convs without bias), forward
// and backward as three launches instead of the ~45 tiny library kernels torch needs for it (matrix products of 8 x C
// vectors, ReLU, Sigmoid and their autograd) -- on the step's critical path in front of every decoder conv.
// hid[b][0|1][j] keeps relu(W1 avg), relu(W1 mx) for the backward pass.  Sums run in a fixed order: reproducible.
constexpr int GM_T = 256;

__global__ __launch_bounds__(GM_T) void gate_mlp_fwd_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                                           const float* __restrict__ w1, const float* __restrict__ w2, int C, int Ch,
                                                           float* __restrict__ s, float* __restrict__ hid) {
  extern __shared__ float gm_lds[];       // [Ch] h_avg + h_max
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* a = avg + (size_t)b * C;
  const float* m = mx + (size_t)b * C;
  for (int j = wave; j < Ch; j += GM_T / 64) {          // one wave per hidden unit: two C-long dot products
    const float* w = w1 + (size_t)j * C;
    float da = 0.f, dm = 0.f;
    for (int c = lane; c < C; c += 64) { const float wv = w[c]; da += wv * a[c]; dm += wv * m[c]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { da += __shfl_xor(da, d, 64); dm += __shfl_xor(dm, d, 64); }
    if (lane == 0) {
      const float ha = fmaxf(da, 0.f), hm = fmaxf(dm, 0.f);
      hid[((size_t)b * 2 + 0) * Ch + j] = ha;
      hid[((size_t)b * 2 + 1) * Ch + j] = hm;
      gm_lds[j] = ha + hm;                              // W2 is linear and has no bias: W2 ha + W2 hm = W2 (ha + hm)
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += GM_T) {
    const float* w = w2 + (size_t)c * Ch;
    float z = 0.f;
    for (int j = 0; j < Ch; ++j) z += w[j] * gm_lds[j];
    s[(size_t)b * C + c] = 1.f / (1.f + expf(-z));
  }
}

// per image: dz = ds s (1 - s); dh = W2^T dz, masked by the two ReLUs; davg = W1^T dh_a, dmax = W1^T dh_m.
// dzs[b][c] and dhs[b][0|1][j] are kept for the weight-gradient launch.
__global__ __launch_bounds__(GM_T) void gate_mlp_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                           const float* __restrict__ hid, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, int C, int Ch, float* __restrict__ davg,
                                                           float* __restrict__ dmax, float* __restrict__ dzs, float* __restrict__ dhs) {
  extern __shared__ float gm_lds[];       // [C] dz, then [2][Ch] dh
  float* dz = gm_lds;
  float* dh = gm_lds + C;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int c = tid; c < C; c += GM_T) {
    const float sv = s[(size_t)b * C + c], v = ds[(size_t)b * C + c] * sv * (1.f - sv);
    dz[c] = v;
    dzs[(size_t)b * C + c] = v;
  }
  __syncthreads();
  for (int j = wave; j < Ch; j += GM_T / 64) {          // dh[j] = sum_c w2[c][j] dz[c]
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc += w2[(size_t)c * Ch + j] * dz[c];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) {
      const float da = hid[((size_t)b * 2 + 0) * Ch + j] > 0.f ? acc : 0.f, dm = hid[((size_t)b * 2 + 1) * Ch + j] > 0.f ? acc : 0.f;
      dh[j] = da; dh[Ch + j] = dm;
      dhs[((size_t)b * 2 + 0) * Ch + j] = da;
      dhs[((size_t)b * 2 + 1) * Ch + j] = dm;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += GM_T) {
    float ga = 0.f, gm = 0.f;
    for (int j = 0; j < Ch; ++j) { const float wv = w1[(size_t)j * C + c]; ga += wv * dh[j]; gm += wv * dh[Ch + j]; }
    davg[(size_t)b * C + c] = ga;
    dmax[(size_t)b * C + c] = gm;
  }
}

// dw2[c][j] = sum_b dz[b][c] (ha + hm)[b][j];  dw1[j][c] = sum_b dh_a[b][j] avg[b][c] + dh_m[b][j] mx[b][c]  (b in order)
__global__ __launch_bounds__(GM_T) void gate_mlp_wgrad_kernel(const float* __restrict__ dzs, const float* __restrict__ dhs,
                                                             const float* __restrict__ hid, const float* __restrict__ avg,
                                                             const float* __restrict__ mx, int B, int C, int Ch,
                                                             float* __restrict__ dw1, float* __restrict__ dw2) {
  const long long n = (long long)C * Ch;
  for (long long i = blockIdx.x * (long long)GM_T + threadIdx.x; i < 2 * n; i += (long long)gridDim.x * GM_T) {
    if (i < n) {                                        // dw1, [Ch][C]: consecutive threads -> consecutive c
      const int j = (int)(i / C), c = (int)(i - (long long)j * C);
      float g = 0.f;
      for (int b = 0; b < B; ++b)
        g += dhs[((size_t)b * 2 + 0) * Ch + j] * avg[(size_t)b * C + c] + dhs[((size_t)b * 2 + 1) * Ch + j] * mx[(size_t)b * C + c];
      dw1[i] = g;
    } else {                                            // dw2, [C][Ch]
      const long long k = i - n;
      const int c = (int)(k / Ch), j = (int)(k - (long long)c * Ch);
      float g = 0.f;
      for (int b = 0; b < B; ++b) g += dzs[(size_t)b * C + c] * (hid[((size_t)b * 2 + 0) * Ch + j] + hid[((size_t)b * 2 + 1) * Ch + j]);
      dw2[k] = g;
    }
  }
}

extern "C" int jspsr_gate_mlp_forward(const float* avg, const float* mx, const float* w1, const float* w2, int B, int C, int Ch,
                                      float* s, float* hid, jspsr_stream_t stream) {
  if (!avg || !mx || !w1 || !w2 || !s || !hid || B <= 0 || C <= 0 || Ch <= 0 || Ch > 4096)
    return fail(JSPSR_EINVAL, "gate_mlp_forward: bad arguments");
  hipLaunchKernelGGL(gate_mlp_fwd_kernel, dim3(B), dim3(GM_T), Ch * sizeof(float), static_cast<hipStream_t>(stream), avg, mx, w1, w2,
                     C, Ch, s, hid);
  return check_launch("gate_mlp_forward");
}

extern "C" size_t jspsr_gate_mlp_backward_workspace_bytes(int B, int C, int Ch) {
  if (B <= 0 || C <= 0 || Ch <= 0) return 0;
  return ((size_t)B * C + (size_t)B * 2 * Ch) * sizeof(float);
}

extern "C" int jspsr_gate_mlp_backward(const float* ds, const float* s, const float* hid, const float* avg, const float* mx,
                                       const float* w1, const float* w2, int B, int C, int Ch, float* davg, float* dmax,
                                       float* dw1, float* dw2, void* workspace, jspsr_stream_t stream) {
  if (!ds || !s || !hid || !avg || !mx || !w1 || !w2 || !davg || !dmax || !dw1 || !dw2 || !workspace || B <= 0 || C <= 0 ||
      Ch <= 0 || (size_t)(C + 2 * Ch) * sizeof(float) > 64 * 1024)
    return fail(JSPSR_EINVAL, "gate_mlp_backward: bad arguments");
  float* dzs = static_cast<float*>(workspace);
  float* dhs = dzs + (size_t)B * C;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gate_mlp_bwd_kernel, dim3(B), dim3(GM_T), (C + 2 * Ch) * sizeof(float), st, ds, s, hid, w1, w2, C, Ch, davg, dmax,
                     dzs, dhs);
  if (int e = check_launch("gate_mlp_backward")) return e;
  const long long n2 = 2LL * C * Ch;
  const int blocks = (int)((n2 + GM_T - 1) / GM_T > 2048 ? 2048 : (n2 + GM_T - 1) / GM_T);
  hipLaunchKernelGGL(gate_mlp_wgrad_kernel, dim3(blocks), dim3(GM_T), 0, st, dzs, dhs, hid, avg, mx, B, C, Ch, dw1, dw2);
  return check_launch("gate_mlp_wgrad");
}
