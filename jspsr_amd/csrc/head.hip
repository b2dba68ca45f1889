// K1c -- the generator's two 1x1 heads (reference models/components/spn.py:41-52,66-68: conv_weight 9 rows, conv_offset
// 16 rows; LRRU.py:238-247) as ONE convolution that WRITES PLANES: (B,H,W,Cin) NHWC feature -> (B,25,H,W) fp32, planes
// 0..8 affinity logits, 9..24 the sixteen learned offsets -- exactly the operand layout of the propagation kernel at the
// PostProcessor.forward boundary (prop_dma.hip), so the in-model propagation step IS the roofline kernel: no padding
// channels, no transposes (rounds 2-3 fed the step from a 32-channel NHWC head: 22 % of its bytes were padding).
//
// Operands are swapped against the usual implicit GEMM so that a lane holds ONE PIXEL COLUMN of the product:
//   forward   D[channel n][pixel p] = sum_c W[n][c] x[p][c]      A = W (rows n, zero beyond 25), B = x straight from global
//             memory: lane (p = l & 31, h = l >> 5) reads the 16-byte chunks of ITS half of the pixel's channels (the k order
//             inside an MFMA is permuted identically for A and B, which is all a dot product needs); an accumulator
//             register then holds one channel of 32 consecutive pixels = a 128-byte plane segment per half-wave store;
//   backward  D[channel c][pixel p] = sum_n W[n][c] g[n][p]      A = W^T with the row -> channel map chosen so that lane
//             (p, h) ends up with the CONTIGUOUS channels h Cin/2 .. of its pixel (16-byte NHWC stores), B = the gradient
//             planes (coalesced 128-byte plane segments).  The same pass writes the gradient once more as a 32-channel
//             NHWC tensor in the compute dtype -- the G operand of the weight-gradient kernel (jspsr_conv2d_wgrad) -- and
//             the per-plane sums (the bias gradient) as one partial row per workgroup, folded in fixed order.
// bf16: v_mfma_f32_32x32x16_bf16 (inputs rounded to bf16, fp32 accumulate -- as every conv of the bf16 path);
// fp32: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain -- the 1e-4 parity path).
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int NP = 25;          // planes: 9 affinity logits + 16 offsets
constexpr int WPB = 4;          // waves per workgroup

struct HeadArgs {
  const void* x;        // (B*HW, x_cs) NHWC feature, first channel x_coff
  const float* w;       // [25][Cin] fp32 masters (rows 0..8 conv_weight, 9..24 conv_offset)
  const float* bias;    // [25]
  float* planes;        // forward: (B,25,HW) out
  const float* g;       // backward: (B,25,HW) gradient planes
  void* dx;             // backward: (B*HW, dx_cs) NHWC, first channel dx_coff (may be NULL: no data gradient wanted)
  void* gn;             // backward: (B*HW, 32) NHWC copy of g in the compute dtype (channels 25..31 zero)
  float* partial;       // backward: [gridDim.x][32] per-plane sums
  int x_cs, x_coff, dx_cs, dx_coff;
  int HW;               // pixels per image, a multiple of 32
  long long groups;     // B * HW / 32
};

__device__ __forceinline__ unsigned pack2(float lo, float hi) {      // v_cvt_pk_bf16_f32 (round to nearest even)
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
  const u32x4 u = {pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
  return __builtin_bit_cast(bf16x8, u);
}

// A wave-private LDS image of 32 pixel rows x NC 16-byte chunks, XOR-swizzled so that both access shapes are conflict-free
// (a ds_*_b128 lane group of 16 lanes must cover 16 distinct 16-byte slots of the 256-byte bank window):
//   written   "column-wise": lane (p, h) holds chunks of ITS pixel -- 16 consecutive lanes = 16 pixels, one chunk index;
//   read back "row-wise":    lane l = chunk l % NC of pixel l / NC -- consecutive lanes = consecutive bytes of the NHWC
//                            tensor, so a wave store instruction writes 1 KiB contiguous instead of 64 scattered 16-byte
//                            pieces (which cost the L2 eight write requests per line: 345 -> 2xx us on the backward).
template <int NC>
struct RowTile {
  static_assert(NC == 4 || NC == 8 || NC == 16 || NC == 32, "chunks per pixel row");
  static constexpr int BYTES = 32 * NC * 16;
  static __device__ __forceinline__ int off(int p, int c) {
    const int sw = NC >= 16 ? (p & 15) : ((p >> (NC == 8 ? 1 : 2)) & (NC - 1));
    return (p * NC + ((c & ~15) | ((c & 15) ^ sw))) * 16;
  }
  // store the image to `base` (+ q * pitch_bytes per pixel row q): NC / 2 wave instructions
  static __device__ __forceinline__ void flush(const char* img, char* base, size_t pitch_bytes, int lane) {
#pragma unroll
    for (int i = 0; i < NC / 2; ++i) {
      const int q = i * (64 / NC) + lane / NC, c = lane % NC;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + off(q, c));
      *reinterpret_cast<u32x4*>(base + (size_t)q * pitch_bytes + c * 16) = v;
    }
  }
};

// ---- forward ----------------------------------------------------------------------------------------------------------
template <typename T, int CIN>
__global__ __launch_bounds__(WPB * 64) void head_fwd_kernel(const HeadArgs A) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int HALF = CIN / 2;                 // channels a lane covers
  constexpr int NCH = BF ? HALF / 8 : HALF / 4; // its 16-byte chunks
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const long long wave0 = (long long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * WPB;

  // A fragments, once per wave: row m = p of W (zero beyond 25), this lane's k-half
  bf16x8 wa_b[BF ? NCH : 1];
  float wa_f[BF ? 1 : HALF];
  {
    const float* wr = A.w + (size_t)(p < NP ? p : 0) * CIN + h * HALF;
    if constexpr (BF) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = p < NP ? wr[8 * j + i] : 0.f;
        wa_b[j] = pack8(f);
      }
    } else {
#pragma unroll
      for (int s = 0; s < HALF; ++s) wa_f[s] = p < NP ? wr[s] : 0.f;
    }
  }
  // accumulator register r of lane (p, h) = channel 8 (r / 4) + 4 h + r % 4 of pixel p
  float bias_r[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = 8 * (r / 4) + 4 * h + (r % 4);
    bias_r[r] = c < NP ? A.bias[c] : 0.f;
  }

#pragma unroll 1
  for (long long gidx = wave0; gidx < A.groups; gidx += nwaves) {
    const long long gp0 = gidx * 32;
    const int b = (int)(gp0 / A.HW);
    const int pix = (int)(gp0 - (long long)b * A.HW) + p;
    const T* xp = reinterpret_cast<const T*>(A.x) + (size_t)(gp0 + p) * A.x_cs + A.x_coff + h * HALF;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias_r[r];
    if constexpr (BF) {
      bf16x8 xb[NCH];
#pragma unroll
      for (int j = 0; j < NCH; ++j) xb[j] = *reinterpret_cast<const bf16x8*>(xp + 8 * j);
#pragma unroll
      for (int j = 0; j < NCH; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_b[j], xb[j], acc, 0, 0, 0);
    } else {
      f32x4 xv[NCH];
#pragma unroll
      for (int j = 0; j < NCH; ++j) xv[j] = *reinterpret_cast<const f32x4*>(xp + 4 * j);
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa_f[4 * j + i], xv[j][i], acc, 0, 0, 0);
      }
    }
    float* op = A.planes + (size_t)b * NP * A.HW + pix;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 8 * (r / 4) + 4 * h + (r % 4);
      if (c < NP) op[(size_t)c * A.HW] = acc[r];
    }
  }
}

// ---- backward: data gradient + NHWC copy of g + per-plane sums -------------------------------------------------------------
// row m of A-tile t <-> channel (CIN/2) ((m >> 2) & 1) + 16 t + 4 (m >> 3) + (m & 3): accumulator register r of lane (p, h)
// (MFMA row 8 (r / 4) + 4 h + r % 4) is then channel (CIN/2) h + 16 t + r
template <int CIN>
__device__ __forceinline__ int tile_channel(int t, int m) {
  return (CIN / 2) * ((m >> 2) & 1) + 16 * t + 4 * (m >> 3) + (m & 3);
}

template <typename T, int CIN, bool DX>
__global__ __launch_bounds__(WPB * 64) void head_bwd_kernel(const HeadArgs A) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int NTL = CIN / 32;                 // 32-channel tiles of the data gradient
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const long long wave0 = (long long)blockIdx.x * WPB + wave, nwaves = (long long)gridDim.x * WPB;
  __shared__ float red[WPB][32];
  constexpr int ES = BF ? 2 : 4;
  using DxTile = RowTile<CIN * ES / 16>;       // the data gradient's 32 pixel rows
  using GnTile = RowTile<32 * ES / 16>;        // the 32-channel NHWC copy's
  __shared__ __attribute__((aligned(16))) char tiles[WPB][(DX ? DxTile::BYTES : 0) + GnTile::BYTES];
  char* const gn_img = tiles[wave];
  char* const dx_img = tiles[wave] + GnTile::BYTES;

  // plane of this lane's value v (bf16: v = 8 s + i -> n = 16 s + 8 h + i; fp32: n = 16 h + v): see the file header
  auto plane_of = [&](int v) { return BF ? 16 * (v >> 3) + 8 * h + (v & 7) : 16 * h + v; };

  bf16x8 wt_b[(BF && DX) ? NTL : 1][2];
  float wt_f[(!BF && DX) ? NTL : 1][16];
  if constexpr (DX) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int c = tile_channel<CIN>(t, p);
      if constexpr (BF) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float f[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int n = 16 * s + 8 * h + i;
            f[i] = n < NP ? A.w[(size_t)n * CIN + c] : 0.f;
          }
          wt_b[t][s] = pack8(f);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int n = 16 * h + s;
          wt_f[t][s] = n < NP ? A.w[(size_t)n * CIN + c] : 0.f;
        }
      }
    }
  }
  float bsum[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) bsum[v] = 0.f;

#pragma unroll 1
  for (long long gidx = wave0; gidx < A.groups; gidx += nwaves) {
    const long long gp0 = gidx * 32;
    const int b = (int)(gp0 / A.HW);
    const int pix = (int)(gp0 - (long long)b * A.HW) + p;
    const float* gpl = A.g + (size_t)b * NP * A.HW + pix;
    float gv[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int n = plane_of(v);
      gv[v] = n < NP ? gpl[(size_t)n * A.HW] : 0.f;
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) bsum[v] += gv[v];
    // results go through the wave's LDS images (column-wise in, row-wise out: RowTile) and leave as contiguous rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous group's flush has read the images (wave-private: program order)
    if constexpr (BF) {
      bf16x8 gb[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = gv[8 * s + i];
        gb[s] = pack8(f);
        *reinterpret_cast<bf16x8*>(gn_img + GnTile::off(p, 2 * s + h)) = gb[s];       // channels 16 s + 8 h ..
      }
      if constexpr (DX) {
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt_b[t][0], gb[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt_b[t][1], gb[1], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const u32x4 o = {pack2(acc[8 * q + 0], acc[8 * q + 1]), pack2(acc[8 * q + 2], acc[8 * q + 3]),
                             pack2(acc[8 * q + 4], acc[8 * q + 5]), pack2(acc[8 * q + 6], acc[8 * q + 7])};
            *reinterpret_cast<u32x4*>(dx_img + DxTile::off(p, h * (CIN / 16) + 2 * t + q)) = o;    // channels h CIN/2 + 16 t + 8 q ..
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = {gv[4 * q], gv[4 * q + 1], gv[4 * q + 2], gv[4 * q + 3]};
        *reinterpret_cast<f32x4*>(gn_img + GnTile::off(p, 4 * h + q)) = o;              // channels 16 h + 4 q ..
      }
      if constexpr (DX) {
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt_f[t][s], gv[s], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 o = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *reinterpret_cast<f32x4*>(dx_img + DxTile::off(p, h * (CIN / 8) + 4 * t + q)) = o;     // channels h CIN/2 + 16 t + 4 q ..
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the images are complete (the LDS operations of one wave complete in order)
    GnTile::flush(gn_img, reinterpret_cast<char*>(A.gn) + (size_t)gp0 * 32 * ES, (size_t)32 * ES, lane);
    if constexpr (DX)
      DxTile::flush(dx_img, reinterpret_cast<char*>(A.dx) + ((size_t)gp0 * A.dx_cs + A.dx_coff) * ES, (size_t)A.dx_cs * ES, lane);
  }

  // per-plane sums of this workgroup: lanes of one half hold the same 16 planes
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    float s = bsum[v];
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (p == 0) red[wave][plane_of(v)] = s;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) s += red[w][threadIdx.x];
    A.partial[(size_t)blockIdx.x * 32 + threadIdx.x] = s;
  }
}

// partial [rows][32] -> dbias[25], fixed order, fp64
__global__ __launch_bounds__(1024) void head_dbias_fold_kernel(const float* __restrict__ partial, int rows, float* __restrict__ dbias) {
  __shared__ double red[32][33];
  const int n = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  double s = 0.0;
  for (int r = r0; r < rows; r += 32) s += (double)partial[(size_t)r * 32 + n];
  red[r0][n] = s;
  __syncthreads();
  if (threadIdx.x < NP) {
    double t = 0.0;
    for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
    dbias[threadIdx.x] = (float)t;
  }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

int head_grid(long long groups) {
  static const int per_cu = env_int("JSPSR_HEAD_WGS", 4);
  static const int cus = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    return cu;
  }();
  long long g = (groups + WPB - 1) / WPB;
  const long long cap = (long long)cus * per_cu;
  if (g > cap) g = cap;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

bool cin_ok(int Cin) { return Cin == 32 || Cin == 64 || Cin == 128; }

template <typename F>
void by_cin(int Cin, F&& f) {
  if (Cin == 32) f(std::integral_constant<int, 32>{});
  else if (Cin == 64) f(std::integral_constant<int, 64>{});
  else f(std::integral_constant<int, 128>{});
}

int check_shape(const char* who, int dtype, int B, int HW, int Cin, int cs, int coff, const void* x) {
  if (dtype != JSPSR_F32 && dtype != JSPSR_BF16) return jspsr::fail(JSPSR_EINVAL, "%s: bad dtype %d", who, dtype);
  if (B <= 0 || HW <= 0 || HW % 32) return jspsr::fail(JSPSR_EINVAL, "%s: H*W = %d must be a positive multiple of 32", who, HW);
  if (!cin_ok(Cin)) return jspsr::fail(JSPSR_EINVAL, "%s: Cin = %d (built: 32, 64, 128)", who, Cin);
  const int e = dtype == JSPSR_BF16 ? 8 : 4;
  if (cs < coff + Cin || cs % e || coff % e) return jspsr::fail(JSPSR_EINVAL, "%s: channel pitch %d / offset %d (Cin %d, 16-byte chunks)", who, cs, coff, Cin);
  if (!jspsr::aligned16(x)) return jspsr::fail(JSPSR_EALIGN, "%s: tensor not 16-byte aligned", who);
  return JSPSR_OK;
}

}  // namespace

extern "C" int jspsr_head_ok(int dtype, int B, int H, int W, int Cin) {
  return (dtype == JSPSR_F32 || dtype == JSPSR_BF16) && B > 0 && H > 0 && W > 0 && ((long long)H * W) % 32 == 0 &&
         (long long)H * W < (1LL << 31) && cin_ok(Cin);
}

extern "C" int jspsr_head_forward(int dtype, const void* x, int x_cstride, int x_coff, int Cin, const float* w25, const float* b25,
                                  float* planes, int B, int H, int W, jspsr_stream_t stream) {
  if (!x || !w25 || !b25 || !planes) return jspsr::fail(JSPSR_EINVAL, "head_forward: null pointer");
  if (!jspsr_head_ok(dtype, B, H, W, Cin)) return jspsr::fail(JSPSR_EINVAL, "head_forward: unsupported shape (jspsr_head_ok)");
  const int HW = H * W;
  if (int e = check_shape("head_forward", dtype, B, HW, Cin, x_cstride, x_coff, x)) return e;
  if (!jspsr::aligned4(planes)) return jspsr::fail(JSPSR_EALIGN, "head_forward: planes not 4-byte aligned");
  HeadArgs A{};
  A.x = x; A.w = w25; A.bias = b25; A.planes = planes; A.x_cs = x_cstride; A.x_coff = x_coff; A.HW = HW;
  A.groups = (long long)B * HW / 32;
  const dim3 grid(head_grid(A.groups)), block(WPB * 64);
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_cin(Cin, [&](auto C) {
    if (dtype == JSPSR_BF16) hipLaunchKernelGGL((head_fwd_kernel<__bf16, decltype(C)::value>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((head_fwd_kernel<float, decltype(C)::value>), grid, block, 0, s, A);
  });
  return jspsr::check_launch("head_forward");
}

extern "C" size_t jspsr_head_backward_workspace_bytes(int B, int H, int W) {
  (void)B; (void)H; (void)W;
  return (size_t)4096 * 32 * sizeof(float);
}

extern "C" int jspsr_head_backward(int dtype, const float* grad_planes, const float* w25, int Cin, void* grad_x, int gx_cstride,
                                   int gx_coff, void* grad_nhwc32, float* grad_b25, void* workspace, int B, int H, int W,
                                   jspsr_stream_t stream) {
  if (!grad_planes || !w25 || !grad_nhwc32 || !grad_b25 || !workspace) return jspsr::fail(JSPSR_EINVAL, "head_backward: null pointer");
  if (!jspsr_head_ok(dtype, B, H, W, Cin)) return jspsr::fail(JSPSR_EINVAL, "head_backward: unsupported shape (jspsr_head_ok)");
  const int HW = H * W;
  if (grad_x) {
    if (int e = check_shape("head_backward", dtype, B, HW, Cin, gx_cstride, gx_coff, grad_x)) return e;
  }
  if (!jspsr::aligned16(grad_nhwc32) || !jspsr::aligned16(workspace) || !jspsr::aligned4(grad_planes))
    return jspsr::fail(JSPSR_EALIGN, "head_backward: alignment");
  HeadArgs A{};
  A.g = grad_planes; A.w = w25; A.dx = grad_x; A.dx_cs = gx_cstride; A.dx_coff = gx_coff; A.gn = grad_nhwc32;
  A.partial = static_cast<float*>(workspace); A.HW = HW;
  A.groups = (long long)B * HW / 32;
  const int rows = head_grid(A.groups);
  const dim3 grid(rows), block(WPB * 64);
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_cin(Cin, [&](auto C) {
    constexpr int CI = decltype(C)::value;
    if (dtype == JSPSR_BF16) {
      if (grad_x) hipLaunchKernelGGL((head_bwd_kernel<__bf16, CI, true>), grid, block, 0, s, A);
      else hipLaunchKernelGGL((head_bwd_kernel<__bf16, CI, false>), grid, block, 0, s, A);
    } else {
      if (grad_x) hipLaunchKernelGGL((head_bwd_kernel<float, CI, true>), grid, block, 0, s, A);
      else hipLaunchKernelGGL((head_bwd_kernel<float, CI, false>), grid, block, 0, s, A);
    }
  });
  if (int e = jspsr::check_launch("head_backward")) return e;
  hipLaunchKernelGGL(head_dbias_fold_kernel, dim3(1), dim3(1024), 0, s, A.partial, rows, grad_b25);
  return jspsr::check_launch("head_dbias_fold");
}
