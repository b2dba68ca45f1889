// Implicit-GEMM convolution on the gfx950 matrix cores (MFMA), NHWC activations.
//
// GEMM view: M = pixels of the output grid, N = output channels, K = taps x input channels.
//   A[m][k]  gathered on the fly from the NHWC input (zero outside the raster) -> LDS
//   B[n][k]  weights pre-packed k-contiguous per output channel ("OHWI")       -> LDS
//   C[m][n]  fp32 accumulators in registers (32x32 MFMA tiles), epilogue to NHWC.
// One kernel covers (reference op it replaces in brackets):
//   * forward conv, any kernel size / stride / padding           [nn.Conv2d, basics.py:11-20,39-47]
//   * data gradient of a conv and ConvTranspose2d forward         [autograd of the above; basics.py:69-77]
//     -- "transposed" tap walk (input row = base - tap), one launch per stride phase so that no
//        MFMA work is spent on structurally-zero taps.
// LDS image: rows of 128 B of K (+ pad), 16-byte chunks; how a lane picks its chunk depends on the product (conv.hip `Mma`):
//   fp32:  144-byte rows, a chunk is 4 k-values; lane (r = l & 31, h = l >> 5) of a 32-row block reads chunk 2s + h of
//          row r at sub-step s -> 4 x v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain)
//   bf16:  160-byte rows, a chunk is 8 k-values; lane (r = l & 15, q = l >> 4) of a 16-row block reads chunk 4s + q of
//          row r -> 1 x v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  Round 3 (was 32x32x16): same cycles and LDS
//          traffic per flop, but the chip holds a higher clock on the 16x16x32 shape.
// (the k order inside a sub-step is permuted identically for A and B, which is all a dot product needs).
#pragma once
#include "common.h"

namespace jspsr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16 = unsigned short;

struct ConvGeom {
  // tensors
  int B, IH, IW, Cin;     // gathered tensor, NHWC
  int OH, OW, Cout;       // written tensor, NHWC (full extent)
  int in_cstride;         // channel pitch of a gathered pixel (>= Cin; lets a conv read a channel slice)
  int in_coff;            // first channel of the slice
  int out_cstride;        // channel pitch of a written pixel (>= Cout; lets a conv write into a concat buffer)
  int out_coff;
  // m-space: pixels enumerated by this launch (MB = B)
  int MH, MW;
  // gathered row/col of m-pixel (y', x') at tap (ty, tx):  iy = y'*iy_mul + iy_add + sign*ty
  int iy_mul, iy_add, ix_mul, ix_add, sign;
  int nty, ntx;           // taps walked
  int ky0, kx0, kstep;    // weight tap of walk index (ty, tx) = (ky0 + kstep*ty, kx0 + kstep*tx)
  int KH, KW;             // weight tap grid (packed weights are [N][KH][KW][Cin])
  // written pixel of m-pixel (y', x'): (y'*oy_mul + oy_add, x'*ox_mul + ox_add)
  int oy_mul, oy_add, ox_mul, ox_add;
  int relu;               // epilogue ReLU
  int accumulate;         // lab switch (JSPSR_CONV_NOXCD)
  const void* addend;     // optional tensor on the written grid: out = [relu](acc * scale + bias + addend)
  int add_cstride;        // its channel pitch (channel 0 of the addend = channel out_coff of the written slice)
  const float* scale;     // optional per-output-channel factor (inference: BatchNorm folded into the epilogue)
  const float* in_affine; // optional [2][Cin] (scale | shift): the gathered tensor is read as [relu](x * scale + shift),
  int in_relu;            // applied while the patch is staged (patch kernel only); zero padding stays zero
  FastDiv fd_ntn, fd_ttx, fd_tty;   // patch kernel: division by its N-tile / tile-column / tile-row counts (set by its launcher)
  // Round 4 -- the reduce pass of a BatchNorm backward in the epilogue of the data gradient that PRODUCES its input gradient
  // (patch kernel, 16-byte epilogue path): red_x = the BatchNorm's saved pre-normalisation tensor on the written grid (channel
  // pitch red_cstride, channel 0 = channel out_coff of the written slice... of ITS OWN tensor: first channel 0), red_par =
  // [4][Cout] floats (a | b | is | mis: mask = a z + b > 0, xhat = z is + mis), red_out = partial rows
  // [jspsr_conv2d_stats_rows][2][Cout]: sum dz and sum dz xhat over each 8x16-pixel tile, dz = the STORED gradient where the
  // ReLU was open.  NULL: off.
  const void* red_x;
  int red_cstride;
  const float* red_par;
  float* red_out;
};

// ---- shared by the register-resident kernels (conv64.hip, conv128.hip) ------------------------------------------------------
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {      // one v_cvt_pk_bf16_f32
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// max(x, 0) on the bit pattern: one v_max_i32 (a negative float is a negative integer; -0 and negative NaNs become +0)
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// 8 bf16 + 8 bf16 (fp32 add, one rounding), optional ReLU
__device__ __forceinline__ u32x4 add_bf16x8(u32x4 a, u32x4 b, bool relu) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16);
    float hi = __uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(b[i] & 0xffff0000u);
    if (relu) { lo = relu_bits(lo); hi = relu_bits(hi); }
    r[i] = pack_bf16(lo, hi);
  }
  return r;
}


// K2r (conv64.hip): 3x3 64->64 unit-stride bf16 conv / data gradient, weights resident in registers
bool conv64_resident_ok(const ConvGeom& g, const void* in, const void* wgt, const float* bias, const void* out);
int launch_conv64_resident(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g,
                           hipStream_t s);

// K2q (conv128.hip): 3x3 128->128 unit-stride bf16 conv / data gradient, the weight matrix resident in one CU's registers
bool conv128_resident_ok(const ConvGeom& g, const void* in, const void* wgt, const float* bias, const void* out);
int launch_conv128_resident(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g, hipStream_t s);

}  // namespace jspsr

// [relu](x * sc + sh) on 16 bytes of T (4 fp32 / 8 bf16 channels), fp32 arithmetic, one rounding to the storage type
template <typename T, int N>
__device__ __forceinline__ uint4 affine_relu16(const uint4& v, const float (&sc)[N], const float (&sh)[N], bool relu) {
  static_assert(N * sizeof(T) == 16, "one 16-byte chunk of channels");
  uint4 r;
  if constexpr (sizeof(T) == 4) {
    float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[i] = f[i] * sc[i] + sh[i];
      if (relu) f[i] = fmaxf(f[i], 0.f);
    }
    r = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  } else {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float lo = __uint_as_float(w[i] << 16) * sc[2 * i] + sh[2 * i];
      float hi = __uint_as_float(w[i] & 0xffff0000u) * sc[2 * i + 1] + sh[2 * i + 1];
      if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
      const __bf16 bl = (__bf16)lo, bh = (__bf16)hi;
      o[i] = (unsigned)__builtin_bit_cast(unsigned short, bl) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
    }
    r = make_uint4(o[0], o[1], o[2], o[3]);
  }
  return r;
}

