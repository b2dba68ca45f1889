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
// LDS image: rows of 128 B (+16 B pad -> conflict-free ds_read_b128), 16-byte chunks; lane
// (r = l & 31, h = l >> 5) of a 32-row MFMA tile reads chunk 2s+h of row r at sub-step s.
//   fp32:  a chunk is 4 k-values  -> 4 x v_mfma_f32_32x32x2_f32   (exact fp32 FMA chain)
//   bf16:  a chunk is 8 k-values  -> 1 x v_mfma_f32_32x32x16_bf16 (fp32 accumulate)
// (the k order inside a 32-byte pair is permuted identically for A and B, which is all a
// dot product needs).
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
};

}  // namespace jspsr
