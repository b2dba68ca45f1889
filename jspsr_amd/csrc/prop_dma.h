// Shared pieces of the persistent LDS-DMA propagation kernels (prop_dma.hip: planar fp32 operands; prop_head_dma.hip:
// operands straight from the generator head's bf16 NHWC output): LDS-DMA issue from inline asm, the double-buffered
// workgroup layout, the nine-tap corner gather out of the staged DEM tile.
#pragma once
#include "prop_tile.h"

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DW = 64;                 // tile width in pixels = one wave's row
constexpr int DLW = DW + 2 * HALO;     // staged DEM row: 80 floats


// NW waves (rows per pass), RP passes per tile: a tile is 64 x (NW RP) pixels and shares ONE staged DEM tile + halo.
// OPB: bytes of one row's operand buffer (whole 1 KiB DMA pieces).
template <int NW, int OPB, int RP = 1>
struct DmaCfg {
  static constexpr int TH = NW * RP;
  static constexpr int LH = TH + 2 * HALO;
  static constexpr int CHUNKS = LH * (DLW / 4);        // 16-byte chunks of the DEM tile
  static constexpr int PIECES = (CHUNKS + (DLW + 4) / 4 + 63) / 64;      // + a zero pad of one row + 2 behind the tile (gather_taps)
  static constexpr int DPW = (PIECES + NW - 1) / NW;   // DEM pieces per wave
  static constexpr int DEMB = PIECES * 1024;
  static constexpr int SMEM = 2 * DEMB + 2 * NW * OPB;
  static_assert(LH * DLW + DLW + 2 <= DEMB / 4, "zero pad behind the staged DEM tile (gather_taps)");
};

template <int OC>
__device__ __forceinline__ constexpr int dch(int k, int c) {     // offset channel of (tap k, component c)
  return OC == 18 ? 2 * k + c : 2 * (k < 4 ? k : k - 1) + c;
}

template <bool NTL>
__device__ __forceinline__ void dma_piece(unsigned lds_dst, const void* src) {
  // lane l's 16 bytes land at lds_dst + 16 l.  M0 is written in the statement that uses it; nothing else in these
  // kernels uses M0 (checked at build time: csrc/Makefile, check_m0).
  if (NTL) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" : : "s"(lds_dst), "v"(src) : "memory");
  else     asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_dst), "v"(src) : "memory");
}

// Lab builds only (tools/lab/build_k1d_variants.sh): -DK1D_STAMPS accumulates per-phase cycle counts of every wave of
// the backward kernel behind the partial rows of the workspace (wait / barrier / issue / compute / store);
// -DK1D_NOCOMPUTE replaces gather + arithmetic by copies (wrong results: the structure's streaming ceiling).
#ifdef K1D_STAMPS
#define K1D_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp[i] += now_ - last_; last_ = now_; } while (0)
#else
#define K1D_STAMP(i) do { } while (0)
#endif

#ifndef K1D_NTS
#define K1D_NTS 1          // gradient planes leave as non-temporal stores
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// NT_ taps' corner fetches at once.  Same values as corners_fast() (prop_tile.h), fewer instructions per tap:
//   * floor + convert = v_cvt_flr_i32_f32, the fractional part = v_fract_f32 (differs from p - floor(p) only for p a
//     hair below an integer: 1 - 2^-24 instead of the rounded 1.0);
//   * the four LDS reads are issued unconditionally; a sample outside tile + halo reads the ZERO pad behind the staged
//     tile (ZPAD: the DEM pieces' out-of-range lanes land zeros there every tile), so no select zeroes anything;
//   * every rarer case -- a tap beyond the halo but near the raster (bounds-checked global reads), a tap outside the
//     raster, NaN / inf coordinates (the tap contributes 0) -- sits behind ONE wave-level test per call.
template <int LH, int LW, int NT_, int ZPAD>
__device__ __forceinline__ void gather_taps(const float* __restrict__ lds, const float* __restrict__ img, int H, int W, int ly0, int lx0,
                                            const float (&py)[NT_], const float (&px)[NT_], Corners (&c)[NT_]) {
  unsigned out = 0;
#pragma unroll
  for (int k = 0; k < NT_; ++k) {
    // fmaxf / fminf return the non-NaN operand: NaN and +-inf coordinates become huge finite ones (out of every range)
    const float cy = fminf(fmaxf(py[k], -1.0e9f), 1.0e9f), cx = fminf(fmaxf(px[k], -1.0e9f), 1.0e9f);
    int y0, x0;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(y0) : "v"(cy));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(x0) : "v"(cx));
    c[k].ly = __builtin_amdgcn_fractf(cy);
    c[k].lx = __builtin_amdgcn_fractf(cx);
    const unsigned ry = (unsigned)y0 - (unsigned)ly0, rx = (unsigned)x0 - (unsigned)lx0;
    const bool inl = ry < (unsigned)(LH - 1) && rx < (unsigned)(LW - 1);
    const float* p = lds + (inl ? ry * LW + rx : (unsigned)ZPAD);
    c[k].v00 = p[0];
    c[k].v01 = p[1];
    c[k].v10 = p[LW];
    c[k].v11 = p[LW + 1];
    if (!inl) out |= 1u << k;
  }
  if (__builtin_amdgcn_ballot_w64(out != 0) != 0) {
#pragma unroll
    for (int k = 0; k < NT_; ++k) {
      if ((out >> k) & 1u) {
        const bool near = (py[k] > -2.f) && (py[k] < (float)(H + 1)) && (px[k] > -2.f) && (px[k] < (float)(W + 1));   // false for NaN
        if (near) {
          const int y0 = (int)floorf(py[k]), x0 = (int)floorf(px[k]);
          const bool y0ok = (unsigned)y0 < (unsigned)H, y1ok = (unsigned)(y0 + 1) < (unsigned)H;
          const bool x0ok = (unsigned)x0 < (unsigned)W, x1ok = (unsigned)(x0 + 1) < (unsigned)W;
          const float* q = img + (ptrdiff_t)y0 * W + x0;
          if (y0ok && x0ok) c[k].v00 = q[0];
          if (y0ok && x1ok) c[k].v01 = q[1];
          if (y1ok && x0ok) c[k].v10 = q[W];
          if (y1ok && x1ok) c[k].v11 = q[W + 1];
        } else {
          c[k].ly = c[k].lx = 0.f;   // keep inf/nan coordinates out of the arithmetic: the tap contributes 0
        }
      }
    }
  }
}

}  // namespace
