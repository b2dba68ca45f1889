// K2w -- weight gradient of a convolution on MFMA (the autograd of nn.Conv2d / nn.ConvTranspose2d
// w.r.t. the weight; reference call site: loss.backward(), train/train_utils.py:217).
//
//   dW[r][ky][kx][c] = sum over m-pixels (b,oy,ox) of  G[b,oy,ox,r] * X[b, oy*s-p+ky, ox*s-p+kx, c]
//
// GEMM view: rows = channels of G (the tensor living on the conv's OUTPUT grid), columns = flattened
// (tap, channel of X), reduction = pixels.  Both operand tiles are staged in LDS exactly as they
// lie in memory ([pixel][channel], channel contiguous):
//   fp32: an MFMA 32x32x2 operand is one value per lane, lanes = consecutive channels of one pixel
//         row -> conflict-free ds_read_b32;
//   bf16: an MFMA 32x32x16 operand is 8 consecutive PIXELS per lane -> ds_read_b64_tr_b16
//         (hardware transposing LDS read), two per operand; row pitch padded so the 4 rows of a
//         block fall in different banks.
// Split over pixel ranges (grid.z) so every layer fills the chip; slabs are summed in a fixed
// order by a second kernel that also writes the master (R, C, KH, KW) fp32 layout -> bit-reproducible.
#include "conv_igemm.h"

#include <type_traits>

namespace {

using namespace jspsr;

constexpr int NT = 256;
using s16x4 = __attribute__((ext_vector_type(4))) short;

struct WgradGeom {
  int B, OH, OW;            // m-space (grid of G)
  int Cg, g_cs, g_coff;     // G channels / pitch / offset
  int IH, IW;               // grid of X
  int Cx, x_cs, x_coff;     // X channels (chunk-padded) / pitch / offset
  int KH, KW, stride, pad;
  int Ktot;                 // KH*KW*Cx
  int m_per_split;          // pixels per grid.z slice (multiple of the stage depth)
  long long M;
};

template <typename T> struct WT;
template <> struct WT<float> { static constexpr int EPC = 4, BKM = 32; };
template <> struct WT<__bf16> { static constexpr int EPC = 8, BKM = 64; };

// bf16 product of the weight-gradient kernels.  -DWGRADLAB_MFMA16 (kernel-lab timing build, WRONG results): the same flops and
// operand reads as two 16x16x32 products, to price the conversion the conv kernels got in round 3 (conv.hip `Mma`).
#ifdef WGRADLAB_MFMA16
#define WG_NM 6
#else
#define WG_NM 3      // MFMAs per (k-block, tap row) group of the patch kernel's pinned schedule
#endif
__device__ __forceinline__ f32x16 wg_mma(const bf16x8& a, const bf16x8& b, f32x16 acc) {
#ifndef WGRADLAB_MFMA16
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#else
  f32x4 c0 = {acc[0], acc[1], acc[2], acc[3]}, c1 = {acc[4], acc[5], acc[6], acc[7]};
  c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
  acc[0] = c0[0]; acc[1] = c0[1]; acc[2] = c0[2]; acc[3] = c0[3];
  acc[4] = c1[0]; acc[5] = c1[1]; acc[6] = c1[2]; acc[7] = c1[3];
  return acc;
#endif
}

__host__ __device__ constexpr int pitch_bytes(int row_bytes) {
  return row_bytes + ((row_bytes % 128 == 0) ? 64 : 0);
}

template <typename T, int BMC, int BNC, int WGM, int WGN>
__global__ __launch_bounds__(NT, 2) void wgrad_kernel(const T* __restrict__ G, const T* __restrict__ X,
                                                     float* __restrict__ ws, WgradGeom g) {
  constexpr int EPC = WT<T>::EPC, BKM = WT<T>::BKM;
  constexpr int WTM = BMC / WGM, WTN = BNC / WGN, MI = WTM / 32, NI = WTN / 32;
  constexpr int GP = pitch_bytes(BMC * sizeof(T)), XP = pitch_bytes(BNC * sizeof(T));
  constexpr int CPR_G = BMC / EPC, CPR_X = BNC / EPC;  // 16-byte chunks per tile row
  constexpr int G_IT = (BKM * CPR_G + NT - 1) / NT, X_IT = BKM * CPR_X / NT;
  constexpr int G_RSTEP = NT / CPR_G, X_RSTEP = NT / CPR_X;
  constexpr int GS_BYTES = BKM * GP, XS_BYTES = BKM * XP;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  static_assert(WGM * WGN == 4 && MI >= 1 && NI >= 1 && NT % CPR_G == 0 && NT % CPR_X == 0, "tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Gs = smem;                    // [2][BKM][GP]
  char* Xs = smem + 2 * GS_BYTES;     // [2][BKM][XP]

  const int tid = threadIdx.x;
  // 1-D grid, numbered so that one XCD (private L2) gets a contiguous run of (slice, column group, row
  // tile) triples: the column groups (taps) of one pixel slice share their G rows and overlapping X rows
  const int gx = (g.Cg + BMC - 1) / BMC, gy = (g.Ktot + BNC - 1) / BNC;
  const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
  const int bx = lin % gx, by = (lin / gx) % gy, bz = lin / (gx * gy);
  const int co0 = bx * BMC, n0 = by * BNC;
  const long long m_begin = (long long)bz * g.m_per_split;
  const long long m_end = (m_begin + g.m_per_split < g.M) ? m_begin + g.m_per_split : g.M;
  const int KT = (int)((m_end - m_begin + BKM - 1) / BKM);

  // ---- G: rows are linear in m.  Descriptor base = first row of this slice, range = the slice:
  // rows past m_end are out of range and come back as zeros (which also silences the X rows there).
  const int gc = tid % CPR_G, gr0 = tid / CPR_G;
  const int g_row_bytes = g.g_cs * (int)sizeof(T);
  const char* gbase = reinterpret_cast<const char*>(G) + ((size_t)m_begin * g.g_cs + g.g_coff) * sizeof(T);
  const unsigned g_range = (unsigned)((m_end - m_begin) * (long long)g_row_bytes);
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gbase), 0, g_range, 0x00020000);
  unsigned goff[G_IT];
#pragma unroll
  for (int i = 0; i < G_IT; ++i) {
    const int r = gr0 + i * G_RSTEP;
    goff[i] = (r < BKM && co0 + gc * EPC < g.Cg) ? (unsigned)(r * g_row_bytes + (co0 + gc * EPC) * (int)sizeof(T)) : OOB;
  }

  // ---- X: chunk xc is a fixed (tap, channel) column group; rows walk the output grid.
  const int xc = tid % CPR_X, xr0 = tid / CPR_X;
  const int kcol = n0 + xc * EPC;
  const bool x_colok = kcol < g.Ktot;
  const int tap = x_colok ? kcol / g.Cx : 0, cix = x_colok ? kcol % g.Cx : 0;
  const int ky = tap / g.KW, kx = tap % g.KW;
  const int x_pix_bytes = g.x_cs * (int)sizeof(T);
  const long long b_first = m_begin / ((long long)g.OH * g.OW);
  const char* xbase = reinterpret_cast<const char*>(X) + ((size_t)b_first * g.IH * g.IW * g.x_cs + g.x_coff) * sizeof(T);
  const long long x_left = ((long long)g.B - b_first) * g.IH * g.IW * (long long)x_pix_bytes;
  const unsigned x_range = x_left > 0xFFFFFF00LL ? 0xFFFFFF00u : (unsigned)x_left;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase), 0, x_range, 0x00020000);
  // per row: output coords, byte offset of input row (iy, ix = kx - pad) and its validity
  int xoy[X_IT], xox[X_IT], xb[X_IT];
  unsigned xrow[X_IT];   // may be "negative" (mod 2^32) when kx < pad; adding a valid ix brings it back
  bool xrow_ok[X_IT];
  auto row_setup = [&](int i) {
    const int iy = xoy[i] * g.stride - g.pad + ky;
    const long long pix = ((long long)(xb[i] - b_first) * g.IH + iy) * g.IW + (kx - g.pad);
    const long long off = pix * x_pix_bytes + cix * (long long)sizeof(T);
    xrow_ok[i] = x_colok && (unsigned)iy < (unsigned)g.IH && xb[i] < g.B &&
                 off + (long long)g.IW * x_pix_bytes < 0xFFFFFF00LL;
    xrow[i] = (unsigned)off;
  };
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const long long m = m_begin + xr0 + i * X_RSTEP;
    xox[i] = (int)(m % g.OW);
    const long long tq = m / g.OW;
    xoy[i] = (int)(tq % g.OH);
    xb[i] = (int)(tq / g.OH);
    row_setup(i);
  }

  uint4 greg[2][G_IT], xreg[2][X_IT];
  int g_soff = 0;   // scalar: advances by one stage of rows
  auto load_stage = [&](auto SET) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < G_IT; ++i) {
      // the range check covers voffset only; rows past the slice are caught by folding the stage
      // advance into voffset when it would leave the range
      const unsigned off = goff[i] == OOB ? OOB : goff[i] + (unsigned)g_soff;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off, 0, 0);
      greg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    g_soff += BKM * g_row_bytes;
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int ix = xox[i] * g.stride - g.pad + kx;
      const bool ok = xrow_ok[i] && (unsigned)ix < (unsigned)g.IW;
      const unsigned off = ok ? xrow[i] + (unsigned)(xox[i] * g.stride * x_pix_bytes) : OOB;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
      xreg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
      xox[i] += BKM;               // advance this row by one stage of pixels
      if (xox[i] >= g.OW) {
        do {
          xox[i] -= g.OW;
          if (++xoy[i] == g.OH) { xoy[i] = 0; ++xb[i]; }
        } while (xox[i] >= g.OW);
        row_setup(i);
      }
    }
  };
  char* const g_st = Gs + gr0 * GP + gc * 16;
  char* const x_st = Xs + xr0 * XP + xc * 16;
  auto store_stage = [&](auto SET, auto BUF) {
    constexpr int set = decltype(SET)::value, buf = decltype(BUF)::value;
#pragma unroll
    for (int i = 0; i < G_IT; ++i)
      if (G_IT * G_RSTEP <= BKM || gr0 + i * G_RSTEP < BKM)
        *reinterpret_cast<uint4*>(g_st + buf * GS_BYTES + i * G_RSTEP * GP) = greg[set][i];
#pragma unroll
    for (int i = 0; i < X_IT; ++i)
      *reinterpret_cast<uint4*>(x_st + buf * XS_BYTES + i * X_RSTEP * XP) = xreg[set][i];
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = lane & 31, lh = lane >> 5;
  // transposing reads (bf16): lane (4q+p) of a 16-lane group addresses row q, columns 4p..4p+3 of a
  // 4-pixel x 16-channel block and receives column (lane&15) of the 4 rows.
  const int grp16 = (lane >> 4) & 1, li = lane & 15, q = li >> 2, pp = li & 3;
  const char* const g_ld = sizeof(T) == 4 ? Gs + lh * GP + (wm * WTM + lr) * 4
                                           : Gs + (8 * lh + q) * GP + (wm * WTM + 16 * grp16 + 4 * pp) * 2;
  const char* const x_ld = sizeof(T) == 4 ? Xs + lh * XP + (wn * WTN + lr) * 4
                                           : Xs + (8 * lh + q) * XP + (wn * WTN + 16 * grp16 + 4 * pp) * 2;

  auto compute = [&](auto BUF) {
    constexpr int buf = decltype(BUF)::value;
    if constexpr (sizeof(T) == 4) {
#pragma unroll 4
      for (int s = 0; s < BKM / 2; ++s) {
        float a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          a[mi] = *reinterpret_cast<const float*>(g_ld + buf * GS_BYTES + 2 * s * GP + mi * 32 * 4);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          b[ni] = *reinterpret_cast<const float*>(x_ld + buf * XS_BYTES + 2 * s * XP + ni * 32 * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < BKM / 16; ++kb) {
        bf16x8 a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const char* p = g_ld + buf * GS_BYTES + kb * 16 * GP + mi * 32 * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * GP));
          a[mi] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const char* p = x_ld + buf * XS_BYTES + kb * 16 * XP + ni * 32 * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * XP));
          b[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = wg_mma(a[mi], b[ni], acc[mi][ni]);
      }
    }
  };
  // [loads of stage t+2] -> MFMAs of stage t -> [stage t+1 regs -> LDS] -> barrier; branch-free so
  // the compiler's vmcnt waits are exact counts (see conv.hip)
  auto step = [&](auto CUR, auto NXT) {
    load_stage(CUR);
    compute(CUR);
    store_stage(NXT, NXT);
    __syncthreads();
  };
  load_stage(S0{});
  load_stage(S1{});
  store_stage(S0{}, S0{});
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    step(S0{}, S1{});
    step(S1{}, S0{});
  }
  if (kt < KT) step(S0{}, S1{});

  float* slab = ws + (size_t)bz * g.Cg * g.Ktot;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int k = n0 + wn * WTN + ni * 32 + lr;
    if (k >= g.Ktot) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wm * WTM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (co < g.Cg) slab[(size_t)co * g.Ktot + k] = acc[mi][ni][e];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 layers (every residual-block conv): all nine taps from one staging.
//
// A workgroup owns a 64-channel tile of G, a 64-channel chunk of X and a column strip of the image (U output
// pixels wide) over a range of rows, and walks down the strip one output row per step.  Per step it stages the
// G row segment (U px) and ONE new X row segment (U + 2 px, with the left/right neighbours); the previous two X
// rows are still in a 4-slot LDS ring.  The nine taps are nine MFMA accumulators fed from the same G fragment
// and nine shifted views of the ring (ky = ring slot, kx = +0/1/2 pixel rows in LDS), so each operand element
// is loaded from memory and stored to LDS once per (G tile, X chunk) instead of once per tap: 3.3x fewer
// staging bytes than the generic kernel above, whose main loop is bound by the VGPR->LDS store path.
// Slabs have the generic kernel's [G channel][tap*Cx + c] layout and go through the same ordered reduce.
// ---------------------------------------------------------------------------------------------
struct PatchGeom {
  int B, H, W;              // output grid == input grid
  int Cg, g_cs, g_coff;
  int Cx, x_cs, x_coff;
  int Ktot;                 // 9 * Cx
  int strips_x;             // W / U
  int rows_per_blk, row_blks;
  int pairs_g, pairs_x;     // tiles of G channels / chunks of X channels
  const float* x_affine;    // optional [2][Cx] (scale | shift): X is read as [relu](x * scale + shift) while it is staged
  int x_relu;               // (the conv's input was never materialised: see conv.hip, in_affine); padding stays zero
};

template <typename T> struct WP;
template <> struct WP<float> { static constexpr int U = 32; };
template <> struct WP<__bf16> { static constexpr int U = 64; };

// XAFF: X is read through x_affine (+ ReLU) while it is staged -- a separate instantiation, so the plain kernel keeps
// its register allocation (with the transform compiled in, the bf16 kernel sits at the 256-VGPR limit: every launch
// measured 13 % slower, transform used or not)
template <typename T, bool XAFF>
__global__ __launch_bounds__(NT, 2) void wgrad_patch_kernel(const T* __restrict__ G, const T* __restrict__ X,
                                                           float* __restrict__ ws, PatchGeom g) {
  constexpr int EPC = WT<T>::EPC, U = WP<T>::U, TC = 64;      // 64-channel tiles on both sides
  constexpr int P = pitch_bytes(TC * sizeof(T));               // LDS bytes per pixel row
  constexpr int CPR = TC / EPC;                                // 16-byte chunks per pixel row
  constexpr int G_IT = U * CPR / NT, X_IT = ((U + 2) * CPR + NT - 1) / NT, RSTEP = NT / CPR;
  constexpr int GS_BYTES = U * P, XS_BYTES = (U + 2) * P;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  static_assert(U * CPR % NT == 0 && NT % CPR == 0, "tile");
  __shared__ __attribute__((aligned(16))) char Gs[2 * GS_BYTES];
  __shared__ __attribute__((aligned(16))) char Xs[4 * XS_BYTES];

  const int tid = threadIdx.x;
  const int npair = g.pairs_g * g.pairs_x;
  const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
  const int pair = lin % npair, unit = lin / npair;            // the pairs of one unit share its G / X rows in L2
  const int pg = pair % g.pairs_g, px = pair / g.pairs_g;
  const int rb = unit % g.row_blks, strip = unit / g.row_blks;
  const int sx = strip % g.strips_x, b = strip / g.strips_x;
  const int co0 = pg * TC, cx0 = px * TC, ox0 = sx * U;
  const int y0 = rb * g.rows_per_blk;
  const int y1 = (y0 + g.rows_per_blk < g.H) ? y0 + g.rows_per_blk : g.H;

  // descriptors: one image each; voffset = (row*W + col) * pitch + channel, anything outside -> OOB -> zeros
  const int g_pix = g.g_cs * (int)sizeof(T), x_pix = g.x_cs * (int)sizeof(T);
  const char* gbase = reinterpret_cast<const char*>(G) + ((size_t)b * g.H * g.W * g.g_cs + g.g_coff) * sizeof(T);
  const char* xbase = reinterpret_cast<const char*>(X) + ((size_t)b * g.H * g.W * g.x_cs + g.x_coff) * sizeof(T);
  const unsigned g_range = (unsigned)((long long)g.H * g.W * g_pix), x_range = (unsigned)((long long)g.H * g.W * x_pix);
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gbase), 0, g_range, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase), 0, x_range, 0x00020000);

  const int ch = tid % CPR, r0 = tid / CPR;                    // this thread's chunk column and first pixel
  const bool g_ok = co0 + ch * EPC < g.Cg, x_ok = cx0 + ch * EPC < g.Cx;
  unsigned gcol[G_IT], xcol[X_IT];                             // byte offset of (pixel in row, channel), or OOB
#pragma unroll
  for (int i = 0; i < G_IT; ++i)
    gcol[i] = g_ok ? (unsigned)((ox0 + r0 + i * RSTEP) * g_pix + (co0 + ch * EPC) * (int)sizeof(T)) : OOB;
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int j = r0 + i * RSTEP, ix = ox0 - 1 + j;            // LDS pixel j holds input column ox0 - 1 + j
    xcol[i] = (x_ok && j < U + 2 && (unsigned)ix < (unsigned)g.W)
                  ? (unsigned)(ix * x_pix + (cx0 + ch * EPC) * (int)sizeof(T)) : OOB;
  }
  const unsigned g_rowb = (unsigned)(g.W * g_pix), x_rowb = (unsigned)(g.W * x_pix);

  uint4 greg[2][G_IT], xreg[2][X_IT];
  float xsc[EPC], xsh[EPC];                                    // this thread's channel chunk never changes
  if (XAFF && x_ok) {
    const float* ap = g.x_affine + cx0 + ch * EPC;
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      const float4 a = *reinterpret_cast<const float4*>(ap + e), b2 = *reinterpret_cast<const float4*>(ap + g.Cx + e);
      xsc[e] = a.x; xsc[e + 1] = a.y; xsc[e + 2] = a.z; xsc[e + 3] = a.w;
      xsh[e] = b2.x; xsh[e + 1] = b2.y; xsh[e + 2] = b2.z; xsh[e + 3] = b2.w;
    }
  }
  auto load_g = [&](auto SET, int oy) {
    constexpr int set = decltype(SET)::value;
    const bool ok = oy < y1;                                   // uniform
#pragma unroll
    for (int i = 0; i < G_IT; ++i) {
      const unsigned off = (ok && gcol[i] != OOB) ? gcol[i] + (unsigned)oy * g_rowb : OOB;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off, 0, 0);
      greg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  auto load_x = [&](auto SET, int iy) {
    constexpr int set = decltype(SET)::value;
    const bool ok = (unsigned)iy < (unsigned)g.H;              // uniform: rows above / below the raster are zeros
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const unsigned off = (ok && xcol[i] != OOB) ? xcol[i] + (unsigned)iy * x_rowb : OOB;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
      xreg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  char* const g_st = Gs + r0 * P + ch * 16;
  char* const x_st = Xs + r0 * P + ch * 16;
  auto store_g = [&](auto SET, int buf) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < G_IT; ++i) *reinterpret_cast<uint4*>(g_st + buf * GS_BYTES + i * RSTEP * P) = greg[set][i];
  };
  auto store_x = [&](auto SET, int slot, int iy) {           // iy: the input row the register set holds
    constexpr int set = decltype(SET)::value;
    const bool rowok = (unsigned)iy < (unsigned)g.H, rl = g.x_relu != 0;
#pragma unroll
    for (int i = 0; i < X_IT; ++i)
      if ((X_IT * RSTEP <= U + 2) || r0 + i * RSTEP < U + 2) {
        uint4 v = xreg[set][i];
        if (XAFF) v = (rowok && xcol[i] != OOB) ? affine_relu16<T>(v, xsc, xsh, rl) : make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(x_st + slot * XS_BYTES + i * RSTEP * P) = v;
      }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;                     // 2 x 2 waves of 32 x 32 per tap
  const int lr = lane & 31, lh = lane >> 5;
  const int grp16 = (lane >> 4) & 1, li = lane & 15, q = li >> 2, pp = li & 3;
  const int g_ld = sizeof(T) == 4 ? lh * P + (wm * 32 + lr) * 4 : (8 * lh + q) * P + (wm * 32 + 16 * grp16 + 4 * pp) * 2;
  const int x_ld = sizeof(T) == 4 ? lh * P + (wn * 32 + lr) * 4 : (8 * lh + q) * P + (wn * 32 + 16 * grp16 + 4 * pp) * 2;

  // one output row: G buffer `gbuf`, X rows iy = oy-1, oy, oy+1 in ring slots (oy + ky) & 3
  auto compute = [&](int gbuf, int oy) {
    const char* gp = Gs + gbuf * GS_BYTES + g_ld;
    const char* xp[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) xp[ky] = Xs + ((oy + ky) & 3) * XS_BYTES + x_ld;
    if constexpr (sizeof(T) == 4) {
#pragma unroll 2
      for (int s2 = 0; s2 < U / 2; ++s2) {
        const float a = *reinterpret_cast<const float*>(gp + 2 * s2 * P);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float bv = *reinterpret_cast<const float*>(xp[ky] + (2 * s2 + kx) * P);
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[ky * 3 + kx], 0, 0, 0);
          }
      }
    } else {
      // 12 groups of (k-block, tap row): the fragments of group i+1 are requested before the three MFMAs of group i
      // are issued (order pinned below; left alone the scheduler puts each read right before its use)
      constexpr int NG = (U / 16) * 3;
      bf16x8 a[2], bfr[2][3];
      auto rd_tr = [](const char* p_) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p_));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p_ + 4 * P));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      auto rd = [&](int grp) {
        const int kb = grp / 3, ky = grp % 3;
        if (ky == 0) a[kb & 1] = rd_tr(gp + kb * 16 * P);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) bfr[grp & 1][kx] = rd_tr(xp[ky] + (kb * 16 + kx) * P);
      };
      rd(0);
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        if (grp + 1 < NG) rd(grp + 1);
        const int kb = grp / 3, ky = grp % 3;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc[ky * 3 + kx] = wg_mma(a[kb & 1], bfr[grp & 1][kx], acc[ky * 3 + kx]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        if (grp + 1 < NG) {
          if ((grp + 1) % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, WG_NM, 0);
      }
    }
  };

  // prologue: X rows y0-1 and y0 straight into the ring, then the two-deep register pipeline of
  // (G row oy, X row oy+1) pairs.  Steps are branch-free (rows past the range load as zeros, unused).
  load_x(S0{}, y0 - 1);
  load_x(S1{}, y0);
  store_x(S0{}, (y0 + 0) & 3, y0 - 1);
  store_x(S1{}, (y0 + 1) & 3, y0);
  load_g(S0{}, y0);
  load_x(S0{}, y0 + 1);
  load_g(S1{}, y0 + 1);
  load_x(S1{}, y0 + 2);
  store_g(S0{}, 0);
  store_x(S0{}, (y0 + 2) & 3, y0 + 1);
  __syncthreads();
  // step for row oy (stage parity CUR): loads of row oy+2 -> MFMAs of row oy -> row oy+1 regs -> LDS -> barrier
  auto step = [&](auto CUR, auto NXT, int oy) {
    constexpr int cur = decltype(CUR)::value;
    load_g(CUR, oy + 2);
    load_x(CUR, oy + 3);
    compute(cur, oy);
    store_g(NXT, cur ^ 1);
    store_x(NXT, (oy + 3) & 3, oy + 2);
    __syncthreads();
  };
  int oy = y0;
  for (; oy + 1 < y1; oy += 2) {
    step(S0{}, S1{}, oy);
    step(S1{}, S0{}, oy + 1);
  }
  if (oy < y1) step(S0{}, S1{}, oy);

  float* slab = ws + (size_t)unit * g.Cg * g.Ktot;
  const int c = cx0 + wn * 32 + lr;
  if (c < g.Cx) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (co < g.Cg) slab[(size_t)co * g.Ktot + t * g.Cx + c] = acc[t][e];
      }
  }
}

struct PatchPlan {
  bool ok;
  int rows_per_blk, row_blks, units;
};

// Units (strip x row range) per (G tile, X chunk) pair: one full wave of workgroups over the chip, but rows
// per workgroup >= 8 so the two halo rows stay a small part of the staging.
template <typename T>
PatchPlan make_patch_plan(int B, int H, int W, int Cg, int Cx) {
  PatchPlan p{};
  constexpr int U = WP<T>::U;
  if (W % U || Cg < 32 || Cx < 32) return p;
  const long long pairs = (long long)((Cg + 63) / 64) * ((Cx + 63) / 64);
  const long long strips = (long long)B * (W / U);
  // one workgroup per CU: alone on the chip 512 (two per CU) measured 5-10 % faster, but in the training step the kernel
  // runs on an auxiliary stream beside the data-gradient chain, and with 75 KB of LDS per workgroup a second one would
  // take the slot a conv workgroup (64.5 KB) can otherwise share the CU through (step: 64.8 -> 64.1 ms)
  static const int target = [] { const char* e = getenv("JSPSR_WGRAD_UNITS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
  long long want = (target + pairs * strips - 1) / (pairs * strips);   // row blocks per strip
  if (want < 1) want = 1;
  int rows = (int)((H + want - 1) / want);
  if (rows < 8) rows = H < 8 ? H : 8;
  p.rows_per_blk = rows;
  p.row_blks = (H + rows - 1) / rows;
  const long long units = strips * p.row_blks;
  if (units * pairs > 0x7fffffffLL || units > 0x7fffffffLL) return p;
  p.units = (int)units;
  p.ok = true;
  return p;
}

bool wgrad_patch_enabled() {
  static const bool on = [] { const char* e = getenv("JSPSR_WGRAD_NOPATCH"); return !(e && atoi(e)); }();
  return on;
}

// dW[r][c][ky][kx] (+)= sum_z ws[z][r][(ky*KW+kx)*Cx + c],  r < R, c < C  (fixed summation order).
// A (32 x 8)-thread workgroup owns 128 consecutive slab elements -- FOUR per lane, one 16-byte load per slab (round 4: the
// 4-byte loads of rounds 1-3 ran at 1.9 TB/s over ~3 GB of slabs per step); its 8 z-lanes each sum every 8th slab, then
// fold in a fixed order (the same order per element as before: same bits).  The scattered 4-byte writes into the
// (R, C, KH, KW) master layout are the small side.  Ktot is a multiple of 4 (Cx is a whole number of 16-byte chunks).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int R,
                                                          int C, int KH, int KW, int Cg, int Cx, int splits,
                                                          int accumulate) {
  __shared__ float4 red[8][32];
  const int Ktot = KH * KW * Cx;
  const long long total = (long long)R * Ktot;
  const size_t slab = (size_t)Cg * Ktot;
  const int ex = threadIdx.x & 31, zl = threadIdx.x >> 5;
  for (long long base = (long long)blockIdx.x * 128; base < total; base += (long long)gridDim.x * 128) {
    const long long idx = base + 4 * ex;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total)
      for (int z = zl; z < splits; z += 8) {   // slab rows of G channels < R are a prefix
        const float4 v = *reinterpret_cast<const float4*>(ws + z * slab + idx);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    red[zl][ex] = s;
    __syncthreads();
    if (zl == 0 && idx < total) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = red[q][ex];
        t[0] += v.x; t[1] += v.y; t[2] += v.z; t[3] += v.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long i = idx + j;
        const int r = (int)(i / Ktot), k = (int)(i % Ktot);
        const int tap = k / Cx, c = k % Cx;
        if (c < C) {
          const size_t dst = ((size_t)r * C + c) * (KH * KW) + tap;
          dW[dst] = accumulate ? dW[dst] + t[j] : t[j];
        }
      }
    }
    __syncthreads();
  }
}

struct Plan {
  int bmc, splits, m_per_split;
};

template <typename T>
Plan make_plan(long long M, int Cg, int Ktot) {
  Plan p;
  p.bmc = Cg > 64 ? 128 : (Cg > 32 ? 64 : 32);
  const int BKM = WT<T>::BKM;
  const long long tiles = (long long)((Cg + p.bmc - 1) / p.bmc) * ((Ktot + 127) / 128);
  long long splits = (768 + tiles - 1) / tiles;         // aim at ~3 workgroups per CU
  const long long max_splits = (M + 4 * BKM - 1) / (4 * BKM);  // >= 4 stages per slice
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  long long per = (M + splits - 1) / splits;
  per = (per + BKM - 1) / BKM * BKM;
  p.m_per_split = (int)per;
  p.splits = (int)((M + per - 1) / per);
  return p;
}

template <typename T, int BMC, int WGM, int WGN>
int launch_w(const void* G, const void* X, float* ws, const WgradGeom& g, int splits, hipStream_t s) {
  constexpr int BNC = 128, BKM = WT<T>::BKM;
  constexpr size_t lds = 2 * BKM * (pitch_bytes(BMC * sizeof(T)) + pitch_bytes(BNC * sizeof(T)));
  auto kern = wgrad_kernel<T, BMC, BNC, WGM, WGN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const long long nblk = (long long)((g.Cg + BMC - 1) / BMC) * ((g.Ktot + BNC - 1) / BNC) * splits;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, s, static_cast<const T*>(G), static_cast<const T*>(X), ws, g);
  return check_launch("conv2d_wgrad");
}

template <typename T>
int run(const void* G, const void* X, float* dW, int R, int C, float* ws, WgradGeom& g, int accumulate, hipStream_t s,
        const float* x_affine, int x_relu) {
  if (g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && g.IH == g.OH && g.IW == g.OW && wgrad_patch_enabled()) {
    const PatchPlan pp = make_patch_plan<T>(g.B, g.OH, g.OW, g.Cg, g.Cx);
    const long long img_g = (long long)g.OH * g.OW * g.g_cs * (long long)sizeof(T);
    const long long img_x = (long long)g.IH * g.IW * g.x_cs * (long long)sizeof(T);
    if (pp.ok && img_g < 0xF0000000LL && img_x < 0xF0000000LL) {
      PatchGeom q{};
      q.B = g.B; q.H = g.OH; q.W = g.OW;
      q.Cg = g.Cg; q.g_cs = g.g_cs; q.g_coff = g.g_coff;
      q.Cx = g.Cx; q.x_cs = g.x_cs; q.x_coff = g.x_coff;
      q.Ktot = g.Ktot; q.strips_x = g.OW / WP<T>::U;
      q.rows_per_blk = pp.rows_per_blk; q.row_blks = pp.row_blks;
      q.pairs_g = (g.Cg + 63) / 64; q.pairs_x = (g.Cx + 63) / 64;
      q.x_affine = x_affine; q.x_relu = x_relu;
      const long long nblk = (long long)pp.units * q.pairs_g * q.pairs_x;
      if (x_affine)
        hipLaunchKernelGGL((wgrad_patch_kernel<T, true>), dim3((unsigned)nblk), dim3(NT), 0, s, static_cast<const T*>(G),
                           static_cast<const T*>(X), ws, q);
      else
        hipLaunchKernelGGL((wgrad_patch_kernel<T, false>), dim3((unsigned)nblk), dim3(NT), 0, s, static_cast<const T*>(G),
                           static_cast<const T*>(X), ws, q);
      if (int e = check_launch("conv2d_wgrad_patch")) return e;
      const long long total = (long long)R * g.Ktot;
      const int blocks = (int)((total + 127) / 128 < 4096 ? (total + 127) / 128 : 4096);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, dW, R, C, g.KH, g.KW, g.Cg, g.Cx,
                         pp.units, accumulate);
      return check_launch("conv2d_wgrad_reduce");
    }
  }
  if (x_affine) return fail(JSPSR_EINVAL, "conv2d_wgrad: x_affine needs the nine-tap kernel (3x3, stride 1, pad 1): ask jspsr_conv2d_wgrad_x_affine_ok first");
  const Plan p = make_plan<T>(g.M, g.Cg, g.Ktot);
  g.m_per_split = p.m_per_split;
  // 32-bit buffer offsets: one slice of G, and the images of X one slice touches, must stay < 3.75 GiB
  const long long g_slice = (long long)p.m_per_split * g.g_cs * (long long)sizeof(T);
  const long long imgs = p.m_per_split / ((long long)g.OH * g.OW) + 2;
  const long long x_span = imgs * g.IH * g.IW * (long long)g.x_cs * (long long)sizeof(T);
  if (g_slice >= 0xF0000000LL || (x_span >= 0xF0000000LL && (long long)g.B * g.IH * g.IW * g.x_cs * (long long)sizeof(T) >= 0xF0000000LL))
    return fail(JSPSR_EINVAL, "conv2d_wgrad: tensor slice exceeds the 32-bit offset range of one launch");
  int e;
  if (p.bmc == 128) e = launch_w<T, 128, 2, 2>(G, X, ws, g, p.splits, s);
  else if (p.bmc == 64) e = launch_w<T, 64, 2, 2>(G, X, ws, g, p.splits, s);
  else e = launch_w<T, 32, 1, 4>(G, X, ws, g, p.splits, s);
  if (e) return e;
  const long long total = (long long)R * g.Ktot;
  const int blocks = (int)((total + 127) / 128 < 4096 ? (total + 127) / 128 : 4096);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, dW, R, C, g.KH, g.KW, g.Cg, g.Cx,
                     p.splits, accumulate);
  return check_launch("conv2d_wgrad_reduce");
}

}  // namespace

extern "C" size_t jspsr_conv2d_wgrad_workspace_bytes(int dtype, int B, int OH, int OW, int Cg, int Cx, int KH, int KW) {
  if (B <= 0 || OH <= 0 || OW <= 0 || Cg <= 0 || Cx <= 0 || KH <= 0 || KW <= 0) return 0;
  const long long M = (long long)B * OH * OW;
  const int Ktot = KH * KW * Cx;
  const Plan p = dtype == JSPSR_BF16 ? make_plan<__bf16>(M, Cg, Ktot) : make_plan<float>(M, Cg, Ktot);
  size_t splits = (size_t)p.splits;
  if (KH == 3 && KW == 3) {  // the 3x3 stride-1 path slices differently (the caller's stride is not known here)
    const PatchPlan pp = dtype == JSPSR_BF16 ? make_patch_plan<__bf16>(B, OH, OW, Cg, Cx) : make_patch_plan<float>(B, OH, OW, Cg, Cx);
    if (pp.ok && (size_t)pp.units > splits) splits = (size_t)pp.units;
  }
  return splits * Cg * Ktot * sizeof(float);
}

extern "C" int jspsr_conv2d_wgrad(int dtype, const void* G, int Cg, int g_cstride, int g_coff, const void* X, int Cx,
                                  int x_cstride, int x_coff, float* dW, int R, int C, int B, int OH, int OW,
                                  int IH, int IW, int KH, int KW, int stride, int pad, int accumulate,
                                  const float* x_affine, int x_relu, void* workspace, jspsr_stream_t stream) {
  if (dtype != JSPSR_F32 && dtype != JSPSR_BF16) return fail(JSPSR_EINVAL, "conv2d_wgrad: bad dtype");
  if (x_affine && !aligned16(x_affine)) return fail(JSPSR_EALIGN, "conv2d_wgrad: x_affine must be 16-byte aligned");
  if (!G || !X || !dW || !workspace) return fail(JSPSR_EINVAL, "conv2d_wgrad: null pointer");
  const int epc = dtype == JSPSR_F32 ? 4 : 8;
  if (Cg <= 0 || Cx <= 0 || Cg % epc || Cx % epc || g_cstride % epc || g_coff % epc || x_cstride % epc || x_coff % epc ||
      g_cstride < g_coff + Cg || x_cstride < x_coff + Cx)
    return fail(JSPSR_EINVAL, "conv2d_wgrad: channel counts/pitches/offsets must be multiples of %d", epc);
  if (R <= 0 || R > Cg || C <= 0 || C > Cx) return fail(JSPSR_EINVAL, "conv2d_wgrad: R/C exceed the padded channel counts");
  if (B <= 0 || OH <= 0 || OW <= 0 || IH <= 0 || IW <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    return fail(JSPSR_EINVAL, "conv2d_wgrad: bad geometry");
  if (!aligned16(G) || !aligned16(X) || !aligned16(workspace)) return fail(JSPSR_EALIGN, "conv2d_wgrad: pointers must be 16-byte aligned");
  WgradGeom g{};
  g.B = B; g.OH = OH; g.OW = OW; g.Cg = Cg; g.g_cs = g_cstride; g.g_coff = g_coff;
  g.IH = IH; g.IW = IW; g.Cx = Cx; g.x_cs = x_cstride; g.x_coff = x_coff;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Cx;
  g.M = (long long)B * OH * OW;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  return dtype == JSPSR_F32 ? run<float>(G, X, dW, R, C, ws, g, accumulate, s, x_affine, x_relu)
                            : run<__bf16>(G, X, dW, R, C, ws, g, accumulate, s, x_affine, x_relu);
}

extern "C" int jspsr_conv2d_wgrad_x_affine_ok(int dtype, int B, int OH, int OW, int Cg, int Cx, int KH, int KW, int stride, int pad) {
  if ((dtype != JSPSR_F32 && dtype != JSPSR_BF16) || KH != 3 || KW != 3 || stride != 1 || pad != 1 || !wgrad_patch_enabled()) return 0;
  const PatchPlan pp = dtype == JSPSR_BF16 ? make_patch_plan<__bf16>(B, OH, OW, Cg, Cx) : make_patch_plan<float>(B, OH, OW, Cg, Cx);
  const long long es = dtype == JSPSR_BF16 ? 2 : 4;
  return pp.ok && (long long)OH * OW * Cg * es < 0xF0000000LL && (long long)OH * OW * Cx * es < 0xF0000000LL;
}
