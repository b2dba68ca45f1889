// K2r -- 3x3, 64 -> 64 channel, unit-stride bf16 convolution with the WEIGHTS RESIDENT IN REGISTERS.
//
// The generator's and layer1's full-resolution convs (basics.py:39-47,111-117 at 64 channels; forward and data gradient)
// have K = 9 * 64 = 576 and N = 64: the whole weight matrix is 73.7 KB.  conv_patch_kernel streams it from L2 once per
// 256-pixel tile behind a barrier per tap and pays a cold patch load per tile (PMC, round 1: 28-34 % of the wave cycles
// parked on the stage barrier / weight vmcnt).  Here a PERSISTENT workgroup per CU (8 waves = 4 pixel groups x 2
// channel groups, each wave 64 pixels x 32 channels) loads its 32 weight rows ONCE into 144 VGPRs (36 MFMA
// fragments) and walks tiles of 16x16 pixels:
//   * the 18x18-pixel input patch of tile t+1 arrives by LDS-DMA (buffer_load_dwordx4 ... lds, 47 pieces of 7 pixels)
//     into the second patch buffer while tile t is multiplied -- no staging registers, no ds_write pass, no cold
//     prologue.  A piece costs its wave 100-200 issue cycles (MI355X_MICROARCH.md; measured here ~270 with 25
//     instructions of address arithmetic and EXEC masking around it, wherever it is placed), so a piece is three
//     instructions: the per-lane source offsets of interior tiles are tile-invariant registers;
//   * LDS image: pixel-major rows of 128 + 16 bytes -- conv_patch_kernel's conflict-free image (rotated tile rows).
//     LDS-DMA writes lane-linear (16 bytes per lane, lane order), so a piece is 7 pixels = 63 lanes x 16 bytes: every
//     ninth lane lands zeros in the pad, lane 63 writes the first 16 bytes of the NEXT piece's first pixel (the same
//     bytes that piece's lane 0 writes).  Pieces abut at 1008-byte steps and pixel P sits at 144 P exactly, so a tap is an
//     IMMEDIATE offset of the fragment read -- no address arithmetic in the tap loop;
//   * the tap loop is 72 MFMAs (v_mfma_f32_32x32x16_bf16) per wave with one patch-fragment read each and NO barrier;
//     one barrier per tile (patch t+1 landed / everyone is done with patch t-1);
//   * waves 4-7 run half a tile behind waves 0-3 (their SIMD partners): one wave of each SIMD is on the matrix pipe
//     while the other writes its previous tile out;
//   * two epilogues.  Without statistics (data gradient, plain and bias / scale convs) the MFMA operands are SWAPPED, so
//     a lane holds sixteen CHANNELS of one pixel: v_cvt_pk_bf16_f32 pairs them, v_permlane32_swap joins the two
//     half-waves' groups of four into 16-byte NHWC pieces, and the tile goes out from registers -- no LDS, ~40 vector
//     instructions (+ 32 fused multiply-adds against a 512-byte LDS table for a per-channel scale / bias).  With
//     BatchNorm statistics (training forward) a lane holds one channel of sixteen pixels, the per-channel sums are
//     in-lane adds, and the tile is transposed through a wave-private LDS scratch; the statistics of a tile are folded
//     across the four pixel groups two barriers later.  (A register form of the statistics -- a halving DPP /
//     ds_swizzle / v_permlane16_swap butterfly over the 32 lanes of a half-wave -- was built and measured 5-10 %
//     slower than the LDS form: 730-830 vs 845-905 TF/s.)
// Out-of-image patch pixels fail the buffer descriptor's range check and land as zeros (the conv's zero padding).
#include "conv_igemm.h"

#include <atomic>

#include <cstdlib>
#include <type_traits>

namespace {

using namespace jspsr;

constexpr int R_NTH = 512;                 // 8 waves
constexpr int R_PW = 18, R_NPIX = R_PW * R_PW;
constexpr int R_PITCH = 144;                              // LDS bytes per patch pixel: 128 + 16 (conflict-free ds_read_b128)
constexpr int R_PPP = 7;                                  // pixels per DMA piece: 7 x 144 = 1008 B = 63 lanes x 16 B
constexpr int R_PIECEB = R_PPP * R_PITCH;
constexpr int R_PIECES = (R_NPIX + R_PPP - 1) / R_PPP;    // 47 pieces per patch
// Who requests the pieces, and when.  Waves 0-3 ("early") multiply the tile in the first half of a period, waves 4-7 in
// the second (after writing their previous tile out).  Measured on one box (tools/lab/build_k2r_variants.sh o<order>e<n>):
//   plain kernel: early waves request AFTER their tap loop and epilogue, 6 pieces each: 1145 TF/s (before the loop: 980)
//   statistics kernel (long LDS epilogue): early waves request BEFORE their tap loop, 8 pieces each: 880 (after: 750)
#ifdef K2R_ORDER
constexpr int order_of(bool) { return K2R_ORDER; }
#else
constexpr int order_of(bool stats) { return stats ? 0 : 2; }
#endif
#ifdef K2R_EARLY
constexpr int early_pieces(bool) { return K2R_EARLY; }
#else
constexpr int early_pieces(bool stats) { return stats ? 8 : 6; }
#endif
constexpr int R_PATCHB = R_PIECES * R_PIECEB + 16;        // 47392 B per patch buffer (+16: lane 63 of the last piece)
constexpr int R_SCRP = 80;                                // scratch row pitch: 64 B of channels + 16
constexpr int R_SCRB = 64 * R_SCRP;                       // per wave
constexpr int R_OFF_SCR = 2 * R_PATCHB;
constexpr int R_OFF_RED = R_OFF_SCR + 8 * R_SCRB;         // [3][4][2][64] floats
constexpr int R_RUNB = 16;                                // two published run starts of the dynamic tile queue (+ pad), last bytes of the segment
constexpr int R_LDS_STATS = R_OFF_RED + 3 * 4 * 2 * 64 * 4 + R_RUNB;
constexpr int R_LDS_PLAIN = 2 * R_PATCHB + R_RUNB;
constexpr int R_OFF_PAR = 2 * R_PATCHB;                   // affine kernel: [2][64] floats, per-channel scale | bias
constexpr int R_LDS_AFFINE = R_OFF_PAR + 2 * 64 * 4 + R_RUNB;
constexpr int R_RUN = 4;                                  // tiles per claim of the dynamic queue (x-neighbours: incremental coordinates)
#ifndef K2R_LA
#define K2R_LA 4      // patch fragments requested ahead of their MFMA (lab builds: -DK2R_LA=n)
#endif
constexpr int R_ROT = 14;                                 // see conv_patch_kernel: row r of the tile is rotated by 14 r
constexpr int NK = 72, LA = K2R_LA;                       // MFMA steps per tile; look-ahead

struct Tile { int bimg, tyi, txi; };
struct PatchSrc { i32x4 desc; int oy0, ox0; bool interior; };

template <int SIGN, int MODE>      // MODE 0: plain, 1: + BatchNorm statistics, 2: + per-channel scale / bias
__global__ __launch_bounds__(R_NTH) void conv64_resident_kernel(const __bf16* __restrict__ in, const __bf16* __restrict__ wgt,
                                                                  const float* __restrict__ bias, __bf16* __restrict__ out,
                                                                  float* __restrict__ stats, ConvGeom g, int ntiles,
                                                                  unsigned* __restrict__ ticket) {
  constexpr bool STATS = MODE == 1, AFFINE = MODE == 2;
  constexpr int LDSB = MODE == 1 ? R_LDS_STATS : (MODE == 2 ? R_LDS_AFFINE : R_LDS_PLAIN);
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = (wave >> 1) & 3, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int ttx = (g.MW + 15) >> 4, tty = (g.MH + 15) >> 4;
  const int G = gridDim.x;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;      // LDS byte address of the dynamic segment
  const int v = xcd_contiguous(blockIdx.x, G);
  const bool late = wave >= 4;
  constexpr int ORDER = order_of(STATS), R_PE = early_pieces(STATS), R_PL = (R_PIECES - 4 * R_PE + 3) / 4;
  constexpr int R_PIT = R_PE > R_PL ? R_PE : R_PL;      // pieces per wave: waves 0-3 R_PE each, waves 4-7 share the rest

  // ---- weights: 36 fragments (9 taps x 4 sub-steps) of this wave's 32 output channels, once ------------------------
  bf16x8 breg[9][4];
  {
    const char* wrow = reinterpret_cast<const char*>(wgt) + (size_t)(wn * 32 + lr) * (9 * 64 * 2) + lh * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) breg[t][s] = *reinterpret_cast<const bf16x8*>(wrow + t * 128 + s * 32);
  }

  // Tile coordinates advance by G tiles per period: carried incrementally (a few scalar adds) instead of two divisions
  // per use -- the scalar unit shares this wave's issue slot with everything else it does.
  const int dGx = G % ttx, dGy = (G / ttx) % tty, dGb = G / (ttx * tty);
  auto advance = [&](Tile c) {
    c.txi += dGx;
    int carry = c.txi >= ttx;
    c.txi -= carry ? ttx : 0;
    c.tyi += dGy + carry;
    carry = c.tyi >= tty;
    c.tyi -= carry ? tty : 0;
    c.bimg += dGb + carry;
    return c;
  };

  // ---- patch DMA: lane l of piece i fills position l % 9 of pixel 7 i + l / 9 (position 8 = the 16-byte pad) --------
  const int pix_bytes = g.in_cstride * 2;
  const int lpix = lane / 9, lp16 = (lane - 9 * lpix) * 16;      // lane 63: pixel 7 (the next piece's first), position 0
  const int back = SIGN < 0 ? 2 : 0;     // reversed walk: the patch starts two pixels earlier
  // source offset of this lane in its i-th piece, from the patch origin: tile-invariant (pad lanes and pixels beyond the
  // patch: an offset that fails the descriptor's range check -> zeros)
  // wave w < 4 owns pieces w + 4 i (i < R_PE), wave w >= 4 pieces 4 R_PE + (w - 4) + 4 i (i < R_PL)
  const int piece0 = late ? 4 * R_PE + (wave - 4) : wave, npiece = late ? R_PL : R_PE;
  unsigned dsrc[R_PIT];
#pragma unroll
  for (int i = 0; i < R_PIT; ++i) {
    const int P = (piece0 + 4 * i) * R_PPP + lpix;
    const int py = P / R_PW, px = P - py * R_PW;
    dsrc[i] = (lp16 < 128 && P < R_NPIX) ? (unsigned)((py * g.IW + px) * pix_bytes + lp16) : 0xFFFFFFF0u;
  }
  auto patch_src = [&](Tile tc) {
    PatchSrc p;
    p.oy0 = tc.tyi * 16 + g.iy_add - back;
    p.ox0 = tc.txi * 16 + g.ix_add - back;
    const long long opix0 = ((long long)tc.bimg * g.IH + p.oy0) * g.IW + p.ox0;     // may lie outside the raster
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(in) + (opix0 * g.in_cstride + g.in_coff) * 2;
    p.desc = i32x4{(int)(unsigned)b64, (int)((unsigned)(b64 >> 32) & 0xffffu), (int)0xFFFFFF00u, 0x00020000};
    p.interior = p.oy0 >= 0 && p.ox0 >= 0 && p.oy0 + R_PW <= g.IH && p.ox0 + R_PW <= g.IW;    // wave-uniform
    return p;
  };
  // This wave's pieces of one patch.  A lane whose pixel lies outside the image gets an offset that fails the descriptor's
  // range check: `buffer_load ... lds` then writes ZEROS for it (tools/lab/lds_dma_oob.hip) -- the conv's zero padding;
  // only tiles on the image border pay for that test.  Issued from inline asm (no memory clobber: the bytes land in the
  // OTHER patch buffer, read after the next barrier): hipcc would order every later LDS read of this wave behind a
  // builtin LDS-DMA with vmcnt(0), and behind a clobbering statement with lgkmcnt(0); the waits that matter are placed
  // by hand.  M0 is written in the statement that uses it (nothing else in this kernel uses M0).
  auto issue_patch = [&](const PatchSrc& p, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < R_PIT; ++i) {
      if (i < npiece && piece0 + 4 * i < R_PIECES) {     // wave-uniform
        unsigned src = dsrc[i];
        if (!p.interior) {
          int lpix_ = lpix;
          asm volatile("" : "+v"(lpix_));                       // border tiles only: recomputed, not kept in registers
          const int P = (piece0 + 4 * i) * R_PPP + lpix_;
          const int py = (int)(__umul24(P, 3641) >> 16), px = P - (int)__umul24(py, R_PW);        // P / 18 for P < 400
          const bool in_img = (unsigned)(p.oy0 + py) < (unsigned)g.IH && (unsigned)(p.ox0 + px) < (unsigned)g.IW;
          src = in_img ? src : 0xFFFFFFF0u;
        }
        const unsigned dst = lds0 + buf * R_PATCHB + (piece0 + 4 * i) * R_PIECEB;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(src), "s"(p.desc));
      }
    }
  };

  // ---- patch-fragment plan ---------------------------------------------------------------------------------------
  int a_base[2];      // LDS byte offset of this lane's fragment at tap 0 (reversed walk: at the LAST tap), sub-step 0
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int R = wm * 64 + mi * 32 + lr, dy = R >> 4;
    a_base[mi] = (dy * R_PW + ((R + R_ROT * dy) & 15)) * R_PITCH + lh * 16;
  }

  // ---- the tap loop: 9 taps x 4 sub-steps x 2 pixel blocks, weights from registers ------------------------------------
  f32x16 acc[2];
  auto mfma_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
    const char* const pa0 = smem + buf * R_PATCHB + a_base[0];
    const char* const pa1 = smem + buf * R_PATCHB + a_base[1];
    bf16x8 a[LA];
#pragma unroll
    for (int k = 0; k < NK + LA; ++k) {
      if (k >= LA) {                     // consumes ring slot k % LA before the read below refills it
        const int kk = k - LA, tap = kk >> 3, s = (kk >> 1) & 3, mi = kk & 1;
        if (STATS) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk % LA], breg[tap][s], acc[mi], 0, 0, 0);   // rows = pixels
        else       acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[tap][s], a[kk % LA], acc[mi], 0, 0, 0);   // rows = channels
      }
      if (k < NK) {
        const int tap = k >> 3, s = (k >> 1) & 3, mi = k & 1;
        const int tp = (tap / 3) * R_PW + tap % 3;                       // patch pixel offset of the tap
        a[k % LA] = *reinterpret_cast<const bf16x8*>((mi ? pa1 : pa0) + (SIGN > 0 ? tp : 2 * R_PW + 2 - tp) * R_PITCH + s * 32);
      }
    }
    // pin the order: LA reads up front, then one read behind every MFMA (left alone, the scheduler sinks each read to
    // just before its use and the MFMA waits out the LDS latency)
    __builtin_amdgcn_sched_group_barrier(0x100, LA, 0);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (k + LA < NK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };

  // ---- epilogue constants ---------------------------------------------------------------------------------------
  const bool relu_first = g.relu && !g.addend, relu_last = g.relu && g.addend;
  char* const scr = smem + R_OFF_SCR + wave * R_SCRB;
  float* const red = reinterpret_cast<float*>(smem + R_OFF_RED);
  const int ncol = wn * 32 + lr;
  const int tty8 = (g.MH + 7) >> 3;

  auto flush_stats = [&](Tile tc, int par) __attribute__((always_inline)) {
    // rows of the statistics buffer are numbered by 8x16-pixel tiles (jspsr_conv2d_stats_rows): pixel groups 0,1 are
    // the upper half of this 16x16 tile, 2,3 the lower
    if (tid < 256) {
      const int half = tid >> 7, which = (tid >> 6) & 1, col = tid & 63;
      const int row = 2 * tc.tyi + half;
      if (row < tty8) {
        const float* r0 = red + ((par * 4 + 2 * half) * 2 + which) * 64 + col;
        stats[((size_t)((tc.bimg * tty8 + row) * ttx + tc.txi) * 2 + which) * 64 + col] = r0[0] + r0[2 * 64];
      }
    }
  };

  // 16-byte NHWC pieces go out through buffer descriptors over the batch image's slice: 32-bit offsets, and a pixel
  // outside the written raster gets an offset that fails the range check (the store / addend load is dropped)
  auto out_desc = [&](int bimg) {
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(out) + ((long long)bimg * g.OH * g.OW * g.out_cstride + g.out_coff + wn * 32) * 2;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(b64), 0, 0xFFFFFF00u, 0x00020000);
  };
  auto add_desc = [&](int bimg) {
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(g.addend) + ((long long)bimg * g.OH * g.OW * g.add_cstride + wn * 32) * 2;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(b64), 0, g.addend ? 0xFFFFFF00u : 0u, 0x00020000);
  };

  // Statistics form: a lane holds channel `ncol` of 2 x 16 pixels (rows of the MFMA result)
  auto epilogue_stats = [&](Tile tc, int par) __attribute__((always_inline)) {
    const int ty0 = tc.tyi * 16, tx0 = tc.txi * 16;
    const bool whole = ty0 + 16 <= g.MH && tx0 + 16 <= g.MW;      // wave-uniform: no pixel of the tile is outside
    if (stats) {
      float s = 0.f, q = 0.f;
      if (whole) {
        f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};              // two rows per v_pk_add_f32 / v_pk_fma_f32
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const f32x2 a2 = {acc[mi][e], acc[mi][e + 1]};
            s2 += a2;
            q2 = __builtin_elementwise_fma(a2, a2, q2);
          }
        s = s2[0] + s2[1];
        q = q2[0] + q2[1];
      } else {
        int lh4 = 4 * lh;
        asm volatile("" : "+v"(lh4));                       // keep the 32 row coordinates out of the loop-invariant registers
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int R = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + lh4, dy = R >> 4;
            const int y = ty0 + dy, x = tx0 + ((R + R_ROT * dy) & 15);
            if (y < g.MH && x < g.MW) {
              const float a0 = acc[mi][e];
              s += a0;
              q += a0 * a0;
            }
          }
      }
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lh == 0) {
        red[((par * 4 + wm) * 2 + 0) * 64 + ncol] = s;
        red[((par * 4 + wm) * 2 + 1) * 64 + ncol] = q;
      }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float o = acc[mi][e];
        *reinterpret_cast<__bf16*>(scr + row * R_SCRP + lr * 2) = (__bf16)o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private scratch: program order is enough
    const int c16 = lane & 3, prow = lane >> 2;
    const __amdgpu_buffer_rsrc_t orsrc = out_desc(tc.bimg), adrsrc = add_desc(tc.bimg);
    // requests first (addend pieces, then the scratch rows), stores last: a load's result is never awaited with a
    // store of this wave in flight
    unsigned ooff[4], aoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int dy = wm * 4 + j;                              // tile row of scratch rows 16 j .. 16 j + 15
      const int y = ty0 + dy, x = tx0 + ((prow + R_ROT * dy) & 15);
      const bool inside = whole || (y < g.MH && x < g.MW);
      const unsigned pix = (unsigned)(y * g.OW + x);
      ooff[j] = inside ? pix * (unsigned)(g.out_cstride * 2) + c16 * 16u : 0xFFFFFFF0u;
      aoff[j] = inside ? pix * (unsigned)(g.add_cstride * 2) + c16 * 16u : 0xFFFFFFF0u;
    }
    if (g.addend) {
      u32x4 av[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) av[j] = __builtin_amdgcn_raw_buffer_load_b128(adrsrc, aoff[j], 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 o = *reinterpret_cast<const u32x4*>(scr + (j * 16 + prow) * R_SCRP + c16 * 16);
        __builtin_amdgcn_raw_buffer_store_b128(add_bf16x8(o, av[j], relu_last), orsrc, ooff[j], 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 o = *reinterpret_cast<const u32x4*>(scr + (j * 16 + prow) * R_SCRP + c16 * 16);
        __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, ooff[j], 0, 0);
      }
    }
  };

  // Epilogue (MFMA operands swapped: rows = channels): lane (lr, lh) holds, of pixel 32 mi + lr of this wave's 64, the channels
  // (e & 3) + 8 (e >> 2) + 4 lh of this wave's 32.  Groups of four go to bf16 pairs; v_permlane32_swap hands the lower
  // half-wave the upper one's group 2 j (channels 8 j + 4 .. 8 j + 7) in exchange for its own group 2 j + 1: the lower
  // half-wave then stores channels 0-7 and 16-23 of its pixel, the upper one 8-15 and 24-31, 16 bytes each.
  auto epilogue_regs = [&](Tile tc) __attribute__((always_inline)) {
    const int ty0 = tc.tyi * 16, tx0 = tc.txi * 16;
    const bool whole = ty0 + 16 <= g.MH && tx0 + 16 <= g.MW;
    const __amdgpu_buffer_rsrc_t orsrc = out_desc(tc.bimg), adrsrc = add_desc(tc.bimg);
    unsigned ooff[2], aoff[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int R = wm * 64 + mi * 32 + lr, dy = R >> 4;
      const int y = ty0 + dy, x = tx0 + ((R + R_ROT * dy) & 15);
      const bool inside = whole || (y < g.MH && x < g.MW);
      const unsigned pix = (unsigned)(y * g.OW + x);
      ooff[mi] = inside ? pix * (unsigned)(g.out_cstride * 2) + lh * 16u : 0xFFFFFFF0u;
      aoff[mi] = inside ? pix * (unsigned)(g.add_cstride * 2) + lh * 16u : 0xFFFFFFF0u;
    }
    u32x4 av[2][2];
    if (g.addend) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int h = 0; h < 2; ++h) av[mi][h] = __builtin_amdgcn_raw_buffer_load_b128(adrsrc, aoff[mi], h * 32, 0);
    }
    if constexpr (AFFINE) {      // per-channel scale / bias of this lane's sixteen channels, from the table the prologue put in LDS
      const float* tab = reinterpret_cast<const float*>(smem + R_OFF_PAR) + wn * 32 + 4 * lh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4v s4 = *reinterpret_cast<const f32x4v*>(tab + 8 * q), b4 = *reinterpret_cast<const f32x4v*>(tab + 64 + 8 * q);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[mi][4 * q + i] = acc[mi][4 * q + i] * s4[i] + b4[i];
      }
    }
    if (relu_first) {      // wave-uniform
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][e] = relu_bits(acc[mi][e]);
    }
    unsigned d[2][8];      // bf16 pairs: d[mi][2 q], d[mi][2 q + 1] = group q (four channels) of pixel block mi
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; e += 2) d[mi][e >> 1] = pack_bf16(acc[mi][e], acc[mi][e + 1]);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      // swap (group 0 | 1) and (group 2 | 3) between the half-waves
#pragma unroll
      for (int q = 0; q < 4; q += 2)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const auto r = __builtin_amdgcn_permlane32_swap(d[mi][2 * q + w], d[mi][2 * q + 2 + w], false, false);
          d[mi][2 * q + w] = r[0];
          d[mi][2 * q + 2 + w] = r[1];
        }
      // now this lane holds [d0 d1 d2 d3] = channels 8 lh + 0..7 and [d4 d5 d6 d7] = channels 16 + 8 lh + 0..7
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 o = {d[mi][4 * h], d[mi][4 * h + 1], d[mi][4 * h + 2], d[mi][4 * h + 3]};
        if (g.addend) o = add_bf16x8(o, av[mi][h], relu_last);
        __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, ooff[mi], h * 32, 0);
      }
    }
  };
  auto epilogue = [&](Tile tc, int par) __attribute__((always_inline)) {
    if constexpr (STATS) epilogue_stats(tc, par); else epilogue_regs(tc);
  };

  // Waves w and w + 4 share a SIMD.  Run in lockstep they would both be in their MFMA phase, then both in their
  // epilogue, and the matrix pipe would idle through every epilogue.  So waves 4-7 run HALF A TILE BEHIND: in the period
  // of tile t (between two barriers) waves 0-3 multiply tile t (and request tile t + G's patch) and then write tile t
  // out, while waves 4-7 first write out their part of tile t - G (accumulators kept across the barrier) and then
  // multiply tile t.  Statistics of a tile are therefore complete one period late: three parities of the fold buffer,
  // folded two periods after the tile.
  if constexpr (AFFINE) {      // absent scale -> 1, absent bias -> 0; visible to every wave after the first barrier of the loop
    if (tid < 128) reinterpret_cast<float*>(smem + R_OFF_PAR)[tid] = tid < 64 ? (g.scale ? g.scale[tid] : 1.f) : (bias ? bias[tid - 64] : 0.f);
  }
  // Tile walk.  Static (ticket == nullptr): tile v, v + G, v + 2G ... -- fine alone on the chip.  Dynamic: the workgroups
  // draw runs of R_RUN x-neighbouring tiles from one global ticket, so a workgroup that got its CU late (a weight-gradient
  // workgroup of another stream held the LDS) simply draws fewer runs instead of making the launch wait for its share.
  // The draw is two tiles of latency away from its use: lane 0 issues the atomic when the walk ENTERS a run (for the
  // run after the next), publishes the returned start in LDS one tile later (behind the vmcnt(0) this wave waits on
  // anyway), and the barrier of the tile after that makes it visible -- R_RUN >= 3 tiles before it is needed.
  const bool dyn = ticket != nullptr;
  int* const s_run = reinterpret_cast<int*>(smem + LDSB - R_RUNB);
  if (dyn) {
    if (tid == 0) {
      s_run[0] = (int)__hip_atomic_fetch_add(ticket, (unsigned)R_RUN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_run[1] = (int)__hip_atomic_fetch_add(ticket, (unsigned)R_RUN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  auto tile_of = [&](int t_) { return Tile{(t_ / ttx) / tty, (t_ / ttx) % tty, t_ % ttx}; };
  int t = dyn ? s_run[0] : v, it = 0;
  int krun = 0, nrun = 0;                 // dynamic: position of tile t in its run, number of that run
  int tn, kn, nn;                         // the same for the NEXT tile (tn >= ntiles: none)
  unsigned pend = 0;                      // lane 0: a drawn run start not yet published
  int pend_slot = -1;
  auto step_walk = [&](int t_, int k_, int n_, Tile c_, int& t2, int& k2, int& n2, Tile& c2) {
    if (!dyn) {
      t2 = t_ + G; k2 = 0; n2 = 0; c2 = advance(c_);
    } else if (k_ + 1 < R_RUN && t_ + 1 < ntiles) {
      t2 = t_ + 1; k2 = k_ + 1; n2 = n_;
      c2 = c_;
      if (++c2.txi == ttx) { c2.txi = 0; if (++c2.tyi == tty) { c2.tyi = 0; ++c2.bimg; } }
    } else {
      n2 = n_ + 1; k2 = 0;
      t2 = s_run[n2 & 1];                 // published at least one barrier ago
      c2 = tile_of(t2 < ntiles ? t2 : 0);
      if (tid == 0) {                     // entering run n2: draw run n2 + 1 into the slot run n2 - 1 no longer needs
        pend = __hip_atomic_fetch_add(ticket, (unsigned)R_RUN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_slot = (n2 + 1) & 1;
      }
    }
  };
  Tile tcur = tile_of(t < ntiles ? t : 0), tprev = tcur, tprev2 = tcur, tnext;
  step_walk(t, krun, nrun, tcur, tn, kn, nn, tnext);
  if (t < ntiles) {
    const PatchSrc p0 = patch_src(tcur);
    issue_patch(p0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef K2R_STAMPS      // lab build: cycles per phase, printed by waves 0 and 4 of workgroup 0 (tools/lab/build_k2r_variants.sh)
  unsigned long long st_bar = 0, st_pre = 0, st_mfma = 0, st_vm = 0, st_post = 0, s0, s1;
#define K2R_STAMP(acc_) do { s1 = __builtin_readcyclecounter(); acc_ += s1 - s0; s0 = s1; } while (0)
#else
#define K2R_STAMP(acc_) do { } while (0)
#endif
  for (; t < ntiles; ++it) {
    const int buf = it & 1;
#ifdef K2R_STAMPS
    s0 = __builtin_readcyclecounter();
#endif
    // every wave has waited for its own DMA pieces after its MFMA phase (below): patch `buf` is complete once all have
    // arrived; nobody reads patch `buf ^ 1` any more.  The raw barrier does not wait for the stores of the epilogue.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    K2R_STAMP(st_bar);
    if (STATS && stats && it > 1) flush_stats(tprev2, (it - 2) % 3);
    // Waves 0-3 request their pieces of the next patch before multiplying, waves 4-7 after writing their previous tile
    // out: either way a whole tap loop lies between the request and the wait.
    const bool more = tn < ntiles;
    const PatchSrc nxt = patch_src(tnext);
    if constexpr (ORDER == 0) {
    if (late) {
      if (it > 0) epilogue(tprev, (it - 1) % 3);
      if (more) issue_patch(nxt, buf ^ 1);
    } else if (more) {
      issue_patch(nxt, buf ^ 1);
    }
    K2R_STAMP(st_pre);
    mfma_tile(buf);
    K2R_STAMP(st_mfma);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the pieces requested above (long landed); no store is younger
    K2R_STAMP(st_vm);
    if (!late) epilogue(tcur, it % 3);
    } else {
    // waves 0-3 multiply FIRST (their SIMD partners are in their epilogue), then request their pieces and write out
    if (late) {
      if (it > 0) epilogue(tprev, (it - 1) % 3);
      if (more) issue_patch(nxt, buf ^ 1);
    }
    K2R_STAMP(st_pre);
    mfma_tile(buf);
    K2R_STAMP(st_mfma);
    if (!late) {
      if (ORDER == 1 && more) issue_patch(nxt, buf ^ 1);
      epilogue(tcur, it % 3);
      if (ORDER == 2 && more) issue_patch(nxt, buf ^ 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces (and, in waves 0-3, its stores)
    K2R_STAMP(st_vm);
    }
    // (every vmcnt(0) above has passed: the drawn run start has returned) publish it, then move on one tile
    if (dyn && tid == 0 && pend_slot >= 0 && pend_slot < 2) { s_run[pend_slot] = (int)pend; pend_slot += 2; }   // (+2: published; the slot number stays readable)
    tprev2 = tprev; tprev = tcur; tcur = tnext;
    t = tn; krun = kn; nrun = nn;
    if (t < ntiles) step_walk(t, krun, nrun, tcur, tn, kn, nn, tnext);
    K2R_STAMP(st_post);
  }
#ifdef K2R_STAMPS
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4))
    printf("K2R wave %d tiles %d: barrier %llu  pre-mfma %llu  mfma %llu  vmcnt %llu  post-mfma %llu (cycles per tile)\n", wave, it,
           st_bar / it, st_pre / it, st_mfma / it, st_vm / it, st_post / it);
#endif
  if (dyn && tid == 0) {
    // last workgroup out re-arms the ticket for the launch that gets this slot next (1024 launches from now).  Every
    // draw of this workgroup has returned before it signs off (a straggling add after the reset would cost the next
    // user its first run).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)G - 1u) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (late && it > 0) epilogue(tprev, (it - 1) % 3);
  if (STATS && stats && it > 0) {
    __syncthreads();
    if (it > 1) flush_stats(tprev2, (it - 2) % 3);
    flush_stats(tprev, (it - 1) % 3);
  }
}

// Tickets of the dynamic tile queue: (next tile, workgroups done) pairs, handed out round-robin per launch; the last
// workgroup of a launch zeroes its pair again, and a slot comes round after 1024 launches.
__device__ unsigned k2r_ring[2 * 1024];

unsigned* next_ticket() {
  static unsigned* base = [] {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(k2r_ring)) == hipSuccess ? static_cast<unsigned*>(p) : nullptr;
  }();
  static std::atomic<unsigned> n{0};
  return base ? base + 2 * (n.fetch_add(1, std::memory_order_relaxed) % 1024u) : nullptr;
}

template <int SIGN, int MODE>
void launch_k2r(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g, int ntiles, int grid,
                hipStream_t s) {
  // dynamic tile queue (opt-in: JSPSR_CONV_DYNQ=1), once every workgroup has several runs to draw.  Measured in the
  // multi-stream step on one box, three interleaved runs each: 65.9 / 64.2 / 65.5 ms static, 68.0 / 66.3 / 65.4 ms dynamic
  // (profiles/r03_k2r_dynamic_queue_ab.txt) -- the late workgroups the queue relieves were not what the step waits for.
  static const int dynq = [] { const char* e = getenv("JSPSR_CONV_DYNQ"); return e ? atoi(e) : 0; }();
  const int forced = conv_dynq_override();      // jspsr_conv_dynamic_queue(): data-parallel runs switch the queue on (RCCL holds CUs beside the backward pass)
  unsigned* ticket = ((forced >= 0 ? forced : dynq) && (long long)ntiles >= 4LL * R_RUN * grid) ? next_ticket() : nullptr;
  constexpr int lds = MODE == 1 ? R_LDS_STATS : (MODE == 2 ? R_LDS_AFFINE : R_LDS_PLAIN);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_resident_kernel<SIGN, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv64_resident_kernel<SIGN, MODE>), dim3(grid), dim3(R_NTH), lds, s, static_cast<const __bf16*>(in),
                     static_cast<const __bf16*>(wgt), bias, static_cast<__bf16*>(out), stats, g, ntiles, ticket);
}

}  // namespace

namespace jspsr {

bool conv64_resident_ok(const ConvGeom& g, const void* in, const void* wgt, const float* bias, const void* out) {
  static const int enabled = [] { const char* e = getenv("JSPSR_CONV_RESIDENT"); return e ? atoi(e) : 1; }();
  if (!enabled) return false;
  if (g.Cin != 64 || g.Cout != 64 || g.nty != 3 || g.ntx != 3 || g.KH != 3 || g.KW != 3) return false;
  if (g.iy_mul != 1 || g.ix_mul != 1 || g.oy_mul != 1 || g.ox_mul != 1 || g.oy_add != 0 || g.ox_add != 0) return false;
  if (g.ky0 != 0 || g.kx0 != 0 || g.kstep != 1 || g.in_affine || g.red_out) return false;
  if (g.in_cstride % 8 || g.in_coff % 8 || g.out_cstride % 8 || g.out_coff % 8) return false;
  if (!aligned16(in) || !aligned16(wgt) || !aligned16(out)) return false;
  if (g.addend && (!aligned16(g.addend) || g.add_cstride % 8)) return false;
  if ((long long)(g.IW + 20) * 20 * g.in_cstride * 2 >= 0x7fffffffLL) return false;    // 32-bit offsets inside a patch
  if ((long long)g.OH * g.OW * g.out_cstride * 2 >= 0xF0000000LL || (long long)g.OH * g.OW * g.add_cstride * 2 >= 0xF0000000LL)
    return false;                                                                       // ... and inside one image of the result
  const long long tiles = (long long)g.B * ((g.MH + 15) / 16) * ((g.MW + 15) / 16);
  static const int min_tiles = [] { const char* e = getenv("JSPSR_CONV_RESIDENT_MIN"); return e ? atoi(e) : 512; }();      // measured: ties the patch kernel at 256 tiles, +15-25 % at 512, +50 % at 8192
  return tiles >= min_tiles && tiles < 0x7fffffffLL;
}

int launch_conv64_resident(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g,
                           hipStream_t s) {
  const int ntiles = g.B * ((g.MH + 15) / 16) * ((g.MW + 15) / 16);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return fail((int)hipErrorInvalidDevice, "conv: device query failed");
    ncu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  const int grid = ntiles < ncu ? ntiles : ncu;
  const int mode = stats ? 1 : ((bias || g.scale) ? 2 : 0);      // the C ABI refuses statistics together with bias / scale
  if (g.sign > 0) {
    if (mode == 0) launch_k2r<1, 0>(in, wgt, bias, out, stats, g, ntiles, grid, s);
    else if (mode == 1) launch_k2r<1, 1>(in, wgt, bias, out, stats, g, ntiles, grid, s);
    else launch_k2r<1, 2>(in, wgt, bias, out, stats, g, ntiles, grid, s);
  } else {
    if (mode == 0) launch_k2r<-1, 0>(in, wgt, bias, out, stats, g, ntiles, grid, s);
    else if (mode == 1) launch_k2r<-1, 1>(in, wgt, bias, out, stats, g, ntiles, grid, s);
    else launch_k2r<-1, 2>(in, wgt, bias, out, stats, g, ntiles, grid, s);
  }
  return check_launch("conv64_resident");
}

}  // namespace jspsr
