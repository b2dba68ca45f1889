// K2q -- 3x3, 128 -> 128 channel, unit-stride bf16 convolution / data gradient with the WHOLE weight matrix resident in
// the register files of one CU, and every patch fragment read from LDS feeding FOUR matrix products.
//
// The 128-channel BasicBlocks of the encoder (basics.py:88-123 at 2 x nf ... at full and half resolution; forward and data
// gradient) have K = 9 * 128 = 1152 and N = 128: the weight matrix is 295 KB, more than one wave's registers but not more
// than a CU's (4 SIMDs x 512 registers x 64 lanes x 4 B = 512 KB).  conv_patch_kernel streams it through LDS once per
// 256-pixel tile: one 1 KB fragment read per MFMA plus the weight staging writes and a barrier per stage -- the chip then
// holds ~1.65 GHz under that load (profiles/r03_conv_patch_wave_life.txt) and the layer sits at 0.31-0.37 of the MFMA
// peak.  What lowers the energy per flop is fewer LDS bytes and fewer instructions per MFMA, so:
//   * ONE workgroup of FOUR waves per CU, one wave per SIMD with the full 512-register budget (256 VGPR + 256 AGPR).
//     Wave (cg, kh) owns 64 output channels (cg) x one half of the input channels (kh): 9 taps x 2 k-steps x 4 channel
//     blocks = 72 v_mfma_f32_16x16x32_bf16 A-operands = 288 registers, loaded ONCE per launch;
//   * tiles of 8 x 16 pixels; every wave multiplies ALL 128 pixels of the tile by its slice: one ds_read_b128 of a
//     16-pixel x 32-channel patch fragment feeds 4 MFMAs (the wave's 4 channel blocks) -- a quarter of the LDS read bytes
//     per flop of the patch kernel, no weight bytes through LDS at all, no stage barriers;
//   * the two K-halves of a channel group meet once per tile, in the MIDDLE of the tap loop: a wave first multiplies the 4 tile
//     rows its partner finishes, writes their partial sums to the partner's inbox in LDS (16 KB, lane-linear ds_write_b128),
//     and after one barrier starts the 4 rows it finishes itself FROM the partner's partial sums (ds_read_b128 straight into
//     the accumulators: no addition pass, no zero start).  Operands are swapped (A = weights), so a lane holds 16 channels of
//     one pixel: the weight rows are assigned to MFMA rows such that those are two runs of 8 contiguous channels -> two
//     16-byte NHWC stores per pixel, no cross-lane traffic;
//   * the 10 x 18-pixel input patch (46 080 B) of tile t+1 arrives by LDS-DMA (45 pieces of 1 KiB, interleaved with
//     the tap loop of tile t) into the other patch buffer.  LDS image: 256 B per pixel, NO padding -- 16-byte chunk c of
//     patch pixel P sits at chunk position c ^ 2 (P & 7): the DMA writes lane-linear, so the swizzle is applied on the
//     SOURCE side (a per-lane, tile-invariant source offset) and costs nothing; the 16 pixels of a fragment read are
//     consecutive P, which makes every 16-lane group of the ds_read_b128 cover 16 distinct bank quads.  18 P = 2 P (mod 8),
//     so the swizzle term depends on the patch row only through (row & 3): 12 base registers (row & 3, tap column), the
//     k-step is an XOR of 64, four patch rows are an immediate offset of 18 432, and the two patch buffers sit 64 KiB apart
//     (one more XOR) -- no address arithmetic in the tap loop beyond those XORs;
//   * BatchNorm statistics (training forward): per-channel sums over a lane's four pixels are in-lane adds, the 16 pixel
//     columns are the 16 lanes of a DPP row (4 row_ror adds per value), the two K-half waves fold through 2 KB of LDS and
//     one of them writes the tile's row of the partial-statistics buffer one period later.
// A third instantiation applies a per-channel scale / bias before the ReLU / addend (inference with the BatchNorm folded into the
// conv; bias + ReLU convs): 1 KB of LDS table, two ds_read_b128 per eight channels in the epilogue.
// LDS: 2 x 46 080 (patches, at 0 and 65 536) + 4 x 16 384 (inboxes) + 2 048 (statistics fold) + 1 024 (dump) + 1 024 (scale | bias)
// of 163 840 bytes.
// The MFMAs are issued from inline asm ("a" constraints: the register allocator keeps MFMA A/B operands in VGPRs whatever the
// pressure and spills the rest of the weights to scratch), which also fixes their order against the fragment reads and the
// DMA pieces; two hazards the compiler does not see inside an asm statement are covered by hand (s_nop after the last MFMA
// before its result is read; s_nop 4 before a vector-memory instruction whose SGPR operand a VALU instruction may have
// written -- the cause of a memory fault in the first version of the dynamic walk).
// Where a wave's cycles go (in-kernel stamps, tools/lab/k2q_stamps.py, 8 x 512^2, ~2.0 GHz held): 15 900 per tile against
// 9 216 of bare MFMA issue -- the tap loop 12 400 (9 970 with no patch traffic at all: ~165 cycles per LDS-DMA piece,
// the same when the pieces are staged through registers instead (K2Q_DMA_MODE 3), wherever they are placed, and whether their
// address arithmetic is 25 instructions or 3), the meeting ~1 500, the epilogue 1 750 (+ 2 300 with statistics, + 1 200 with
// an addend).  With ONE wave per SIMD nothing overlaps these; two waves per SIMD do not fit the weights.
// Out-of-image patch pixels fail the buffer descriptor's range check and land as zeros (the conv's zero padding).
#include "conv_igemm.h"

#include <atomic>

#include <cstdlib>
#include <type_traits>

namespace {

using namespace jspsr;

constexpr int Q_NTH = 256;                                  // 4 waves, one per SIMD
constexpr int Q_TH = 8, Q_TW = 16;                          // tile
constexpr int Q_PH = Q_TH + 2, Q_PW = Q_TW + 2, Q_NPIX = Q_PH * Q_PW;      // 10 x 18 patch
constexpr int Q_PIXB = 256;                                 // LDS bytes per patch pixel: 128 bf16, swizzled, no padding
constexpr int Q_PATCHB = Q_NPIX * Q_PIXB;                   // 46 080
constexpr int Q_PIECES = Q_PATCHB / 1024;                   // 45 LDS-DMA pieces of 64 lanes x 16 B
static_assert(Q_PIECES * 1024 == Q_PATCHB, "whole pieces");
constexpr int Q_NP = (Q_PIECES + 3) / 4;                    // pieces per wave (wave w: pieces w, w + 4, ...)
constexpr int Q_BUF1 = 65536;                               // the second patch buffer: one address bit away
constexpr int Q_XCHB = 16384;                               // inbox of one wave: 4 rows x 4 channel blocks x 1 KiB
constexpr int Q_OFF_RED = Q_PATCHB + Q_XCHB;                // [2 kh][2][128] floats (statistics fold), inside the first 64 KiB
static_assert(Q_OFF_RED + 2048 <= Q_BUF1, "the first 64 KiB hold patch 0, inbox 0 and the fold");
constexpr int Q_OFF_RUN = Q_OFF_RED + 2048;                // two published run starts of the dynamic tile queue
static_assert(Q_OFF_RUN + 16 <= Q_BUF1, "run starts inside the first 64 KiB");
constexpr int Q_RUN = 4;                                    // tiles per draw (x-neighbours: their halo columns stay in this XCD's L2)
constexpr int Q_OFF_DUMP = Q_BUF1 + Q_PATCHB + 3 * Q_XCHB;  // 1 KiB nobody reads: where the piece a wave does not have lands
constexpr int Q_OFF_PAR = Q_OFF_DUMP + 1024;                // affine form: [2][128] floats, per-channel scale | bias
constexpr int Q_LDS = Q_OFF_PAR + 1024;                     // 162 816
static_assert(Q_LDS <= 163840, "LDS of one CU");
constexpr int Q_ROWS4 = 4 * Q_PW * Q_PIXB;                  // four patch rows: 18 432 (an immediate offset)
#ifndef K2Q_LA
#define K2Q_LA 4      // patch fragments requested ahead of their MFMAs (lab builds: -DK2Q_LA=n)
#endif
#ifndef K2Q_DMA_MODE
#define K2Q_DMA_MODE 1   // 0: all LDS-DMA pieces before the tap loop; 1: interleaved with it; 2: none (timing lab: WRONG results);
#endif                   // 3: pieces staged through registers (buffer_load -> VGPR -> ds_write_b128), interleaved
#ifndef K2Q_LD_EVERY
#define K2Q_LD_EVERY 9    // mode 3: a piece is requested every so many fragment reads ...
#endif
#ifndef K2Q_LD_DIST
#define K2Q_LD_DIST 27    // ... and written to LDS this many reads later (three pieces = 12 registers in flight)
#endif
#ifndef K2Q_DMA_EVERY
#define K2Q_DMA_EVERY 6  // a piece every so many fragment reads, from the start of the tap loop: the last of the 12 is requested half a
#endif                   // tile before the wait for it (4 .. 12 measured equal: profiles/r04_k2q_wave_life.txt)
constexpr int Q_NK = 144, Q_LAH = K2Q_LA;                   // fragment reads per tile and wave; look-ahead

__device__ __forceinline__ int inbox_off(int wave) { return wave == 0 ? Q_PATCHB : Q_BUF1 + Q_PATCHB + (wave - 1) * Q_XCHB; }

// v[c] += v[c] of the lane SH places further round the 16-lane DPP row (row_ror): after 8, 4, 2, 1 every lane holds the row's sum
template <int SH, int N>
__device__ __forceinline__ void row_sum(float (&v)[N]) {
#pragma unroll
  for (int c = 0; c < N; ++c)
    v[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[c]), 0x120 + SH, 0xf, 0xf, false));
}

// f(integral_constant<int, I>) for I = B .. E-1, in order: loop indices that template arguments can be made from
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// Four products of one patch fragment with the four channel blocks of a wave, issued from inline asm: the register
// allocator keeps MFMA A/B operands in VGPRs whatever the pressure (and spills the rest of the 288 weight registers to
// scratch); an "a" constraint is the only way to have them live in the AGPR half of the file.  FIRST: the accumulators are
// written without being read (srcC = 0).  WA: weights in AGPRs ("a") or VGPRs ("v").  asm volatile: program order is issue
// order -- the fragment reads and the DMA pieces stay where the source puts them.
template <bool FIRST, bool WA>
__device__ __forceinline__ void mfma4(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const i32x4& w0, const i32x4& w1, const i32x4& w2,
                                      const i32x4& w3, const i32x4& b) {
#define K2Q_MFMA4(OUT, CC, WC)                                                                                           \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %8, " CC "0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, " CC "1\n\t"        \
               "v_mfma_f32_16x16x32_bf16 %2, %6, %8, " CC "2\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %8, " CC "3"             \
               : OUT(c0), OUT(c1), OUT(c2), OUT(c3) : WC(w0), WC(w1), WC(w2), WC(w3), "v"(b))
  if constexpr (FIRST) {
    // "0" + "0" etc. would read as operand numbers: the first product of an accumulator takes the literal 0 as srcC
    if constexpr (WA)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %8, 0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, 0\n\t"
                   "v_mfma_f32_16x16x32_bf16 %2, %6, %8, 0\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %8, 0"
                   : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b));
    else
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %8, 0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, 0\n\t"
                   "v_mfma_f32_16x16x32_bf16 %2, %6, %8, 0\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %8, 0"
                   : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b));
  } else {
    if constexpr (WA) K2Q_MFMA4("+v", "%", "a");
    else K2Q_MFMA4("+v", "%", "v");
  }
#undef K2Q_MFMA4
}

typedef const i32x4 __attribute__((address_space(3)))* lds_frag_t;      // a patch fragment by its LDS address (no generic-pointer base add per read)

struct QTile { int bimg, tyi, txi; };
struct QSrc { i32x4 desc; __amdgpu_buffer_rsrc_t rsrc; int oy0, ox0; bool interior; };

template <int SIGN, int MODE, int KH>      // MODE 0: plain (+ addend / ReLU), 1: + BatchNorm statistics, 2: + per-channel scale / bias
__device__ __forceinline__ void conv128_body(const __bf16* __restrict__ in, const __bf16* __restrict__ wgt, const float* __restrict__ bias,
                                             __bf16* __restrict__ out, float* __restrict__ stats, const ConvGeom& g, int ntiles, unsigned* __restrict__ ticket, char* smem) {
  constexpr bool STATS = MODE == 1, AFFINE = MODE == 2;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int cg = wave & 1;                                  // (KH = wave >> 1)
  const int n = lane & 15, q = lane >> 4;
  const int ttx = (g.MW + Q_TW - 1) / Q_TW, tty = (g.MH + Q_TH - 1) / Q_TH;
  const int G = gridDim.x;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  if (lds0 != 0) __builtin_trap();      // the XOR steps of the fragment addresses (k-step: 64, patch buffer: 65 536) assume the dynamic segment starts at 0: this kernel has no static LDS
  const int v = xcd_contiguous(blockIdx.x, G);

  // ---- weights: 72 A-operands (9 taps x 2 k-steps x 4 channel blocks), once -------------------------------------------------
  // MFMA row m of channel block cb is output channel 64 cg + 32 (cb >> 1) + 8 (m >> 2) + 4 (cb & 1) + (m & 3): the result lane
  // (n, q) -- rows 4 q .. 4 q + 3 of every block -- then holds channels 32 h + 8 q + 0 .. 7 (h = cb >> 1) of pixel n
  i32x4 breg[9][2][4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int ch = 64 * cg + 32 * (cb >> 1) + 8 * (n >> 2) + 4 * (cb & 1) + (n & 3);
    const char* wrow = reinterpret_cast<const char*>(wgt) + (size_t)ch * (9 * 128 * 2) + (64 * KH + 8 * q) * 2;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) breg[t][s][cb] = *reinterpret_cast<const i32x4*>(wrow + t * 256 + s * 64);
  }

  const int dGx = G % ttx, dGy = (G / ttx) % tty, dGb = G / (ttx * tty);
  auto advance = [&](QTile c) {
    c.txi += dGx;
    int carry = c.txi >= ttx;
    c.txi -= carry ? ttx : 0;
    c.tyi += dGy + carry;
    carry = c.tyi >= tty;
    c.tyi -= carry ? tty : 0;
    c.bimg += dGb + carry;
    return c;
  };

  // ---- patch DMA: lane (n, q) of piece i fills chunk POSITION n of patch pixel P = 4 i + q with chunk n ^ 2 (P & 7) ----------
  const int pix_bytes = g.in_cstride * 2;
  constexpr int back = SIGN < 0 ? 2 : 0;     // reversed walk: the patch starts two pixels earlier
  auto patch_src = [&](QTile tc, bool more) {
    QSrc p;
    p.oy0 = tc.tyi * Q_TH + g.iy_add - back;
    p.ox0 = tc.txi * Q_TW + g.ix_add - back;
    const long long opix0 = ((long long)tc.bimg * g.IH + p.oy0) * g.IW + p.ox0;     // may lie outside the raster
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(in) + (opix0 * g.in_cstride + g.in_coff) * 2;
    // no next tile: an empty descriptor -- every lane fails the range check and the pieces land as zeros in the idle buffer
    p.desc = i32x4{(int)(unsigned)b64, (int)((unsigned)(b64 >> 32) & 0xffffu), more ? (int)0xFFFFFF00u : 0, 0x00020000};
    p.interior = p.oy0 >= 0 && p.ox0 >= 0 && p.oy0 + Q_PH <= g.IH && p.ox0 + Q_PW <= g.IW;    // wave-uniform
    p.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(b64), 0, more ? 0xFFFFFF00u : 0u, 0x00020000);
    return p;
  };
  // (s_nop 4 between the M0 write and the DMA: 1 wait state for M0, 5 for a descriptor SGPR that a VALU instruction -- a
  // v_readfirstlane of tile coordinates that came through LDS, dynamic walk -- may have written just before the statement)
  // One piece, branch-free (it sits between the MFMAs of the tap loop).  A lane whose pixel lies outside the image gets an
  // offset that fails the descriptor's range check: `buffer_load ... lds` then writes ZEROS for it -- the conv's zero padding.
  // Issued from inline asm without a memory clobber (the bytes land in the OTHER patch buffer, read after the next
  // barrier); M0 is written in the statement that uses it and by nothing else in this file (csrc/check_m0.sh).
  // (the swizzle term is the same for all pieces of a lane: 16 i = 0 (mod 8))
  const int p0lane = 4 * wave + q;
  const unsigned swz16 = (unsigned)((n ^ (2 * (p0lane & 7))) << 4);
  // Interior tiles (all but the image border's) take a lane's source offset of piece i from a register and issue the piece in
  // three instructions; border tiles recompute the lane's patch coordinates for the range test behind a scalar branch -- with
  // 24-bit multiplies (full rate), each kept apart from the add behind it by an empty asm (left alone, the compiler makes
  // quarter-rate v_mad_u64_u32 of them and an EXEC-masked block around those), and no lane branch.  With ONE wave per SIMD
  // every instruction in the tap loop that is not an MFMA is time the matrix pipe idles.
  unsigned dsrc[Q_NP];
#pragma unroll
  for (int i = 0; i < Q_NP; ++i) {
    const int P = p0lane + 16 * i, py = P / Q_PW, px = P - py * Q_PW;
    dsrc[i] = (unsigned)((py * g.IW + px) * pix_bytes) + swz16;
  }
  struct PieceXY { int py, px; unsigned off; };
  auto piece_xy = [&](int i) __attribute__((always_inline)) {       // tile-invariant, recomputed: patch pixel of this lane in piece i
    int p0 = p0lane;
    asm volatile("" : "+v"(p0));                               // opaque: the compiler must not hoist twelve sets of coordinates into registers
    const int P = p0 + 16 * i;
    PieceXY c;
    c.py = (int)(__umul24(P, 3641) >> 16);                     // P / 18 for P < 400
    unsigned t = __umul24(c.py, Q_PW);
    asm("" : "+v"(t));
    c.px = P - (int)t;
    unsigned t2 = __umul24(c.py, g.IW);                        // 10 IW + 18 and the pixel pitch are < 2^24 (conv128_resident_ok)
    asm("" : "+v"(t2));
    c.off = t2 + (unsigned)c.px;
    return c;
  };
  auto piece_src = [&](const QSrc& p, const PieceXY& c) __attribute__((always_inline)) {
    unsigned b = __umul24(c.off, pix_bytes);
    asm("" : "+v"(b));
    unsigned src = b + swz16;
    src = (unsigned)(p.oy0 + c.py) < (unsigned)g.IH ? src : 0xFFFFFFF0u;
    src = (unsigned)(p.ox0 + c.px) < (unsigned)g.IW ? src : 0xFFFFFFF0u;
    return src;
  };
  auto piece_dst = [&](unsigned bufoff, int i) __attribute__((always_inline)) {      // wave-uniform; the dump for the 12th piece of waves 1-3
#ifdef K2Q_LAB_TODUMP      // timing lab (WRONG results): every piece is fetched and lands in the dump -- the traffic without new patch data
    return (unsigned)Q_OFF_DUMP;
#endif
    return (wave + 4 * i < Q_PIECES) ? bufoff + (unsigned)(wave + 4 * i) * 1024u : (unsigned)Q_OFF_DUMP;
  };
  auto issue_piece = [&](const QSrc& p, unsigned bufoff, int i) __attribute__((always_inline)) {
    unsigned src = dsrc[i];
    if (!p.interior) src = piece_src(p, piece_xy(i));
    const unsigned dst = lds0 + piece_dst(bufoff, i);
#if defined(K2Q_LAB_NOISSUE)      // timing lab (WRONG results): the address arithmetic without the DMA instruction
    asm volatile("" : : "s"(dst), "v"(src), "s"(p.desc));
#elif defined(K2Q_LAB_OOB)        // timing lab (WRONG results): the DMA instruction with every lane out of range -- zeros land, nothing is fetched
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(src | 0xF0000000u), "s"(p.desc));
#else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(src), "s"(p.desc));
#endif
  };

  // ---- patch-fragment plan: lane (n, q) reads chunk 8 KH + 4 s + q of patch pixel 18 R + tx + n ---------------------------------
  unsigned abase[4][3];      // [R & 3][tap column], k-step 0, patch buffer 0
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
      abase[b][tx] = (unsigned)((Q_PW * b + tx + n) * Q_PIXB + (((8 * KH + q) ^ (2 * ((2 * b + tx + n) & 7))) << 4));

  // The addend of a tile (the gradient arriving along the shortcut) is read in the epilogue, where nothing hides an HBM miss:
  // one more DMA piece early in the tap loop touches this wave's 64 lines of it (pixel (4 KH + q, n), 128 B each) and lands
  // in the dump -- the epilogue's loads then hit L2.  No addend: an empty descriptor.
  auto prefetch_addend = [&](QTile tc) __attribute__((always_inline)) {
    const int y = tc.tyi * Q_TH + 4 * KH + q, x = tc.txi * Q_TW + n;
    const unsigned off = (y < g.MH && x < g.MW) ? (unsigned)(y * g.OW + x) * (unsigned)(g.add_cstride * 2) : 0xFFFFFFF0u;
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(g.addend) + ((long long)tc.bimg * g.OH * g.OW * g.add_cstride + 64 * cg) * 2;
    const i32x4 d = {(int)(unsigned)b64, (int)((unsigned)(b64 >> 32) & 0xffffu), g.addend ? (int)0xFFFFFF00u : 0, 0x00020000};
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds0 + (unsigned)Q_OFF_DUMP), "v"(off), "s"(d));
  };

  // ---- the two K-halves of a channel group meet in the MIDDLE of the tap loop ----------------------------------------------------
  // tile rows 4 KH .. 4 KH + 3 are this wave's to finish; the other four go to wave ^ 2 (same channels, other K-half).  A wave
  // multiplies the rows it gives away FIRST, writes their partial sums to the partner's inbox, and after one barrier starts the
  // rows it keeps FROM the partner's partial sums (the inbox is read straight into the accumulators: no addition pass, no
  // zero start) -- with one wave per SIMD every instruction that is not an MFMA is time the matrix pipe idles.
  char* const give = smem + inbox_off(wave ^ 2) + lane * 16;
  const char* const take = smem + inbox_off(wave) + lane * 16;
  f32x4 acc[8][4];
  auto meet = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the accumulators are read by ordinary instructions next: the compiler does not know an MFMA wrote them
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<f32x4*>(give + (i * 4 + cb) * 1024) = acc[4 * (1 - KH) + i][cb];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[4 * KH + i][cb] = *reinterpret_cast<const f32x4*>(take + (i * 4 + cb) * 1024);
  };

  // ---- the tap loop: 144 fragment reads x 4 MFMAs, weights from registers -----------------------------------------------------
  auto mfma_tile = [&](const QSrc& nxt, unsigned nbufoff, QTile tc) __attribute__((always_inline)) {
    i32x4 a[Q_LAH];
    constexpr int NSTG = (K2Q_LD_DIST + K2Q_LD_EVERY - 1) / K2Q_LD_EVERY;
    u32x4 stg[NSTG];       // mode 3: pieces on their way from memory to LDS
    static_for<0, Q_NK + Q_LAH>([&](auto K) __attribute__((always_inline)) {
      constexpr int k = decltype(K)::value;
      // step kk: half = kk / 72 (0: the rows given away, 1: the rows kept), then 18 (tap, k-step) pairs x 4 rows
      if constexpr (k >= Q_LAH) {        // consumes ring slot k % LA before the read below refills it
        constexpr int kk = k - Q_LAH, half = kk / 72, st = (kk % 72) >> 2, tap = st >> 1, s = st & 1;
        constexpr int r = (half == 0 ? 4 * (1 - KH) : 4 * KH) + (kk & 3);
        if constexpr (kk == 72) meet();
        // taps 0-7: weights in AGPRs (64 operands = all 256 of them), tap 8: in VGPRs
        mfma4<(half == 0 && st == 0), (tap < 8)>(acc[r][0], acc[r][1], acc[r][2], acc[r][3], breg[tap][s][0], breg[tap][s][1], breg[tap][s][2],
                                                 breg[tap][s][3], a[kk % Q_LAH]);
      }
      if constexpr (k < Q_NK) {
        constexpr int half = k / 72, st = (k % 72) >> 2, tap = st >> 1, s = st & 1;
        constexpr int r = (half == 0 ? 4 * (1 - KH) : 4 * KH) + (k & 3);
        constexpr int ty = SIGN > 0 ? tap / 3 : 2 - tap / 3, tx = SIGN > 0 ? tap % 3 : 2 - tap % 3;
        constexpr int R = r + ty;
        a[k % Q_LAH] = *reinterpret_cast<lds_frag_t>((abase[R & 3][tx] ^ (unsigned)(s * 64)) + (unsigned)((R >> 2) * Q_ROWS4));
        if constexpr (K2Q_DMA_MODE == 1 && k % K2Q_DMA_EVERY == 2 && k / K2Q_DMA_EVERY < Q_NP) issue_piece(nxt, nbufoff, k / K2Q_DMA_EVERY);
        if constexpr (!STATS && k == 1) prefetch_addend(tc);
        if constexpr (K2Q_DMA_MODE == 3) {
          constexpr int LD = K2Q_LD_EVERY, WD = K2Q_LD_DIST;
          static_assert(2 + LD * (Q_NP - 1) + WD < Q_NK, "the last piece is written inside the loop");
          if constexpr (k >= 2 + WD && (k - 2 - WD) % LD == 0 && (k - 2 - WD) / LD < Q_NP) {
            constexpr int j = (k - 2 - WD) / LD;
            *reinterpret_cast<u32x4*>(smem + piece_dst(nbufoff, j) + lane * 16) = stg[j % NSTG];
          }
          if constexpr (k >= 2 && (k - 2) % LD == 0 && (k - 2) / LD < Q_NP) {
            constexpr int j = (k - 2) / LD;
            stg[j % NSTG] = __builtin_amdgcn_raw_buffer_load_b128(nxt.rsrc, piece_src(nxt, piece_xy(j)), 0, 0);
          }
        }
      }
    });
    // the accumulators are read by ordinary instructions next: the compiler does not know an MFMA wrote them
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  };

  const bool relu_first = g.relu && !g.addend, relu_last = g.relu && g.addend;
  float* const red = reinterpret_cast<float*>(smem + Q_OFF_RED);      // [2 kh][2][128]
  auto out_desc = [&](int bimg) {
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(out) + ((long long)bimg * g.OH * g.OW * g.out_cstride + g.out_coff + 64 * cg) * 2;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(b64), 0, 0xFFFFFF00u, 0x00020000);
  };
  auto add_desc = [&](int bimg) {
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(g.addend) + ((long long)bimg * g.OH * g.OW * g.add_cstride + 64 * cg) * 2;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(b64), 0, g.addend ? 0xFFFFFF00u : 0u, 0x00020000);
  };

  // Lane (n, q) holds, of pixel (row 4 KH + i, column n) of the tile, channels 32 h + 8 q + 0..7 (h = 0, 1) of this wave's 64:
  // acc[.][2 h] are the first four, acc[.][2 h + 1] the last four.
  auto epilogue = [&](QTile tc) __attribute__((always_inline)) {
    const int ty0 = tc.tyi * Q_TH + 4 * KH, x = tc.txi * Q_TW + n;
    const __amdgpu_buffer_rsrc_t orsrc = out_desc(tc.bimg), adrsrc = add_desc(tc.bimg);
    unsigned ooff[4], aoff[4];
    bool inside[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = ty0 + i;
      inside[i] = y < g.MH && x < g.MW;
      const unsigned pix = (unsigned)(y * g.OW + x);
      ooff[i] = inside[i] ? pix * (unsigned)(g.out_cstride * 2) + q * 16u : 0xFFFFFFF0u;
      aoff[i] = inside[i] ? pix * (unsigned)(g.add_cstride * 2) + q * 16u : 0xFFFFFFF0u;
    }
    if constexpr (STATS) {
      if (stats) {
        // per-channel sum and sum of squares over this wave's 4 x 16 pixels: rows in-lane, the 16 columns across the DPP row;
        // eight channels (one 16-byte run) at a time
        const bool whole = tc.tyi * Q_TH + Q_TH <= g.MH && tc.txi * Q_TW + Q_TW <= g.MW;      // wave-uniform
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float s1[8], s2[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
          if (whole) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const float a0 = acc[4 * KH + i][2 * h + (c >> 2)][c & 3];
                s1[c] += a0;
                s2[c] = __builtin_fmaf(a0, a0, s2[c]);
              }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float m = inside[i] ? 1.f : 0.f;
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const float a0 = acc[4 * KH + i][2 * h + (c >> 2)][c & 3] * m;
                s1[c] += a0;
                s2[c] = __builtin_fmaf(a0, a0, s2[c]);
              }
            }
          }
          row_sum<8>(s1); row_sum<8>(s2);
          row_sum<4>(s1); row_sum<4>(s2);
          row_sum<2>(s1); row_sum<2>(s2);
          row_sum<1>(s1); row_sum<1>(s2);
          if (n == 0) {
            float* r1 = red + (KH * 2 + 0) * 128 + 64 * cg + 32 * h + 8 * q;
            float* r2 = red + (KH * 2 + 1) * 128 + 64 * cg + 32 * h + 8 * q;
            *reinterpret_cast<f32x4*>(r1) = f32x4{s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<f32x4*>(r1 + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
            *reinterpret_cast<f32x4*>(r2) = f32x4{s2[0], s2[1], s2[2], s2[3]};
            *reinterpret_cast<f32x4*>(r2 + 4) = f32x4{s2[4], s2[5], s2[6], s2[7]};
          }
        }
      }
    }
    if constexpr (AFFINE) {      // acc * scale + bias for this lane's sixteen channels, from the table the prologue put in LDS
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* tab = reinterpret_cast<const float*>(smem + Q_OFF_PAR) + 64 * cg + 32 * h + 8 * q;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(tab + 4 * u), b4 = *reinterpret_cast<const f32x4*>(tab + 128 + 4 * u);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * KH + i][2 * h + u][j] = acc[4 * KH + i][2 * h + u][j] * s4[j] + b4[j];
        }
      }
    }
    // Two whole branches (wave-uniform): the addend loads are issued AND consumed inside one of them.  (Issued under one `if`
    // and consumed under another, the compiler's wait-count bookkeeping carries them as possibly pending into the next
    // tile, whose first MFMAs overwrite their registers: it then waits at the head of every tap loop with a small vmcnt --
    // for the stores of the tile before.)
    if (!STATS && g.addend) {
      u32x4 av[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) av[i][h] = __builtin_amdgcn_raw_buffer_load_b128(adrsrc, aoff[i], h * 64, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 lo = acc[4 * KH + i][2 * h], hi = acc[4 * KH + i][2 * h + 1];
          const u32x4 o = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
          __builtin_amdgcn_raw_buffer_store_b128(add_bf16x8(o, av[i][h], relu_last), orsrc, ooff[i], h * 64, 0);
        }
    } else {
      if (!STATS && relu_first) {      // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * KH + i][cb][j] = relu_bits(acc[4 * KH + i][cb][j]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 lo = acc[4 * KH + i][2 * h], hi = acc[4 * KH + i][2 * h + 1];
          const u32x4 o = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
          __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, ooff[i], h * 64, 0);
        }
    }
  };
  // the tile's row of the partial-statistics buffer (rows are 8 x 16-pixel tiles: jspsr_conv2d_stats_rows), one period late
  auto flush_stats = [&](QTile tc) __attribute__((always_inline)) {
    const int which = tid >> 7, col = tid & 127;
    stats[((size_t)((tc.bimg * tty + tc.tyi) * ttx + tc.txi) * 2 + which) * 128 + col] = red[which * 128 + col] + red[(2 + which) * 128 + col];
  };

  if constexpr (AFFINE) {      // absent scale -> 1, absent bias -> 0; visible to every wave after the first barrier of the loop
    reinterpret_cast<float*>(smem + Q_OFF_PAR)[tid] = tid < 128 ? (g.scale ? g.scale[tid] : 1.f) : (bias ? bias[tid - 128] : 0.f);
  }
  // Tile walk.  Static (ticket == nullptr): tile v, v + G, v + 2G ... -- fine alone on the chip.  Dynamic: the workgroups draw
  // runs of Q_RUN x-neighbouring tiles from one global ticket.  A workgroup of this kernel needs a WHOLE CU (all of its LDS,
  // every register of its four SIMDs): beside the weight-gradient kernels of the other streams some CUs come free late, and
  // with the static walk the launch would end that much late -- here such a workgroup simply draws fewer runs.  A draw is
  // issued by thread 0 when the walk ENTERS a run (for the run after the next), published in LDS at the end of that tile
  // (the wait-count there has covered it) and read behind a barrier at least one tile later.
  const bool dyn = ticket != nullptr;
  int* const s_run = reinterpret_cast<int*>(smem + Q_OFF_RUN);
  if (dyn) {
    if (tid == 0) {
      s_run[0] = (int)__hip_atomic_fetch_add(ticket, (unsigned)Q_RUN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_run[1] = (int)__hip_atomic_fetch_add(ticket, (unsigned)Q_RUN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  auto tile_of = [&](int t_) { return QTile{(t_ / ttx) / tty, (t_ / ttx) % tty, t_ % ttx}; };
  int t = dyn ? s_run[0] : v, it = 0;
  int krun = 0, nrun = 0;                 // dynamic: position of tile t in its run, number of that run
  unsigned pend = 0;                      // thread 0: a drawn run start not yet published
  int pend_slot = -1;
  QTile tcur = tile_of(t < ntiles ? t : 0), tprev = tcur;
  unsigned bufoff = 0;
  if (t < ntiles) {
    const QSrc p0 = patch_src(tcur, true);
#pragma unroll
    for (int i = 0; i < Q_NP; ++i) issue_piece(p0, 0u, i);
  }
  // the weights and the first patch are in.  As a builtin: the compiler's own wait-count bookkeeping then knows that no load of
  // its own is pending at the loop head (left to itself it waits for the weight loads INSIDE the loop with small counts, and
  // every tile would wait there for the stores of the tile before)
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  asm volatile("" ::: "memory");
#ifdef K2Q_STAMPS      // lab build: cycles per phase, printed by waves 0 and 2 of workgroup 0 (tools/lab/build_k2q_variants.sh stamps=-DK2Q_STAMPS)
  unsigned long long st_bar = 0, st_mfma = 0, st_epi = 0, st_vm = 0, s0, s1;
  const unsigned long long c_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();
#define K2Q_STAMP(acc_) do { s1 = __builtin_readcyclecounter(); acc_ += s1 - s0; s0 = s1; } while (0)
#else
#define K2Q_STAMP(acc_) do { } while (0)
#endif
  for (; t < ntiles; ++it) {
#ifdef K2Q_STAMPS
    s0 = __builtin_readcyclecounter();
#endif
    // every wave has waited for its own pieces of this patch (below); once all have arrived it is complete, nobody reads
    // the other patch buffer or an inbox any more
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    K2Q_STAMP(st_bar);
    if (STATS && stats && it > 0) flush_stats(tprev);
    int tn, kn = 0, nn = nrun;
    QTile tnext;
    if (!dyn) {
      tn = t + G;
      tnext = advance(tcur);
    } else if (krun + 1 < Q_RUN && t + 1 < ntiles) {
      tn = t + 1;
      kn = krun + 1;
      tnext = tcur;
      if (++tnext.txi == ttx) { tnext.txi = 0; if (++tnext.tyi == tty) { tnext.tyi = 0; ++tnext.bimg; } }
    } else {
      nn = nrun + 1;
      tn = s_run[nn & 1];                 // published at least one barrier ago
      tnext = tile_of(tn < ntiles ? tn : 0);
    }
    if (dyn && tid == 0 && krun == 0) {   // entering run nrun: draw run nrun + 2 into the slot run nrun no longer needs
      // from inline asm, in place in `pend`: as a builtin the compiler waits for the returned value (vmcnt(0): the stores of the
      // tile before and the round trip) at the end of this block, where the loop-carried variable is merged; the value is
      // first read at the end of the tile, behind the wait-count that covers it
      // (s_nop 4: the compiler may have just produced the pointer's SGPRs with a VALU instruction -- v_readlane from its
      // SGPR-spill register -- and a vector-memory instruction must not read such an SGPR within 5 wait states; the compiler
      // does not look into an asm statement for that hazard)
      asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "+v"(pend) : "v"(0u), "v"((unsigned)Q_RUN), "s"(ticket));
      pend_slot = nrun & 1;
    }
    const bool more = tn < ntiles;
    const QSrc nxt = patch_src(tnext, more);
    if (K2Q_DMA_MODE == 0) {
#pragma unroll
      for (int i = 0; i < Q_NP; ++i) issue_piece(nxt, bufoff ^ Q_BUF1, i);
    }
    mfma_tile(nxt, bufoff ^ Q_BUF1, tcur);
    K2Q_STAMP(st_mfma);
    epilogue(tcur);
    K2Q_STAMP(st_epi);
    // the other patch buffer next: one XOR per base register
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) abase[b][tx] ^= (unsigned)Q_BUF1;
    bufoff ^= Q_BUF1;
    // this wave's pieces of the next patch have landed once all but its 8 youngest vector-memory operations (the stores of
    // the epilogue; the pieces and the addend loads are older) are done
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    K2Q_STAMP(st_vm);
    if (dyn && tid == 0 && pend_slot >= 0) { s_run[pend_slot] = (int)pend; pend_slot = -1; }
    tprev = tcur;
    tcur = tnext;
    t = tn; krun = kn; nrun = nn;
  }
#ifdef K2Q_STAMPS
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 2) && it > 0) {
    const unsigned long long dc = __builtin_readcyclecounter() - c_begin, dr = __builtin_amdgcn_s_memrealtime() - r_begin;
    printf("K2Q wave %d tiles %d: barrier %llu  tap loop incl. the meeting %llu  epilogue %llu  vmcnt %llu (cycles per tile); clock %.0f MHz\n",
           wave, it, st_bar / it, st_mfma / it, st_epi / it, st_vm / it, 100.0 * (double)dc / (double)dr);
  }
#endif
  if (dyn && tid == 0) {
    // last workgroup out re-arms the ticket for the launch that gets this slot next (1024 launches from now); every draw of
    // this workgroup has returned before it signs off
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned old = __hip_atomic_fetch_add(ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)G - 1u) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (STATS && stats && it > 0) {
    __syncthreads();
    flush_stats(tprev);
  }
}

template <int SIGN, int MODE>
__global__ __launch_bounds__(Q_NTH) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv128_resident_kernel(
    const __bf16* __restrict__ in, const __bf16* __restrict__ wgt, const float* __restrict__ bias, __bf16* __restrict__ out,
    float* __restrict__ stats, ConvGeom g, int ntiles, unsigned* __restrict__ ticket) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  // the K-half of a wave decides which accumulators it keeps: compile-time per branch (wave-uniform, scalar branch)
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 7) == 0) conv128_body<SIGN, MODE, 0>(in, wgt, bias, out, stats, g, ntiles, ticket, smem);
  else conv128_body<SIGN, MODE, 1>(in, wgt, bias, out, stats, g, ntiles, ticket, smem);
}

// Tickets of the dynamic tile queue: (next tile, workgroups done) pairs, handed out round-robin per launch; the last
// workgroup of a launch zeroes its pair again, and a slot comes round after 1024 launches.
__device__ unsigned k2q_ring[2 * 1024];

unsigned* next_ticket() {
  static unsigned* base = [] {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(k2q_ring)) == hipSuccess ? static_cast<unsigned*>(p) : nullptr;
  }();
  static std::atomic<unsigned> n{0};
  return base ? base + 2 * (n.fetch_add(1, std::memory_order_relaxed) % 1024u) : nullptr;
}

template <int SIGN, int MODE>
void launch_k2q(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g, int ntiles, int grid, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv128_resident_kernel<SIGN, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS);
    attr_set = true;
  }
  // dynamic tile queue (opt-in: JSPSR_CONV_DYNQ128=1) once every workgroup has several runs to draw.  Measured on one box:
  // alone on the chip the static walk is 5 % faster (8 x 512^2 forward 486 vs 510 us: tile coordinates that come through LDS
  // are per-lane values to the compiler, and neighbouring tiles no longer share an XCD's L2); in the multi-stream step
  // 63.6 / 64.3 / 64.4 ms static against 63.1 / 64.3 / 64.4 dynamic, three interleaved runs each -- no difference.
  static const int dynq = [] { const char* e = getenv("JSPSR_CONV_DYNQ128"); return e ? atoi(e) : 0; }();
  const int forced = conv_dynq_override();      // jspsr_conv_dynamic_queue(): data-parallel runs switch the queue on (RCCL holds CUs beside the backward pass)
  unsigned* ticket = ((forced >= 0 ? forced : dynq) && (long long)ntiles >= 4LL * Q_RUN * grid) ? next_ticket() : nullptr;
  hipLaunchKernelGGL((conv128_resident_kernel<SIGN, MODE>), dim3(grid), dim3(Q_NTH), Q_LDS, s, static_cast<const __bf16*>(in),
                     static_cast<const __bf16*>(wgt), bias, static_cast<__bf16*>(out), stats, g, ntiles, ticket);
}

}  // namespace

namespace jspsr {

bool conv128_resident_ok(const ConvGeom& g, const void* in, const void* wgt, const float* bias, const void* out) {
  static const int enabled = [] { const char* e = getenv("JSPSR_CONV_RESIDENT128"); return e ? atoi(e) : 1; }();
  if (!enabled) return false;
  if (g.Cin != 128 || g.Cout != 128 || g.nty != 3 || g.ntx != 3 || g.KH != 3 || g.KW != 3) return false;
  if (g.iy_mul != 1 || g.ix_mul != 1 || g.oy_mul != 1 || g.ox_mul != 1 || g.oy_add != 0 || g.ox_add != 0) return false;
  if (g.ky0 != 0 || g.kx0 != 0 || g.kstep != 1 || g.in_affine || g.red_out) return false;
  // (the statistics form is the plain training forward: jspsr_conv2d_forward refuses statistics with bias / scale / ReLU / addend)
  if (bias && !aligned4(bias)) return false;
  if (g.sign < 0 && (bias || g.scale)) return false;      // (see launch_conv128_resident)
  if (g.in_cstride % 8 || g.in_coff % 8 || g.out_cstride % 8 || g.out_coff % 8) return false;
  if (!aligned16(in) || !aligned16(wgt) || !aligned16(out)) return false;
  if (g.addend && (!aligned16(g.addend) || g.add_cstride % 8)) return false;
  if ((long long)(g.IW + 20) * 12 * g.in_cstride * 2 >= 0x7fffffffLL || g.IW >= (1 << 20) || g.in_cstride >= (1 << 20)) return false;    // 32-bit offsets inside a patch, 24-bit factors
  if ((long long)g.OH * g.OW * g.out_cstride * 2 >= 0xF0000000LL || (long long)g.OH * g.OW * g.add_cstride * 2 >= 0xF0000000LL)
    return false;                                                                       // ... and inside one image of the result
  const long long tiles = (long long)g.B * ((g.MH + Q_TH - 1) / Q_TH) * ((g.MW + Q_TW - 1) / Q_TW);
  static const int min_tiles = [] { const char* e = getenv("JSPSR_CONV_RESIDENT128_MIN"); return e ? atoi(e) : 2048; }();
  return tiles >= min_tiles && tiles < 0x7fffffffLL;
}

int launch_conv128_resident(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g, hipStream_t s) {
  const int ntiles = g.B * ((g.MH + Q_TH - 1) / Q_TH) * ((g.MW + Q_TW - 1) / Q_TW);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return fail((int)hipErrorInvalidDevice, "conv: device query failed");
    ncu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  const int grid = ntiles < ncu ? ntiles : ncu;
  static const int trace = [] { const char* e = getenv("JSPSR_CONV_TRACE128"); return e ? atoi(e) : 0; }();      // lab: the geometry of each launch
  if (trace)
    fprintf(stderr, "K2q sign %d B %d %dx%d in pitch %d off %d out pitch %d off %d addend %d (pitch %d) relu %d stats %d in %p out %p\n", g.sign, g.B,
            g.MH, g.MW, g.in_cstride, g.in_coff, g.out_cstride, g.out_coff, g.addend != nullptr, g.add_cstride, g.relu, stats != nullptr, in, out);
  const int mode = stats ? 1 : ((bias || g.scale) ? 2 : 0);      // the C ABI refuses statistics together with bias / scale
  if (g.sign > 0) {
    if (mode == 0) launch_k2q<1, 0>(in, wgt, bias, out, stats, g, ntiles, grid, s);
    else if (mode == 1) launch_k2q<1, 1>(in, wgt, bias, out, stats, g, ntiles, grid, s);
    else launch_k2q<1, 2>(in, wgt, bias, out, stats, g, ntiles, grid, s);
  } else {
    // the reversed walk exists in the plain form only: a data gradient has no statistics, and conv128_resident_ok leaves the
    // transposed convs with a bias / scale (none of this shape in the models) to the patch kernel -- four instantiations, not six
    if (mode != 0) return fail(JSPSR_EINVAL, "conv128_resident: the reversed walk has no statistics / affine form");
    launch_k2q<-1, 0>(in, wgt, bias, out, stats, g, ntiles, grid, s);
  }
  return check_launch("conv128_resident");
}

}  // namespace jspsr
