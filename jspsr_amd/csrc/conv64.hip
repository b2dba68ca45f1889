// K2r -- 3x3, 64 -> 64 channel, unit-stride bf16 convolution with the WEIGHTS RESIDENT IN REGISTERS.
//
// The generator's and layer1's full-resolution convs (basics.py:39-47,111-117 at 64 channels; forward and data gradient)
// have K = 9 * 64 = 576 and N = 64: the whole weight matrix is 73.7 KB.  conv_patch_kernel streams it from L2 once per
// 256-pixel tile behind a barrier per tap and pays a cold patch load per tile (PMC, round 1: 28-34 % of the wave cycles
// parked on the stage barrier / weight vmcnt).  Here a PERSISTENT workgroup per CU (8 waves = 4 pixel groups x 2
// channel groups, each wave 64 pixels x 32 channels) loads its 32 weight rows ONCE into 144 VGPRs (36 MFMA B
// fragments) and walks tiles of 16x16 pixels:
//   * the 18x18-pixel input patch of tile t+1 arrives by LDS-DMA (buffer_load_dwordx4 ... lds, 41 one-KiB pieces) into the
//     second patch buffer while tile t is multiplied -- no staging registers, no ds_write pass, no cold prologue;
//   * LDS image: pixel-major 128-byte rows, 16-byte chunk c of pixel P at chunk position c ^ ((P >> 1) & 7).  LDS-DMA
//     writes lane-linear, so the swizzle is applied on the SOURCE address; every ds_read_b128 lane group of every tap
//     then covers 16 distinct bank quads (checked exhaustively: all taps, waves, sub-steps, both walk directions);
//   * the tap loop is 72 MFMAs (v_mfma_f32_32x32x16_bf16) per wave with one A-fragment read each and NO barrier; one
//     barrier per tile (patch t+1 landed / everyone is done with patch t-1);
//   * epilogue per wave through a private LDS scratch (transpose to 16-byte NHWC pieces), no workgroup barrier; the
//     BatchNorm statistics of tile t are folded across the four pixel groups after the next tile's barrier.
// Out-of-image patch pixels fail the buffer descriptor's range check and land as zeros (the conv's zero padding).
#include "conv_igemm.h"

#include <cstdlib>

namespace {

using namespace jspsr;

constexpr int R_NTH = 512;                 // 8 waves
constexpr int R_PW = 18, R_NPIX = R_PW * R_PW;
constexpr int R_PIECES = (R_NPIX * 8 + 63) / 64;          // 41 one-KiB DMA pieces per patch
constexpr int R_PIT = (R_PIECES + 7) / 8;                 // pieces per wave (wave w owns pieces w, w+8, ...)
constexpr int R_PATCHB = R_PIECES * 1024;                 // 41984 B per patch buffer (multiple of 128)
constexpr int R_SCRP = 80;                                // scratch row pitch: 64 B of channels + 16
constexpr int R_SCRB = 64 * R_SCRP;                       // per wave
constexpr int R_OFF_SCR = 2 * R_PATCHB;
constexpr int R_OFF_RED = R_OFF_SCR + 8 * R_SCRB;         // [2][4][2][64] floats
constexpr int R_LDS = R_OFF_RED + 2 * 4 * 2 * 64 * 4;
constexpr int R_ROT = 14;                                 // see conv_patch_kernel: row r of the tile is rotated by 14 r

typedef __attribute__((address_space(3))) void* lptr_t;

template <int SIGN>
__global__ __launch_bounds__(R_NTH) void conv64_resident_kernel(const __bf16* __restrict__ in, const __bf16* __restrict__ wgt,
                                                                  const float* __restrict__ bias, __bf16* __restrict__ out,
                                                                  float* __restrict__ stats, ConvGeom g, int ntiles) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int ttx = (g.MW + 15) >> 4, tty = (g.MH + 15) >> 4;
  const int G = gridDim.x;
  const int v = xcd_contiguous(blockIdx.x, G);

  // ---- weights: 36 B fragments (9 taps x 4 sub-steps) of this wave's 32 output channels, once -------------------
  bf16x8 breg[9][4];
  {
    const char* wrow = reinterpret_cast<const char*>(wgt) + (size_t)(wn * 32 + lr) * (9 * 64 * 2) + lh * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) breg[t][s] = *reinterpret_cast<const bf16x8*>(wrow + t * 128 + s * 32);
  }

  // ---- patch DMA plan: lane `lane` of piece i fills 16-byte unit u = 64 i + lane = (pixel P, position q) ----------
  const int pix_bytes = g.in_cstride * 2;
  unsigned dsrc[R_PIT];        // byte offset of (pixel, chunk) from the patch origin
  unsigned dyx[R_PIT];         // py | px << 8
#pragma unroll
  for (int i = 0; i < R_PIT; ++i) {
    const int u = (wave + 8 * i) * 64 + lane, P = u >> 3, q = u & 7;
    const int c = q ^ ((P >> 1) & 7);
    const int py = P / R_PW, px = P - py * R_PW;
    dsrc[i] = (unsigned)((py * g.IW + px) * pix_bytes + c * 16);
    dyx[i] = (unsigned)(py | (px << 8));
    if (P >= R_NPIX) dsrc[i] = 0xFFFFFFF0u;      // beyond the patch: zeros into the slack of the buffer
  }
  const int back = SIGN < 0 ? 2 : 0;     // reversed walk: the patch starts two pixels earlier

  auto tile_coords = [&](int t, int& bimg, int& tyi, int& txi) {
    txi = t % ttx;
    const int r = t / ttx;
    tyi = r % tty;
    bimg = r / tty;
  };
  // A lane whose pixel lies outside the image (or beyond the patch) gets an offset that fails the descriptor's range
  // check: `buffer_load ... lds` then writes ZEROS for it (tools/lab/lds_dma_oob.hip) -- the conv's zero padding.
  auto issue_patch = [&](int t, int buf) __attribute__((always_inline)) {
    int bimg, tyi, txi;
    tile_coords(t, bimg, tyi, txi);
    const int oy0 = tyi * 16 + g.iy_add - back, ox0 = txi * 16 + g.ix_add - back;
    const long long opix0 = ((long long)bimg * g.IH + oy0) * g.IW + ox0;     // may lie outside the raster
    const char* base = reinterpret_cast<const char*>(in) + (opix0 * g.in_cstride + g.in_coff) * 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xFFFFFF00u, 0x00020000);
#pragma unroll
    for (int i = 0; i < R_PIT; ++i) {
      if (wave + 8 * i < R_PIECES) {     // wave-uniform
        const int py = dyx[i] & 0xff, px = dyx[i] >> 8;
        const bool ok = (unsigned)(oy0 + py) < (unsigned)g.IH && (unsigned)(ox0 + px) < (unsigned)g.IW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(smem + buf * R_PATCHB + (wave + 8 * i) * 1024), 16,
                                                 ok ? dsrc[i] : 0xFFFFFFF0u, 0, 0, 0);
      }
    }
  };

  // ---- A-fragment plan ------------------------------------------------------------------------------------------
  int Pl[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int R = wm * 64 + mi * 32 + lr, dy = R >> 4;
    Pl[mi] = dy * R_PW + ((R + R_ROT * dy) & 15) + (SIGN < 0 ? 2 * R_PW + 2 : 0);
  }
  const int lh16 = lh << 4;

  // ---- epilogue constants ---------------------------------------------------------------------------------------
  const int ncol = wn * 32 + lr;
  const float bv = bias ? bias[ncol] : 0.f, sv = g.scale ? g.scale[ncol] : 1.f;
  const bool relu_first = g.relu && !g.addend, relu_last = g.relu && g.addend;
  char* const scr = smem + R_OFF_SCR + wave * R_SCRB;
  float* const red = reinterpret_cast<float*>(smem + R_OFF_RED);
  const int tty8 = (g.MH + 7) >> 3;

  auto flush_stats = [&](int t, int par) __attribute__((always_inline)) {
    // rows of the statistics buffer are numbered by 8x16-pixel tiles (jspsr_conv2d_stats_rows): pixel groups 0,1 are
    // the upper half of this 16x16 tile, 2,3 the lower
    if (tid < 256) {
      int bimg, tyi, txi;
      tile_coords(t, bimg, tyi, txi);
      const int half = tid >> 7, which = (tid >> 6) & 1, col = tid & 63;
      const int row = 2 * tyi + half;
      if (row < tty8) {
        const float* r0 = red + ((par * 4 + 2 * half) * 2 + which) * 64 + col;
        stats[((size_t)((bimg * tty8 + row) * ttx + txi) * 2 + which) * 64 + col] = r0[0] + r0[2 * 64];
      }
    }
  };

  int t = v, it = 0;
  if (t < ntiles) issue_patch(t, 0);
  for (; t < ntiles; t += G, ++it) {
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces (and stores) have completed
    __syncthreads();                                       // patch `buf` has landed; everyone has left tile t - G
    if (stats && it > 0) flush_stats(t - G, buf ^ 1);
    if (t + G < ntiles) issue_patch(t + G, buf ^ 1);

    // ---- 9 taps x 4 sub-steps x 2 row blocks, weights from registers ---------------------------------------------
    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
    int pl0 = Pl[0], pl1 = Pl[1];
    asm volatile("" : "+v"(pl0), "+v"(pl1));      // keep the 18 tap addresses out of the loop-invariant registers
    const char* const pbase = smem + buf * R_PATCHB;
    auto a_addr = [&](int tap, int mi) __attribute__((always_inline)) {
      const int tp = (tap / 3) * R_PW + tap % 3;
      const int P = (mi ? pl1 : pl0) + SIGN * tp;
      return (P << 7) + ((((P << 3) & 0x70)) ^ lh16);
    };
    constexpr int NK = 72, LA = 4;     // MFMA steps; fragment reads in flight ahead of their MFMA
    bf16x8 a[LA];
    int ad[2];
#pragma unroll
    for (int k = 0; k < NK + LA; ++k) {
      if (k >= LA) {                     // consumes ring slot k % LA before the read below refills it
        const int kk = k - LA, tap = kk >> 3, s = (kk >> 1) & 3, mi = kk & 1;
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk % LA], breg[tap][s], acc[mi], 0, 0, 0);
      }
      if (k < NK) {
        const int tap = k >> 3, s = (k >> 1) & 3, mi = k & 1;
        if (s == 0) ad[mi] = a_addr(tap, mi);
        a[k % LA] = *reinterpret_cast<const bf16x8*>(pbase + (ad[mi] ^ (s * 32)));
      }
    }
    // pin the order: LA reads up front, then one read behind every MFMA (left alone, the scheduler sinks each read
    // to just before its use and the MFMA waits out the LDS latency)
    __builtin_amdgcn_sched_group_barrier(0x100, LA, 0);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (k + LA < NK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }

    // ---- epilogue: this wave's 64 pixels x 32 channels -------------------------------------------------------------
    int bimg, tyi, txi;
    tile_coords(t, bimg, tyi, txi);
    const int ty0 = tyi * 16, tx0 = txi * 16;
    if (stats) {
      float s = 0.f, q = 0.f;
      if (ty0 + 16 <= g.MH && tx0 + 16 <= g.MW) {          // interior tile (wave-uniform): every row counts
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float a0 = acc[mi][e];
            s += a0;
            q += a0 * a0;
          }
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
        red[((buf * 4 + wm) * 2 + 0) * 64 + ncol] = s;
        red[((buf * 4 + wm) * 2 + 1) * 64 + ncol] = q;
      }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float o = acc[mi][e] * sv + bv;
        if (relu_first) o = fmaxf(o, 0.f);
        *reinterpret_cast<__bf16*>(scr + row * R_SCRP + lr * 2) = (__bf16)o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private scratch: program order is enough
    const int c16 = lane & 3, prow = lane >> 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int dy = wm * 4 + j;                              // tile row of scratch rows 16 j .. 16 j + 15
      const int y = ty0 + dy, x = tx0 + ((prow + R_ROT * dy) & 15);
      uint4 o = *reinterpret_cast<const uint4*>(scr + (j * 16 + prow) * R_SCRP + c16 * 16);
      if (y < g.MH && x < g.MW) {
        const long long opix = ((long long)bimg * g.OH + y) * g.OW + x;
        if (g.addend) {
          const uint4 ad4 = *reinterpret_cast<const uint4*>(static_cast<const char*>(g.addend) +
                                                           (opix * g.add_cstride + wn * 32) * 2 + c16 * 16);
          const unsigned* pa = &o.x;
          const unsigned* pb = &ad4.x;
          unsigned pr[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float lo = __uint_as_float(pa[i] << 16) + __uint_as_float(pb[i] << 16);
            float hi = __uint_as_float(pa[i] & 0xffff0000u) + __uint_as_float(pb[i] & 0xffff0000u);
            if (relu_last) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            const __bf16 l = (__bf16)lo, h = (__bf16)hi;
            pr[i] = (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
          }
          o = make_uint4(pr[0], pr[1], pr[2], pr[3]);
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + (opix * g.out_cstride + g.out_coff + wn * 32) * 2 + c16 * 16) = o;
      }
    }
  }
  if (stats && it > 0) {
    __syncthreads();
    flush_stats(t - G, (it - 1) & 1);
  }
}

}  // namespace

namespace jspsr {

bool conv64_resident_ok(const ConvGeom& g, const void* in, const void* wgt, const void* out) {
  static const int enabled = [] { const char* e = getenv("JSPSR_CONV_RESIDENT"); return e ? atoi(e) : 1; }();
  if (!enabled) return false;
  if (g.Cin != 64 || g.Cout != 64 || g.nty != 3 || g.ntx != 3 || g.KH != 3 || g.KW != 3) return false;
  if (g.iy_mul != 1 || g.ix_mul != 1 || g.oy_mul != 1 || g.ox_mul != 1 || g.oy_add != 0 || g.ox_add != 0) return false;
  if (g.ky0 != 0 || g.kx0 != 0 || g.kstep != 1 || g.in_affine) return false;
  if (g.in_cstride % 8 || g.in_coff % 8 || g.out_cstride % 8 || g.out_coff % 8) return false;
  if (!aligned16(in) || !aligned16(wgt) || !aligned16(out)) return false;
  if (g.addend && (!aligned16(g.addend) || g.add_cstride % 8)) return false;
  if ((long long)(g.IW + 20) * 20 * g.in_cstride * 2 >= 0x7fffffffLL) return false;    // 32-bit offsets inside a patch
  const long long tiles = (long long)g.B * ((g.MH + 15) / 16) * ((g.MW + 15) / 16);
  static const int min_tiles = [] { const char* e = getenv("JSPSR_CONV_RESIDENT_MIN"); return e ? atoi(e) : 1024; }();
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_resident_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_resident_kernel<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
  }
  const int grid = ntiles < ncu ? ntiles : ncu;
  if (g.sign > 0)
    hipLaunchKernelGGL(conv64_resident_kernel<1>, dim3(grid), dim3(R_NTH), R_LDS, s, static_cast<const __bf16*>(in),
                       static_cast<const __bf16*>(wgt), bias, static_cast<__bf16*>(out), stats, g, ntiles);
  else
    hipLaunchKernelGGL(conv64_resident_kernel<-1>, dim3(grid), dim3(R_NTH), R_LDS, s, static_cast<const __bf16*>(in),
                       static_cast<const __bf16*>(wgt), bias, static_cast<__bf16*>(out), stats, g, ntiles);
  return check_launch("conv64_resident");
}

}  // namespace jspsr
