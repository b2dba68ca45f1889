// K1d -- the fused propagation step on planar fp32 operands (PostProcessor.forward and its autograd, reference
// models/components/spn.py:99-118) as a PERSISTENT, LDS-DMA staged stream.  Same arithmetic as prop.hip (which stays
// the general path: W % 4 != 0, unaligned operands); this file is the access shape.
//
// Why: the one-pixel-per-lane kernels of prop.hip touch 26 / 52 plane streams 4 bytes per lane and keep at most one
// row pass of loads in flight per wave; their traffic is 1.00x algorithmic but they stop at 0.58-0.63 of HBM peak.
// Here every byte moves 16 bytes per lane and a whole tile is in flight while the previous one is computed:
//   * a workgroup of NW waves per CU walks a contiguous run of 64 x NW pixel tiles (XCD-contiguous runs); wave w
//     owns row w of the tile;
//   * operands: `global_load_lds_dwordx4` (LDS-DMA, no VGPR destination): one wave instruction = 64 lanes x 16 B =
//     the 256-byte row segment of FOUR planes, landing lane-linear in the wave's private buffer [plane][64 px];
//     9 affinities + 16|18 offsets (+ grad_out) = 25..28 planes = 7 pieces per row;
//   * the DEM tile + 8-pixel halo: `buffer_load_dwordx4 ... lds`, one piece per wave; a lane outside the raster
//     fails the descriptor's range check and the DMA writes zeros -- the sampler's border rule;
//   * double buffered: tile t+1's pieces are issued right after the barrier that opens tile t and stay in flight
//     behind counted `s_waitcnt vmcnt(N)` (inline asm: behind the builtin hipcc would drain them with vmcnt(0) at the
//     next LDS read); one raw s_barrier per tile (the DEM tile is the only shared data);
//   * compute: one pixel per lane out of LDS (conflict-free ds_read_b32, plane = immediate offset);
//   * backward: the 25|27 gradient planes go back IN PLACE over the operands in LDS and leave as 16-byte
//     non-temporal stores, again four plane segments per wave instruction;
//   * parameter gradients: fp64 per lane across the whole run, one reduction per workgroup (no atomics).
#include "prop_tile.h"

#include <cstdlib>
#include <initializer_list>

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DW = 64;                 // tile width in pixels = one wave's row
constexpr int DLW = DW + 2 * HALO;     // staged DEM row: 80 floats
constexpr int OPB = 7 * 1024;          // one wave's operand buffer: up to 28 planes x 64 px x 4 B = 7 DMA pieces

struct DmaArgs {
  const float* dem;
  const float* weight;
  const float* offset;
  const float* gout;
  const float* wk;
  const float* b0;
  float* out;
  float* gweight;
  float* goffset;
  float* partial;      // rows of NRED floats behind a 16-byte header holding the row count
  float scale;
  int B, H, W, tiles_x, tiles_y, ntiles;
};

template <int NW>
struct DmaCfg {
  static constexpr int LH = NW + 2 * HALO;
  static constexpr int CHUNKS = LH * (DLW / 4);        // 16-byte chunks of the DEM tile
  static constexpr int PIECES = (CHUNKS + 63) / 64;
  static constexpr int DPW = (PIECES + NW - 1) / NW;   // DEM pieces per wave
  static constexpr int DEMB = PIECES * 1024;
  static constexpr int SMEM = 2 * DEMB + 2 * NW * OPB;
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

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// The nine taps' corner fetches at once: 36 LDS reads issued back to back at clamped indices (zeroed by select when the
// sample lies outside tile + halo), then ONE wave-level test for the rare lanes that need the bounds-checked global
// fallback (a tap more than HALO pixels outside the tile but still near the raster).  Same values as corners_fast().
template <int LH, int LW>
__device__ __forceinline__ void gather9(const float* __restrict__ lds, const float* __restrict__ img, int H, int W, int ly0, int lx0,
                                        const float (&py)[9], const float (&px)[9], Corners (&c)[9]) {
  unsigned fbmask = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float fy = floorf(py[k]), fx = floorf(px[k]);
    c[k].ly = py[k] - fy;
    c[k].lx = px[k] - fx;
    // fmaxf / fminf return the non-NaN operand: NaN and +-inf coordinates become huge finite ones (out of every range)
    const int y0 = (int)fminf(fmaxf(fy, -1.0e9f), 1.0e9f), x0 = (int)fminf(fmaxf(fx, -1.0e9f), 1.0e9f);
    const int ry = y0 - ly0, rx = x0 - lx0;
    const bool inl = (unsigned)ry < (unsigned)(LH - 1) && (unsigned)rx < (unsigned)(LW - 1);
    const bool near = (py[k] > -2.f) && (py[k] < (float)(H + 1)) && (px[k] > -2.f) && (px[k] < (float)(W + 1));   // false for NaN
    const float* p = lds + (inl ? ry * LW + rx : 0);
    const float t00 = p[0], t01 = p[1], t10 = p[LW], t11 = p[LW + 1];
    c[k].v00 = inl ? t00 : 0.f;
    c[k].v01 = inl ? t01 : 0.f;
    c[k].v10 = inl ? t10 : 0.f;
    c[k].v11 = inl ? t11 : 0.f;
    if (near && !inl) fbmask |= 1u << k;
    if (!near && !inl) c[k].ly = c[k].lx = 0.f;   // keep inf/nan coordinates out of the arithmetic: the tap contributes 0
  }
  if (__builtin_amdgcn_ballot_w64(fbmask != 0) != 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if ((fbmask >> k) & 1u) {
        const int y0 = (int)floorf(py[k]), x0 = (int)floorf(px[k]);      // near the raster: finite and small
        const bool y0ok = (unsigned)y0 < (unsigned)H, y1ok = (unsigned)(y0 + 1) < (unsigned)H;
        const bool x0ok = (unsigned)x0 < (unsigned)W, x1ok = (unsigned)(x0 + 1) < (unsigned)W;
        const float* q = img + (ptrdiff_t)y0 * W + x0;
        if (y0ok && x0ok) c[k].v00 = q[0];
        if (y0ok && x1ok) c[k].v01 = q[1];
        if (y1ok && x0ok) c[k].v10 = q[W];
        if (y1ok && x1ok) c[k].v11 = q[W + 1];
      }
    }
  }
}

template <int OC, int NW, bool BWD, bool NTL>
__global__ __launch_bounds__(NW * 64) void prop_dma_kernel(const DmaArgs A) {
  using C = DmaCfg<NW>;
  constexpr int LH = C::LH;
  constexpr int NPL = 9 + OC + (BWD ? 1 : 0);      // operand planes staged per row
  constexpr int NPIECE = (NPL + 3) / 4;
  constexpr int NOUTPL = 9 + OC;                   // gradient planes written per row (backward)
  constexpr int NOPIECE = (NOUTPL + 3) / 4;
  constexpr int NST = BWD ? NOPIECE : 1;           // vector-memory stores a valid row issues per tile
  static_assert(NPIECE <= 7, "operand buffer is 7 pieces");
  __shared__ __attribute__((aligned(1024))) char smem[C::SMEM];
  __shared__ double red[NW][NRED];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  const int H = A.H, W = A.W;
  const size_t P = (size_t)H * W;

  // this workgroup's contiguous run of tiles (neighbouring runs on one XCD: they share halo rows in its L2)
  const int G = gridDim.x;
  const int run = jspsr::xcd_contiguous(blockIdx.x, G);
  const int tq = A.ntiles / G, trem = A.ntiles - tq * G;       // runs differ by at most one tile
  const int t_begin = run * tq + min(run, trem), t_end = t_begin + tq + (run < trem ? 1 : 0);
  int b, ty, tx;
  {
    const int per_img = A.tiles_x * A.tiles_y;
    b = t_begin / per_img;
    const int r = t_begin - b * per_img;
    ty = r / A.tiles_x;
    tx = r - ty * A.tiles_x;
  }

  // ---- tile-invariant lane plans ------------------------------------------------------------------------------------
  // operand / gradient piece i: plane 4 i + lane / 16, 16-byte chunk lane % 16 of the row segment
  const int lq = lane >> 4, lc = lane & 15;
  unsigned loff[NPIECE];
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) {
    const int p = 4 * i + lq;
    const int ch = p < 9 ? p : (p < 9 + OC ? p - 9 : 0);
    loff[i] = (unsigned)((size_t)ch * P * 4) + lc * 16;
  }
  // DEM piece j of this wave: chunk q = (wave + j NW) * 64 + lane of the (LH x 80) tile, row q / 20, column 4 (q % 20)
  int doff[C::DPW], dcx[C::DPW];
#pragma unroll
  for (int j = 0; j < C::DPW; ++j) {
    const int q = (wave + j * NW) * 64 + lane;
    const int rr = q / (DLW / 4), cc = (q - rr * (DLW / 4)) * 4;
    doff[j] = (rr * W + cc) * 4;
    dcx[j] = q < C::CHUNKS ? cc : (1 << 30);      // beyond the tile: never inside the raster
  }

  auto issue_tile = [&](int ib, int ity, int itx, int buf) __attribute__((always_inline)) {
    const int y0 = ity * NW, x0 = itx * DW;
    {
      const unsigned long long db = reinterpret_cast<unsigned long long>(A.dem + (size_t)ib * P);
      const i32x4 desc = i32x4{(int)(unsigned)db, (int)((unsigned)(db >> 32) & 0xffffu), (int)(unsigned)(P * 4), 0x00020000};
      const int origin = ((y0 - HALO) * W + (x0 - HALO)) * 4;
#pragma unroll
      for (int j = 0; j < C::DPW; ++j) {
        const int piece = wave + j * NW;
        if (piece < C::PIECES) {     // wave-uniform
          // rows above / below the raster fail the range check by themselves (negative or >= P*4); columns do not
          const unsigned off = (unsigned)(x0 - HALO + dcx[j]) < (unsigned)W ? (unsigned)(origin + doff[j]) : 0xFFFFFFF0u;
          const unsigned dst = lds0 + buf * C::DEMB + piece * 1024;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(off), "s"(desc) : "memory");
        }
      }
    }
    const int y = y0 + wave;
    if (y < H) {       // wave-uniform
      const size_t pix = (size_t)y * W + x0;
      const char* wb = reinterpret_cast<const char*>(A.weight + (size_t)ib * 9 * P + pix);
      const char* ob = reinterpret_cast<const char*>(A.offset + (size_t)ib * OC * P + pix);
      const char* gb = BWD ? reinterpret_cast<const char*>(A.gout + (size_t)ib * P + pix) : wb;
      const bool colok = x0 + lc * 4 < W;
      const unsigned dst0 = lds0 + 2 * C::DEMB + (buf * NW + wave) * OPB;
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) {
        const int p = 4 * i + lq;
        const char* base = p < 9 ? wb : (p < 9 + OC ? ob : gb);
        if (colok && p < NPL) dma_piece<NTL>(dst0 + i * 1024, base + loff[i]);
      }
    }
  };

  float wreg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wreg[k] = A.wk[k];
  const float bias = BWD ? 0.f : A.b0[0];
  double dsum[NRED];
#pragma unroll
  for (int i = 0; i < NRED; ++i) dsum[i] = 0.0;

#ifdef K1D_STAMPS
  unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
  const unsigned long long first_ = last_;
#endif
  issue_tile(b, ty, tx, 0);
  bool counted = false;      // the previous tile of this wave issued its NST stores after the pieces waited for here
  int buf = 0;
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
    // this wave's pieces of tile t have landed (all but the NST younger stores of tile t-1); then everybody's
    K1D_STAMP(5);
    if (counted) wait_vm<NST>(); else wait_vm<0>();
    K1D_STAMP(0);
    __builtin_amdgcn_s_barrier();
    K1D_STAMP(1);
    // tile t+1 into the other buffers: every wave is past its reads of them (tile t-1) -- it is past the barrier
    int nb = b, nty = ty, ntx = tx + 1;
    if (ntx == A.tiles_x) { ntx = 0; if (++nty == A.tiles_y) { nty = 0; ++nb; } }
    if (t + 1 < t_end) issue_tile(nb, nty, ntx, buf ^ 1);
    K1D_STAMP(2);

    const int y0 = ty * NW, x0 = tx * DW;
    const int y = y0 + wave, x = x0 + lane;
    counted = y < H;
    if (y < H) {       // wave-uniform
      const float* dl = reinterpret_cast<const float*>(smem + buf * C::DEMB);
      float* ob = reinterpret_cast<float*>(smem + 2 * C::DEMB + (buf * NW + wave) * OPB);
      const float* img = A.dem + (size_t)b * P;
      const int ly0 = y0 - HALO, lx0 = x0 - HALO;
      const size_t pix = (size_t)y * W + x0;
      if (x < W) {
        float a[9], oy[9], ox[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = ob[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if (OC == 18 || k != 4) {
            oy[k] = ob[(9 + dch<OC>(k, 0)) * 64 + lane];
            ox[k] = ob[(9 + dch<OC>(k, 1)) * 64 + lane];
          } else {
            oy[k] = ox[k] = 0.f;
          }
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += a[k];
        const float mean = s / 9.f;
        float py[9], px[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          py[k] = (float)(y - 1 + k / 3) + oy[k];
          px[k] = (float)(x - 1 + k % 3) + ox[k];
        }
        Corners cr[9];
#ifdef K1D_NOCOMPUTE
#pragma unroll
        for (int k = 0; k < 9; ++k) { cr[k].v00 = py[k]; cr[k].v01 = px[k]; cr[k].v10 = cr[k].v11 = 0.f; cr[k].ly = cr[k].lx = 0.5f; }
#else
        gather9<LH, DLW>(dl, img, H, W, ly0, lx0, py, px, cr);
#endif
        if (!BWD) {
          float acc = bias;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const Corners& c = cr[k];
            const float hy = 1.f - c.ly, hx = 1.f - c.lx;
            const float S = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
            acc += wreg[k] * (a[k] - mean) * S;
          }
          A.out[(size_t)b * P + pix + lane] = acc + A.scale * dl[(y - ly0) * DLW + (x - lx0)];
        } else {
          const float gj = ob[(9 + OC) * 64 + lane];
          float gm[9], gy[9], gx[9];
          float gsum = 0.f;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const Corners& c = cr[k];
            const float hy = 1.f - c.ly, hx = 1.f - c.lx;
            const float S = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
            const float dSdy = hx * (c.v10 - c.v00) + c.lx * (c.v11 - c.v01);
            const float dSdx = hy * (c.v01 - c.v00) + c.ly * (c.v11 - c.v10);
            const float m = a[k] - mean;
            const float coef = gj * wreg[k] * m;
            gy[k] = coef * dSdy;
            gx[k] = coef * dSdx;
            const float gmk = gj * wreg[k] * S;
            gm[k] = gmk;
            gsum += gmk;
            dsum[k] += (double)(gj * m * S);
          }
          gsum /= 9.f;
          // results over the operands, in place (every operand of this pixel is in registers by now)
#pragma unroll
          for (int k = 0; k < 9; ++k) ob[k * 64 + lane] = gm[k] - gsum;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            if (OC == 18 || k != 4) {
              ob[(9 + dch<OC>(k, 0)) * 64 + lane] = gy[k];
              ob[(9 + dch<OC>(k, 1)) * 64 + lane] = gx[k];
            }
          }
          dsum[9] += (double)gj;
        }
      }
      K1D_STAMP(3);
      if (BWD) {
        // the row's gradient planes, four 256-byte plane segments per wave instruction (same wave wrote them: LDS
        // operations of one wave complete in order)
        char* gwb = reinterpret_cast<char*>(A.gweight + (size_t)b * 9 * P + pix);
        char* gob = reinterpret_cast<char*>(A.goffset + (size_t)b * OC * P + pix);
        const bool colok = x0 + lc * 4 < W;
#pragma unroll
        for (int i = 0; i < NOPIECE; ++i) {
          const int p = 4 * i + lq;
          if (colok && p < NOUTPL) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ob) + i * 1024 + lane * 16);
            char* dst = (p < 9 ? gwb : gob) + loff[i];
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
          }
        }
      }
    }
    K1D_STAMP(4);
    b = nb; ty = nty; tx = ntx;
  }
#ifdef K1D_STAMPS
  if (BWD && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(A.partial + 4 + 4096 * NRED) + ((size_t)blockIdx.x * NW + wave) * 8;
    for (int i = 0; i < 6; ++i) o[i] = stamp[i];
    o[6] = __builtin_amdgcn_s_memtime() - first_;
    o[7] = (unsigned long long)(t_end - t_begin);
  }
#endif

  if (BWD) {
#pragma unroll
    for (int i = 0; i < NRED; ++i) {
      double v = dsum[i];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
      if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NRED) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w][threadIdx.x];
      A.partial[4 + (size_t)blockIdx.x * NRED + threadIdx.x] = (float)v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int*>(A.partial)[0] = G;
  }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

int num_cus() {
  static const int n = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    return cu;
  }();
  return n;
}

}  // namespace
namespace jspsr { int prop_dma_max_rows(); }
namespace {

struct Plan {
  int nw, grid;
  bool ntl;
};

Plan make_plan(int B, int H, int W, DmaArgs& A) {
  static const int nw_env = env_int("JSPSR_PROP_NW", 8);
  static const int ntl_env = env_int("JSPSR_PROP_NTL", 1);
  static const int wgs_env = env_int("JSPSR_PROP_WGS", 0);     // workgroups per CU (0 = what the LDS admits)
  Plan p;
  p.nw = nw_env == 4 ? 4 : 8;
  p.ntl = ntl_env != 0;
  A.B = B; A.H = H; A.W = W;
  A.tiles_x = (W + DW - 1) / DW;
  A.tiles_y = (H + p.nw - 1) / p.nw;
  const long long n = (long long)B * A.tiles_x * A.tiles_y;
  A.ntiles = (int)n;
  const int per_cu = wgs_env > 0 ? wgs_env : (p.nw == 8 ? 1 : 2);
  const long long cap = (long long)num_cus() * per_cu;
  p.grid = (int)(n < cap ? n : cap);
  if (p.grid > jspsr::prop_dma_max_rows()) p.grid = jspsr::prop_dma_max_rows();
  return p;
}

template <int OC, bool BWD>
void launch(const Plan& p, const DmaArgs& A, hipStream_t s) {
  const dim3 grid(p.grid);
  if (p.nw == 8) {
    if (p.ntl) hipLaunchKernelGGL((prop_dma_kernel<OC, 8, BWD, true>), grid, dim3(512), 0, s, A);
    else       hipLaunchKernelGGL((prop_dma_kernel<OC, 8, BWD, false>), grid, dim3(512), 0, s, A);
  } else {
    if (p.ntl) hipLaunchKernelGGL((prop_dma_kernel<OC, 4, BWD, true>), grid, dim3(256), 0, s, A);
    else       hipLaunchKernelGGL((prop_dma_kernel<OC, 4, BWD, false>), grid, dim3(256), 0, s, A);
  }
}

}  // namespace

namespace jspsr {

// Rows of parameter-gradient partial sums the DMA backward may write (= the cap on its grid).
int prop_dma_max_rows() { return 4096; }

// Shapes and pointers the DMA path takes: 16-byte rows and 32-bit lane offsets.
bool prop_dma_ok(int B, int H, int W, int oc, std::initializer_list<const void*> ptrs) {
  static const int on = env_int("JSPSR_PROP_DMA", 1);
  if (!on || W % 4 != 0) return false;
  const long long P = (long long)H * W;
  if (P * 4 * (oc + 10) >= (1LL << 32) || P * 4 >= (1LL << 30) || (long long)B * ((W + 63) / 64) * ((H + 3) / 4) > 0x7fffffffLL) return false;
  for (const void* q : ptrs)
    if (!aligned16(q)) return false;
  return true;
}

int prop_dma_forward(const float* dem, const float* weight, const float* offset, int oc, const float* wk, const float* b0,
                     float scale, float* out, int B, int H, int W, hipStream_t s) {
  DmaArgs A{};
  A.dem = dem; A.weight = weight; A.offset = offset; A.wk = wk; A.b0 = b0; A.out = out; A.scale = scale;
  const Plan p = make_plan(B, H, W, A);
  if (oc == 18) launch<18, false>(p, A, s); else launch<16, false>(p, A, s);
  return check_launch("prop_forward (dma)");
}

int prop_dma_backward(const float* gout, const float* dem, const float* weight, const float* offset, int oc, const float* wk,
                      float* gweight, float* goffset, float* partial, int B, int H, int W, hipStream_t s) {
  DmaArgs A{};
  A.dem = dem; A.weight = weight; A.offset = offset; A.gout = gout; A.wk = wk; A.gweight = gweight; A.goffset = goffset;
  A.partial = partial;
  const Plan p = make_plan(B, H, W, A);
  if (oc == 18) launch<18, true>(p, A, s); else launch<16, true>(p, A, s);
  return check_launch("prop_backward (dma)");
}

}  // namespace jspsr
