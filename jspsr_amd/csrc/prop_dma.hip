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
//   * parameter gradients: per lane across the whole run, one fp64 reduction per workgroup (no atomics).
#include "prop_dma.h"

#include <cstdlib>
#include <initializer_list>

namespace {

constexpr int OPB = 7 * 1024;          // one row's operand buffer: up to 28 planes x 64 px x 4 B = 7 DMA pieces

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
  // elements between consecutive images of weight / offset (and of their gradients): 9 P / OC P for the two tensors of the
  // public boundary, 25 P for both when they are planes 0..8 / 9..24 of ONE (B,25,H,W) head tensor (SIG form)
  size_t wbs, obs;
  size_t ps;      // elements between consecutive planes of weight / offset and of their gradients: H W, or a padded pitch
};

// One tile t of a workgroup's run, operand / DEM buffers t % 2.  Two roles:
//   mover:   wait for its pieces of tile t | barrier | request tile t+1 into the other buffers
//   compute: barrier | tile t, one pixel per lane, results in place | backward: the row's 25|27 gradient planes out
// SPLIT = false: NW waves, each plays both roles for its row (8 waves per CU: LDS holds two 7 KB buffers per row).
// SPLIT = true:  NW compute waves (0 .. NW-1) + NW mover waves (NW .. 2 NW-1); mover NW + w loads what compute wave w
//                consumes, so a compute wave never pays the issue time of an LDS-DMA instruction (200-400 cycles each
//                while the memory pipeline is backed up), and the tile's bytes are requested as soon as the barrier falls.
// SIG: the nine affinity planes hold LOGITS (the generator head's raw output, spn.py:41-43): the sigmoid is applied here in
// fp32, and the backward writes d/d(logit) = d/d(affinity) * a (1 - a) -- the in-model form (jspsr_prop_logits_*).
template <int OC, int NW, bool BWD, bool NTL, bool SPLIT, int RP, bool SIG>
__global__ __launch_bounds__((SPLIT ? 2 : 1) * NW * 64, SPLIT ? 4 : 2) void prop_dma_kernel(const DmaArgs A) {
  using C = DmaCfg<NW, OPB, RP>;
  constexpr int LH = C::LH, TH = C::TH;
  constexpr int NPL = 9 + OC + (BWD ? 1 : 0);      // operand planes staged per row
  constexpr int NPIECE = (NPL + 3) / 4;
  constexpr int NOUTPL = 9 + OC;                   // gradient planes written per row (backward)
  constexpr int NOPIECE = (NOUTPL + 3) / 4;
  // vector-memory stores a valid row issues per tile -- the counted wait of the symmetric form relies on EVERY one of
  // them being issued, i.e. on each store piece having a lane whose predicate holds on every valid row: lane sq = sc = 0
  // (plane 4 i < NOUTPL for every i < NOPIECE, asserted below; x0 < W for every tile of the grid)
  constexpr int NST = BWD ? NOPIECE : 1;
  static_assert(4 * (NOPIECE - 1) < NOUTPL, "every store piece needs an always-active lane (counted vmcnt waits)");
  static_assert(NPIECE <= 7, "operand buffer is 7 pieces");
  __shared__ __attribute__((aligned(1024))) char smem[C::SMEM];
  __shared__ double red[NW][NRED];

  const int lane = threadIdx.x & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool mover = !SPLIT || wave_all >= NW, computes = !SPLIT || wave_all < NW;
  const int wave = wave_all >= NW ? wave_all - NW : wave_all;      // the tile row this wave serves
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
#ifdef K1D_STAMPS
  unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
  const unsigned long long first_ = last_;
#endif

  // ---- mover: tile-invariant lane plans ------------------------------------------------------------------------------
  // operand piece i: plane 4 i + lane / 16, 16-byte chunk lane % 16 of the row segment
  const int lq = lane >> 4, lc = lane & 15;
  unsigned loff[NPIECE];
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) {
    const int p = 4 * i + lq;
    const int ch = p < 9 ? p : (p < 9 + OC ? p - 9 : 0);
    loff[i] = (unsigned)((size_t)ch * A.ps * 4) + lc * 16;
  }
  // DEM piece j of this wave: chunk q = (wave + j NW) * 64 + lane of the (LH x 80) tile, row q / 20, column 4 (q % 20)
  int doff[C::DPW], dcx[C::DPW];
#pragma unroll
  for (int j = 0; j < C::DPW; ++j) {
    const int q = (wave + j * NW) * 64 + lane;
    const int rr = q / (DLW / 4), cc = (q - rr * (DLW / 4)) * 4;
    doff[j] = (rr * W + cc) * 4;
    dcx[j] = q < C::CHUNKS ? cc : (1 << 30);      // beyond the tile: never inside the raster (the DMA lands zeros: ZPAD)
  }

  // the DEM tile + halo of tile (ib, ity, itx) into DEM buffer `dbuf`: this wave's pieces
  auto issue_dem = [&](int ib, int ity, int itx, int dbuf) __attribute__((always_inline)) {
    const int y0 = ity * TH, x0 = itx * DW;
    const unsigned long long db = reinterpret_cast<unsigned long long>(A.dem + (size_t)ib * P);
    // (wave-uniform by construction; readfirstlane makes the compiler see it, or the descriptor lands in VGPRs)
    const i32x4 desc = i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)db), __builtin_amdgcn_readfirstlane((int)((unsigned)(db >> 32) & 0xffffu)),
                             __builtin_amdgcn_readfirstlane((int)(unsigned)(P * 4)), 0x00020000};
    const int origin = ((y0 - HALO) * W + (x0 - HALO)) * 4;
#pragma unroll
    for (int j = 0; j < C::DPW; ++j) {
      const int piece = wave + j * NW;
      if (piece < C::PIECES) {     // wave-uniform
        // rows above / below the raster fail the range check by themselves (negative or >= P*4); columns do not
        const unsigned off = (unsigned)(x0 - HALO + dcx[j]) < (unsigned)W ? (unsigned)(origin + doff[j]) : 0xFFFFFFF0u;
        const unsigned dst = lds0 + dbuf * C::DEMB + piece * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(off), "s"(desc) : "memory");
      }
    }
  };
  // this wave's row y of image ib, 64 pixels from x0, into operand buffer `obuf`
  auto issue_row = [&](int ib, int y, int x0, int obuf) __attribute__((always_inline)) {
    if (y < H) {       // wave-uniform
      const size_t pix = (size_t)y * W + x0;
      const char* wb = reinterpret_cast<const char*>(A.weight + (size_t)ib * A.wbs + pix);
      const char* ob = reinterpret_cast<const char*>(A.offset + (size_t)ib * A.obs + pix);
      const char* gb = BWD ? reinterpret_cast<const char*>(A.gout + (size_t)ib * P + pix) : wb;
      const bool colok = x0 + lc * 4 < W;
      const unsigned dst0 = lds0 + 2 * C::DEMB + (obuf * NW + wave) * OPB;
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) {
        const int p = 4 * i + lq;
        const char* base = p < 9 ? wb : (p < 9 + OC ? ob : gb);
        if (colok && p < NPL) dma_piece<NTL>(dst0 + i * 1024, base + loff[i]);
      }
    }
  };

  // ---- compute: constants and running sums -----------------------------------------------------------------------------
  float wreg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wreg[k] = A.wk[k];
  const float bias = BWD ? 0.f : A.b0[0];
  // parameter-gradient partial sums: fp32 per LANE over the run's tiles (one value per tile: a few dozen additions
  // of like-sized terms), fp64 from there on (across lanes, waves, workgroups -- where the cancellation is)
  float dsum[NRED];
#pragma unroll
  for (int i = 0; i < NRED; ++i) dsum[i] = 0.f;

  // Slots: (tile t, pass p), p = 0 .. RP-1: rows ty TH + p NW + wave.  Operand buffers alternate per slot, DEM buffers per
  // tile.  Symmetric waves meet at ONE barrier per tile (the DEM tile is the only shared data; within a tile every wave
  // runs its RP rows at its own pace); split waves also hand rows from mover to compute wave, so they meet every slot.
  if (mover) {
    issue_dem(b, ty, tx, 0);
    issue_row(b, ty * TH + wave, tx * DW, 0);
  }
  bool counted = false;      // (not SPLIT) this wave issued NST stores behind the pieces it waits for next
  int buf = 0, dbuf = 0;
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t, dbuf ^= 1) {
    int nb = b, nty = ty, ntx = tx + 1;
    if (ntx == A.tiles_x) { ntx = 0; if (++nty == A.tiles_y) { nty = 0; ++nb; } }
#pragma unroll 1
  for (int ps = 0; ps < RP; ++ps, buf ^= 1) {
    K1D_STAMP(5);
    if (mover) {
      // this wave's pieces of this slot have landed (all but its NST younger stores of the previous slot)
      if (!SPLIT && counted) wait_vm<NST>(); else wait_vm<0>();
    }
    K1D_STAMP(0);
    if (SPLIT || ps == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the previous slot's LDS traffic is done (a raw barrier waits for nothing)
      __builtin_amdgcn_s_barrier();
    }
    K1D_STAMP(1);
    // the next slot into the other operand buffer; at a tile's first pass also the NEXT tile's DEM tile into the other
    // DEM buffer (every wave is past the barrier, i.e. past its reads of that buffer)
    if (mover) {
      if (ps == 0 && t + 1 < t_end) issue_dem(nb, nty, ntx, dbuf ^ 1);
      if (ps + 1 < RP) issue_row(b, ty * TH + (ps + 1) * NW + wave, tx * DW, buf ^ 1);
      else if (t + 1 < t_end) issue_row(nb, nty * TH + wave, ntx * DW, buf ^ 1);
    }
    K1D_STAMP(2);

    const int y0 = ty * TH, x0 = tx * DW;
    const int y = y0 + ps * NW + wave, x = x0 + lane;
    counted = y < H;
    if (computes && y < H) {       // wave-uniform
      float* ob = reinterpret_cast<float*>(smem + 2 * C::DEMB + (buf * NW + wave) * OPB);
      if (x < W) {
        const float* dl = reinterpret_cast<const float*>(smem + dbuf * C::DEMB);
        const float* img = A.dem + (size_t)b * P;
        const int ly0 = y0 - HALO, lx0 = x0 - HALO;
        float a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = SIG ? __builtin_amdgcn_rcpf(1.f + __expf(-ob[k * 64 + lane])) : ob[k * 64 + lane];   // v_exp + v_rcp
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += a[k];
        const float mean = s / 9.f;
        const float gj = BWD ? ob[(9 + OC) * 64 + lane] : 0.f;
        float gm[9];
        float acc = bias, gsum = 0.f;
        // TG taps at a time (3 bounds the live registers to 128 per lane: four waves per SIMD in the SPLIT form)
        constexpr int TG = SPLIT ? 3 : 9;
#pragma unroll
        for (int g3 = 0; g3 < 9 / TG; ++g3) {
          float py[TG], px[TG];
#pragma unroll
          for (int j = 0; j < TG; ++j) {
            const int k = TG * g3 + j;
            float oy = 0.f, ox = 0.f;
            if (OC == 18 || k != 4) {
              oy = ob[(9 + dch<OC>(k, 0)) * 64 + lane];
              ox = ob[(9 + dch<OC>(k, 1)) * 64 + lane];
            }
            py[j] = (float)(y - 1 + k / 3) + oy;
            px[j] = (float)(x - 1 + k % 3) + ox;
          }
          Corners cr[TG];
#ifdef K1D_NOCOMPUTE
#pragma unroll
          for (int j = 0; j < TG; ++j) { cr[j].v00 = py[j]; cr[j].v01 = px[j]; cr[j].v10 = cr[j].v11 = 0.f; cr[j].ly = cr[j].lx = 0.5f; }
#else
          gather_taps<LH, DLW, TG, LH * DLW>(dl, img, H, W, ly0, lx0, py, px, cr);
#endif
#pragma unroll
          for (int j = 0; j < TG; ++j) {
            const int k = TG * g3 + j;
            const Corners& c = cr[j];
            const float hy = 1.f - c.ly, hx = 1.f - c.lx;
            const float S = hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
            const float m = a[k] - mean;
            if (!BWD) {
              acc += wreg[k] * m * S;
            } else {
              const float dSdy = hx * (c.v10 - c.v00) + c.lx * (c.v11 - c.v01);
              const float dSdx = hy * (c.v01 - c.v00) + c.ly * (c.v11 - c.v10);
              const float coef = gj * wreg[k] * m;
              if (OC == 18 || k != 4) {      // in place: this tap's offsets are in registers, nobody else reads them
                ob[(9 + dch<OC>(k, 0)) * 64 + lane] = coef * dSdy;
                ob[(9 + dch<OC>(k, 1)) * 64 + lane] = coef * dSdx;
              }
              const float gmk = gj * wreg[k] * S;
              gm[k] = gmk;
              gsum += gmk;
              dsum[k] += gj * m * S;
            }
          }
        }
        if (!BWD) {
          A.out[(size_t)b * P + (size_t)y * W + x] = acc + A.scale * dl[(y - ly0) * DLW + (x - lx0)];
        } else {
          gsum /= 9.f;
#pragma unroll
          for (int k = 0; k < 9; ++k) ob[k * 64 + lane] = SIG ? (gm[k] - gsum) * a[k] * (1.f - a[k]) : gm[k] - gsum;
          dsum[9] += gj;
        }
      }
      K1D_STAMP(3);
      if (BWD) {
        // the row's finished gradient planes: four 256-byte plane segments per wave instruction, non-temporal (this wave
        // wrote them: the LDS operations of one wave complete in order)
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));      // recomputed per tile: hoisted out of the loop the seven lane offsets cost seven registers
        const int sq = lane_ >> 4, sc = lane_ & 15;
        const char* obc = reinterpret_cast<const char*>(ob);
        const size_t pix = (size_t)y * W + x0;
        char* gwb = reinterpret_cast<char*>(A.gweight + (size_t)b * A.wbs + pix);
        char* gob = reinterpret_cast<char*>(A.goffset + (size_t)b * A.obs + pix);
        const bool colok = x0 + sc * 4 < W;
        const unsigned plane_b = (unsigned)(A.ps * 4);
        // two batches (4 + 3 pieces): all seven lifted at once cost 28 registers the compute part has no room for
#pragma unroll
        for (int i0 = 0; i0 < NOPIECE; i0 += 4) {
          f32x4 v[4];
#pragma unroll
          for (int i = i0; i < i0 + 4 && i < NOPIECE; ++i) v[i - i0] = *reinterpret_cast<const f32x4*>(obc + i * 1024 + lane_ * 16);
#pragma unroll
          for (int i = i0; i < i0 + 4 && i < NOPIECE; ++i) {
            const int p = 4 * i + sq;
            if (colok && p < NOUTPL) {
              f32x4* dst = reinterpret_cast<f32x4*>((p < 9 ? gwb : gob) + ((unsigned)(p < 9 ? p : p - 9) * plane_b + sc * 16));
              if (K1D_NTS) __builtin_nontemporal_store(v[i - i0], dst); else *dst = v[i - i0];
            }
          }
          asm volatile("" ::: "memory");
        }
      }
    }
    K1D_STAMP(4);
  }
    b = nb; ty = nty; tx = ntx;
  }
#ifdef K1D_STAMPS
  if (BWD && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(A.partial + 4 + 4096 * NRED) + ((size_t)blockIdx.x * 2 * NW + wave_all) * 8;
    for (int i = 0; i < 6; ++i) o[i] = stamp[i];
    o[6] = __builtin_amdgcn_s_memtime() - first_;
    o[7] = (unsigned long long)(t_end - t_begin);
  }
#endif
  if (BWD) {
    if (computes) {
#pragma unroll
      for (int i = 0; i < NRED; ++i) {
        double v = (double)dsum[i];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[wave][i] = v;
      }
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
  int nw, rp, grid;
  bool ntl, split;
};

Plan make_plan(int B, int H, int W, bool bwd, DmaArgs& A, bool lab_shapes = true) {
  // defaults = what measured best on MI355X; the environment is for A/B measurements
  static const int nw_env = env_int("JSPSR_PROP_NW", 4);
  static const int ntl_env = env_int("JSPSR_PROP_NTL", -1);
  static const int split_env = env_int("JSPSR_PROP_SPLIT", -1);
  static const int wgs_env = env_int("JSPSR_PROP_WGS", 0);     // workgroups per CU (0 = what the LDS admits)
  Plan p;
  static const int rp_env = env_int("JSPSR_PROP_RP", 0);        // rows per wave per tile (1, 2 or 4)
  p.nw = (lab_shapes && nw_env == 8) ? 8 : 4;      // (the SIG form is built for 4-wave, one-row-per-wave tiles only)
  p.ntl = ntl_env < 0 ? true : ntl_env != 0;
  p.split = split_env < 0 ? !bwd : split_env != 0;
  // several rows per wave under one staged DEM tile: symmetric waves only (they then meet once per TILE; split waves hand
  // rows over every slot anyway, and their 128-register budget has no room for the second loop level)
  p.rp = (p.nw == 8 || p.split || !lab_shapes) ? 1 : (rp_env == 2 || rp_env == 4 ? rp_env : 1);      // measured equal within the noise (profiles/r03_k1_dma_rows_per_tile.txt): 1
  A.B = B; A.H = H; A.W = W;
  A.tiles_x = (W + DW - 1) / DW;
  A.tiles_y = (H + p.nw * p.rp - 1) / (p.nw * p.rp);
  const long long n = (long long)B * A.tiles_x * A.tiles_y;
  A.ntiles = (int)n;
  const int per_cu = wgs_env > 0 ? wgs_env : (p.nw == 8 ? 1 : 2);
  const long long cap = (long long)num_cus() * per_cu;
  p.grid = (int)(n < cap ? n : cap);
  if (p.grid > jspsr::prop_dma_max_rows()) p.grid = jspsr::prop_dma_max_rows();
  return p;
}

template <int OC, bool BWD, int NW, bool NTL, int RP, bool SIG = false>
void launch4(const Plan& p, const DmaArgs& A, hipStream_t s) {
  const dim3 grid(p.grid);
  if constexpr (RP == 1) {
    if (p.split) { hipLaunchKernelGGL((prop_dma_kernel<OC, NW, BWD, NTL, true, 1, SIG>), grid, dim3(2 * NW * 64), 0, s, A); return; }
  }
  hipLaunchKernelGGL((prop_dma_kernel<OC, NW, BWD, NTL, false, RP, SIG>), grid, dim3(NW * 64), 0, s, A);
}

template <int OC, bool BWD, bool NTL>
void launch3(const Plan& p, const DmaArgs& A, hipStream_t s) {
  if (p.nw == 8) launch4<OC, BWD, 8, NTL, 1>(p, A, s);
  else if (p.rp == 4) launch4<OC, BWD, 4, NTL, 4>(p, A, s);
  else if (p.rp == 2) launch4<OC, BWD, 4, NTL, 2>(p, A, s);
  else launch4<OC, BWD, 4, NTL, 1>(p, A, s);
}

template <int OC, bool BWD>
void launch(const Plan& p, const DmaArgs& A, hipStream_t s) {
  if (p.ntl) launch3<OC, BWD, true>(p, A, s); else launch3<OC, BWD, false>(p, A, s);
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
  A.wbs = (size_t)9 * H * W; A.obs = (size_t)oc * H * W; A.ps = (size_t)H * W;
  const Plan p = make_plan(B, H, W, false, A);
  if (oc == 18) launch<18, false>(p, A, s); else launch<16, false>(p, A, s);
  return check_launch("prop_forward (dma)");
}

int prop_dma_backward(const float* gout, const float* dem, const float* weight, const float* offset, int oc, const float* wk,
                      float* gweight, float* goffset, float* partial, int B, int H, int W, hipStream_t s) {
  DmaArgs A{};
  A.dem = dem; A.weight = weight; A.offset = offset; A.gout = gout; A.wk = wk; A.gweight = gweight; A.goffset = goffset;
  A.partial = partial;
  A.wbs = (size_t)9 * H * W; A.obs = (size_t)oc * H * W; A.ps = (size_t)H * W;
  const Plan p = make_plan(B, H, W, true, A);
  if (oc == 18) launch<18, true>(p, A, s); else launch<16, true>(p, A, s);
  return check_launch("prop_backward (dma)");
}

// The in-model form: ONE (B,25,H,W) fp32 tensor, planes 0..8 affinity LOGITS, 9..24 the sixteen learned offsets -- the
// generator head's output as jspsr_head_forward writes it; the gradient in the same layout.  Same kernel, SIG = true.
int prop_dma_logits_forward(const float* dem, const float* head, const float* wk, const float* b0, float scale, float* out,
                            int B, int H, int W, hipStream_t s) {
  DmaArgs A{};
  static const int pad = env_int("JSPSR_LAB_PLANE_PAD", 0);      // lab: plane pitch H W + pad floats (the caller allocates accordingly)
  A.ps = (size_t)H * W + pad;
  A.dem = dem; A.weight = head; A.offset = head + 9 * A.ps; A.wk = wk; A.b0 = b0; A.out = out; A.scale = scale;
  A.wbs = A.obs = 25 * A.ps;
  const Plan p = make_plan(B, H, W, false, A, false);
  if (p.ntl) launch4<16, false, 4, true, 1, true>(p, A, s); else launch4<16, false, 4, false, 1, true>(p, A, s);
  return check_launch("prop_logits_forward (dma)");
}

int prop_dma_logits_backward(const float* gout, const float* dem, const float* head, const float* wk, float* ghead, float* partial,
                             int B, int H, int W, hipStream_t s) {
  DmaArgs A{};
  static const int pad = env_int("JSPSR_LAB_PLANE_PAD", 0);
  A.ps = (size_t)H * W + pad;
  A.dem = dem; A.weight = head; A.offset = head + 9 * A.ps; A.gout = gout; A.wk = wk;
  A.gweight = ghead; A.goffset = ghead + 9 * A.ps; A.partial = partial;
  A.wbs = A.obs = 25 * A.ps;
  const Plan p = make_plan(B, H, W, true, A, false);
  if (p.ntl) launch4<16, true, 4, true, 1, true>(p, A, s); else launch4<16, true, 4, false, 1, true>(p, A, s);
  return check_launch("prop_logits_backward (dma)");
}

}  // namespace jspsr
