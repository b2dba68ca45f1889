// K1hd -- the head-fed propagation step (K1h, prop_head.hip: sigmoid + zero centre offset + PostProcessor.forward on
// the merged 1x1 head's NHWC output, reference models/components/spn.py:43,66-73,99-118) for bf16 heads, as a
// persistent LDS-DMA stream in the mould of prop_dma.hip.  What the benchmarked step runs.
//
// K1h gave each pixel to 4 neighbouring lanes (one 16-byte load each) and replicated the per-pixel arithmetic over
// the group; here a row segment of 64 pixels x 64 bytes is FOUR contiguous 1 KiB LDS-DMA pieces, one pixel per lane:
//   * a workgroup walks a contiguous run of 64 x NW pixel tiles; wave w (or mover wave NW + w) owns row w;
//   * per row: the 4 head pieces (+ 16 lanes of grad_out in the backward) land in the row's buffer; the DEM tile +
//     halo by `buffer_load ... lds` (zeros outside the raster); tile t+1 is requested while tile t is computed;
//   * a lane reads its pixel's 64 bytes with four ds_read_b128.  Pixel p's 16-byte chunk c sits in slot
//     c ^ ((p >> 2) & 3) of the pixel (swizzled through the DMA's per-lane SOURCE address -- the LDS image of a piece is
//     lane-linear): lanes p, p + 4, p + 8, p + 12 of a ds_read_b128 lane group then hit four different bank quads;
//   * backward: the head's gradient (sigmoid derivative included, unused channels exact zeros) replaces the head IN
//     PLACE and leaves as four 1 KiB non-temporal stores per row.
// Channel order of the head ("tap-major", ops.merge_heads): c = 4 t + j, t = the 8 learned taps (k = t < 4 ? t : t + 1),
// j = 0 affinity logit, 1 dy, 2 dx, 3: the centre tap's logit for t == 0, unused otherwise.
#include "prop_dma.h"

#include <cstdlib>
#include <initializer_list>

namespace {

constexpr int HOPB_F = 4 * 1024;       // one row's buffer, forward: 64 px x 64 B of head = 4 DMA pieces
constexpr int HOPB_B = 4 * 1024 + 256; // backward: + the row's 256 B of grad_out (a quarter piece) -- three workgroups fit a CU's 160 KB

struct HdArgs {
  const float* dem;
  const __bf16* head;
  const float* gout;
  const float* wk;
  const float* b0;
  float* out;
  __bf16* ghead;
  float* partial;      // rows of NRED floats behind a 16-byte header holding the row count
  float scale;
  int B, H, W, tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }   // v_exp + v_rcp

__device__ __forceinline__ void unpack8(const u32x4& raw, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(raw[i] << 16);
    f[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
  }
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <int NW, bool BWD, bool SPLIT>
__global__ __launch_bounds__((SPLIT ? 2 : 1) * NW * 64, SPLIT ? 4 : 2) void prop_head_dma_kernel(const HdArgs A) {
  constexpr int HOPB = BWD ? HOPB_B : HOPB_F;
  using C = DmaCfg<NW, HOPB>;
  constexpr int LH = C::LH;
  constexpr int NST = BWD ? 4 : 1;                 // vector-memory stores a valid row issues per tile
  __shared__ __attribute__((aligned(256))) char smem[C::SMEM];
  __shared__ double red[BWD ? NW : 1][NRED];

  const int lane = threadIdx.x & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool mover = !SPLIT || wave_all >= NW, computes = !SPLIT || wave_all < NW;
  const int wave = wave_all >= NW ? wave_all - NW : wave_all;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  const int H = A.H, W = A.W;
  const size_t P = (size_t)H * W;

  const int G = gridDim.x;
  const int run = jspsr::xcd_contiguous(blockIdx.x, G);
  const int tq = A.ntiles / G, trem = A.ntiles - tq * G;
  const int t_begin = run * tq + min(run, trem), t_end = t_begin + tq + (run < trem ? 1 : 0);
  int b, ty, tx;
  {
    const int per_img = A.tiles_x * A.tiles_y;
    b = t_begin / per_img;
    const int r = t_begin - b * per_img;
    ty = r / A.tiles_x;
    tx = r - ty * A.tiles_x;
  }

  // ---- mover: lane plans ---------------------------------------------------------------------------------------------
  // head piece i, lane L: LDS slot L & 3 of pixel 16 i + (L >> 2) receives that pixel's chunk (L & 3) ^ ((pixel >> 2) & 3)
  const int ppx = lane >> 2, pslot = lane & 3;
  unsigned hoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = 16 * i + ppx;
    hoff[i] = (unsigned)(p * 64 + ((pslot ^ ((p >> 2) & 3)) * 16));
  }
  int doff[C::DPW], dcx[C::DPW];
#pragma unroll
  for (int j = 0; j < C::DPW; ++j) {
    const int q = (wave + j * NW) * 64 + lane;
    const int rr = q / (DLW / 4), cc = (q - rr * (DLW / 4)) * 4;
    doff[j] = (rr * W + cc) * 4;
    dcx[j] = q < C::CHUNKS ? cc : (1 << 30);
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
        if (piece < C::PIECES) {
          const unsigned off = (unsigned)(x0 - HALO + dcx[j]) < (unsigned)W ? (unsigned)(origin + doff[j]) : 0xFFFFFFF0u;
          const unsigned dst = lds0 + buf * C::DEMB + piece * 1024;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(off), "s"(desc) : "memory");
        }
      }
    }
    const int y = y0 + wave;
    if (y < H) {
      const size_t pix = ((size_t)ib * H + y) * W + x0;
      const char* hb = reinterpret_cast<const char*>(A.head) + pix * 64;
      const unsigned dst0 = lds0 + 2 * C::DEMB + (buf * NW + wave) * HOPB;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (x0 + 16 * i + ppx < W) dma_piece<true>(dst0 + i * 1024, hb + hoff[i]);
      if (BWD) {
        const char* gb = reinterpret_cast<const char*>(A.gout + pix);
        if (lane < 16 && x0 + lane * 4 < W) dma_piece<true>(dst0 + 4096, gb + lane * 16);
      }
    }
  };

  float wreg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wreg[k] = A.wk[k];
  const float bias = BWD ? 0.f : A.b0[0];
  float dsum[NRED];      // fp32 per lane over the run's tiles, fp64 across lanes / waves / workgroups (prop_dma.hip)
#pragma unroll
  for (int i = 0; i < NRED; ++i) dsum[i] = 0.f;

  if (mover) issue_tile(b, ty, tx, 0);
  bool counted = false;
  int buf = 0;
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
    if (mover) {
      if (!SPLIT && counted) wait_vm<NST>(); else wait_vm<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int nb = b, nty = ty, ntx = tx + 1;
    if (ntx == A.tiles_x) { ntx = 0; if (++nty == A.tiles_y) { nty = 0; ++nb; } }
    if (mover && t + 1 < t_end) issue_tile(nb, nty, ntx, buf ^ 1);

    const int y0 = ty * NW, x0 = tx * DW;
    const int y = y0 + wave, x = x0 + lane;
    // the counted wait above assumes EVERY one of the row's NST stores was issued: in the backward a store piece whose 16
    // pixels all lie beyond W has no active lane and is skipped (execz), so a ragged last column tile (W - x0 < 64) waits
    // for everything instead (vmcnt(4) could otherwise return with the next tile's pieces still in flight)
    counted = y < H && (!BWD || x0 + DW <= W);
    if (computes && y < H) {
      char* ob = smem + 2 * C::DEMB + (buf * NW + wave) * HOPB;
      const int sw = (lane >> 2) & 3;                      // this pixel's chunk c lives in slot c ^ sw
      if (x < W) {
        const float* dl = reinterpret_cast<const float*>(smem + buf * C::DEMB);
        const float* img = A.dem + (size_t)b * P;
        const int ly0 = y0 - HALO, lx0 = x0 - HALO;
        // the pixel's 32 channels: chunk j = taps 2 j, 2 j + 1
        float f[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) unpack8(*reinterpret_cast<const u32x4*>(ob + lane * 64 + ((j ^ sw) * 16)), f[j]);
        const float gj = BWD ? reinterpret_cast<const float*>(ob + 4096)[lane] : 0.f;
        // affinities (fp32 sigmoid of the stored logits), tap k = 0..8 row-major, centre k = 4 from channel 3
        float a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int tt = k < 4 ? k : k - 1;
          a[k] = sigmoid_fast(k == 4 ? f[0][3] : f[tt >> 1][(tt & 1) * 4]);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += a[k];
        const float mean = s * (1.f / 9.f);
        const float dc = dl[(y - ly0) * DLW + (x - lx0)];     // dem[y][x]: the centre tap's sample, and the residual
        float gm[9], gyx[8][2];
        float acc = bias, gsum = 0.f;
#pragma unroll
        for (int g3 = 0; g3 < 3; ++g3) {
          // taps 3 g3 .. 3 g3 + 2; the centre tap (k = 4) samples dem[y][x] itself: no gather
          float py[3], px[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int k = 3 * g3 + j, tt = k < 4 ? k : k - 1;
            const float oy = k == 4 ? 0.f : f[tt >> 1][(tt & 1) * 4 + 1];
            const float ox = k == 4 ? 0.f : f[tt >> 1][(tt & 1) * 4 + 2];
            py[j] = (float)(y - 1 + k / 3) + oy;
            px[j] = (float)(x - 1 + k % 3) + ox;
          }
          Corners cr[3];
          gather_taps<LH, DLW, 3, LH * DLW>(dl, img, H, W, ly0, lx0, py, px, cr);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int k = 3 * g3 + j, tt = k < 4 ? k : k - 1;
            const Corners& c = cr[j];
            const float hy = 1.f - c.ly, hx = 1.f - c.lx;
            const float S = k == 4 ? dc : hy * hx * c.v00 + hy * c.lx * c.v01 + c.ly * hx * c.v10 + c.ly * c.lx * c.v11;
            const float m = a[k] - mean;
            if (!BWD) {
              acc += wreg[k] * m * S;
            } else {
              if (k != 4) {
                const float dSdy = hx * (c.v10 - c.v00) + c.lx * (c.v11 - c.v01);
                const float dSdx = hy * (c.v01 - c.v00) + c.ly * (c.v11 - c.v10);
                const float coef = gj * wreg[k] * m;
                gyx[tt][0] = coef * dSdy;
                gyx[tt][1] = coef * dSdx;
              }
              const float gmk = gj * wreg[k] * S;
              gm[k] = gmk;
              gsum += gmk;
              dsum[k] += gj * m * S;
            }
          }
        }
        if (!BWD) {
          A.out[(size_t)b * P + (size_t)y * W + x] = acc + A.scale * dc;
        } else {
          gsum *= (1.f / 9.f);
          dsum[9] += gj;
          float dl_[9];        // gradient of the logits: (gm - mean gm) * sigmoid'
#pragma unroll
          for (int k = 0; k < 9; ++k) dl_[k] = (gm[k] - gsum) * a[k] * (1.f - a[k]);
          // the head's gradient over the head, in place (same swizzled slots); unused channels exact zeros
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u32x4 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int tt = 2 * j + h, k = tt < 4 ? tt : tt + 1;
              o[2 * h] = pack2(dl_[k], gyx[tt][0]);
              o[2 * h + 1] = pack2(gyx[tt][1], tt == 0 ? dl_[4] : 0.f);
            }
            *reinterpret_cast<u32x4*>(ob + lane * 64 + ((j ^ sw) * 16)) = o;
          }
        }
      }
      if (BWD) {
        // the row's gradient: four contiguous 1 KiB pieces, un-swizzled through the destination address
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        const int sp = lane_ >> 2, ss = lane_ & 3;
        char* gh = reinterpret_cast<char*>(A.ghead) + (((size_t)b * H + y) * W + x0) * 64;
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4*>(ob + i * 1024 + lane_ * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = 16 * i + sp;
          if (x0 + p < W) __builtin_nontemporal_store(v[i], reinterpret_cast<u32x4*>(gh + p * 64 + ((ss ^ ((p >> 2) & 3)) * 16)));
        }
      }
    }
    b = nb; ty = nty; tx = ntx;
  }
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

template <bool BWD>
void launch(HdArgs& A, int B, int H, int W, hipStream_t s) {
  // defaults = what measured best on MI355X; the environment is for A/B measurements
  static const int split_env = env_int("JSPSR_PROP_HEAD_SPLIT", -1);
  static const int wgs_env = env_int("JSPSR_PROP_HEAD_WGS", 0);
  constexpr int NW = 4;
  // forward: symmetric waves, 3 workgroups per CU (31.1 us; mover / compute waves at 2 per CU: 32.2); backward: mover /
  // compute waves, 2 per CU (61.7 us; symmetric at 3 per CU: 63.2) -- profiles/r03_k1h_dma_variants.txt
  const bool split = split_env < 0 ? BWD : split_env != 0;
  A.B = B; A.H = H; A.W = W;
  A.tiles_x = (W + DW - 1) / DW;
  A.tiles_y = (H + NW - 1) / NW;
  const long long n = (long long)B * A.tiles_x * A.tiles_y;
  A.ntiles = (int)n;
  long long grid = (long long)num_cus() * (wgs_env > 0 ? wgs_env : (split ? 2 : 3));
  if (grid > n) grid = n;
  if (grid > 4096) grid = 4096;
  if (split) hipLaunchKernelGGL((prop_head_dma_kernel<NW, BWD, true>), dim3((unsigned)grid), dim3(2 * NW * 64), 0, s, A);
  else       hipLaunchKernelGGL((prop_head_dma_kernel<NW, BWD, false>), dim3((unsigned)grid), dim3(NW * 64), 0, s, A);
}

}  // namespace

namespace jspsr {

// bf16 heads with 16-byte rows of the fp32 companions (dem, out / grad_out): what the models produce
bool prop_head_dma_ok(int B, int H, int W, std::initializer_list<const void*> ptrs) {
  static const int on = env_int("JSPSR_PROP_HEAD_DMA", 1);
  if (!on || W % 4 != 0) return false;
  const long long P = (long long)H * W;
  if (P * 4 >= (1LL << 30) || (long long)B * ((W + 63) / 64) * ((H + 3) / 4) > 0x7fffffffLL) return false;
  for (const void* q : ptrs)
    if (!aligned16(q)) return false;
  return true;
}

int prop_head_dma_forward(const float* dem, const void* head, const float* wk, const float* b0, float scale, float* out, int B,
                          int H, int W, hipStream_t s) {
  HdArgs A{};
  A.dem = dem; A.head = static_cast<const __bf16*>(head); A.wk = wk; A.b0 = b0; A.scale = scale; A.out = out;
  launch<false>(A, B, H, W, s);
  return check_launch("prop_head_forward (dma)");
}

int prop_head_dma_backward(const float* gout, const float* dem, const void* head, const float* wk, void* ghead, float* partial,
                           int B, int H, int W, hipStream_t s) {
  HdArgs A{};
  A.dem = dem; A.head = static_cast<const __bf16*>(head); A.gout = gout; A.wk = wk; A.ghead = static_cast<__bf16*>(ghead);
  A.partial = partial;
  launch<true>(A, B, H, W, s);
  return check_launch("prop_head_backward (dma)");
}

}  // namespace jspsr
