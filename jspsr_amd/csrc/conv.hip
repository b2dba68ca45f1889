// K2 -- implicit-GEMM convolution (forward, data gradient, transposed conv) on MFMA.
// Design notes: conv_igemm.h.  Entry points: include/jspsr_hip.h (jspsr_conv2d_*, jspsr_pack_weight).
#include "conv_igemm.h"

#include <cstdlib>
#include <type_traits>

namespace {

using namespace jspsr;

constexpr int NT = 256;    // 4 waves
constexpr int NCH = 8;     // 16-byte chunks per LDS row per stage (128 B of K per row)
template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPC = 4; };
template <> struct Elem<__bf16> { static constexpr int EPC = 8; };

// How a wave multiplies, per storage type: the MFMA block edge, the LDS row pitch that makes its fragment reads
// conflict-free, which 16-byte chunk of which row a lane reads, and which accumulator element sits where.
//   fp32: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), 144-byte rows, lane (r = l & 31, h = l >> 5) reads chunk 2s + h
//   bf16: v_mfma_f32_16x16x32_bf16, 160-byte rows, lane (r = l & 15, q = l >> 4) reads chunk 4s + q.
// Round 3: bf16 moved from 32x32x16 to 16x16x32 products -- the same cycles, LDS reads and registers per flop, but the
// chip holds a higher clock on them (MI355X_MICROARCH.md 'DVFS give-back' (7); measured here as a timing build:
// profiles/r03_conv_patch_movers_mfma16_lab.txt, -3.5 % on the whole step).  A ds_read_b128 lane group holds the 16 rows
// of a block with chunk c for eight of them and c + 1 for the other eight: rows (base + r) * pitch + chunk are 16 distinct
// 16-byte slots of the 256-byte bank window for ANY base exactly when pitch / 16 = 2 (mod 4) -> 160 bytes.
template <typename T> struct Mma;
template <> struct Mma<float> {
  static constexpr int BLK = 32, ROWP = 144, NE = 16, SUB = NCH / 2, SUBB = 32;   // block edge, row pitch, acc elements, sub-steps per stage, bytes per sub-step
  using acc_t = f32x16;
  static __device__ __forceinline__ int frag_row(int lane) { return lane & 31; }
  static __device__ __forceinline__ int frag_chunk(int lane) { return lane >> 5; }
  static __device__ __forceinline__ int acc_row(int lane, int e) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 31; }
};
template <> struct Mma<__bf16> {
  static constexpr int BLK = 16, ROWP = 160, NE = 4, SUB = NCH / 4, SUBB = 64;
  using acc_t = f32x4;
  static __device__ __forceinline__ int frag_row(int lane) { return lane & 15; }
  static __device__ __forceinline__ int frag_chunk(int lane) { return lane >> 4; }
  static __device__ __forceinline__ int acc_row(int lane, int e) { return 4 * (lane >> 4) + e; }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 15; }
};

// bf16 on the 32x32x16 product (the form before round 3): kept for the single-buffered narrow tiles of the implicit-GEMM
// kernel, which are HBM / latency bound and live on four resident workgroups per SIMD at <= 128 VGPRs -- the 16x16 blocks
// hold twice the fragments per accumulator there (12 more registers -> 3 per SIMD: 64->64 1x1 at 8x512^2 0.104 -> 0.122 ms)
struct Mma32bf16 {
  static constexpr int BLK = 32, ROWP = 144, NE = 16, SUB = NCH / 2, SUBB = 32;
  using acc_t = f32x16;
  static __device__ __forceinline__ int frag_row(int lane) { return lane & 31; }
  static __device__ __forceinline__ int frag_chunk(int lane) { return lane >> 5; }
  static __device__ __forceinline__ int acc_row(int lane, int e) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 31; }
};
template <typename T, int NBUF> using IgemmMma = std::conditional_t<(NBUF == 1 && sizeof(T) == 2), Mma32bf16, Mma<T>>;

__device__ __forceinline__ void mma_chunk(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_chunk(f32x16& acc, const float4& a, const float4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_chunk(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

template <typename T> struct Frag;
template <> struct Frag<float> { using type = float4; };
template <> struct Frag<__bf16> { using type = bf16x8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }


// BatchNorm statistics from the accumulators (fp32, before rounding to the storage type): per output
// channel, sum and sum of squares over the tile's valid rows -> one partial row per M-tile,
// stats[(tile * 2 + {0,1}) * Cout + n].  The lanes that hold the same column (2 for the 32x32 products, 4 for the
// 16x16 ones) are folded by shuffles, the WGM wave rows through LDS; fixed order -> reproducible.
template <typename M, int MI, int NI, int WGM, int WTM, int WTN, int BN, typename OutPix>
__device__ __forceinline__ void stats_epilogue(const typename M::acc_t (&acc)[MI][NI], float* __restrict__ stats, char* scratch,
                                               int tile, int Cout, int n0, int wm, int wn, int lane,
                                               OutPix&& out_pixel, int zero_tile = -1) {
  float* red = reinterpret_cast<float*>(scratch);   // [WGM][2][BN]
  float s[NI], q[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) s[ni] = q[ni] = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < M::NE; ++e) {
      const int row = wm * WTM + mi * M::BLK + M::acc_row(lane, e);
      if (out_pixel(row) < 0) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float v = acc[mi][ni][e];
        s[ni] += v;
        q[ni] += v * v;
      }
    }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
    for (int d = 32; d >= M::BLK; d >>= 1) {
      s[ni] += __shfl_xor(s[ni], d, 64);
      q[ni] += __shfl_xor(q[ni], d, 64);
    }
    if (lane < M::BLK) {
      red[(wm * 2 + 0) * BN + wn * WTN + ni * M::BLK + lane] = s[ni];
      red[(wm * 2 + 1) * BN + wn * WTN + ni * M::BLK + lane] = q[ni];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BN; i += blockDim.x) {
    const int which = i / BN, col = i % BN;
    if (n0 + col >= Cout) continue;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WGM; ++w) t += red[(w * 2 + which) * BN + col];
    stats[((size_t)tile * 2 + which) * Cout + n0 + col] = t;
    if (zero_tile >= 0) stats[((size_t)zero_tile * 2 + which) * Cout + n0 + col] = 0.f;  // row of a tile folded into `tile`
  }
  __syncthreads();
}

// 16 bytes of T -> EPC floats
template <typename T>
__device__ __forceinline__ void unpack_chunk(const uint4& v, float (&f)[Elem<T>::EPC]) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(w[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
}

// 16 bytes of T plus 16 bytes of T, element-wise (fp32 add, one rounding): the epilogue's `+ addend`
template <typename T>
__device__ __forceinline__ uint4 add_packed(uint4 a, uint4 b) {
  if constexpr (sizeof(T) == 4) {
    float4 x = __builtin_bit_cast(float4, a), y = __builtin_bit_cast(float4, b);
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    return __builtin_bit_cast(uint4, x);
  } else {
    const unsigned* pa = &a.x;
    const unsigned* pb = &b.x;
    uint4 r;
    unsigned* pr = &r.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = __uint_as_float(pa[i] << 16) + __uint_as_float(pb[i] << 16);
      const float hi = __uint_as_float(pa[i] & 0xffff0000u) + __uint_as_float(pb[i] & 0xffff0000u);
      const __bf16 l = (__bf16)lo, h = (__bf16)hi;
      pr[i] = (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
    }
    return r;
  }
}

// max(x, 0) on 16 bytes of T
template <typename T>
__device__ __forceinline__ uint4 relu_packed(uint4 a) {
  if constexpr (sizeof(T) == 4) {
    float4 x = __builtin_bit_cast(float4, a);
    x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
    return __builtin_bit_cast(uint4, x);
  } else {
    unsigned* p = &a.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {           // bf16: sign bit set -> negative (or -0) -> 0
      if (p[i] & 0x00008000u) p[i] &= 0xffff0000u;
      if (p[i] & 0x80000000u) p[i] &= 0x0000ffffu;
    }
    return a;
  }
}

// UNI: Cin is a multiple of the stage depth BK, so every 16-byte chunk of a stage belongs to the
// same tap -> the tap walk is wave-uniform (scalar registers, scalar offset of the buffer loads).
// NBUF: LDS staging buffers.  2 = one barrier per stage; 1 = two barriers per stage but half the LDS,
// which doubles the resident workgroups for the narrow (N <= 64) tiles -- those layers are HBM/latency
// bound at full resolution (input read + output write dominate), so overlap across workgroups wins.
template <typename T, int BM, int BN, int WGM, int WGN, bool UNI, int NBUF>
__global__ __launch_bounds__(NT, NBUF == 1 ? 4 : 2) void conv_igemm_kernel(const T* __restrict__ in,
                                                          const T* __restrict__ wgt,
                                                          const float* __restrict__ bias,
                                                          T* __restrict__ out, float* __restrict__ stats, ConvGeom g) {
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int EPC = Elem<T>::EPC, BK = NCH * EPC;
  using M = IgemmMma<T, NBUF>;
  constexpr int ROWB = M::ROWP, BLK = M::BLK;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / BLK, NI = WTN / BLK;
  constexpr int A_IT = BM * NCH / NT, B_IT = (BN * NCH + NT - 1) / NT;
  static_assert(MI >= 1 && NI >= 1 && BM * NCH % NT == 0, "tile");
  using frag_t = typename Frag<T>::type;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int AS_BYTES = BM * ROWB, BS_BYTES = BN * ROWB;   // one buffer
  char* As = smem;                                  // [NBUF][BM][ROWB]
  char* Bs = smem + NBUF * AS_BYTES;                // [NBUF][BN][ROWB]

  // ---- tile decode ---------------------------------------------------------------------------
  // An M-tile is a TLH x 16 block of pixels of ONE image of the m-grid (row r of the tile is pixel
  // (ty0 + r/16, tx0 + r%16)): pixel coordinates come from shifts, no per-row table or division.
  constexpr int TLW = 16, TLH = BM / TLW;
  const int tid = threadIdx.x;
  const int ntn = (g.Cout + BN - 1) / BN;
  const int ttx = (g.MW + TLW - 1) / TLW, tty = (g.MH + TLH - 1) / TLH;
  const int t = g.accumulate ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, gridDim.x);  // (lab switch)
  const int n0 = (t % ntn) * BN;
  const int mt = t / ntn;
  const int txi = mt % ttx, tyi = (mt / ttx) % tty, bimg = mt / (ttx * tty);
  const int ty0 = tyi * TLH, tx0 = txi * TLW;

  // ---- staging plan -------------------------------------------------------------------------
  // This thread stages 16-byte chunk `ch` of tile rows r0 + 32*i, i.e. pixels (ty0 + r0/16 + 2i,
  // tx0 + r0%16).  Everything constant along K is folded into per-row 32-bit byte offsets relative to
  // a workgroup-uniform base (buffer descriptors: an out-of-range offset makes the hardware return
  // zeros without touching memory -- that is the zero padding and every other mask) and per-row
  // validity bits over the tap walk (separable: bit ty of ymask AND bit tx of xmask).
  const int ch = tid & (NCH - 1);
  const int r0 = tid >> 3;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int pix_bytes = g.in_cstride * (int)sizeof(T);
  const int iy_ref = ty0 * g.iy_mul + g.iy_add, ix_ref = tx0 * g.ix_mul + g.ix_add;   // tap (0,0) of tile row 0
  const long long refpix = ((long long)bimg * g.IH + iy_ref) * g.IW + ix_ref;
  const long long reach = (long long)(g.nty > 0 ? g.nty - 1 : 0) * g.IW + (g.ntx > 0 ? g.ntx - 1 : 0);
  const long long basepix = refpix - (g.sign < 0 ? reach : 0);   // lowest pixel any tap of row 0 touches
  const char* abase = reinterpret_cast<const char*>(in) + (basepix * g.in_cstride + g.in_coff) * (long long)sizeof(T);
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase), 0, 0xFFFFFF00u, 0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(wgt)), 0, 0xFFFFFF00u, 0x00020000);
  // taps q with 0 <= i0 + sign*q < extent form one contiguous run [lo, hi] of the walk
  auto run = [&](int i0, int extent, int n) -> unsigned {
    int lo, hi;
    if (g.sign > 0) { lo = -i0; hi = extent - 1 - i0; } else { lo = i0 - (extent - 1); hi = i0; }
    lo = lo < 0 ? 0 : lo;
    hi = hi > n - 1 ? n - 1 : hi;
    return hi < lo ? 0u : ((2u << hi) - 1u) & ~((1u << lo) - 1u);
  };
  const int sdx = r0 & (TLW - 1), sdy = r0 >> 4;                  // this thread's pixel column / first row
  const bool colok = tx0 + sdx < g.MW;
  const unsigned xmask = colok ? run(ix_ref + sdx * g.ix_mul, g.IW, g.ntx) : 0u;
  unsigned aoff[A_IT], ymask[A_IT], boff[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int dy = sdy + 2 * i;                                   // rows r0 + 32*i
    const bool rowok = colok && ty0 + dy < g.MH;
    ymask[i] = rowok ? run(iy_ref + dy * g.iy_mul, g.IH, g.nty) : 0u;
    // byte offset of the row's tap-(0,0) pixel from the reference pixel: >= 0 (rows ascend)
    const long long rel = ((long long)dy * g.iy_mul * g.IW + sdx * g.ix_mul) * pix_bytes;
    if (rel > 0xE0000000LL) ymask[i] = 0;   // cannot happen for tiles < 3.5 GiB; stay safe
    aoff[i] = (unsigned)rel + (UNI ? ch * 16u : 0u);
  }
  const int Ktot_w = g.KH * g.KW * g.Cin;  // packed weight row length
  const int K = g.nty * g.ntx * g.Cin;     // walked K (0 for a stride phase that no tap reaches)
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int r = r0 + 32 * i, n = n0 + r;
    boff[i] = (K > 0 && r < BN && n < g.Cout) ? (unsigned)((size_t)n * Ktot_w * sizeof(T)) + (UNI ? ch * 16u : 0u) : OOB;
  }
  const int KT = (K + BK - 1) / BK;

  // tap walk state: of the whole stage (UNI, scalar) or of this thread's chunk
  int ci = UNI ? 0 : ch * EPC, ty = 0, tx = 0;
  auto norm = [&]() {
    while (ci >= g.Cin) {
      ci -= g.Cin;
      if (++tx == g.ntx) { tx = 0; ++ty; }
    }
  };
  if (!UNI) norm();

  // two register sets: the loads of stage t+2 are issued before the MFMAs of stage t, so a global
  // load has two stages of MFMA time to land
  uint4 areg[2][A_IT], breg[2][B_IT];
  auto load_stage = [&](auto SET) {
    constexpr int set = decltype(SET)::value;
    if constexpr (UNI) {
      // past the end of K the walk simply wraps to tap 0: the data is staged but never multiplied
      const int wy = ty < g.nty ? ty : 0, wx = ty < g.nty ? tx : 0;
      // tap displacement measured from the lowest pixel of the walk, so it is never negative
      const int tp = wy * g.IW + wx;
      const int adelta = (g.sign > 0 ? tp : (int)reach - tp) * pix_bytes + ci * (int)sizeof(T);
      const int bdelta = (((g.ky0 + g.kstep * wy) * g.KW + (g.kx0 + g.kstep * wx)) * g.Cin + ci) * (int)sizeof(T);
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const bool ok = ((ymask[i] >> wy) & (xmask >> wx) & 1u) != 0;
        // adelta rides in the scalar offset: not part of the range check, added to the address
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(arsrc, ok ? aoff[i] : OOB, adelta, 0);
        areg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(brsrc, boff[i], bdelta, 0);
        breg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
      }
      ci += BK;
      if (ci >= g.Cin) {   // Cin % BK == 0: exactly one wrap
        ci = 0;
        if (++tx == g.ntx) { tx = 0; ++ty; }
      }
    } else {
      const bool kok = ty < g.nty;
      const int tp = ty * g.IW + tx;
      const unsigned adelta = (unsigned)((g.sign > 0 ? tp : (int)reach - tp) * pix_bytes + ci * (int)sizeof(T));
      const unsigned bdelta = (unsigned)((((g.ky0 + g.kstep * ty) * g.KW + (g.kx0 + g.kstep * tx)) * g.Cin + ci) * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const bool ok = kok && ((ymask[i] >> ty) & (xmask >> tx) & 1u);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(arsrc, ok ? aoff[i] + adelta : OOB, 0, 0);
        areg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(brsrc, (kok && boff[i] != OOB) ? boff[i] + bdelta : OOB, 0, 0);
        breg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
      }
      ci += BK;  // stages are always loaded in order
      norm();
    }
  };
  char* const a_st = As + r0 * ROWB + ch * 16;   // + buf*AS_BYTES + i*32*ROWB: immediates
  char* const b_st = Bs + r0 * ROWB + ch * 16;
  auto store_stage = [&](auto SET, auto BUF) {
    constexpr int set = decltype(SET)::value, buf = NBUF == 1 ? 0 : decltype(BUF)::value;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      *reinterpret_cast<uint4*>(a_st + buf * AS_BYTES + i * 32 * ROWB) = areg[set][i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (B_IT * 32 <= BN || r0 + 32 * i < BN)
        *reinterpret_cast<uint4*>(b_st + buf * BS_BYTES + i * 32 * ROWB) = breg[set][i];
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  typename M::acc_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < M::NE; ++e) acc[mi][ni][e] = 0.f;

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = M::frag_row(lane), lc = M::acc_col(lane);
  const char* const a_ld = As + (wm * WTM + lr) * ROWB + M::frag_chunk(lane) * 16;
  const char* const b_ld = Bs + (wn * WTN + lr) * ROWB + M::frag_chunk(lane) * 16;

  auto compute = [&](auto BUF) {
    constexpr int buf = NBUF == 1 ? 0 : decltype(BUF)::value;
#pragma unroll
    for (int s = 0; s < M::SUB; ++s) {
      frag_t a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const frag_t*>(a_ld + buf * AS_BYTES + mi * BLK * ROWB + s * M::SUBB);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const frag_t*>(b_ld + buf * BS_BYTES + ni * BLK * ROWB + s * M::SUBB);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma_chunk(acc[mi][ni], a[mi], b[ni]);
    }
  };
  // one pipeline step: [issue loads of stage t+2] -> MFMAs of stage t -> [stage t+1 regs -> LDS] -> barrier.
  // Stage t lives in register set t&1 and LDS buffer t&1.  No conditionals inside a step, so hipcc
  // counts the outstanding loads exactly and waits only for the older set (vmcnt(N), never a drain).
#ifndef CONV_ABLATE
#define CONV_ABLATE 0   // lab builds only (tools/lab): 1 = no global loads, 2 = no MFMA/LDS reads, 3 = no LDS writes
#endif
  auto step = [&](auto CUR, auto NXT) {
    if (CONV_ABLATE != 1 && CONV_ABLATE != 5) load_stage(CUR);   // stage t+2 -> set CUR (stage t's copy is already in LDS buffer CUR)
    if (CONV_ABLATE != 2) compute(CUR);      // MFMAs of stage t
    if (NBUF == 1) __syncthreads();          // single buffer: everyone is done reading stage t
    if (CONV_ABLATE != 3) store_stage(NXT, NXT);  // stage t+1 -> LDS buffer NXT
    __syncthreads();
  };
  if (CONV_ABLATE != 5) {
    load_stage(S0{});
    load_stage(S1{});
  }
  store_stage(S0{}, S0{});
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    step(S0{}, S1{});
    step(S1{}, S0{});
  }
  if (kt < KT) step(S0{}, S1{});

  // ---- epilogue ------------------------------------------------------------------------------
  // C row = (e&3) + 8*(e>>2) + 4*lh, C col = lr: a lane holds ONE channel of 16 rows, so direct stores
  // would be 2-/4-byte scatters.  The tile is transposed through LDS (the staging buffers are free
  // now) and written with 16-byte stores: every output row leaves as one contiguous run.
  // written pixel of tile row `row`, -1 if the row is outside the m-grid or the written raster
  auto out_pixel = [&](int row) -> long long {
    const int y = ty0 + (row >> 4), x = tx0 + (row & (TLW - 1));
    const int oy = y * g.oy_mul + g.oy_add, ox = x * g.ox_mul + g.ox_add;
    return (y < g.MH && x < g.MW && oy < g.OH && ox < g.OW) ? ((long long)bimg * g.OH + oy) * g.OW + ox : -1;
  };
  if (stats) {
    if (NBUF == 1) __syncthreads();
    stats_epilogue<M, MI, NI, WGM, WTM, WTN, BN>(acc, stats, smem, mt, g.Cout, n0, wm, wn, lane, out_pixel);
  }
  float bv[NI], sv[NI];
  int ncol[NI];
  const bool relu_first = g.relu && !g.addend, relu_last = g.relu && g.addend;   // ReLU is the last operation
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    ncol[ni] = n0 + wn * WTN + ni * BLK + lc;
    bv[ni] = (bias && ncol[ni] < g.Cout) ? bias[ncol[ni]] : 0.f;
    sv[ni] = (g.scale && ncol[ni] < g.Cout) ? g.scale[ncol[ni]] : 1.f;
  }
  constexpr int OPITCH = BN * (int)sizeof(T) + 16;       // bytes per staged output row (+16: bank spread)
  // the launcher sizes dynamic LDS as max(staging, BM * OPITCH)
  if (NBUF == 1) __syncthreads();
  char* Os = smem;
  // whole 16-byte pieces of channels: a last N tile that is only partly inside Cout still takes the vector path
  // (pieces at or beyond Cout are skipped in the copy loop)
  const bool vec_ok = (g.Cout % EPC == 0) && ((g.out_cstride * (int)sizeof(T)) % 16 == 0) &&
                      (((g.out_coff + n0) * (int)sizeof(T)) % 16 == 0) &&
                      ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) &&
                      (!g.addend || (((reinterpret_cast<uintptr_t>(g.addend) & 15u) == 0) &&
                                     ((g.add_cstride * (int)sizeof(T)) % 16 == 0) && ((n0 * (int)sizeof(T)) % 16 == 0)));
  if (vec_ok) {
    constexpr int CPRO = BN * (int)sizeof(T) / 16;       // 16-byte chunks per output row
    constexpr int O_IT = BM * CPRO / NT, A_PRE = O_IT <= 8 ? O_IT : 1;
    static_assert(BM * CPRO % NT == 0, "epilogue rows");
    // the addend's 16-byte pieces are requested up front: they fly during the accumulator -> LDS transpose and
    // the barrier instead of one exposed memory latency per piece in the copy loop below
    uint4 areg[A_PRE];
    const bool add_pre = g.addend && O_IT <= 8;
    if (add_pre) {
#pragma unroll
      for (int j = 0; j < A_PRE; ++j) {
        const int i = tid + j * NT, row = i / CPRO, c16 = i % CPRO;
        const long long opix = out_pixel(row);
        areg[j] = make_uint4(0, 0, 0, 0);
        if (opix >= 0 && n0 + c16 * EPC < g.Cout)
          areg[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(g.addend) +
                                                    (opix * g.add_cstride + n0) * (long long)sizeof(T) + c16 * 16);
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < M::NE; ++e) {
        const int row = wm * WTM + mi * BLK + M::acc_row(lane, e);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float v = acc[mi][ni][e] * sv[ni] + bv[ni];
          if (relu_first) v = fmaxf(v, 0.f);
          *reinterpret_cast<T*>(Os + row * OPITCH + (wn * WTN + ni * BLK + lc) * (int)sizeof(T)) = (T)v;
        }
      }
    __syncthreads();
    char* const obase = reinterpret_cast<char*>(out + g.out_coff + n0);
#pragma unroll
    for (int j = 0; j < O_IT; ++j) {
      const int i = tid + j * NT, row = i / CPRO, c16 = i % CPRO;
      const long long opix = out_pixel(row);
      if (opix < 0 || n0 + c16 * EPC >= g.Cout) continue;
      uint4 v = *reinterpret_cast<const uint4*>(Os + row * OPITCH + c16 * 16);
      if (CONV_ABLATE == 4 && v.x != 0x12345678u) continue;   // lab: no output stores
      if (add_pre)
        v = add_packed<T>(v, areg[j < A_PRE ? j : 0]);
      else if (g.addend)
        v = add_packed<T>(v, *reinterpret_cast<const uint4*>(static_cast<const char*>(g.addend) +
                                                              (opix * g.add_cstride + n0) * (long long)sizeof(T) + c16 * 16));
      if (relu_last) v = relu_packed<T>(v);
      *reinterpret_cast<uint4*>(obase + opix * g.out_cstride * (long long)sizeof(T) + c16 * 16) = v;
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int e = 0; e < M::NE; ++e) {
        const int row = wm * WTM + mi * BLK + M::acc_row(lane, e);
        const long long opix = out_pixel(row);
        if (opix < 0) continue;
        T* orow = out + opix * g.out_cstride + g.out_coff;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (ncol[ni] >= g.Cout) continue;
          float v = acc[mi][ni][e] * sv[ni] + bv[ni];
          if (relu_first) v = fmaxf(v, 0.f);
          if (g.addend) v = (float)(T)v + (float)static_cast<const T*>(g.addend)[opix * g.add_cstride + ncol[ni]];
          if (relu_last) v = fmaxf(v, 0.f);
          orow[ncol[ni]] = (T)v;
        }
      }
    }
  }
}

// Lab builds only (tools/lab/build_conv_variants.sh stamps="-DCONVLAB_STAMPS"): per-phase cycle sums of every wave of the
// patch kernel, added into one device array that jspsr_lab_conv_stamps() reads and clears; the product build has none of it.
#ifdef CONVLAB_STAMPS
constexpr int CONVLAB_SLOTS = 1 << 17;           // one row of 16 per wave (rows reused modulo: later launches overwrite)
__device__ unsigned long long convlab_stamp[CONVLAB_SLOTS * 16];
#define CL_NOW(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#if CONVLAB_STAMPS == 2      // per-stage phases (these stamps slow the kernel several times over: read the ratios only)
#define CL_STAMP(i) do { unsigned long long now_; CL_NOW(now_); cl_stamp[i] += now_ - cl_last; cl_last = now_; } while (0)
#else                        // 1: four stamps in a wave's life (start, first stage ready, main loop done, end)
#define CL_STAMP(i) do { } while (0)
#endif
#else
#define CL_STAMP(i) do { } while (0)
#endif
#if defined(CONVLAB_STAMPS) && CONVLAB_STAMPS == 3     // 3: the prologue and the epilogue in pieces (a dozen stamps per wave)
#define CL_STAMP3(i) do { unsigned long long now_; CL_NOW(now_); cl_stamp[i] += now_ - cl_last; cl_last = now_; } while (0)
#else
#define CL_STAMP3(i) do { } while (0)
#endif

__host__ __device__ constexpr int patch_it(int BM, int NTH) { return ((BM / 16 + 2) * 18 * NCH + NTH - 1) / NTH; }

// ---------------------------------------------------------------------------------------------
// Patch variant for unit-stride tap walks (forward stride-1 convs, every data-gradient /
// transposed-conv phase) with Cin a multiple of the stage depth.  The implicit-GEMM kernel above
// re-stages the A tile once per TAP (9 x for a 3x3); here the (TLH+nty-1) x (16+ntx-1) input patch
// of the current 64-/32-channel chunk is staged in LDS ONCE and every tap reads its A operand from
// it at a shifted row, so global->LDS traffic for A drops from ntaps x to ~1.4 x.  Only the weight
// tile still streams per tap (two-deep register prefetch).  The next chunk's patch is prefetched
// into registers at the first tap of the current chunk and written at its last.
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN == 8 ? 1 : 2) void conv_patch_kernel(const T* __restrict__ in, const T* __restrict__ wgt,
                                                          const float* __restrict__ bias, T* __restrict__ out,
                                                          float* __restrict__ stats, ConvGeom g) {
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves");
  constexpr int NTH = 64 * WGM * WGN, RPI = NTH / NCH;   // threads; patch pixels / weight rows per staging iteration
  constexpr int TLW = 16, TLH = BM / TLW;
  constexpr int EPC = Elem<T>::EPC, BK = NCH * EPC;
  using M = Mma<T>;
  constexpr int ROWB = M::ROWP, BLK = M::BLK;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / BLK, NI = WTN / BLK;
  constexpr int B_IT = (BN * NCH + NTH - 1) / NTH;
  constexpr int P_IT = patch_it(BM, NTH);         // patch chunks per thread: (TLH+2)*(16+2) pixels * 8 chunks <= P_IT * 256
  constexpr int MAXPIX = P_IT * NTH / NCH;    // 192 (8x16 tile) / 352 (16x16 tile) patch pixels
  constexpr int BS_BYTES = BN * ROWB;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  using frag_t = typename Frag<T>::type;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ps = smem;                          // [MAXPIX][ROWB]   input patch of the current channel chunk
  char* Bs = smem + MAXPIX * ROWB;          // [2][BN][ROWB]    weight tile of the current / next tap

#ifdef CONVLAB_STAMPS
  unsigned long long cl_entry, cl_stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cl_last = 0, cl_first = 0;
  CL_NOW(cl_entry);
  cl_last = cl_entry;
#endif
  const int tid = threadIdx.x;
  const int ntn = (g.Cout + BN - 1) / BN;
  const int ttx = (g.MW + TLW - 1) / TLW, tty = (g.MH + TLH - 1) / TLH;
  const int t = xcd_contiguous(blockIdx.x, gridDim.x);
  // (divisions by launch constants: common.h FastDiv -- five runtime divisions at the head of every workgroup were most
  // of the 3.7 k cycles a wave spent before its first load, profiles/r03_conv_patch_wave_life.txt)
  const int mt = (int)fastdiv((unsigned)t, g.fd_ntn);
  const int n0 = (t - mt * ntn) * BN;
  const int mrow = (int)fastdiv((unsigned)mt, g.fd_ttx), txi = mt - mrow * ttx;
  const int bimg = (int)fastdiv((unsigned)mrow, g.fd_tty), tyi = mrow - bimg * tty;
  const int ty0 = tyi * TLH, tx0 = txi * TLW;

  CL_STAMP3(4);                  // tile coordinates (first kernel arguments read)
  const int PW = TLW + g.ntx - 1, PH = TLH + g.nty - 1, npix = PW * PH;
  const int pix_bytes = g.in_cstride * (int)sizeof(T);
  // patch origin in the gathered raster (reversed walk: the patch starts nty-1 / ntx-1 pixels earlier)
  const int oy0 = ty0 + g.iy_add - (g.sign < 0 ? g.nty - 1 : 0);
  const int ox0 = tx0 + g.ix_add - (g.sign < 0 ? g.ntx - 1 : 0);
  const long long opix0 = ((long long)bimg * g.IH + oy0) * g.IW + ox0;    // may lie outside the raster
  const char* abase = reinterpret_cast<const char*>(in) + (opix0 * g.in_cstride + g.in_coff) * (long long)sizeof(T);
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase), 0, 0xFFFFFF00u, 0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(wgt)), 0, 0xFFFFFF00u, 0x00020000);

  // patch staging plan: this thread owns 16-byte chunk `ch` of patch pixels p0 + 32*i
  const int ch = tid & (NCH - 1);
  unsigned poff[P_IT];       // global byte offset of the pixel (+ chunk) from the patch origin, OOB if padding
  int plds[P_IT];            // LDS byte offset, -1 if this slot is beyond the patch
  {
    // patch pixel pp = py * PW + px, walked in steps of RPI: PW is 16, 17 or 18 (1..3 taps across), so the row / column
    // split needs no division -- at most RPI / 16 conditional subtractions at the start, one per step after
    const int qd = g.ntx == 1 ? RPI / 16 : g.ntx == 2 ? RPI / 17 : RPI / 18, rd = RPI - qd * PW;
    int pp = tid >> 3, py = 0, px = pp;
#pragma unroll
    for (int k = 0; k < RPI / 16; ++k)
      if (px >= PW) { px -= PW; ++py; }
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const bool inpatch = pp < npix;
      const int iy = oy0 + py, ix = ox0 + px;
      const bool ok = inpatch && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
      poff[i] = ok ? (unsigned)((py * g.IW + px) * pix_bytes) + ch * 16u : OOB;
      plds[i] = inpatch ? pp * ROWB + ch * 16 : -1;
      pp += RPI;
      py += qd;
      px += rd;
      if (px >= PW) { px -= PW; ++py; }
    }
  }
  const int r0 = tid >> 3;
  const int Ktot_w = g.KH * g.KW * g.Cin;
  const int ntaps = g.nty * g.ntx, nchunks = g.Cin / BK;
  const int KT = ntaps * nchunks;
  unsigned boff[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int r = r0 + RPI * i, n = n0 + r;
    boff[i] = (KT > 0 && r < BN && n < g.Cout) ? (unsigned)((size_t)n * Ktot_w * sizeof(T)) + ch * 16u : OOB;
  }

  uint4 preg[P_IT], breg[2][B_IT];
  // Input transform (g.in_affine): the gathered tensor is the PRE-normalisation output of a BatchNorm; its per-channel
  // (scale | shift) and the ReLU are applied here, between the patch registers and LDS, so the normalised activation
  // never exists in memory (conv -> BN -> ReLU -> conv, basics.py:111-117).  Padding (OOB slots) stays zero.
  float asc[EPC], ash[EPC];
  auto load_patch = [&](int chunk) __attribute__((always_inline)) {
    const int soff = chunk * BK * (int)sizeof(T);
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(arsrc, poff[i], soff, 0);
      preg[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    if (g.in_affine) {
      const float* ap = g.in_affine + chunk * BK + ch * EPC;
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        const float4 a = *reinterpret_cast<const float4*>(ap + e), b = *reinterpret_cast<const float4*>(ap + g.Cin + e);
        asc[e] = a.x; asc[e + 1] = a.y; asc[e + 2] = a.z; asc[e + 3] = a.w;
        ash[e] = b.x; ash[e + 1] = b.y; ash[e + 2] = b.z; ash[e + 3] = b.w;
      }
    }
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
    if (g.in_affine) {
      const bool rl = g.in_relu != 0;
#pragma unroll
      for (int i = 0; i < P_IT; ++i)
        if (plds[i] >= 0)
          *reinterpret_cast<uint4*>(Ps + plds[i]) = poff[i] != OOB ? affine_relu16<T>(preg[i], asc, ash, rl) : make_uint4(0u, 0u, 0u, 0u);
      return;
    }
#pragma unroll
    for (int i = 0; i < P_IT; ++i)
      if (plds[i] >= 0) *reinterpret_cast<uint4*>(Ps + plds[i]) = preg[i];
  };
  // weight-tile walk (chunk-major, taps inner): state of the NEXT stage to load
  int lty = 0, ltx = 0, lchunk = 0;
  auto load_b = [&](auto SET) {
    constexpr int set = decltype(SET)::value;
    const bool past = lchunk >= nchunks;       // past the end: re-read stage 0 (staged, never multiplied)
    const int wy = past ? 0 : lty, wx = past ? 0 : ltx, wc = past ? 0 : lchunk;
    const int bdelta = (((g.ky0 + g.kstep * wy) * g.KW + (g.kx0 + g.kstep * wx)) * g.Cin + wc * BK) * (int)sizeof(T);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(brsrc, boff[i], bdelta, 0);
      breg[set][i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    if (++ltx == g.ntx) { ltx = 0; if (++lty == g.nty) { lty = 0; ++lchunk; } }
  };
  char* const b_st = Bs + r0 * ROWB + ch * 16;
  auto store_b = [&](auto SET, auto BUF) {
    constexpr int set = decltype(SET)::value, buf = decltype(BUF)::value;
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (B_IT * RPI <= BN || r0 + RPI * i < BN)
        *reinterpret_cast<uint4*>(b_st + buf * BS_BYTES + i * RPI * ROWB) = breg[set][i];
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  typename M::acc_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < M::NE; ++e) acc[mi][ni][e] = 0.f;

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = M::frag_row(lane), lc = M::acc_col(lane);
  // A operand: MFMA row R of the tile is tile pixel (dy, dx) = (R / 16, (R + rot * dy) % 16); at tap (ty,tx) it reads
  // patch pixel (dy + oy(ty), dx + ox(tx)).  The rotation by `rot` per tile row undoes the skew of the patch pitch: a
  // 32-row MFMA block spans two tile rows PW = 16 + ntx - 1 patch pixels apart, and with 144-byte pixel rows two
  // lanes of one ds_read_b128 lane group share a bank exactly when their patch pixel indices agree mod 16 -- with
  // dx = R % 16 the second row's indices are shifted by PW - 16 against the first's and 2 of 16 lanes collide
  // (measured: 23 % of the kernel's LDS cycles were bank conflicts); rotated, every group covers 16 distinct residues.
  // (16-row blocks -- the bf16 products -- are one tile row each: nothing to undo, rot = 0)
  const int rot = BLK == 16 ? 0 : (16 - ((PW - TLW) & (TLW - 1))) & (TLW - 1);
  int a_ld[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = wm * WTM + mi * BLK + lr;
    a_ld[mi] = ((row >> 4) * PW + ((row + rot * (row >> 4)) & (TLW - 1))) * ROWB + M::frag_chunk(lane) * 16;
  }
  const char* const b_ld = Bs + (wn * WTN + lr) * ROWB + M::frag_chunk(lane) * 16;

  int cty = 0, ctx = 0, cchunk = 0;   // stage being computed
  auto compute = [&](auto BUF) {
    constexpr int buf = decltype(BUF)::value;
    const int oy = g.sign > 0 ? cty : g.nty - 1 - cty, ox = g.sign > 0 ? ctx : g.ntx - 1 - ctx;
    const int tapoff = (oy * PW + ox) * ROWB;
    // fragments of sub-step s+1 are requested before the MFMAs of sub-step s are issued, so an LDS read has a
    // whole sub-step of matrix work to land in (the scheduler, left alone, places each read right before its use)
    frag_t a[2][MI], b[2][NI];
    if constexpr (M::SUB == 2) {
      // 16x16x32 products: two k-steps of MI x NI products per stage.  Issued whole, a k-step is MI + NI = 8 reads in
      // front of 16 MFMAs -- all 16 reads of the stage before its first product, in a burst from every wave at once
      // (measured: +6 % wave cycles against the 32x32 schedule).  So each k-step is split by block rows into two halves:
      //   R(b k0, a-top k0) R(a-bottom k0) M(k0 top) R(b k1, a-top k1) M(k0 bottom) R(a-bottom k1) M(k1 top) M(k1 bottom)
      constexpr int MH = MI / 2;
      static_assert(MI % 2 == 0, "two block rows at least");
      auto rd_b = [&](int set, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[set][ni] = *reinterpret_cast<const frag_t*>(b_ld + buf * BS_BYTES + ni * BLK * ROWB + s * M::SUBB);
      };
      auto rd_a = [&](int set, int s, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = h * MH; mi < (h + 1) * MH; ++mi) a[set][mi] = *reinterpret_cast<const frag_t*>(Ps + a_ld[mi] + tapoff + s * M::SUBB);
      };
      auto mm = [&](int set, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = h * MH; mi < (h + 1) * MH; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma_chunk(acc[mi][ni], a[set][mi], b[set][ni]);
      };
      rd_b(0, 0); rd_a(0, 0, 0); rd_a(0, 0, 1);
      mm(0, 0);
      rd_b(1, 1); rd_a(1, 1, 0);
      mm(0, 1);
      rd_a(1, 1, 1);
      mm(1, 0);
      mm(1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, NI + MH, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, MH, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MH * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NI + MH, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MH * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, MH, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MH * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MH * NI, 0);
      return;
    }
    auto frags = [&](int set, int s) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[set][mi] = *reinterpret_cast<const frag_t*>(Ps + a_ld[mi] + tapoff + s * M::SUBB);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[set][ni] = *reinterpret_cast<const frag_t*>(b_ld + buf * BS_BYTES + ni * BLK * ROWB + s * M::SUBB);
    };
    frags(0, 0);
#pragma unroll
    for (int s = 0; s < M::SUB; ++s) {
      if (s + 1 < M::SUB) frags((s + 1) & 1, s + 1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma_chunk(acc[mi][ni], a[s & 1][mi], b[s & 1][ni]);
    }
    // pin the order: reads(0) reads(1) mfma(0) reads(2) mfma(1) ... mfma(last)
    constexpr int NR = MI + NI, NM = MI * NI * (sizeof(T) == 4 ? 4 : 1);
    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
    for (int s = 0; s < M::SUB; ++s) {
      if (s + 1 < M::SUB) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    }
  };
  auto step = [&](auto CUR, auto NXT) {
    load_b(CUR);                                               // weights of stage s+2
    const bool first_tap = (cty == 0 && ctx == 0), more = cchunk + 1 < nchunks;
    if (first_tap && more) load_patch(cchunk + 1);             // next chunk's patch: lands during this chunk's taps
    CL_STAMP(0);
    compute(CUR);
    CL_STAMP(1);
    const bool last_tap = (cty == g.nty - 1 && ctx == g.ntx - 1);
    if (last_tap && more) {
      __syncthreads();                                         // everyone is done with this chunk's patch
      store_patch();
    }
    CL_STAMP(2);
    store_b(NXT, NXT);
    CL_STAMP(3);
    __syncthreads();
    CL_STAMP(4);
    if (++ctx == g.ntx) { ctx = 0; if (++cty == g.nty) { cty = 0; ++cchunk; } }
  };
  CL_STAMP3(5);                  // the tile's plan
  if (KT > 0) {
    load_patch(0);
    load_b(S0{});
    load_b(S1{});
    CL_STAMP3(6);                // loads issued
    store_patch();
    store_b(S0{}, S0{});
  }
  CL_STAMP3(7);                  // first patch + weights waited for and stored
  __syncthreads();
#ifdef CONVLAB_STAMPS
  CL_NOW(cl_last);
  cl_first = cl_last;
#endif
#ifdef CONVLAB_PRIO     // lab: the workgroups of every other dispatch round at raised priority (see tools/lab/build_conv_variants.sh)
  if ((blockIdx.x >> CONVLAB_PRIO) & 1) __builtin_amdgcn_s_setprio(1);
#endif
  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    step(S0{}, S1{});
    step(S1{}, S0{});
  }
  if (kt < KT) step(S0{}, S1{});

#ifdef CONVLAB_STAMPS
  unsigned long long cl_loop;
  CL_NOW(cl_loop);
  cl_last = cl_loop;
#endif
  // ---- epilogue (as above): transpose through LDS, 16-byte stores ----------------------------
  auto out_pixel = [&](int row) -> long long {
    const int y = ty0 + (row >> 4), x = tx0 + ((row + rot * (row >> 4)) & (TLW - 1));
    const int oy = y * g.oy_mul + g.oy_add, ox = x * g.ox_mul + g.ox_add;
    return (y < g.MH && x < g.MW && oy < g.OH && ox < g.OW) ? ((long long)bimg * g.OH + oy) * g.OW + ox : -1;
  };
  if (stats) {
    // partial rows are numbered by 8x16-pixel tiles (jspsr_conv2d_stats_rows); a 16x16 tile reports under its
    // upper half's number and zeroes the lower half's row
    if constexpr (BM == 128) {
      stats_epilogue<M, MI, NI, WGM, WTM, WTN, BN>(acc, stats, smem, mt, g.Cout, n0, wm, wn, lane, out_pixel);
    } else {
      const int tty8 = (g.MH + 7) / 8;
      const int r1 = (bimg * tty8 + 2 * tyi) * ttx + txi;
      const int r2 = (2 * tyi + 1 < tty8) ? r1 + ttx : -1;
      stats_epilogue<M, MI, NI, WGM, WTM, WTN, BN>(acc, stats, smem, r1, g.Cout, n0, wm, wn, lane, out_pixel, r2);
    }
  }
  float bv[NI], sv[NI];
  int ncol[NI];
  const bool relu_first = g.relu && !g.addend, relu_last = g.relu && g.addend;   // ReLU is the last operation
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    ncol[ni] = n0 + wn * WTN + ni * BLK + lc;
    bv[ni] = (bias && ncol[ni] < g.Cout) ? bias[ncol[ni]] : 0.f;
    sv[ni] = (g.scale && ncol[ni] < g.Cout) ? g.scale[ncol[ni]] : 1.f;
  }
  constexpr int OPITCH = BN * (int)sizeof(T) + 16;
  char* Os = smem;
  // whole 16-byte pieces of channels: a last N tile that is only partly inside Cout still takes the vector path
  // (pieces at or beyond Cout are skipped in the copy loop)
  const bool vec_ok = (g.Cout % EPC == 0) && ((g.out_cstride * (int)sizeof(T)) % 16 == 0) &&
                      (((g.out_coff + n0) * (int)sizeof(T)) % 16 == 0) &&
                      ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) &&
                      (!g.addend || (((reinterpret_cast<uintptr_t>(g.addend) & 15u) == 0) &&
                                     ((g.add_cstride * (int)sizeof(T)) % 16 == 0) && ((n0 * (int)sizeof(T)) % 16 == 0)));
  if (vec_ok) {
    constexpr int CPRO = BN * (int)sizeof(T) / 16;       // 16-byte chunks per output row
    constexpr int O_IT = BM * CPRO / NTH, OB = O_IT <= 8 ? O_IT : 8;
    static_assert(BM * CPRO % NTH == 0 && O_IT % OB == 0, "epilogue rows");
    // Where this thread's pieces of the tile go, and the addend's pieces, are worked out and requested BEFORE the
    // accumulator -> LDS transpose (the first OB = 8 of them when fp32 tiles have 16): the 64-bit pixel arithmetic and
    // the addend's memory latency then sit beside the transpose and the barrier, and the copy loop behind the barrier is
    // OB LDS reads in a row followed by OB stores (it was read - address - store per piece: 3.5 k cycles per wave,
    // profiles/r03_conv_patch_wave_life.txt).
    long long ooff[OB];        // byte offset from obase, -1: nothing to store (outside the written grid / beyond Cout)
    uint4 areg[OB];
    uint4 zreg[OB];            // BatchNorm-backward reduce (g.red_out): the saved pre-normalisation values of the pieces
    // Beside a partner wave that keeps the matrix pipe busy a vector instruction of this wave costs ~10 cycles, so the
    // per-piece arithmetic is kept short: the tile's origin on the written grid and its byte offset are wave-uniform
    // (scalar unit, 64-bit), a piece adds a 32-bit pixel delta by one v_mad_u64_u32, and the bounds tests are skipped
    // for tiles that lie wholly inside the grid.
    const int oy_t = ty0 * g.oy_mul + g.oy_add, ox_t = tx0 * g.ox_mul + g.ox_add, ystep = g.oy_mul * g.OW;
    const long long pix_t = ((long long)bimg * g.OH + oy_t) * g.OW + ox_t;
    const unsigned ocs = (unsigned)g.out_cstride * (unsigned)sizeof(T), acs = (unsigned)g.add_cstride * (unsigned)sizeof(T);
    const long long obase_t = pix_t * ocs, abase_t = pix_t * acs + n0 * (long long)sizeof(T);
    const unsigned zcs = (unsigned)g.red_cstride * (unsigned)sizeof(T);
    const long long zbase_t = pix_t * zcs + n0 * (long long)sizeof(T);
    const bool inside = ty0 + TLH <= g.MH && tx0 + TLW <= g.MW && oy_t + (TLH - 1) * g.oy_mul < g.OH &&
                        ox_t + (TLW - 1) * g.ox_mul < g.OW && n0 + BN <= g.Cout;
    auto plan_pieces = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < OB; ++j) {
        const int i = tid + (j0 + j) * NTH, row = i / CPRO, c16 = i % CPRO;
        const int dy = row >> 4, dx = (row + rot * dy) & (TLW - 1);
        const unsigned dpix = (unsigned)(dy * ystep + dx * g.ox_mul);        // < 2^31 (checked by the launcher)
        bool ok = true;
        if (!inside)
          ok = ty0 + dy < g.MH && tx0 + dx < g.MW && oy_t + dy * g.oy_mul < g.OH && ox_t + dx * g.ox_mul < g.OW &&
               n0 + c16 * EPC < g.Cout;
        ooff[j] = ok ? obase_t + (long long)((unsigned long long)dpix * ocs) + c16 * 16 : -1;
        if (g.addend) {
          areg[j] = make_uint4(0, 0, 0, 0);
          if (ok)
            areg[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(g.addend) + abase_t +
                                                      (long long)((unsigned long long)dpix * acs) + c16 * 16);
        }
        if (g.red_out) {
          zreg[j] = make_uint4(0, 0, 0, 0);
          if (ok)
            zreg[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(g.red_x) + zbase_t +
                                                      (long long)((unsigned long long)dpix * zcs) + c16 * 16);
        }
      }
    };
    // BatchNorm-backward reduce: this thread's pieces all hold the same EPC channels (NTH is a multiple of the pieces per
    // row), so their two sums live in registers across the copy loop -- the accumulators are dead by then
    float rs1[EPC], rs2[EPC], rpa[EPC], rpb[EPC], rpi[EPC], rpm[EPC];
    if (g.red_out) {
      static_assert(NTH % CPRO == 0, "a thread's pieces share their channels");
      const int cc = n0 + (tid % CPRO) * EPC;
#pragma unroll
      for (int e2 = 0; e2 < EPC; ++e2) {
        const bool in_c = cc + e2 < g.Cout;
        rs1[e2] = rs2[e2] = 0.f;
        rpa[e2] = in_c ? g.red_par[cc + e2] : 0.f;
        rpb[e2] = in_c ? g.red_par[g.Cout + cc + e2] : 0.f;
        rpi[e2] = in_c ? g.red_par[2 * g.Cout + cc + e2] : 0.f;
        rpm[e2] = in_c ? g.red_par[3 * g.Cout + cc + e2] : 0.f;
      }
    }
    plan_pieces(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < M::NE; ++e) {
        const int row = wm * WTM + mi * BLK + M::acc_row(lane, e);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float v = acc[mi][ni][e] * sv[ni] + bv[ni];
          if (relu_first) v = fmaxf(v, 0.f);
          *reinterpret_cast<T*>(Os + row * OPITCH + (wn * WTN + ni * BLK + lc) * (int)sizeof(T)) = (T)v;
        }
      }
    CL_STAMP3(0);                // accumulators -> LDS
    __syncthreads();
    CL_STAMP3(1);                // barrier
    char* const obase = reinterpret_cast<char*>(out + g.out_coff + n0);
#pragma unroll
    for (int j0 = 0; j0 < O_IT; j0 += OB) {
      if (j0) plan_pieces(j0);
      uint4 ov[OB];
#pragma unroll
      for (int j = 0; j < OB; ++j) {
        const int i = tid + (j0 + j) * NTH, row = i / CPRO, c16 = i % CPRO;
        ov[j] = *reinterpret_cast<const uint4*>(Os + row * OPITCH + c16 * 16);
      }
      CL_STAMP3(2);              // tile pieces read back from LDS
#pragma unroll
      for (int j = 0; j < OB; ++j) {
        if (ooff[j] < 0) continue;
        uint4 v = ov[j];
        if (g.addend) v = add_packed<T>(v, areg[j]);
        if (relu_last) v = relu_packed<T>(v);
        *reinterpret_cast<uint4*>(obase + ooff[j]) = v;
        if (g.red_out) {      // dz = the stored gradient where the ReLU behind the BatchNorm was open; sums of dz and dz * xhat
          float dv[EPC], zv[EPC];
          unpack_chunk<T>(v, dv);
          unpack_chunk<T>(zreg[j], zv);
#pragma unroll
          for (int e2 = 0; e2 < EPC; ++e2) {
            const float dz = rpa[e2] * zv[e2] + rpb[e2] > 0.f ? dv[e2] : 0.f;
            rs1[e2] += dz;
            rs2[e2] += dz * (zv[e2] * rpi[e2] + rpm[e2]);
          }
        }
      }
    }
    if (g.red_out) {
      // fold over the NTH / CPRO threads that share a channel group (fixed order), one partial row per 8x16-pixel tile
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);      // [NTH][2 EPC]
#pragma unroll
      for (int e2 = 0; e2 < EPC; ++e2) {
        red[tid * 2 * EPC + e2] = rs1[e2];
        red[tid * 2 * EPC + EPC + e2] = rs2[e2];
      }
      __syncthreads();
      int r1 = mt, r2 = -1;
      if constexpr (BM != 128) {
        const int tty8 = (g.MH + 7) / 8;
        r1 = (bimg * tty8 + 2 * tyi) * ttx + txi;
        r2 = (2 * tyi + 1 < tty8) ? r1 + ttx : -1;
      }
      for (int i = tid; i < 2 * BN; i += NTH) {
        const int which = i / BN, col = i % BN;
        if (n0 + col >= g.Cout) continue;
        const int cg_ = col / EPC, e2 = col % EPC;
        float tsum = 0.f;
        for (int k2 = 0; k2 < NTH / CPRO; ++k2) tsum += red[(cg_ + CPRO * k2) * 2 * EPC + which * EPC + e2];
        g.red_out[((size_t)r1 * 2 + which) * g.Cout + n0 + col] = tsum;
        if (r2 >= 0) g.red_out[((size_t)r2 * 2 + which) * g.Cout + n0 + col] = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < M::NE; ++e) {
        const int row = wm * WTM + mi * BLK + M::acc_row(lane, e);
        const long long opix = out_pixel(row);
        if (opix < 0) continue;
        T* orow = out + opix * g.out_cstride + g.out_coff;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (ncol[ni] >= g.Cout) continue;
          float v = acc[mi][ni][e] * sv[ni] + bv[ni];
          if (relu_first) v = fmaxf(v, 0.f);
          if (g.addend) v = (float)(T)v + (float)static_cast<const T*>(g.addend)[opix * g.add_cstride + ncol[ni]];
          if (relu_last) v = fmaxf(v, 0.f);
          orow[ncol[ni]] = (T)v;
        }
      }
  }
#ifdef CONVLAB_STAMPS
  if ((tid & 63) == 0) {
    unsigned long long end_;
    CL_NOW(end_);
    unsigned long long* o = convlab_stamp + (size_t)((blockIdx.x * (NTH / 64) + (tid >> 6)) % CONVLAB_SLOTS) * 16;
    for (int i = 0; i < 4; ++i) o[i] = cl_stamp[i] + (i == 3 ? cl_stamp[4] : 0);   // (3: weight store + stage barrier; level 3: 0, 1 = epilogue pieces)
    o[4] = cl_first - cl_entry;     // entry .. first stage staged and the barrier behind it
    o[5] = cl_loop - cl_first;      // main loop
    o[6] = end_ - cl_loop;          // epilogue
    for (int i = 5; i < 8; ++i) o[3 + i] = cl_stamp[i];      // level 3: 8 plan, 9 loads issued, 10 wait + store
    o[11] = cl_stamp[4];                                     // level 3: head of the plan (tile coordinates)
    o[7] = (unsigned long long)KT;  // stages
  }
#endif
}

template <typename T, int BM, int BN, int WGM, int WGN>
int launch_patch(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g, hipStream_t s) {
  constexpr int NTH = 64 * WGM * WGN, TLW = 16, TLH = BM / TLW, MAXPIX = patch_it(BM, NTH) * NTH / NCH;
  const long long nblk = (long long)g.B * ((g.MH + TLH - 1) / TLH) * ((g.MW + TLW - 1) / TLW) * ((g.Cout + BN - 1) / BN);
  if (nblk > 0x7fffffffLL || (long long)TLH * g.oy_mul * g.OW + (long long)TLW * g.ox_mul > 0x7fffffffLL)
    return fail(JSPSR_EINVAL, "conv: grid too large");
  constexpr int ROWB = Mma<T>::ROWP;
  const size_t lds_stage = (size_t)MAXPIX * ROWB + 2 * BN * ROWB, lds_out = (size_t)BM * (BN * sizeof(T) + 16);
  const size_t lds = lds_stage > lds_out ? lds_stage : lds_out;
  ConvGeom gk = g;
  gk.fd_ntn = make_fastdiv((unsigned)((g.Cout + BN - 1) / BN));
  gk.fd_ttx = make_fastdiv((unsigned)((g.MW + TLW - 1) / TLW));
  gk.fd_tty = make_fastdiv((unsigned)((g.MH + TLH - 1) / TLH));
  auto kern = conv_patch_kernel<T, BM, BN, WGM, WGN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NTH), lds, s, static_cast<const T*>(in), static_cast<const T*>(wgt),
                     bias, static_cast<T*>(out), stats, gk);
  return check_launch(BM == 256 ? (BN == 64 ? "conv_patch_16x16" : "conv_patch_16x16x128") : "conv_patch");
}

template <typename T, int BM, int BN, int WGM, int WGN, int NBUF>
int launch_cfg(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g, hipStream_t s) {
  if ((long long)g.B * g.MH * g.MW <= 0) return JSPSR_OK;
  constexpr int TLW = 16, TLH = BM / TLW;
  const long long nblk = (long long)g.B * ((g.MH + TLH - 1) / TLH) * ((g.MW + TLW - 1) / TLW) * ((g.Cout + BN - 1) / BN);
  if (nblk > 0x7fffffffLL) return fail(JSPSR_EINVAL, "conv: grid too large");
  constexpr int ROWB = IgemmMma<T, NBUF>::ROWP;
  const size_t lds_stage = NBUF * (BM + BN) * ROWB, lds_out = (size_t)BM * (BN * sizeof(T) + 16);
  const size_t lds = lds_stage > lds_out ? lds_stage : lds_out;
  constexpr int BK = NCH * Elem<T>::EPC;
  static bool attr_set = false;  // > 64 KiB dynamic LDS needs the opt-in once per kernel
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, BM, BN, WGM, WGN, true, NBUF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, BM, BN, WGM, WGN, false, NBUF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (g.Cin % BK == 0)
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WGM, WGN, true, NBUF>), dim3((unsigned)nblk), dim3(NT), lds, s,
                       static_cast<const T*>(in), static_cast<const T*>(wgt), bias, static_cast<T*>(out), stats, g);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WGM, WGN, false, NBUF>), dim3((unsigned)nblk), dim3(NT), lds, s,
                       static_cast<const T*>(in), static_cast<const T*>(wgt), bias, static_cast<T*>(out), stats, g);
  return check_launch("conv_igemm");
}

template <typename T>
int launch(const void* in, const void* wgt, const float* bias, void* out, float* stats, const ConvGeom& g, hipStream_t s) {
  // narrow tiles: single LDS buffer (4 resident workgroups per CU) while K is short -- the layer is then
  // HBM/latency bound and overlap across workgroups wins; double buffer for long K (measured crossover
  // between K = 576 and K = 2304 on MI355X).  JSPSR_CONV_NBUF overrides for experiments.
  // unit-stride tap walk, <= 3x3 taps, Cin a multiple of the stage depth: stage the input patch once
  static const int no_patch = [] { const char* e = getenv("JSPSR_CONV_NOPATCH"); return e ? atoi(e) : 0; }();
  constexpr int BKT = NCH * Elem<T>::EPC;
  if constexpr (sizeof(T) == 2)
  {
    if (conv64_resident_ok(g, in, wgt, bias, out)) return launch_conv64_resident(in, wgt, bias, out, stats, g, s);
    if (conv128_resident_ok(g, in, wgt, bias, out)) return launch_conv128_resident(in, wgt, bias, out, stats, g, s);
  }
  if ((!no_patch || g.in_affine) && g.iy_mul == 1 && g.ix_mul == 1 && g.nty >= 1 && g.ntx >= 1 && g.nty <= 3 && g.ntx <= 3 &&
      g.nty * g.ntx > 1 && g.Cin % BKT == 0 && (long long)(g.IW + 20) * 12 * g.in_cstride * (long long)sizeof(T) < 0xE0000000LL) {
    static const int tall = [] { const char* e = getenv("JSPSR_CONV_TALL"); return e ? atoi(e) : 1; }();
    const long long tiles16 = (long long)g.B * ((g.MH + 15) / 16) * ((g.MW + 15) / 16);
    if (g.Cout > 64) {
      // 16x16-pixel tile, 8 waves: the weight tile is staged once per 256 pixels instead of once per 128.  Measured
      // on MI355X: +5 % on the 64x64x512-channel layers, -3..5 % on the full-resolution ones (one 8-wave workgroup
      // per CU stalls as a whole at each stage barrier; two 4-wave workgroups cover each other) -> opt-in only
      // (JSPSR_CONV_TALL=2).
      if (tall == 2 && tiles16 * ((g.Cout + 127) / 128) >= 512) return launch_patch<T, 256, 128, 4, 2>(in, wgt, bias, out, stats, g, s);
      // round 4: the 64-channel configuration's 16x16-pixel x 64-channel tile for the wider layers too, once there are
      // >= 1024 of them -- the patch is staged once per 64 output channels instead of once per 128, the weight tile once per
      // 256 pixels instead of 128 (32 % fewer LDS write bytes per output): +3 % (8 x 512^2, 128 -> 128) .. +9.5 % (256 ->
      // 256 data gradient at 128^2) per kernel, 67.03 -> 66.63 ms per single-stream step (profiles/r04_conv_tile_256x64_lab.txt).
      // JSPSR_CONV_TALL=4 keeps the 128 x 128 tile of rounds 1-3.
      if (tall && tall != 4 && tiles16 * ((g.Cout + 63) / 64) >= 1024) return launch_patch<T, 256, 64, 4, 1>(in, wgt, bias, out, stats, g, s);
      return launch_patch<T, 128, 128, 2, 2>(in, wgt, bias, out, stats, g, s);
    }
    if (g.Cout > 32) {
      // 64 output channels: a 16x16-pixel tile (4 waves x 64 px x 64 ch) halves the weight-tile staging and the
      // fragment reads per MFMA of the 8x16 tile; worth it once the image has enough tiles to fill the chip
      if (tall && tiles16 >= 1024) return launch_patch<T, 256, 64, 4, 1>(in, wgt, bias, out, stats, g, s);
      return launch_patch<T, 128, 64, 2, 2>(in, wgt, bias, out, stats, g, s);
    }
    return launch_patch<T, 128, 32, 4, 1>(in, wgt, bias, out, stats, g, s);
  }
  if (g.red_out) return fail(JSPSR_EINVAL, "conv2d_dgrad: the fused BatchNorm-backward reduce needs the patch kernel (jspsr_conv2d_dgrad_reduce_ok)");
  if (g.in_affine) return fail(JSPSR_EINVAL, "conv2d_forward: in_affine needs the patch kernel (stride 1, 2..9 taps, Cin a multiple of %d): ask jspsr_conv2d_in_affine_ok first", BKT);
  static const int nbuf_env = [] { const char* e = getenv("JSPSR_CONV_NBUF"); return e ? atoi(e) : 0; }();
  const int nbuf_narrow = nbuf_env ? nbuf_env : ((long long)g.nty * g.ntx * g.Cin <= 1152 ? 1 : 2);
  if (g.Cout > 64) return launch_cfg<T, 128, 128, 2, 2, 2>(in, wgt, bias, out, stats, g, s);
  if (nbuf_narrow == 2) {
    if (g.Cout > 32) return launch_cfg<T, 128, 64, 2, 2, 2>(in, wgt, bias, out, stats, g, s);
    return launch_cfg<T, 128, 32, 4, 1, 2>(in, wgt, bias, out, stats, g, s);
  }
  if (g.Cout > 32) return launch_cfg<T, 128, 64, 2, 2, 1>(in, wgt, bias, out, stats, g, s);
  return launch_cfg<T, 128, 32, 4, 1, 1>(in, wgt, bias, out, stats, g, s);
}

int check_common(int dtype, const void* a, const void* w, const void* o, int cin, int cs_in, int co_in,
                 int cs_out, int co_out, int cout, const char* what) {
  if (dtype != JSPSR_F32 && dtype != JSPSR_BF16) return fail(JSPSR_EINVAL, "%s: dtype must be JSPSR_F32 or JSPSR_BF16", what);
  if (!a || !w || !o) return fail(JSPSR_EINVAL, "%s: null pointer", what);
  const int epc = dtype == JSPSR_F32 ? 4 : 8;
  if (cin <= 0 || cin % epc || cs_in % epc || co_in % epc || cs_in < co_in + cin)
    return fail(JSPSR_EINVAL, "%s: gathered channels (C=%d pitch=%d off=%d) must be multiples of %d", what, cin, cs_in, co_in, epc);
  if (cout <= 0 || cs_out < co_out + cout) return fail(JSPSR_EINVAL, "%s: bad output channel slice", what);
  if (!aligned16(a) || !aligned16(w)) return fail(JSPSR_EALIGN, "%s: input/weight pointers must be 16-byte aligned", what);
  return JSPSR_OK;
}

// (O, I, KH, KW) fp32 master -> packed rows.  mode 0: [o][ky][kx][i_pad]; mode 1: [i][ky][kx][o_pad]
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int O, int I, int KH, int KW,
                                   int mode, int Cpad, long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cpad);
    long long r = idx / Cpad;
    const int kx = (int)(r % KW); r /= KW;
    const int ky = (int)(r % KH); r /= KH;
    const int n = (int)r;
    float v = 0.f;
    if (mode == 0) { if (c < I) v = w[(((size_t)n * I + c) * KH + ky) * KW + kx]; }
    else           { if (c < O) v = w[(((size_t)c * I + n) * KH + ky) * KW + kx]; }
    out[idx] = (T)v;
  }
}

// All conv weights of a model in ONE launch (after the optimizer step).  Descriptor i owns blocks
// [blk0_i, blk0_{i+1}) of PACK_CHUNK elements each (`start` holds blk0, in blocks); a block finds its descriptor once
// (binary search over <= a few hundred entries, wave-uniform), then walks its chunk like pack_weight_kernel.
constexpr int PACK_CHUNK = 2048;
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const jspsr_pack_desc* __restrict__ descs, int n) {
  int lo = 0, hi = n - 1;
  const long long b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].start <= b) lo = mid; else hi = mid - 1;
  }
  const jspsr_pack_desc d = descs[lo];
  const long long l0 = (b - d.start) * PACK_CHUNK;
  const float* __restrict__ w = d.w;
#pragma unroll
  for (int k = 0; k < PACK_CHUNK / 256; ++k) {
    const long long l = l0 + k * 256 + threadIdx.x;
    if (l >= d.total) break;
    const int c = (int)(l % d.c_pad);
    long long r = l / d.c_pad;
    const int kx = (int)(r % d.KW); r /= d.KW;
    const int ky = (int)(r % d.KH); r /= d.KH;
    const int nn = (int)r;
    float v = 0.f;
    if (d.mode == 0) { if (c < d.I) v = w[(((size_t)nn * d.I + c) * d.KH + ky) * d.KW + kx]; }
    else             { if (c < d.O) v = w[(((size_t)c * d.I + nn) * d.KH + ky) * d.KW + kx]; }
    if (d.dtype == JSPSR_F32) static_cast<float*>(d.out)[l] = v;
    else static_cast<__bf16*>(d.out)[l] = (__bf16)v;
  }
}

}  // namespace

extern "C" int jspsr_pack_weights_multi(const jspsr_pack_desc* descs, int n, long long total_blocks, jspsr_stream_t stream) {
  if (!descs || n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return fail(JSPSR_EINVAL, "pack_weights_multi: bad arguments");
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), descs, n);
  return check_launch("pack_weights_multi");
}

extern "C" int jspsr_pack_chunk(void) { return PACK_CHUNK; }

extern "C" int jspsr_pack_weight(int dtype, const float* w, void* packed, int O, int I, int KH, int KW,
                                 int mode, int c_pad, jspsr_stream_t stream) {
  if (!w || !packed || O <= 0 || I <= 0 || KH <= 0 || KW <= 0 || (mode != 0 && mode != 1))
    return fail(JSPSR_EINVAL, "pack_weight: bad arguments");
  const int C = mode == 0 ? I : O, N = mode == 0 ? O : I;
  if (c_pad < C) return fail(JSPSR_EINVAL, "pack_weight: c_pad %d < %d", c_pad, C);
  const long long total = (long long)N * KH * KW * c_pad;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == JSPSR_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, w, static_cast<float*>(packed), O, I, KH, KW, mode, c_pad, total);
  else if (dtype == JSPSR_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, w, static_cast<__bf16*>(packed), O, I, KH, KW, mode, c_pad, total);
  else
    return fail(JSPSR_EINVAL, "pack_weight: bad dtype");
  return check_launch("pack_weight");
}

extern "C" int jspsr_conv2d_stats_rows(int B, int OH, int OW) {
  if (B <= 0 || OH <= 0 || OW <= 0) return 0;
  return B * ((OH + 7) / 8) * ((OW + 15) / 16);   // one partial row per 8x16-pixel M-tile
}

extern "C" int jspsr_conv2d_forward(int dtype, const void* in, const void* wpack, const float* bias, void* out,
                                    int B, int IH, int IW, int Cin, int in_cstride, int in_coff, int Cout,
                                    int out_cstride, int out_coff, int KH, int KW, int stride, int pad, int relu,
                                    float* stats, const float* scale, const void* addend, int add_cstride,
                                    const float* in_affine, int in_relu, jspsr_stream_t stream) {
  if (int e = check_common(dtype, in, wpack, out, Cin, in_cstride, in_coff, out_cstride, out_coff, Cout, "conv2d_forward")) return e;
  if (in_affine && !aligned16(in_affine)) return fail(JSPSR_EALIGN, "conv2d_forward: in_affine must be 16-byte aligned");
  if (addend && add_cstride < Cout) return fail(JSPSR_EINVAL, "conv2d_forward: addend pitch %d < %d channels", add_cstride, Cout);
  if (B <= 0 || IH <= 0 || IW <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    return fail(JSPSR_EINVAL, "conv2d_forward: bad geometry");
  ConvGeom g{};
  g.B = B; g.IH = IH; g.IW = IW; g.Cin = Cin;
  g.OH = (IH + 2 * pad - KH) / stride + 1;
  g.OW = (IW + 2 * pad - KW) / stride + 1;
  if (g.OH <= 0 || g.OW <= 0) return fail(JSPSR_EINVAL, "conv2d_forward: empty output");
  g.Cout = Cout; g.in_cstride = in_cstride; g.in_coff = in_coff; g.out_cstride = out_cstride; g.out_coff = out_coff;
  g.MH = g.OH; g.MW = g.OW;
  g.iy_mul = stride; g.iy_add = -pad; g.ix_mul = stride; g.ix_add = -pad; g.sign = 1;
  g.nty = KH; g.ntx = KW; g.ky0 = 0; g.kx0 = 0; g.kstep = 1; g.KH = KH; g.KW = KW;
  g.oy_mul = 1; g.oy_add = 0; g.ox_mul = 1; g.ox_add = 0; g.relu = relu;
  g.accumulate = getenv("JSPSR_CONV_NOXCD") ? 1 : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stats && (bias || relu || scale || addend))
    return fail(JSPSR_EINVAL, "conv2d_forward: statistics are taken from the raw accumulators (no bias / scale / addend / ReLU)");
  g.scale = scale; g.addend = addend; g.add_cstride = add_cstride;
  g.in_affine = in_affine; g.in_relu = in_relu;
  return dtype == JSPSR_F32 ? launch<float>(in, wpack, bias, out, stats, g, s) : launch<__bf16>(in, wpack, bias, out, stats, g, s);
}

extern "C" int jspsr_conv2d_in_affine_ok(int dtype, int Cin, int KH, int KW, int stride) {
  const int bk = NCH * (dtype == JSPSR_F32 ? 4 : 8);
  return (dtype == JSPSR_F32 || dtype == JSPSR_BF16) && stride == 1 && KH >= 1 && KW >= 1 && KH <= 3 && KW <= 3 && KH * KW > 1 &&
         Cin > 0 && Cin % bk == 0;
}

// Can the data gradient of this conv carry the reduce pass of the BatchNorm behind its input (ConvGeom::red_*)?  The patch
// kernel's 16-byte epilogue does: 3x3, stride 1, pad 1, gathered channels a multiple of the stage depth, written channels
// whole 16-byte chunks -- and only where the launch would take the patch kernel anyway (the 64 -> 64 bf16 layers on >= 512
// tiles go to K2r, whose register budget has no room for the sums: DESIGN.md section 5).
extern "C" int jspsr_conv2d_dgrad_reduce_ok(int dtype, int B, int IH, int IW, int Cg, int Cin, int KH, int KW, int stride, int pad) {
  if (dtype != JSPSR_F32 && dtype != JSPSR_BF16) return 0;
  static const int off = [] { const char* e = getenv("JSPSR_BN_REDUCE_FUSE"); return e && atoi(e) == 0; }();
  if (off) return 0;
  const int epc = dtype == JSPSR_F32 ? 4 : 8, bkt = NCH * epc;
  if (KH != 3 || KW != 3 || stride != 1 || pad != 1 || Cg <= 0 || Cg % bkt || Cin <= 0 || Cin % epc || B <= 0 || IH <= 0 || IW <= 0) return 0;
  static const int no_patch = [] { const char* e = getenv("JSPSR_CONV_NOPATCH"); return e ? atoi(e) : 0; }();
  if (no_patch) return 0;
  if (dtype == JSPSR_BF16 && Cg == 64 && Cin == 64) {
    static const int resident = [] { const char* e = getenv("JSPSR_CONV_RESIDENT"); return e ? atoi(e) : 1; }();
    static const int min_tiles = [] { const char* e = getenv("JSPSR_CONV_RESIDENT_MIN"); return e ? atoi(e) : 512; }();
    if (resident && (long long)B * ((IH + 15) / 16) * ((IW + 15) / 16) >= min_tiles) return 0;
  }
  return 1;
}

extern "C" int jspsr_conv2d_dgrad(int dtype, const void* gout, const void* wpack_t, const float* bias, void* gin,
                                  int B, int OH, int OW, int Cg, int g_cstride, int g_coff, int IH, int IW,
                                  int Cin, int in_cstride, int in_coff, int KH, int KW, int stride, int pad,
                                  int relu, const void* addend, int add_cstride, const float* scale,
                                  const void* red_x, int red_cstride, const float* red_par, float* red_out,
                                  jspsr_stream_t stream) {
  if (int e = check_common(dtype, gout, wpack_t, gin, Cg, g_cstride, g_coff, in_cstride, in_coff, Cin, "conv2d_dgrad")) return e;
  if (addend && add_cstride < Cin) return fail(JSPSR_EINVAL, "conv2d_dgrad: addend pitch %d < %d channels", add_cstride, Cin);
  if (red_out) {
    if (!red_x || !red_par || !jspsr_conv2d_dgrad_reduce_ok(dtype, B, IH, IW, Cg, Cin, KH, KW, stride, pad))
      return fail(JSPSR_EINVAL, "conv2d_dgrad: the fused BatchNorm-backward reduce needs jspsr_conv2d_dgrad_reduce_ok(...) != 0");
    const int epc = dtype == JSPSR_F32 ? 4 : 8;
    if (red_cstride < Cin || red_cstride % epc || in_cstride % epc || in_coff % epc || !aligned16(red_x) || !aligned16(gin) ||
        (addend && (!aligned16(addend) || add_cstride % epc)))
      return fail(JSPSR_EALIGN, "conv2d_dgrad: fused reduce needs 16-byte pieces (pitches / offsets multiples of %d, aligned tensors)", epc);
  }
  if (B <= 0 || OH <= 0 || OW <= 0 || IH <= 0 || IW <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    return fail(JSPSR_EINVAL, "conv2d_dgrad: bad geometry");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // one launch per stride phase (py,px): written pixels y = stride*y' + py see only taps
  // ky == (py + pad) mod stride; gathered row = (y + pad - ky)/stride = y' + cy - t.
  for (int py = 0; py < stride; ++py)
    for (int px = 0; px < stride; ++px) {
      ConvGeom g{};
      g.B = B; g.IH = OH; g.IW = OW; g.Cin = Cg;       // gathered tensor = gout
      g.OH = IH; g.OW = IW; g.Cout = Cin;              // written tensor = gin
      g.in_cstride = g_cstride; g.in_coff = g_coff; g.out_cstride = in_cstride; g.out_coff = in_coff;
      g.MH = (IH - py + stride - 1) / stride;
      g.MW = (IW - px + stride - 1) / stride;
      if (g.MH <= 0 || g.MW <= 0) continue;
      g.ky0 = (py + pad) % stride; g.kx0 = (px + pad) % stride; g.kstep = stride;
      g.nty = g.ky0 < KH ? (KH - g.ky0 + stride - 1) / stride : 0;
      g.ntx = g.kx0 < KW ? (KW - g.kx0 + stride - 1) / stride : 0;
      if (g.nty == 0 || g.ntx == 0) g.nty = g.ntx = 0;
      g.iy_mul = 1; g.iy_add = (py + pad - g.ky0) / stride; g.ix_mul = 1; g.ix_add = (px + pad - g.kx0) / stride;
      g.sign = -1; g.KH = KH; g.KW = KW;
      g.oy_mul = stride; g.oy_add = py; g.ox_mul = stride; g.ox_add = px; g.relu = relu;
      g.addend = addend; g.add_cstride = add_cstride; g.scale = scale;
      g.red_x = red_x; g.red_cstride = red_cstride; g.red_par = red_par; g.red_out = red_out;
      const int e = dtype == JSPSR_F32 ? launch<float>(gout, wpack_t, bias, gin, nullptr, g, s)
                                       : launch<__bf16>(gout, wpack_t, bias, gin, nullptr, g, s);
      if (e) return e;
    }
  return JSPSR_OK;
}

#ifdef CONVLAB_STAMPS
extern "C" int jspsr_lab_conv_stamps(unsigned long long* out16, int reset) {      // sums over the wave rows written so far
  static unsigned long long host[CONVLAB_SLOTS * 16];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(convlab_stamp), sizeof(host)) != hipSuccess) return 1;
  for (int i = 0; i < 16; ++i) out16[i] = 0;
  for (int r = 0; r < CONVLAB_SLOTS; ++r)
    if (host[r * 16 + 7]) { for (int i = 0; i < 16; ++i) out16[i] += host[r * 16 + i]; }
  if (reset) { for (size_t i = 0; i < sizeof(host) / 8; ++i) host[i] = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(convlab_stamp), host, sizeof(host)) != hipSuccess) return 1; }
  return 0;
}
#endif
