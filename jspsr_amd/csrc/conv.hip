// K2 -- implicit-GEMM convolution (forward, data gradient, transposed conv) on MFMA.
// Design notes: conv_igemm.h.  Entry points: include/jspsr_hip.h (jspsr_conv2d_*, jspsr_pack_weight).
#include "conv_igemm.h"

namespace {

using namespace jspsr;

constexpr int NT = 256;    // 4 waves
constexpr int NCH = 8;     // 16-byte chunks per LDS row per stage (128 B of K per row)
constexpr int ROWB = 144;  // LDS row pitch: 128 B + 16 B pad -> conflict-free ds_read_b128

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPC = 4; };
template <> struct Elem<__bf16> { static constexpr int EPC = 8; };

__device__ __forceinline__ void mma_chunk(f32x16& acc, const float4& a, const float4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_chunk(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

template <typename T> struct Frag;
template <> struct Frag<float> { using type = float4; };
template <> struct Frag<__bf16> { using type = bf16x8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }

struct RowInfo {  // one per tile row (m-pixel), staged in LDS
  int b;          // image index, -1: row beyond M
  int iy0, ix0;   // gathered pixel of tap walk index (0,0)
  int opix;       // written pixel index (b*OH + oy)*OW + ox, -1: nothing to write
};

template <typename T, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(NT, 2) void conv_igemm_kernel(const T* __restrict__ in,
                                                          const T* __restrict__ wgt,
                                                          const float* __restrict__ bias,
                                                          T* __restrict__ out, ConvGeom g) {
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int EPC = Elem<T>::EPC, BK = NCH * EPC;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32;
  constexpr int A_IT = BM * NCH / NT, B_IT = (BN * NCH + NT - 1) / NT;
  static_assert(MI >= 1 && NI >= 1 && BM * NCH % NT == 0, "tile");
  using frag_t = typename Frag<T>::type;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                                  // [2][BM][ROWB]
  char* Bs = smem + 2 * BM * ROWB;                  // [2][BN][ROWB]
  RowInfo* rows = reinterpret_cast<RowInfo*>(smem + 2 * (BM + BN) * ROWB);  // [BM]

  const int tid = threadIdx.x;
  const int M = g.B * g.MH * g.MW;
  const int ntn = (g.Cout + BN - 1) / BN;
  const int nblk = gridDim.x;
  const int t = xcd_contiguous(blockIdx.x, nblk);
  const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;

  if (tid < BM) {
    RowInfo ri;
    const int m = m0 + tid;
    if (m < M) {
      const int xq = m % g.MW, tq = m / g.MW, yq = tq % g.MH, b = tq / g.MH;
      ri.b = b;
      ri.iy0 = yq * g.iy_mul + g.iy_add;
      ri.ix0 = xq * g.ix_mul + g.ix_add;
      const int oy = yq * g.oy_mul + g.oy_add, ox = xq * g.ox_mul + g.ox_add;
      ri.opix = (oy < g.OH && ox < g.OW) ? (b * g.OH + oy) * g.OW + ox : -1;
    } else {
      ri.b = -1;
      ri.iy0 = ri.ix0 = 0;
      ri.opix = -1;
    }
    rows[tid] = ri;
  }
  __syncthreads();

  // this thread's staging slots: chunk `ch` of rows (tid/8 + 32*i)
  const int ch = tid & (NCH - 1);
  const int r0 = tid >> 3;
  int rb[A_IT], riy[A_IT], rix[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const RowInfo ri = rows[r0 + 32 * i];
    rb[i] = ri.b;
    riy[i] = ri.iy0;
    rix[i] = ri.ix0;
  }
  const int Ktot_w = g.KH * g.KW * g.Cin;  // packed weight row length
  const int K = g.nty * g.ntx * g.Cin;     // walked K
  const int KT = (K + BK - 1) / BK;

  // tap walk state of this thread's chunk
  int ci = ch * EPC, ty = 0, tx = 0;
  auto norm = [&]() {
    while (ci >= g.Cin) {
      ci -= g.Cin;
      if (++tx == g.ntx) { tx = 0; ++ty; }
    }
  };
  norm();

  uint4 areg[A_IT], breg[B_IT];
  auto load_stage = [&]() {
    const bool kok = ty < g.nty;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = riy[i] + g.sign * ty, ix = rix[i] + g.sign * tx;
      const bool ok = kok && rb[i] >= 0 && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        const size_t pix = ((size_t)rb[i] * g.IH + iy) * g.IW + ix;
        v = *reinterpret_cast<const uint4*>(in + pix * g.in_cstride + g.in_coff + ci);
      }
      areg[i] = v;
    }
    const int wk = ((g.ky0 + g.kstep * ty) * g.KW + (g.kx0 + g.kstep * tx)) * g.Cin + ci;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int r = r0 + 32 * i;
      const int n = n0 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kok && r < BN && n < g.Cout) v = *reinterpret_cast<const uint4*>(wgt + (size_t)n * Ktot_w + wk);
      breg[i] = v;
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      *reinterpret_cast<uint4*>(As + (buf * BM + r0 + 32 * i) * ROWB + ch * 16) = areg[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (r0 + 32 * i < BN) *reinterpret_cast<uint4*>(Bs + (buf * BN + r0 + 32 * i) * ROWB + ch * 16) = breg[i];
  };
  auto advance = [&]() {
    ci += BK;
    norm();
  };

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

  if (KT > 0) {
    load_stage();
    store_stage(0);
    advance();
  }
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) load_stage();  // global loads for the next stage fly during the MFMAs below
    const char* Ab = As + (buf * BM + wm * WTM + lr) * ROWB + lh * 16;
    const char* Bb = Bs + (buf * BN + wn * WTN + lr) * ROWB + lh * 16;
#pragma unroll
    for (int s = 0; s < NCH / 2; ++s) {
      frag_t a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const frag_t*>(Ab + mi * 32 * ROWB + s * 32);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const frag_t*>(Bb + ni * 32 * ROWB + s * 32);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma_chunk(acc[mi][ni], a[mi], b[ni]);
    }
    if (more) {
      store_stage(buf ^ 1);
      advance();
    }
    __syncthreads();
  }

  // epilogue: C row = (e&3) + 8*(e>>2) + 4*lh, C col = lr
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * WTN + ni * 32 + lr;
    if (n >= g.Cout) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * WTM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int opix = rows[row].opix;
        if (opix < 0) continue;
        float v = acc[mi][ni][e] + bv;
        if (g.relu) v = fmaxf(v, 0.f);
        out[(size_t)opix * g.out_cstride + g.out_coff + n] = (T)v;
      }
    }
  }
}

template <typename T, int BM, int BN, int WGM, int WGN>
int launch_cfg(const void* in, const void* wgt, const float* bias, void* out, const ConvGeom& g, hipStream_t s) {
  const long long M = (long long)g.B * g.MH * g.MW;
  if (M <= 0) return JSPSR_OK;
  const long long nblk = ((M + BM - 1) / BM) * ((g.Cout + BN - 1) / BN);
  if (nblk > 0x7fffffffLL) return fail(JSPSR_EINVAL, "conv: grid too large");
  const size_t lds = 2 * (BM + BN) * ROWB + BM * sizeof(RowInfo);
  auto kern = conv_igemm_kernel<T, BM, BN, WGM, WGN>;
  static bool attr_set = false;  // > 64 KiB dynamic LDS needs the opt-in once per kernel
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, s, static_cast<const T*>(in),
                     static_cast<const T*>(wgt), bias, static_cast<T*>(out), g);
  return check_launch("conv_igemm");
}

template <typename T>
int launch(const void* in, const void* wgt, const float* bias, void* out, const ConvGeom& g, hipStream_t s) {
  if (g.Cout > 64) return launch_cfg<T, 128, 128, 2, 2>(in, wgt, bias, out, g, s);
  if (g.Cout > 32) return launch_cfg<T, 128, 64, 2, 2>(in, wgt, bias, out, g, s);
  return launch_cfg<T, 128, 32, 4, 1>(in, wgt, bias, out, g, s);
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

}  // namespace

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

extern "C" int jspsr_conv2d_forward(int dtype, const void* in, const void* wpack, const float* bias, void* out,
                                    int B, int IH, int IW, int Cin, int in_cstride, int in_coff, int Cout,
                                    int out_cstride, int out_coff, int KH, int KW, int stride, int pad, int relu,
                                    jspsr_stream_t stream) {
  if (int e = check_common(dtype, in, wpack, out, Cin, in_cstride, in_coff, out_cstride, out_coff, Cout, "conv2d_forward")) return e;
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == JSPSR_F32 ? launch<float>(in, wpack, bias, out, g, s) : launch<__bf16>(in, wpack, bias, out, g, s);
}

extern "C" int jspsr_conv2d_dgrad(int dtype, const void* gout, const void* wpack_t, const float* bias, void* gin,
                                  int B, int OH, int OW, int Cg, int g_cstride, int g_coff, int IH, int IW,
                                  int Cin, int in_cstride, int in_coff, int KH, int KW, int stride, int pad,
                                  int relu, jspsr_stream_t stream) {
  if (int e = check_common(dtype, gout, wpack_t, gin, Cg, g_cstride, g_coff, in_cstride, in_coff, Cin, "conv2d_dgrad")) return e;
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
      const int e = dtype == JSPSR_F32 ? launch<float>(gout, wpack_t, bias, gin, g, s) : launch<__bf16>(gout, wpack_t, bias, gin, g, s);
      if (e) return e;
    }
  return JSPSR_OK;
}
