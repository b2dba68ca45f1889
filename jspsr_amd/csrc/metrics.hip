// Evaluation scores on the device: what evaluation/metrics.py computes per tile with a
// handful of torch calls and a host round trip per score, as a short chain of kernels with no host synchronisation:
//   * MeterBase._prepare   metrics.py:147-199   border crop by int(h * border), clamp of the prediction to [0, 1]
//   * ToDEM.descale_data   data/data_utils.py:441-457   v * (max - min) + min, or exp(v * log(max - min)) + min
//   * MeterPSNR (piq.psnr, data_range 1, mean)  metrics.py:229-235   -10 log10(mse + 1e-8)
//   * MeterRMSE            metrics.py:372-384   sqrt(sum dh^2 / n)
//   * MeterMedian          metrics.py:453       torch.median(dh): the LOWER middle element
//   * MeterNMAD            metrics.py:508-510   1.4826 * median |dh - median dh|
//   * MeterLE95            metrics.py:565-568   kthvalue(|dh|, k = 1 + round(0.95 (n - 1)))
// The order statistics are an 8-bit-per-pass radix SELECT on order-preserving 32-bit keys (4 histogram passes + 4
// one-workgroup scans per statistic): exact -- the selected value is an element of the array, bit for bit what a sort
// would give -- and O(n) instead of the sort behind torch.median / torch.kthvalue.
#include "common.h"

#include <cmath>

namespace {

using namespace jspsr;

constexpr int MT = 256;

// float -> unsigned key with the same ordering (NaNs sort above +inf)
__device__ __forceinline__ unsigned order_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct SelState {          // one per statistic, in device memory
  unsigned prefix;         // key bits fixed so far (high bits)
  unsigned long long k;    // rank still to find inside the current prefix class (0-based)
  float center;            // mode 1: |x - center|
  float result;
};

// pass 1: dh[i] = descale(clamp(pred)) - descale(gt) over the cropped window; partial sums {sum (p-g)^2, sum dh^2}
__global__ __launch_bounds__(MT) void metrics_prepare_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                            int H, int W, int bh, int bw, float vmin, float vmax, int elev_log,
                                                            float* __restrict__ dh, float* __restrict__ partial) {
  __shared__ float red[2][MT / 64];
  const int h = H - 2 * bh, w = W - 2 * bw;
  const long long n = (long long)h * w;
  const float lg = logf(vmax - vmin);
  float s0 = 0.f, s1 = 0.f;
  for (long long i = blockIdx.x * (long long)MT + threadIdx.x; i < n; i += (long long)gridDim.x * MT) {
    const int y = (int)(i / w), x = (int)(i - (long long)y * w);
    const size_t j = (size_t)(y + bh) * W + (x + bw);
    const float p = fminf(fmaxf(pred[j], 0.f), 1.f), g = gt[j];
    const float dp = elev_log ? expf(p * lg) + vmin : p * (vmax - vmin) + vmin;
    const float dg = elev_log ? expf(g * lg) + vmin : g * (vmax - vmin) + vmin;
    const float d = dp - dg;
    dh[i] = d;
    s0 += (p - g) * (p - g);
    s1 += d * d;
  }
  float v[2] = {s0, s1};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int d2 = 32; d2 > 0; d2 >>= 1) v[q] += __shfl_xor(v[q], d2, 64);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int k = 0; k < MT / 64; ++k) t += red[threadIdx.x][k];
    partial[(size_t)blockIdx.x * 2 + threadIdx.x] = t;
  }
}

// scores[0] = PSNR, scores[1] = RMSE from the partial rows (fp64, fixed order); initialises the three select states
__global__ __launch_bounds__(MT) void metrics_reduce_kernel(const float* __restrict__ partial, int rows, long long n,
                                                           float* __restrict__ scores, SelState* __restrict__ st,
                                                           unsigned* __restrict__ hist) {
  __shared__ double red[2][MT / 64];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < rows; i += MT) {
    a += (double)partial[(size_t)i * 2];
    b += (double)partial[(size_t)i * 2 + 1];
  }
#pragma unroll
  for (int d2 = 32; d2 > 0; d2 >>= 1) {
    a += __shfl_xor(a, d2, 64);
    b += __shfl_xor(b, d2, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < MT / 64; ++k) { sa += red[0][k]; sb += red[1][k]; }
    scores[0] = (float)(-10.0 * log10(sa / (double)n + 1e-8));
    scores[1] = (float)sqrt(sb / (double)n);
    const unsigned long long kmed = (unsigned long long)((n - 1) / 2);                 // lower median, 0-based
    const unsigned long long k95 = (unsigned long long)llrint(0.95 * (double)(n - 1)); // 1 + round(0.95 (n-1)), 0-based
    st[0] = SelState{0u, kmed, 0.f, 0.f};    // median of dh
    st[1] = SelState{0u, kmed, 0.f, 0.f};    // median of |dh - median|   (center filled in after statistic 0)
    st[2] = SelState{0u, k95, 0.f, 0.f};     // k-th of |dh|
  }
  for (int i = threadIdx.x; i < 256; i += MT) hist[i] = 0u;
}

// mode 0: x; 1: |x - center|; 2: |x|
__device__ __forceinline__ float sel_transform(float x, int mode, float center) {
  return mode == 0 ? x : (mode == 1 ? fabsf(x - center) : fabsf(x));
}

// histogram of byte `pass` (3 = most significant) of the keys whose higher bytes equal the state's prefix
__global__ __launch_bounds__(MT) void select_hist_kernel(const float* __restrict__ x, long long n, int mode, int pass,
                                                        const SelState* __restrict__ st, unsigned* __restrict__ hist) {
  __shared__ unsigned lh[256];
  lh[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned prefix = st->prefix;
  const float center = st->center;
  const int shift = pass * 8;
  const unsigned himask = pass == 3 ? 0u : (0xffffffffu << (shift + 8));
  for (long long i = blockIdx.x * (long long)MT + threadIdx.x; i < n; i += (long long)gridDim.x * MT) {
    const unsigned key = order_key(sel_transform(x[i], mode, center));
    if ((key & himask) == (prefix & himask)) atomicAdd(&lh[(key >> shift) & 0xffu], 1u);
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// one workgroup: find the bin holding rank k, fix its byte, reduce k, clear the histogram; after the last pass the key
// is complete: write the value (scaled) to scores[slot] and, for the median, hand it to the next statistic as center
__global__ __launch_bounds__(256) void select_scan_kernel(unsigned* __restrict__ hist, int pass, SelState* __restrict__ st,
                                                         float scale, float* __restrict__ scores, int slot,
                                                         SelState* __restrict__ next_center) {
  __shared__ unsigned long long cum[256];
  const unsigned c = hist[threadIdx.x];
  cum[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0, k = st->k;
    int bin = 255;
    for (int b = 0; b < 256; ++b) {
      if (k < run + cum[b]) { bin = b; break; }
      run += cum[b];
    }
    st->k = k - run;
    st->prefix |= (unsigned)bin << (pass * 8);
    if (pass == 0) {
      const float v = key_value(st->prefix);
      st->result = v;
      scores[slot] = v * scale;
      if (next_center) next_center->center = v;
    }
  }
  hist[threadIdx.x] = 0u;
}

int metric_blocks(long long n) {
  long long b = (n + MT * 8 - 1) / (MT * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" size_t jspsr_metrics_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const long long n = (long long)H * W;
  return (size_t)n * sizeof(float) + (size_t)metric_blocks(n) * 2 * sizeof(float) + 256 * sizeof(unsigned) + 4 * sizeof(SelState) + 64;
}

extern "C" int jspsr_metrics_forward(const float* pred, const float* gt, int H, int W, float border, float value_min,
                                     float value_max, int elev_log, float* scores, void* workspace, jspsr_stream_t stream) {
  if (!pred || !gt || !scores || !workspace || H <= 0 || W <= 0) return fail(JSPSR_EINVAL, "metrics_forward: bad arguments");
  if (!(border >= 0.f) || border >= 0.5f) return fail(JSPSR_EINVAL, "metrics_forward: border must be in [0, 0.5)");
  if (!(value_max - value_min > 1.f)) return fail(JSPSR_EINVAL, "metrics_forward: value_max - value_min must exceed 1");
  if (!aligned16(workspace)) return fail(JSPSR_EALIGN, "metrics_forward: workspace not 16-byte aligned");
  const int bh = (int)((float)H * border), bw = (int)((float)W * border);     // int(h * border), metrics.py:172-183
  const int h = H - 2 * bh, w = W - 2 * bw;
  if (h <= 0 || w <= 0) return fail(JSPSR_EINVAL, "metrics_forward: nothing left after the border crop");
  const long long n = (long long)h * w;
  const int blocks = metric_blocks(n);
  char* ws = static_cast<char*>(workspace);
  float* dh = reinterpret_cast<float*>(ws);
  size_t off = ((size_t)H * W * sizeof(float) + 15) & ~(size_t)15;
  float* partial = reinterpret_cast<float*>(ws + off);
  off += ((size_t)metric_blocks((long long)H * W) * 2 * sizeof(float) + 15) & ~(size_t)15;
  unsigned* hist = reinterpret_cast<unsigned*>(ws + off);
  off += 256 * sizeof(unsigned);
  SelState* st = reinterpret_cast<SelState*>(ws + off);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(metrics_prepare_kernel, dim3(blocks), dim3(MT), 0, s, pred, gt, H, W, bh, bw, value_min, value_max,
                     elev_log, dh, partial);
  if (int e = check_launch("metrics_prepare")) return e;
  hipLaunchKernelGGL(metrics_reduce_kernel, dim3(1), dim3(MT), 0, s, partial, blocks, n, scores, st, hist);
  if (int e = check_launch("metrics_reduce")) return e;
  // scores[2] = median, [3] = NMAD = 1.4826 * median |dh - median|, [4] = LE95
  const int modes[3] = {0, 1, 2};
  const float scales[3] = {1.f, 1.4826f, 1.f};
  for (int q = 0; q < 3; ++q) {
    for (int pass = 3; pass >= 0; --pass) {
      hipLaunchKernelGGL(select_hist_kernel, dim3(blocks), dim3(MT), 0, s, dh, n, modes[q], pass, st + q, hist);
      hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(256), 0, s, hist, pass, st + q, scales[q], scores, 2 + q,
                         q == 0 ? st + 1 : nullptr);
    }
  }
  return check_launch("metrics_select");
}
