// The two ends of the training step that touch every pixel / every parameter exactly once:
//   * fused L1 + L2 + Sobel-L1 loss, forward and gradient (reference: MultiLoss, losses/loss_schemes.py:
//     55-72 with configs/*.yml:67-70; EdgeLoss = L1 between kornia spatial_gradient(sobel, normalised)
//     of prediction and target, losses/loss_functions.py:171-185).  The Sobel operator is linear, so the
//     edge term is computed on d = pred - gt; replicate padding is a clamp of the neighbour index.
//   * fused multi-tensor AdamW over one flat parameter/gradient buffer (torch.optim.AdamW semantics,
//     utils/common_config.py:241-291, configs/*.yml:71-76): HBM-bound, 16 B read + 12 B written per
//     parameter.
#include "common.h"

namespace {

using namespace jspsr;

constexpr int LT = 256;

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// pass 1: per-pixel d, Sobel of d, partial sums {sum|d|, sum d^2, sum(|gx|+|gy|)}, signs of gx, gy
__global__ __launch_bounds__(LT) void loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                     signed char* __restrict__ sg, float* __restrict__ partial,
                                                     int B, int H, int W) {
  __shared__ float red[3][LT / 64];
  const long long n = (long long)B * H * W;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (long long i = blockIdx.x * (long long)LT + threadIdx.x; i < n; i += (long long)gridDim.x * LT) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const long long base = i - (long long)y * W - x;
    const int ym = y > 0 ? y - 1 : 0, yp = y < H - 1 ? y + 1 : H - 1;
    const int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : W - 1;
    auto D = [&](int yy, int xx) { const long long j = base + (long long)yy * W + xx; return pred[j] - gt[j]; };
    const float d = D(y, x);
    const float a = D(ym, xm), b = D(ym, x), c = D(ym, xp), e = D(y, xm), f = D(y, xp), g = D(yp, xm), h = D(yp, x), k = D(yp, xp);
    const float gx = ((c - a) + 2.f * (f - e) + (k - g)) * 0.125f;
    const float gy = ((g - a) + 2.f * (h - b) + (k - c)) * 0.125f;
    s1 += fabsf(d);
    s2 += d * d;
    s3 += fabsf(gx) + fabsf(gy);
    sg[2 * i] = (signed char)sgn(gx);
    sg[2 * i + 1] = (signed char)sgn(gy);
  }
  float v[3] = {s1, s2, s3};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int d2 = 32; d2 > 0; d2 >>= 1) v[q] += __shfl_xor(v[q], d2, 64);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float t = 0.f;
    for (int w = 0; w < LT / 64; ++w) t += red[threadIdx.x][w];
    partial[(size_t)blockIdx.x * 3 + threadIdx.x] = t;
  }
}

// losses[0..3] = {L1, L2, Grad, Total}
__global__ void loss_finalize_kernel(const float* __restrict__ partial, int rows, long long n, float w1, float w2,
                                     float wg, float* __restrict__ losses) {
  __shared__ double acc[3];
  const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (q < 3) {
    double s = 0.0;
    for (int r = lane; r < rows; r += 64) s += (double)partial[(size_t)r * 3 + q];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) acc[q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double l1 = acc[0] / (double)n, l2 = acc[1] / (double)n, lg = acc[2] / (2.0 * (double)n);
    losses[0] = (float)l1; losses[1] = (float)l2; losses[2] = (float)lg;
    losses[3] = (float)(w1 * l1 + w2 * l2 + wg * lg);
  }
}

// pass 2: d(Total)/d(pred) * upstream scalar gradient.  Adjoint of the replicate-padded Sobel: pixel p
// collects k[delta] * sign(g[q]) from every (q, delta) with clamp(q + delta) == p.
__global__ __launch_bounds__(LT) void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                     const signed char* __restrict__ sg, const float* __restrict__ gscale,
                                                     float w1, float w2, float wg, float* __restrict__ gpred, int B,
                                                     int H, int W) {
  const long long n = (long long)B * H * W;
  const float up = gscale ? gscale[0] : 1.f;
  const float c1 = up * w1 / (float)n, c2 = up * 2.f * w2 / (float)n, cg = up * wg / (2.f * (float)n) * 0.125f;
  // kernels: gx weight kx[dy][dx] = {-1,0,1; -2,0,2; -1,0,1}, gy weight = its transpose
  for (long long i = blockIdx.x * (long long)LT + threadIdx.x; i < n; i += (long long)gridDim.x * LT) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const long long base = i - (long long)y * W - x;
    const float d = pred[i] - gt[i];
    float acc = 0.f;
    for (int qy = y - 1; qy <= y + 1; ++qy) {
      if (qy < 0 || qy >= H) continue;
      for (int qx = x - 1; qx <= x + 1; ++qx) {
        if (qx < 0 || qx >= W) continue;
        const long long j = base + (long long)qy * W + qx;
        const float sx = (float)sg[2 * j], sy = (float)sg[2 * j + 1];
        if (sx == 0.f && sy == 0.f) continue;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
          int ty = qy + dy; ty = ty < 0 ? 0 : (ty > H - 1 ? H - 1 : ty);
          if (ty != y) continue;
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            int tx = qx + dx; tx = tx < 0 ? 0 : (tx > W - 1 ? W - 1 : tx);
            if (tx != x) continue;
            const float kx = (float)dx * (dy == 0 ? 2.f : 1.f);
            const float ky = (float)dy * (dx == 0 ? 2.f : 1.f);
            acc += kx * sx + ky * sy;
          }
        }
      }
    }
    gpred[i] = c1 * sgn(d) + c2 * d + cg * acc;
  }
}

// AdamW, decoupled weight decay, bias-corrected (torch.optim.AdamW, maximize=False, amsgrad=False)
__device__ __forceinline__ void adamw_one(float& P, float G, float& M, float& V, float lr, float beta1, float beta2,
                                          float eps, float wd, float bc1, float bc2s) {
  P *= 1.f - lr * wd;
  M = beta1 * M + (1.f - beta1) * G;
  V = beta2 * V + (1.f - beta2) * G * G;
  P -= (lr / bc1) * M / (sqrtf(V) / bc2s + eps);
}

// `head` scalar elements bring the four (equally misaligned) buffers to a 16-byte boundary, then float4s, then a tail.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, int head,
                                                   float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                   float bc2s) {
  const long long n4 = (n - head) / 4;
  float4* p4 = reinterpret_cast<float4*>(p + head);
  const float4* g4 = reinterpret_cast<const float4*>(g + head);
  float4* m4 = reinterpret_cast<float4*>(m + head);
  float4* v4 = reinterpret_cast<float4*>(v + head);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 P = p4[i];
    const float4 G = g4[i];
    float4 M = m4[i], V = v4[i];
    adamw_one(P.x, G.x, M.x, V.x, lr, beta1, beta2, eps, wd, bc1, bc2s);
    adamw_one(P.y, G.y, M.y, V.y, lr, beta1, beta2, eps, wd, bc1, bc2s);
    adamw_one(P.z, G.z, M.z, V.z, lr, beta1, beta2, eps, wd, bc1, bc2s);
    adamw_one(P.w, G.w, M.w, V.w, lr, beta1, beta2, eps, wd, bc1, bc2s);
    p4[i] = P;
    m4[i] = M;
    v4[i] = V;
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) {  // up to 3 head + 3 tail elements
    const int tail = (int)(n - head - n4 * 4);
    long long i = -1;
    if ((int)threadIdx.x < head) i = threadIdx.x;
    else if ((int)threadIdx.x >= 4 && (int)threadIdx.x - 4 < tail) i = head + n4 * 4 + (threadIdx.x - 4);
    if (i >= 0) {
      float P = p[i], M = m[i], V = v[i];
      adamw_one(P, g[i], M, V, lr, beta1, beta2, eps, wd, bc1, bc2s);
      p[i] = P; m[i] = M; v[i] = V;
    }
  }
}

// The same update with its seven scalars -- lr, beta1, beta2, eps, weight decay, the two bias corrections -- read from DEVICE
// memory: a launch captured in a hipGraph carries its kernel arguments verbatim, and the learning rate and the bias
// corrections change from step to step (jspsr_adamw_step_dev; the caller refreshes the seven floats before every replay).
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long long n, int head,
                                                       const float* __restrict__ hyper) {
  const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2s = hyper[6];
  const long long n4 = (n - head) / 4;
  float4* p4 = reinterpret_cast<float4*>(p + head);
  const float4* g4 = reinterpret_cast<const float4*>(g + head);
  float4* m4 = reinterpret_cast<float4*>(m + head);
  float4* v4 = reinterpret_cast<float4*>(v + head);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 P = p4[i];
    const float4 G = g4[i];
    float4 M = m4[i], V = v4[i];
    adamw_one(P.x, G.x, M.x, V.x, lr, beta1, beta2, eps, wd, bc1, bc2s);
    adamw_one(P.y, G.y, M.y, V.y, lr, beta1, beta2, eps, wd, bc1, bc2s);
    adamw_one(P.z, G.z, M.z, V.z, lr, beta1, beta2, eps, wd, bc1, bc2s);
    adamw_one(P.w, G.w, M.w, V.w, lr, beta1, beta2, eps, wd, bc1, bc2s);
    p4[i] = P;
    m4[i] = M;
    v4[i] = V;
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) {  // up to 3 head + 3 tail elements
    const int tail = (int)(n - head - n4 * 4);
    long long i = -1;
    if ((int)threadIdx.x < head) i = threadIdx.x;
    else if ((int)threadIdx.x >= 4 && (int)threadIdx.x - 4 < tail) i = head + n4 * 4 + (threadIdx.x - 4);
    if (i >= 0) {
      float P = p[i], M = m[i], V = v[i];
      adamw_one(P, g[i], M, V, lr, beta1, beta2, eps, wd, bc1, bc2s);
      p[i] = P; m[i] = M; v[i] = V;
    }
  }
}

int loss_blocks(long long n) {
  long long b = (n + LT - 1) / LT;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" size_t jspsr_loss_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const long long n = (long long)B * H * W;
  return (((size_t)loss_blocks(n) * 3 * sizeof(float) + 15) & ~(size_t)15) + (size_t)n * 2;
}

extern "C" int jspsr_loss_forward(const float* pred, const float* gt, float w1, float w2, float wg, float* losses,
                                  void* workspace, int B, int H, int W, jspsr_stream_t stream) {
  if (!pred || !gt || !losses || !workspace || B <= 0 || H <= 0 || W <= 0) return fail(JSPSR_EINVAL, "loss_forward: bad arguments");
  const long long n = (long long)B * H * W;
  const int blocks = loss_blocks(n);
  float* partial = static_cast<float*>(workspace);
  signed char* sg = static_cast<signed char*>(workspace) + (((size_t)blocks * 3 * sizeof(float) + 15) & ~(size_t)15);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(blocks), dim3(LT), 0, s, pred, gt, sg, partial, B, H, W);
  if (int e = check_launch("loss_forward")) return e;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(192), 0, s, partial, blocks, n, w1, w2, wg, losses);
  return check_launch("loss_finalize");
}

extern "C" int jspsr_loss_backward(const float* pred, const float* gt, const float* grad_total, float w1, float w2,
                                   float wg, float* grad_pred, const void* workspace, int B, int H, int W,
                                   jspsr_stream_t stream) {
  if (!pred || !gt || !grad_pred || !workspace || B <= 0 || H <= 0 || W <= 0) return fail(JSPSR_EINVAL, "loss_backward: bad arguments");
  const long long n = (long long)B * H * W;
  const int blocks = loss_blocks(n);
  const signed char* sg = static_cast<const signed char*>(workspace) + (((size_t)blocks * 3 * sizeof(float) + 15) & ~(size_t)15);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(blocks), dim3(LT), 0, static_cast<hipStream_t>(stream), pred, gt, sg, grad_total,
                     w1, w2, wg, grad_pred, B, H, W);
  return check_launch("loss_backward");
}

extern "C" int jspsr_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int step, jspsr_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return fail(JSPSR_EINVAL, "adamw_step: bad arguments");
  const uintptr_t mis = reinterpret_cast<uintptr_t>(param) & 15;
  if ((mis & 3) || (reinterpret_cast<uintptr_t>(grad) & 15) != mis || (reinterpret_cast<uintptr_t>(exp_avg) & 15) != mis ||
      (reinterpret_cast<uintptr_t>(exp_avg_sq) & 15) != mis)
    return fail(JSPSR_EALIGN, "adamw_step: buffers must be 4-byte aligned and equally offset from a 16-byte boundary");
  int head = (int)(((16 - mis) & 15) >> 2);
  if (head > n) head = (int)n;
  // bias corrections in double on the host, as torch.optim.AdamW computes them (float powf is ~3e-5 relative off at
  // small step counts)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  long long b = (n / 4 + 255) / 256;
  const int blocks = (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad, exp_avg,
                     exp_avg_sq, n, head, lr, beta1, beta2, eps, weight_decay, bc1, bc2s);
  return check_launch("adamw_step");
}

extern "C" int jspsr_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                                    const float* hyper, jspsr_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n <= 0) return fail(JSPSR_EINVAL, "adamw_step_dev: bad arguments");
  const uintptr_t mis = reinterpret_cast<uintptr_t>(param) & 15;
  if ((mis & 3) || (reinterpret_cast<uintptr_t>(grad) & 15) != mis || (reinterpret_cast<uintptr_t>(exp_avg) & 15) != mis ||
      (reinterpret_cast<uintptr_t>(exp_avg_sq) & 15) != mis || (reinterpret_cast<uintptr_t>(hyper) & 3))
    return fail(JSPSR_EALIGN, "adamw_step_dev: buffers must be 4-byte aligned and equally offset from a 16-byte boundary");
  int head = (int)(((16 - mis) & 15) >> 2);
  if (head > n) head = (int)n;
  long long b = (n / 4 + 255) / 256;
  const int blocks = (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
  hipLaunchKernelGGL(adamw_dev_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad, exp_avg,
                     exp_avg_sq, n, head, hyper);
  return check_launch("adamw_step");
}
