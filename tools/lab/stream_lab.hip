// Kernel lab (not product): what can a 27-plane-read / N-plane-write stream reach on MI355X?
// hipcc --offload-arch=gfx950 -O3 -o stream_lab stream_lab.hip && ./stream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NIN, int NOUT, typename V>
__global__ __launch_bounds__(256) void stream_planes(const V* __restrict__ in, V* __restrict__ out, size_t plane_v, size_t n_v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_v; i += (size_t)gridDim.x * blockDim.x) {
    V acc = in[i];
#pragma unroll
    for (int p = 1; p < NIN; ++p) {
      V v = in[p * plane_v + i];
      if constexpr (sizeof(V) == 16) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
      else acc += v;
    }
#pragma unroll
    for (int p = 0; p < NOUT; ++p) out[p * plane_v + i] = acc;
  }
}

// batch-aware variant: planes of one image are contiguous ([B][C][H*W]) like the real operands
template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void stream_bchw(const float* __restrict__ in, float* __restrict__ out, int P, int B) {
  const int tiles = P / 256;
  for (int t = blockIdx.x; t < tiles * B; t += gridDim.x) {
    const int b = t / tiles, i = (t % tiles) * 256 + threadIdx.x;
    const float* ip = in + (size_t)b * NIN * P + i;
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < NIN; ++p) acc += ip[(size_t)p * P];
    float* op = out + (size_t)b * NOUT * P + i;
#pragma unroll
    for (int p = 0; p < NOUT; ++p) op[(size_t)p * P] = acc;
  }
}

template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void stream_strided(const float* __restrict__ in, float* __restrict__ out, int P, int B, size_t PS, size_t IS_in, size_t IS_out) {
  const int tiles = P / 256;
  for (int t = blockIdx.x; t < tiles * B; t += gridDim.x) {
    const int b = t / tiles, i = (t % tiles) * 256 + threadIdx.x;
    const float* ip = in + (size_t)b * IS_in + i;
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < NIN; ++p) acc += ip[(size_t)p * PS];
    float* op = out + (size_t)b * IS_out + i;
#pragma unroll
    for (int p = 0; p < NOUT; ++p) op[(size_t)p * PS] = acc;
  }
}

// interleaved (NHWC-like) operand: 27 floats per pixel contiguous; lane = pixel
template <int NIN>
__global__ __launch_bounds__(256) void stream_nhwc(const float* __restrict__ in, float* __restrict__ out, size_t npix) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    const float* ip = in + i * NIN;
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < NIN; ++p) acc += ip[p];
    out[i] = acc;
  }
}

int main() {
  const int B = 8, H = 512, W = 512, P = H * W;
  const size_t n = (size_t)B * P;
  const int NSET = 3;
  float *in[NSET], *out[NSET];
  for (int s = 0; s < NSET; ++s) {
    CK(hipMalloc(&in[s], n * 30 * 4));
    CK(hipMalloc(&out[s], n * 30 * 4));
    CK(hipMemset(in[s], 0x3c, n * 28 * 4));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 3; ++i) launch(i % NSET);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int it = 30;
    for (int i = 0; i < it; ++i) launch(i % NSET);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.2f us  %7.0f GB/s\n", name, ms / it * 1e3, bytes / (ms / it) / 1e6);
    return 0;
  };
  for (int pad : {0, 64, 192, 1024, 4160}) {
    char nm[128];
    const size_t PS = P + pad;
    snprintf(nm, 128, "27in/1out strided pad=%d grid=8192", pad);
    timeit(nm, 28.0 * n * 4, [&](int s) { hipLaunchKernelGGL((stream_strided<27, 1>), dim3(8192), dim3(256), 0, 0, in[s], out[s], P, B, PS, 27 * PS, 1 * PS); });
    snprintf(nm, 128, "27in/25out strided pad=%d grid=8192", pad);
    timeit(nm, 52.0 * n * 4, [&](int s) { hipLaunchKernelGGL((stream_strided<27, 25>), dim3(8192), dim3(256), 0, 0, in[s], out[s], P, B, PS, 27 * PS, 25 * PS); });
  }
  for (int grid : {2048, 8192}) {
    char nm[128];
    snprintf(nm, 128, "27-interleaved in/1out grid=%d", grid);
    timeit(nm, 28.0 * n * 4, [&](int s) { hipLaunchKernelGGL((stream_nhwc<27>), dim3(grid), dim3(256), 0, 0, in[s], out[s], n); });
  }
  for (int grid : {8192}) {
    char nm[128];
    snprintf(nm, 128, "27in/1out float4 grid=%d", grid);
    timeit(nm, 28.0 * n * 4, [&](int s) { hipLaunchKernelGGL((stream_planes<27, 1, float4>), dim3(grid), dim3(256), 0, 0, (const float4*)in[s], (float4*)out[s], n / 4, n / 4); });
    snprintf(nm, 128, "27in/1out float  grid=%d", grid);
    timeit(nm, 28.0 * n * 4, [&](int s) { hipLaunchKernelGGL((stream_planes<27, 1, float>), dim3(grid), dim3(256), 0, 0, in[s], out[s], n, n); });
    snprintf(nm, 128, "27in/1out float bchw grid=%d", grid);
    timeit(nm, 28.0 * n * 4, [&](int s) { hipLaunchKernelGGL((stream_bchw<27, 1>), dim3(grid), dim3(256), 0, 0, in[s], out[s], P, B); });
    snprintf(nm, 128, "27in/25out float bchw grid=%d", grid);
    timeit(nm, 52.0 * n * 4, [&](int s) { hipLaunchKernelGGL((stream_bchw<27, 25>), dim3(grid), dim3(256), 0, 0, in[s], out[s], P, B); });
    snprintf(nm, 128, "27in/25out float4 grid=%d", grid);
    timeit(nm, 52.0 * n * 4, [&](int s) { hipLaunchKernelGGL((stream_planes<27, 25, float4>), dim3(grid), dim3(256), 0, 0, (const float4*)in[s], (float4*)out[s], n / 4, n / 4); });
    snprintf(nm, 128, "1in/1out float4 (copy) grid=%d", grid);
    timeit(nm, 2.0 * n * 28 * 4, [&](int s) { hipLaunchKernelGGL((stream_planes<1, 1, float4>), dim3(grid), dim3(256), 0, 0, (const float4*)in[s], (float4*)out[s], (size_t)0, n * 28 / 4); });
  }
  return 0;
}
