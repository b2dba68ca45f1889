// Lab probe: what does `buffer_load_dwordx4 ... lds` write to LDS for a lane whose offset fails the descriptor's range
// check?  (conv64.hip relies on the answer for zero padding.)  Prints the LDS image after one piece with odd lanes OOB.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, 1024, 0x00020000);
  const unsigned off = (threadIdx.x & 1) ? 0xFFFFFFF0u : threadIdx.x * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)lds, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  unsigned *src, *out, h[256];
  (void)hipMalloc(&src, 1024); (void)hipMalloc(&out, 1024);
  for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
  (void)hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out);
  (void)hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  int zeros = 0, kept = 0, good = 0;
  for (int l = 0; l < 64; ++l)
    for (int k = 0; k < 4; ++k) {
      const unsigned v = h[l * 4 + k];
      if (l & 1) { zeros += v == 0; kept += v == 0xdeadbeefu; } else good += v == 0x1000u + l * 4 + k;
    }
  printf("in-range dwords correct %d/128; OOB lanes: zero %d/128, untouched %d/128; sample lane1: %08x\n", good, zeros, kept, h[4]);
  return 0;
}
