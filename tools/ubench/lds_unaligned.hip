// Does an LDS dword read at a byte address that is not a multiple of 4 return the four bytes at that address on this chip (the compiler emits ONE
// ds_read_b32 for an align-1 load: gfx950 is built with unaligned DS access)?  Every thread reads at offsets 0 .. 3 from its own slot and compares
// with the bytes.  Prints "unaligned LDS dword reads: ok" or the first mismatch.   build: hipcc --offload-arch=gfx950 -O3 -o build/ubench/lds_unaligned tools/ubench/lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed)) U1 { uint32_t v; };
__global__ void k(uint32_t* out)
{
  __shared__ uint32_t t[256 + 2];
  for (int i = threadIdx.x; i < 258 * 4; i += 256) ((uint8_t*)t)[i] = (uint8_t)i;   // byte b of the array holds b & 255
  __syncthreads();
  for (int off = 0; off < 4; off++) {
    const uint8_t* p = (const uint8_t*)t + threadIdx.x * 4 + off;
    out[threadIdx.x * 4 + off] = ((const U1*)p)->v;
  }
}
int main()
{
  uint32_t* d; uint32_t h[1024];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int i = 0; i < 1024; i++) {
    const int b = (i / 4) * 4 + (i & 3);
    const uint32_t want = (uint32_t)(b & 255) | ((uint32_t)((b + 1) & 255) << 8) | ((uint32_t)((b + 2) & 255) << 16) | ((uint32_t)((b + 3) & 255) << 24);
    if (h[i] != want) { printf("unaligned LDS dword reads: MISMATCH at byte %d: got %08x want %08x\n", b, h[i], want); return 1; }
  }
  printf("unaligned LDS dword reads: ok\n");
  return 0;
}
