// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS project's access widths (MI355X guide, HBM section:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated:
// calibrate on a known byte count in your own access pattern before trusting an absolute").  Kernels that read / write a KNOWN number of bytes
// (1 GiB each, far beyond the 256 MiB Infinity Cache) with the widths the decode kernels use: 1, 4 and 16 bytes per lane.
// build: hipcc --offload-arch=gfx950 -O3 -o build/fetch_calib tools/ubench/fetch_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <typename T> __global__ void k_calib_read(const T* __restrict__ src, size_t n, uint32_t* sink)
{
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = src[i];
    const unsigned char* b = (const unsigned char*)&v;
    for (unsigned k = 0; k < sizeof(T); k += sizeof(T) > 4 ? 4 : sizeof(T)) acc += b[k];
  }
  if (acc == 0x12345678u) *sink = acc;   // (never true for the zero-filled / patterned buffer: keeps the loads alive)
}
template <typename T> __global__ void k_calib_write(T* __restrict__ dst, size_t n, T v)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

int main()
{
  const size_t bytes = size_t(1) << 30;
  void *a = nullptr, *b = nullptr; uint32_t* sink = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  const dim3 grid(256 * 32), block(256);
  hipLaunchKernelGGL(k_calib_read<uint8_t>, grid, block, 0, 0, (const uint8_t*)a, bytes, sink);
  hipLaunchKernelGGL(k_calib_read<uint32_t>, grid, block, 0, 0, (const uint32_t*)a, bytes / 4, sink);
  hipLaunchKernelGGL(k_calib_read<uint4>, grid, block, 0, 0, (const uint4*)a, bytes / 16, sink);
  hipLaunchKernelGGL(k_calib_write<uint8_t>, grid, block, 0, 0, (uint8_t*)b, bytes, (uint8_t)3);
  hipLaunchKernelGGL(k_calib_write<uint32_t>, grid, block, 0, 0, (uint32_t*)b, bytes / 4, 0x03030303u);
  hipLaunchKernelGGL(k_calib_write<uint4>, grid, block, 0, 0, (uint4*)b, bytes / 16, make_uint4(3, 3, 3, 3));
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
  printf("each kernel moved %zu bytes\n", bytes);
  return 0;
}
