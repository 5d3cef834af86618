// queue_overlap.hip — do kernels of two HIP streams overlap the way the stage-overlap pipeline needs?  (dev tool, round 6)
//   stream A: P0, P1, P2 ...   (a "pool": W workgroups of 64 threads that each spin for T ms - resident, few slots)
//   stream B: after P_k (event): X_k = several launches of a huge grid of short workgroups (the pixel kernels)
// Prints the timeline from device timestamps the kernels take themselves.  usage: queue_overlap [pool_wgs] [delay_us]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(64) void k_pool(uint32_t ticks, uint64_t* stamp)
{
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ __launch_bounds__(256) void k_pixels(float* buf, size_t n, uint64_t* stamp)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = __builtin_amdgcn_s_memrealtime();
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) buf[i] = buf[i] * 1.0001f + 1.0f;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void k_delay(uint32_t ticks) { const uint64_t t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64); }
int main(int argc, char** argv)
{
  const int pool = argc > 1 ? atoi(argv[1]) : 6144, delay_us = argc > 2 ? atoi(argv[2]) : 0, prio = argc > 3 ? atoi(argv[3]) : 0;
  hipStream_t A, B;
  int least = 0, greatest = 0; hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
  if (prio) hipStreamCreateWithPriority(&B, hipStreamNonBlocking, least); else hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
  const size_t n = size_t(1) << 30;   // 4 GB of floats per launch: ~1.5 ms each at 6 TB/s; 8 launches per stage
  float* buf; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
  const int K = 4;
  uint64_t* st; hipHostMalloc(&st, sizeof(uint64_t) * 4 * K);
  std::vector<hipEvent_t> ev(K);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int rep = 0; rep < 2; rep++) {
    for (int k = 0; k < K; k++) {
      hipLaunchKernelGGL(k_pool, dim3(pool), dim3(64), 0, A, 2000000u /* 20 ms */, st + 4 * k);
      hipEventRecord(ev[k], A);
      hipStreamWaitEvent(B, ev[k], 0);
      if (delay_us) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, B, (uint32_t)delay_us * 100u);
      for (int j = 0; j < 8; j++) hipLaunchKernelGGL(k_pixels, dim3((unsigned)(n / 256)), dim3(256), 0, B, buf, n, st + 4 * k + 2);
    }
    hipDeviceSynchronize();
  }
  const uint64_t t0 = st[0];
  for (int k = 0; k < K; k++)
    printf("pool %d: %8.2f .. %8.2f ms      last pixel launch of stage %d: %8.2f .. %8.2f ms\n", k, (st[4 * k] - t0) / 1e5, (st[4 * k + 1] - t0) / 1e5, k, (st[4 * k + 2] - t0) / 1e5, (st[4 * k + 3] - t0) / 1e5);
  return 0;
}
