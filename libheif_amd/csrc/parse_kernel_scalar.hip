// parse_kernel_scalar.hip — the LATENCY variant of the CABAC parse kernel: the same parse_core.h, compiled with the arithmetic
// decoder's state (range / value / bits_needed) in SCALAR registers.  The throughput kernels (parse_kernel.hip) keep that state
// in vector registers because 28 waves per CU saturate the CU-shared scalar pipe; a lone still or a small grid puts at most a
// few waves on a CU, the scalar pipe is idle and its dependent-issue latency is what bounds a substream.
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "kernels.h"
#define HIPDEC_PARSE_SCALAR_CABAC 1
#define pcore pcore_scalar          // own namespace: this translation unit's inline functions differ from parse_kernel.hip's
#include "parse_core.h"

namespace hipdec {

__global__ __launch_bounds__(64) void k_parse_scalar(ParseArgs A)
{
  __shared__ pcore::Lds lds;
  const int lane = (int)threadIdx.x;
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(A.ticket, 1u);
  const uint32_t wave_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  for (int i = lane * 8; i < 32 * 32; i += 512) *(uint4*)&lds.coef[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (wave_idx >= A.num_waves) return;
  pcore::parse_wave(A, wave_idx, &lds);
}

void launch_parse_scalar(const ParseArgs& a, hipStream_t s)
{
  if (a.num_waves) hipLaunchKernelGGL(k_parse_scalar, dim3(a.num_waves), dim3(64), 0, s, a);
}

}  // namespace hipdec
