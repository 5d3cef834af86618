// parse_kernel_inter.hip — the CABAC parse kernel for batches that hold P pictures (the samples of a sequence track): the same
// parse_core.h as parse_kernel.hip, compiled WITH the syntax a P slice adds (cu_skip_flag, pred_mode_flag, inter part_mode, prediction_unit,
// mvd_coding, rqt_root_cbf, the inter transform tree; contexts of initType 1 / 2).  The throughput kernels of parse_kernel.hip are built without it:
// the scalar pipe bounds the parser and stills never take these branches.  launch_parse() sends a batch here when the host found a P slice in it
// (ParseArgs::inter).  4:0:0 / 4:2:0 only (the host refuses P slices of other chroma formats), one register budget (sequences are a latency path).
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "kernels.h"
#define HIPDEC_PARSE_CHROMA_GENERAL 0
#define HIPDEC_PARSE_INTER 1
#define pcore pcore_inter            // own namespace: this translation unit's inline functions differ from parse_kernel.hip's
#include "parse_core.h"

namespace hipdec {

__global__ __launch_bounds__(64) void k_parse_inter(ParseArgs A)
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

void launch_parse_inter(const ParseArgs& a, hipStream_t s)
{
  if (!a.num_waves) return;
  hipLaunchKernelGGL(k_parse_inter, dim3(a.num_waves), dim3(64), 0, s, a);
}

}  // namespace hipdec
