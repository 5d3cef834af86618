// parse_kernel_tp.hip — the THROUGHPUT build of the CABAC parser (k_parse_occ8: 8 waves per SIMD, WPP rows as tasks of the work pool): parse_core.h
// with the context variables and the rangeTabLps / transIdxLps tables in LDS (HIPDEC_PARSE_LDS_CTX, parse_bins_lds_gfx950.h).  The latency build
// (k_parse: one lone wave per substream, register-file contexts, shorter dependent chains per bin) stays in parse_kernel.hip.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hevc_device.h"
#include "kernels.h"
#define HIPDEC_PARSE_CHROMA_GENERAL 0   // 4:0:0 / 4:2:0 pictures only; batches with 4:2:2 / 4:4:4 pictures go to parse_kernel_general.hip
#define HIPDEC_PARSE_LDS_CTX 1
#include "parse_core.h"

namespace hipdec {

#ifndef HIPDEC_TP_OCC
#define HIPDEC_TP_OCC 8   // (measurement builds: tools/ab_variant.sh tp7 -DHIPDEC_TP_OCC=7 with HIPDEC_POOL_WAVES=7168)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HIPDEC_TP_OCC, HIPDEC_TP_OCC))) void k_parse_occ8(ParseArgs A)
{
  __shared__ pcore::Lds lds;
  const int lane = (int)threadIdx.x;
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(A.ticket, 1u);
  const uint32_t wave_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  for (int i = lane * 8; i < 32 * 32; i += 512) *(uint4*)&lds.coef[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (wave_idx >= A.num_waves) return;
  // the hand-scheduled statements address Lds by byte offsets from LDS address 0: it is the kernel's only __shared__ object
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) pcore::Lds*)&lds != 0u) { pcore::pc_report(A.status, DEV_ERR_SYNTAX | (int32_t)0x40000000); return; }
  pcore::parse_wave(A, wave_idx, &lds);
}

void launch_parse_throughput(const ParseArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_parse_occ8, dim3(a.num_waves), dim3(64), 0, s, a); }

}  // namespace hipdec
