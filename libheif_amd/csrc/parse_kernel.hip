// parse_kernel.hip — CABAC entropy decoding + syntax parsing, one wavefront per independent substream.
//
// The parser itself is parse_core.h (a wave-uniform scalar instruction stream over lane-indexed
// register files; see the header for the MI355X mapping).  This translation unit is the launch
// shell: work is handed out by an atomic ticket so that a WPP row's predecessor (always an earlier
// substream index) is resident or finished before the row can wait on it, and every wait is
// bounded.
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "kernels.h"
#include "parse_core.h"

namespace hipdec {

__global__ __launch_bounds__(64) void k_parse(ParseArgs A)
{
  __shared__ pcore::Lds lds;
  const int lane = (int)threadIdx.x;
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(A.ticket, 1u);
  const uint32_t wave_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  for (int i = lane * 8; i < 32 * 32; i += 512) *(uint4*)&lds.coef[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (wave_idx >= A.num_waves) return;
  const uint32_t first = pcore::uload32(&A.waves[wave_idx].first), stride = pcore::uload32(&A.waves[wave_idx].stride),
                 end = pcore::uload32(&A.waves[wave_idx].end), lag = pcore::uload32(&A.waves[wave_idx].start_lag);
  for (uint32_t sub = first; sub < end; sub += stride)
    if (pcore::parse_substream(A, sub, stride == 1, lag, &lds)) break;
}

void launch_parse(const ParseArgs& a, hipStream_t s)
{
  if (a.num_waves) hipLaunchKernelGGL(k_parse, dim3(a.num_waves), dim3(64), 0, s, a);
}

}  // namespace hipdec
