// parse_kernel_general.hip — the CABAC parse kernel for batches that hold 4:2:2 or 4:4:4 pictures: the same parse_core.h as
// parse_kernel.hip, compiled WITH the ChromaArrayType 2 / 3 paths (two stacked chroma blocks per unit with their own flags; chroma blocks of
// luma size, per-partition chroma modes, a fifth cbf context).  The throughput kernels of parse_kernel.hip are built without them because the
// scalar pipe bounds the parser and the extra block-loop bookkeeping costs the all-4:2:0 batch 1.5 % (profiles/r03_422_bench_main.json against
// profiles/r03z_final_bench.json); launch_parse() sends a batch here when the host found such a picture in it (ParseArgs::general_chroma).
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "kernels.h"
#define HIPDEC_PARSE_CHROMA_GENERAL 1
#define pcore pcore_general          // own namespace: this translation unit's inline functions differ from parse_kernel.hip's
#include "parse_core.h"

namespace hipdec {

#define HIPDEC_PARSE_BODY                                                                                   \
    __shared__ pcore::Lds lds;                                                                            \
    const int lane = (int)threadIdx.x;                                                                    \
    uint32_t t = 0;                                                                                       \
    if (lane == 0) t = atomicAdd(A.ticket, 1u);                                                           \
    const uint32_t wave_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);                           \
    for (int i = lane * 8; i < 32 * 32; i += 512) *(uint4*)&lds.coef[i] = make_uint4(0, 0, 0, 0);         \
    __syncthreads();                                                                                      \
    if (wave_idx >= A.num_waves) return;                                                                  \
    pcore::parse_wave(A, wave_idx, &lds);                                                                 \

// latency mode (a lone still: all registers) and throughput mode (8 waves per SIMD), as in parse_kernel.hip
__global__ __launch_bounds__(64) void k_parse_gen(ParseArgs A) { HIPDEC_PARSE_BODY }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_parse_gen_occ8(ParseArgs A) { HIPDEC_PARSE_BODY }

void launch_parse_general(const ParseArgs& a, bool throughput, hipStream_t s)
{
  if (!a.num_waves) return;
  if (throughput) hipLaunchKernelGGL(k_parse_gen_occ8, dim3(a.num_waves), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(k_parse_gen, dim3(a.num_waves), dim3(64), 0, s, a);
}

}  // namespace hipdec
