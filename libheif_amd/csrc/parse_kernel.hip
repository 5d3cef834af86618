// parse_kernel.hip — CABAC entropy decoding + syntax parsing, one wavefront per independent substream.
//
// The parser itself is parse_core.h (a wave-uniform scalar instruction stream over lane-indexed
// register files; see the header for the MI355X mapping).  This translation unit is the launch
// shell: work is handed out by an atomic ticket so that a WPP row's predecessor (always an earlier
// substream index) is resident or finished before the row can wait on it, and every wait is
// bounded.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hevc_device.h"
#include "kernels.h"
#define HIPDEC_PARSE_CHROMA_GENERAL 0   // 4:0:0 / 4:2:0 pictures only; batches with 4:2:2 / 4:4:4 pictures go to parse_kernel_general.hip
#include "parse_core.h"

namespace hipdec {

// The kernel body at three register budgets: all registers (latency mode: a lone still), 8 waves per SIMD (throughput mode: the measured
// best on MI355X, profiles/r03_occupancy_sweep.txt) and 6 (80 VGPRs, less scratch: HIPDEC_PARSE_OCCUPANCY=6 selects it for measurements).
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

__global__ __launch_bounds__(64) void k_parse(ParseArgs A) { HIPDEC_PARSE_BODY }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_parse_occ6(ParseArgs A) { HIPDEC_PARSE_BODY }
// (the register-file build at 8 waves per SIMD: round 5's throughput kernel, kept for A/B measurements - HIPDEC_PARSE_CTX=rf; the throughput kernel
//  k_parse_occ8 is parse_kernel_tp.hip's, with LDS-resident contexts)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_parse_occ8_rf(ParseArgs A) { HIPDEC_PARSE_BODY }

void launch_parse(const ParseArgs& a, hipStream_t s)
{
  if (!a.num_waves) return;
  static const int forced = getenv("HIPDEC_PARSE_OCCUPANCY") ? atoi(getenv("HIPDEC_PARSE_OCCUPANCY")) : -1;
  // (round 2 kept a scalar-register variant of the arithmetic decoder for lone stills; with the hand-scheduled statements the vector
  //  form is faster there too — one 4K still: parse 247 -> 209 ms — and the variant is gone)
  // throughput mode (the chip is oversubscribed with parser waves): 8 waves per SIMD; latency mode: all registers
  // (pool mode: 8 waves per SIMD since the scalar / vector rebalancing of round 2 — 895 against 907 ms per 2048 4K stills with 7; before it the
  //  scalar pipe was saturated and the eighth wave bought nothing)
  const int occ = forced >= 0 ? forced : (a.pool ? 8 : (a.num_waves >= 2048 ? 8 : 0));
  if (a.inter) { launch_parse_inter(a, s); return; }
  if (a.general_chroma) { launch_parse_general(a, occ != 0, s); return; }
  static const bool rf = getenv("HIPDEC_PARSE_CTX") && getenv("HIPDEC_PARSE_CTX")[0] == 'r';
  if (occ == 8 && !rf) launch_parse_throughput(a, s);
  else if (occ == 8) hipLaunchKernelGGL(k_parse_occ8_rf, dim3(a.num_waves), dim3(64), 0, s, a);
  else if (occ == 6) hipLaunchKernelGGL(k_parse_occ6, dim3(a.num_waves), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(k_parse, dim3(a.num_waves), dim3(64), 0, s, a);
}

}  // namespace hipdec
