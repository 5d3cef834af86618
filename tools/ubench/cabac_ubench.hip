// cabac_ubench.hip — cycles per CABAC bin of the parser's primitives on one wavefront (dev tool).
//   hipcc --offload-arch=gfx950 -O3 -I libheif_amd/csrc -I include tools/ubench/cabac_ubench.hip -o /tmp/cabac_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "parse_core.h"
using namespace hipdec;
using namespace hipdec::pcore;

__global__ __launch_bounds__(64) void k_bench(const uint8_t* bs, uint32_t nbytes, int mode, int iters, uint64_t* out, uint32_t* sink)
{
  __shared__ Lds lds;
  PS s;
  s.L = &lds; s.err = 0; s.bs = bs; s.slice_qp_y = 30;
  load_tables(s);
  PC_VEC_BEGIN PC_L(s.win) = 0; PC_L(s.win_next) = 0; PC_L(s.ctxA) = 0; PC_L(s.ctxB) = 0; PC_L(s.ctxC) = 0; PC_VEC_END
  init_contexts(s);
  cabac_start(s, 0, nbytes);
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  if (mode == 0) {
    for (int i = 0; i < iters; i++) acc += (uint32_t)decode_bin(s, s.ctxB, i & 31);
  } else if (mode == 1) {
    for (int i = 0; i < iters; i++) acc += (uint32_t)decode_bypass(s);
  } else if (mode == 2) {
    for (int i = 0; i < iters; i++) { acc += (uint32_t)decode_bin(s, s.ctxB, (int)(acc & 31)); }
  } else if (mode == 3) {  // empty loop with a dependent SALU chain
    for (int i = 0; i < iters; i++) { acc = acc * 3 + (uint32_t)i; }
  } else if (mode == 7) {  // 16 dependent SALU ops per iteration
    uint32_t a = (uint32_t)iters;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) { a = a * 5u + 1u; }
    }
    acc += a;
  } else if (mode == 8) {  // 16 dependent VALU ops per iteration (uniform values in vector registers)
    UReg a = pc_vec((uint32_t)iters);
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) { a = a * 5u + 1u; }
    }
    acc += pc_uni(a);
  } else if (mode == 9) {  // 8 SALU + 8 VALU per iteration, independent chains
    uint32_t a = (uint32_t)iters; UReg b = pc_vec((uint32_t)iters + 1u);
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 4; k++) { a = a * 5u + 1u; b = b * 7u + 3u; }
    }
    acc += a + pc_uni(b);
  } else if (mode == 6) {  // taken-branch chain: 8 never-fall-through branches per iteration
    for (int i = 0; i < iters; i++) {
      asm volatile("s_branch 1f\n s_nop 0\n1: s_branch 2f\n s_nop 0\n2: s_branch 3f\n s_nop 0\n3: s_branch 4f\n s_nop 0\n4: s_branch 5f\n s_nop 0\n5: s_branch 6f\n s_nop 0\n6: s_branch 7f\n s_nop 0\n7: s_branch 8f\n s_nop 0\n8:" ::: "memory");
      acc += (uint32_t)i;
    }
  } else if (mode == 4) {  // readlane chain
    for (int i = 0; i < iters; i++) { acc = pc_rdlane(s.t_lps, (int)(acc & 63)) + (uint32_t)i; }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = acc + s.pos + pc_uni(s.range); }
}

int main(int argc, char** argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  const uint32_t nbytes = 1 << 20;
  std::vector<uint8_t> h(nbytes + 1024);
  srand(1);
  for (auto& b : h) b = (uint8_t)(rand() >> 7);
  uint8_t* d; uint64_t* out; uint32_t* sink;
  hipMalloc(&d, h.size()); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  hipMalloc(&out, 8 * 4096); hipMalloc(&sink, 4 * 4096);
  const char* names[] = {"decision bin (ctx cycling)", "bypass bin", "decision bin (ctx data-dependent)", "SALU mul-add chain", "readlane chain", "(removed)", "8 taken branches", "16 SALU ops (mul+add x8)", "16 VALU ops (mul+add x8)", "8 SALU + 8 VALU"};
  for (int blocks : {1, 4096, 8192}) {
    for (int mode = 0; mode < 10; mode++) { if (mode == 5) continue;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k_bench, dim3(blocks), dim3(64), 0, 0, d, nbytes, mode, 1000, out, sink);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_bench, dim3(blocks), dim3(64), 0, 0, d, nbytes, mode, iters, out, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      uint64_t c = 0; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
      printf("blocks %5d  %-36s  %8.1f cycles/op (wave 0, s_memtime)   wall %.3f ms  -> %.1f Mops/s aggregate\n", blocks, names[mode], (double)c / iters, ms,
             (double)blocks * iters / ms / 1e3);
    }
  }
  return 0;
}
