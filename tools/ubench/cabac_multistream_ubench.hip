// cabac_multistream_ubench.hip — the experiment the one-wave-per-substream parser has not tried: SEVERAL CABAC
// substreams per wavefront.  Every active lane is an arithmetic decoder of its own — range / offset / bit count in its VGPR lane, its own
// byte stream, its own context variables in LDS ([context][lane] bytes, rangeTabLps / transIdxLps as LDS tables) — and the wave decodes one
// context-coded bin PER ACTIVE LANE per step, branch-free (both outcomes computed, selected).  `streams` active lanes per wave model the layouts:
// 64 = a substream per lane, 4 = four 16-lane groups (the other 15 lanes of a group are free for vector side jobs), 1 = today's layout.
// What it measures is the bin-decoding ceiling of such a design when the lanes run in lockstep (no syntax divergence): instructions per step are
// independent of `streams`, so bins/s scale with the active lanes; a real parser multiplies by its lane utilisation.  Compare with
// tools/ubench/cabac_ubench.hip mode 0 (the product's decode_bin, one substream per wave).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/cabac_multistream_ubench.hip -o /tmp/cabac_ms && /tmp/cabac_ms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

static const uint8_t h_range_lps[64 * 4] = {
  128,176,208,240, 128,167,197,227, 128,158,187,216, 123,150,178,205, 116,142,169,195, 111,135,160,185, 105,128,152,175, 100,122,144,166,
   95,116,137,158,  90,110,130,150,  85,104,123,142,  81, 99,117,135,  77, 94,111,128,  73, 89,105,122,  69, 85,100,116,  66, 80, 95,110,
   62, 76, 90,104,  59, 72, 86, 99,  56, 69, 81, 94,  53, 65, 77, 89,  51, 62, 73, 85,  48, 59, 69, 80,  46, 56, 66, 76,  43, 53, 63, 72,
   41, 50, 59, 69,  39, 48, 56, 65,  37, 45, 54, 62,  35, 43, 51, 59,  33, 41, 48, 56,  32, 39, 46, 53,  30, 37, 43, 50,  29, 35, 41, 48,
   27, 33, 39, 45,  26, 31, 37, 43,  24, 30, 35, 41,  23, 28, 33, 39,  22, 27, 32, 37,  21, 26, 30, 35,  20, 24, 29, 33,  19, 23, 27, 31,
   18, 22, 26, 30,  17, 21, 25, 28,  16, 20, 23, 27,  15, 19, 22, 25,  14, 18, 21, 24,  14, 17, 20, 23,  13, 16, 19, 22,  12, 15, 18, 21,
   12, 14, 17, 20,  11, 14, 16, 19,  11, 13, 15, 18,  10, 12, 15, 17,  10, 12, 14, 16,   9, 11, 13, 15,   9, 11, 12, 14,   8, 10, 12, 14,
    8,  9, 11, 13,   7,  9, 11, 12,   7,  9, 10, 12,   7,  8, 10, 11,   6,  8,  9, 11,   6,  7,  9, 10,   6,  7,  8,  9,   2,  2,  2,  2};
static const uint8_t h_next_lps[64] = {
   0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9,11,11,12, 13,13,15,15,16,16,18,18,19,19,21,21,22,22,23,24,
  24,25,26,26,27,27,28,29,29,30,30,30,31,32,32,33, 33,33,34,34,35,35,35,36,36,36,37,37,37,38,38,63};

constexpr int NCTX = 48;

// mode 0: every lane uses the same context index each step (lockstep syntax: conflict-free LDS), 1: the index depends on the lane's own bins
__global__ __launch_bounds__(64) void k_multi(const uint8_t* bs, uint32_t stream_bytes, const uint8_t* tables, int streams, int mode, int iters, uint64_t* out,
                                              uint32_t* sink)
{
  __shared__ uint8_t ctx[NCTX][64];            // context variable pStateIdx | valMps << 6 of lane l: ctx[c][l]
  __shared__ uint32_t t_lps[64];               // rangeTabLps[p][0..3] packed
  __shared__ uint8_t t_next[64];               // transIdxLps[p] | 64 where valMps flips
  const int lane = threadIdx.x;
  for (int c = 0; c < NCTX; c++) ctx[c][lane] = (uint8_t)((c * 7 + lane) & 63);
  t_lps[lane] = ((const uint32_t*)tables)[lane];
  t_next[lane] = tables[256 + lane] | (lane == 0 ? 64 : 0);
  __syncthreads();
  const int stride = 64 / streams;             // active lanes 0, stride, 2 * stride, ...
  const bool active = (lane % stride) == 0;
  const uint8_t* my = bs + (size_t)((blockIdx.x * 64u + (uint32_t)lane) % 4096u) * stream_bytes;
  uint32_t pos = 2, range = 510, value = ((uint32_t)my[0] << 8) | my[1];
  int bits = -8;
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  if (active) {
    for (int i = 0; i < iters; i++) {
      const int c = mode == 0 ? (i % NCTX) : (int)((acc * 5u + (uint32_t)i) % NCTX);
      const uint32_t st = ctx[c][lane];
      const uint32_t row = t_lps[st & 63];
      const uint32_t lps = (row >> ((range >> 3) & 24u)) & 255u;
      const uint32_t r_mps = range - lps, scaled = r_mps << 7;
      const bool is_lps = value >= scaled;
      // MPS outcome
      const uint32_t nb_m = 1u - (scaled >> 15);
      const uint32_t st_m = st + (((st & 63u) != 62u) ? 1u : 0u);
      // LPS outcome
      const uint32_t nb_l = (uint32_t)__clz((int)lps) - 23u;
      const uint32_t st_l = ((uint32_t)t_next[st & 63] & 127u) ^ (st & 64u);
      const uint32_t nb = is_lps ? nb_l : nb_m;
      range = (is_lps ? lps : r_mps) << nb;
      value = (is_lps ? value - scaled : value) << nb;
      ctx[c][lane] = (uint8_t)(is_lps ? st_l : st_m);
      acc = (acc << 1) | ((st >> 6) ^ (is_lps ? 1u : 0u));
      bits += (int)nb;
      if (bits >= 0) {                       // per-lane refill: a divergent branch, as it would be in a real parser
        value += (uint32_t)my[pos % stream_bytes] << bits;
        pos++;
        bits -= 8;
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + lane] = acc + range + pos;
}

int main(int argc, char** argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const uint32_t stream_bytes = 8192, nstreams = 4096;
  std::vector<uint8_t> h((size_t)stream_bytes * nstreams + 64);
  srand(1);
  for (auto& b : h) b = (uint8_t)(rand() >> 7);
  std::vector<uint8_t> tab(320);
  for (int p = 0; p < 64; p++) for (int q = 0; q < 4; q++) tab[p * 4 + q] = h_range_lps[p * 4 + q];
  for (int p = 0; p < 64; p++) tab[256 + p] = h_next_lps[p];
  uint8_t *d, *dt; uint64_t* out; uint32_t* sink;
  hipMalloc(&d, h.size()); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  hipMalloc(&dt, tab.size()); hipMemcpy(dt, tab.data(), tab.size(), hipMemcpyHostToDevice);
  hipMalloc(&out, 8 * 16384); hipMalloc(&sink, 4 * 64 * 16384);
  printf("context-coded bins, one per ACTIVE lane per step; %d steps per wave\n", iters);
  for (int blocks : {1, 1024, 4096, 8192}) {
    for (int mode = 0; mode < 2; mode++) {
      for (int streams : {1, 4, 16, 64}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_multi, dim3(blocks), dim3(64), 0, 0, d, stream_bytes, dt, streams, mode, 100, out, sink);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_multi, dim3(blocks), dim3(64), 0, 0, d, stream_bytes, dt, streams, mode, iters, out, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint64_t c = 0; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
        printf("waves %5d  %-28s streams/wave %2d  %7.1f cycles/step (wave 0)  wall %8.3f ms  -> %9.1f G bins/s aggregate\n", blocks,
               mode == 0 ? "ctx index uniform (lockstep)" : "ctx index data-dependent", streams, (double)c / iters, ms,
               (double)blocks * streams * iters / ms / 1e6);
      }
    }
  }
  return 0;
}
