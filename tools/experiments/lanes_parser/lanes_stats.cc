// lanes_stats.cc — CPU-TEST-ONLY instrumentation of the lane-per-substream parser under the SIMT shim (built into libparse_emu_stats.so with
// -DHIPDEC_LANES_STATS; `make -C tests/emu libparse_emu_stats.so`): per iteration of a wave's loop, which syntax states are populated and how
// many lanes decode something.  tools/lanes_stats.py prints the summary.  NOT part of the product.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {
struct Acc {
  std::atomic<uint64_t> iterations{0}, busy_lanes{0}, waiting_lanes{0}, live_lanes{0}, distinct_states{0}, max_wave_iterations{0};
  std::atomic<uint64_t> state_populated[64], state_lanes[64], kind_lanes[8];
} g;
thread_local int t_state[64], t_kind[64], t_wait[64];
thread_local uint64_t t_iter = 0;
}

namespace hipdec { namespace planes {
void hipdec_lanes_stats(int lane, int state, int kind, int waiting)
{
  t_state[lane] = state; t_kind[lane] = kind; t_wait[lane] = waiting;
  if (lane != 63) return;
  const int S_DONE = 39;
  uint64_t seen = 0;
  int busy = 0, wait = 0, live = 0;
  for (int l = 0; l < 64; l++) {
    if (t_state[l] == S_DONE) continue;
    live++;
    if (t_wait[l]) { wait++; continue; }
    if (t_kind[l]) busy++;
    seen |= 1ull << t_state[l];
    g.state_lanes[t_state[l]]++;
    g.kind_lanes[t_kind[l] & 7]++;
  }
  for (int s = 0; s < 64; s++) if ((seen >> s) & 1) g.state_populated[s]++;
  g.iterations++; g.busy_lanes += busy; g.waiting_lanes += wait; g.live_lanes += live; g.distinct_states += __builtin_popcountll(seen);
  t_iter++;
  if (live == 0 || (live == wait && false)) {}
  uint64_t m = g.max_wave_iterations.load();
  while (t_iter > m && !g.max_wave_iterations.compare_exchange_weak(m, t_iter)) {}
}
} }

extern "C" void emu_lanes_stats_reset()
{
  g.iterations = 0; g.busy_lanes = 0; g.waiting_lanes = 0; g.live_lanes = 0; g.distinct_states = 0; g.max_wave_iterations = 0;
  for (auto& a : g.state_populated) a = 0;
  for (auto& a : g.state_lanes) a = 0;
  for (auto& a : g.kind_lanes) a = 0;
}
extern "C" void emu_lanes_stats_get(uint64_t* out /* 6 + 64 + 64 + 8 */)
{
  out[0] = g.iterations; out[1] = g.busy_lanes; out[2] = g.waiting_lanes; out[3] = g.live_lanes; out[4] = g.distinct_states; out[5] = g.max_wave_iterations;
  for (int i = 0; i < 64; i++) { out[6 + i] = g.state_populated[i]; out[70 + i] = g.state_lanes[i]; }
  for (int i = 0; i < 8; i++) out[134 + i] = g.kind_lanes[i];
}
