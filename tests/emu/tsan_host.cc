// CPU-TEST-ONLY: a multi-threaded C++ host on the C ABI (include/heif_hipdec.h) for the ThreadSanitizer build of the WHOLE library on the emulator
// (tools/emu_tsan_host.sh).  Python cannot host that build - libtsan would have to be preloaded into an uninstrumented interpreter - so this program is
// linked with the instrumented host sources of the product (decoder.hip: both coalescers, chains, DPB holds, the resident-plane registry; runtime.hip:
// pools; hevc_headers.hip; batch_layout.hip; plugin.hip; grid_rccl.hip) and the UNinstrumented kernels under the SIMT emulator (the lanes of a workgroup are
// ucontext coroutines, which ThreadSanitizer cannot follow; with HIPEMU_THREADS=1 every launch runs on the launching thread).
//
// What it does: the committed golden stills and golden tracks (tests/golden) are decoded once serially, then by T application threads side by side -
// stills through hipdec_decoder_decode (the still-image coalescer; every other one with tracked planes + hipdec_color_convert, the resident registry),
// tracks sample by sample through hipdec_decoder_next_picture (look-ahead chains, the chain coalescer), a share of the inputs damaged - and every
// picture of every undamaged input must come out with the hashes of the serial pass.  ThreadSanitizer watches the host code while that happens.
// usage: tsan_host <tests/golden> [threads] [rounds] [seed] [damaged_percent] [cold]
//   cold = 1: the application threads come FIRST - nothing of the library has run, so its lazy initialisation (device, pools, knobs read on first use)
//   happens under concurrency - and the serial pass they are compared with runs afterwards
#include "heif_hipdec.h"
#include <dirent.h>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#if defined(__SANITIZE_THREAD__)
// GCC 11's libtsan has no interceptor for pthread_cond_clockwait (added in GCC 12 / LLVM 13), which is what libstdc++'s condition_variable::wait_until
// calls for steady_clock deadlines (glibc >= 2.30): the sanitizer then never sees the waiter give the mutex up, reports "double lock of a mutex" and
// after that a race for everything the mutex protects.  This definition in the executable takes precedence over glibc's and goes through
// pthread_cond_timedwait, which IS intercepted.
#include <pthread.h>
#include <time.h>
extern "C" int pthread_cond_clockwait(pthread_cond_t* c, pthread_mutex_t* m, clockid_t clk, const struct timespec* abstime)
{
  struct timespec now_clk, now_rt;
  clock_gettime(clk, &now_clk);
  clock_gettime(CLOCK_REALTIME, &now_rt);
  long long ns = ((long long)abstime->tv_sec - now_clk.tv_sec) * 1000000000ll + (abstime->tv_nsec - now_clk.tv_nsec);
  if (ns < 0) ns = 0;
  ns += (long long)now_rt.tv_sec * 1000000000ll + now_rt.tv_nsec;
  struct timespec t;
  t.tv_sec = (time_t)(ns / 1000000000ll); t.tv_nsec = (long)(ns % 1000000000ll);
  return pthread_cond_timedwait(c, m, &t);
}
#endif

namespace {

typedef std::vector<uint8_t> Bytes;
typedef std::array<uint64_t, 4> PicHash;   // Y, Cb, Cr, RGB24 (0 where absent)

uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
  const uint8_t* q = (const uint8_t*)p;
  for (size_t i = 0; i < n; i++) { h ^= q[i]; h *= 1099511628211ull; }
  return h;
}

bool read_file(const std::string& path, Bytes& out)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize((size_t)n);
  const bool ok = fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

struct Input {
  std::string name;
  bool track = false;
  std::vector<Bytes> samples;        // a still: one element; a track: its access units in decoding order
  std::vector<PicHash> expected;     // pictures in output order (serial pass)
  std::vector<size_t> batch_of;      // a batch: the stills (indices into the input list) hipdec_batch_* decodes in one launch set
  int grid_rows = 0, grid_cols = 0, tile_w = 0, tile_h = 0;   // a grid photo: samples[0] is the tile every cell shows (hipdec_grid_*: shards on every emulated device)
};

// planes of the picture the decoder currently serves -> hashes; tracked: the resident-plane registry + the colour boundary on top
int hash_planes(hipdec_decoder* d, const hipdec_image_info& I, bool tracked, PicHash& out)
{
  out = PicHash{0, 0, 0, 0};
  const size_t es = I.bit_depth_luma > 8 ? 2 : 1;
  const int n = I.chroma_format_idc ? 3 : 1;
  Bytes plane[3];
  size_t stride[3] = {0, 0, 0};
  for (int c = 0; c < n; c++) {
    const int w = c ? I.chroma_width : I.width, h = c ? I.chroma_height : I.height;
    stride[c] = ((size_t)w * es + 15) & ~size_t(15);
    plane[c].assign(stride[c] * (size_t)h, 0);
    const int rc = tracked ? hipdec_decoder_read_plane_tracked(d, c, plane[c].data(), stride[c]) : hipdec_decoder_read_plane(d, c, plane[c].data(), stride[c]);
    if (rc) return rc;
    uint64_t hsh = 1469598103934665603ull;
    for (int y = 0; y < h; y++) hsh = fnv(plane[c].data() + (size_t)y * stride[c], (size_t)w * es, hsh);
    out[(size_t)c] = hsh;
  }
  if (tracked && I.chroma_format_idc == 1 && I.bit_depth_luma == 8) {   // what libheif's patched colour op does with the planes it was just handed
    hipdec_color_image img{};
    img.width = I.width; img.height = I.height; img.chroma = 1; img.bit_depth = 8;
    for (int c = 0; c < 3; c++) { img.plane[c] = plane[c].data(); img.stride[c] = stride[c]; }
    Bytes mirrored[3];
    if (!(I.width & 1) && !(I.height & 1)) {   // ... with an 'imir' in between, as the patched libheif applies it (resident planes feed the transform, its result the conversion)
      hipdec_color_image outi{};
      for (int c = 0; c < 3; c++) { mirrored[c].assign(plane[c].size(), 0); outi.plane[c] = mirrored[c].data(); outi.stride[c] = stride[c]; }
      const int args[1] = {1};
      const int rc = hipdec_image_transform(&img, HIPDEC_XF_MIRROR, args, &outi);
      if (rc == 0) { for (int c = 0; c < 3; c++) img.plane[c] = mirrored[c].data(); }
      else if (rc != HIPDEC_ERR_UNSUPPORTED) return rc;
    }
    hipdec_nclx nclx{1, I.colour_primaries, I.transfer_characteristics, I.matrix_coeffs, I.full_range_flag};
    const size_t os = ((size_t)I.width * 3 + 15) & ~size_t(15);
    Bytes rgb(os * (size_t)I.height, 0);
    const int rc = hipdec_color_convert(&img, &nclx, 10 /* heif_chroma_interleaved_RGB */, 1, 0, rgb.data(), os, 0);
    if (rc == 0) {
      uint64_t hsh = 1469598103934665603ull;
      for (int y = 0; y < I.height; y++) hsh = fnv(rgb.data() + (size_t)y * os, (size_t)I.width * 3, hsh);
      out[3] = hsh;
    } else if (rc != HIPDEC_ERR_UNSUPPORTED) return rc;
  }
  return 0;
}

// -> 0 and the pictures in output order, or the first error
int play(const Input& in, const std::vector<Bytes>& samples, bool tracked, std::vector<PicHash>& got)
{
  got.clear();
  hipdec_decoder* d = nullptr;
  int rc = hipdec_decoder_new(&d, 0, 0);
  if (rc) return rc;
  hipdec_image_info I{};
  if (!in.batch_of.empty()) {   // the throughput interface: one launch set for n stills, planes read back item by item
    hipdec_decoder_free(d);
    std::vector<const void*> ptrs;
    std::vector<size_t> sizes;
    for (const Bytes& b : samples) { ptrs.push_back(b.data()); sizes.push_back(b.size()); }
    hipdec_batch* b = nullptr;
    rc = hipdec_batch_create(&b, (int)ptrs.size(), ptrs.data(), sizes.data(), 0);
    if (rc) return rc;
    rc = hipdec_batch_run(b, nullptr);
    if (!rc) rc = hipdec_batch_status(b);
    for (int i = 0; i < (int)ptrs.size() && !rc; i++) {
      rc = hipdec_batch_info(b, i, &I);
      PicHash h{0, 0, 0, 0};
      for (int c = 0; c < 3 && !rc; c++) {
        const int w = c ? I.chroma_width : I.width, hh = c ? I.chroma_height : I.height;
        Bytes plane((size_t)w * (size_t)hh, 0);
        rc = hipdec_batch_read_plane(b, i, c, plane.data(), (size_t)w);
        h[(size_t)c] = fnv(plane.data(), plane.size());
      }
      if (!rc) got.push_back(h);
    }
    hipdec_batch_free(b);
    return rc;
  }
  if (in.grid_rows) {
    hipdec_decoder_free(d);
    const int n = in.grid_rows * in.grid_cols;
    std::vector<const void*> ptrs((size_t)n, samples[0].data());
    std::vector<size_t> sizes((size_t)n, samples[0].size());
    hipdec_grid* g = nullptr;
    const int ow = in.grid_cols * in.tile_w - 10, oh = in.grid_rows * in.tile_h - 6;   // (the canvas clips the last column / row of tiles)
    rc = hipdec_grid_create(&g, in.grid_rows, in.grid_cols, ow, oh, ptrs.data(), sizes.data(), nullptr, 0, 0);
    if (rc) return rc;
    int shards = 0;
    rc = hipdec_grid_info(g, &I, &shards);
    static std::atomic<bool> told{false};
    if (!rc && !told.exchange(true)) printf("grid photos: %d shards (HIPEMU_DEVICES emulated devices, one issue thread per device)\n", shards);
    if (!rc) rc = hipdec_grid_decode(g);
    if (!rc) rc = hipdec_grid_wait(g);
    PicHash h{0, 0, 0, 0};
    for (int c = 0; c < 3 && !rc; c++) {
      const int w = c ? (ow + 1) / 2 : ow, hh = c ? (oh + 1) / 2 : oh;
      Bytes plane((size_t)w * (size_t)hh, 0);
      rc = tracked ? hipdec_grid_read_plane_tracked(g, c, plane.data(), (size_t)w) : hipdec_grid_read_plane(g, c, plane.data(), (size_t)w);
      h[(size_t)c] = fnv(plane.data(), plane.size());
    }
    if (!rc) got.push_back(h);
    hipdec_grid_free(g);
    return rc;
  }
  if (!in.track) {
    rc = hipdec_decoder_push_data(d, samples[0].data(), samples[0].size());
    if (!rc) rc = hipdec_decoder_decode(d, &I);
    PicHash h;
    if (!rc) rc = hash_planes(d, I, tracked, h);
    if (!rc) got.push_back(h);
  } else {
    for (size_t k = 0; k <= samples.size() && !rc; k++) {
      const bool flush = k == samples.size();
      if (!flush) {
        rc = hipdec_decoder_push_data(d, samples[k].data(), samples[k].size());
        if (rc) break;
        hipdec_decoder_set_user_data(d, (uintptr_t)(1000 + k));
      }
      for (;;) {
        int have = 0;
        uintptr_t ud = 0;
        rc = hipdec_decoder_next_picture(d, flush ? 1 : 0, &I, &have, &ud);
        if (rc || !have) break;
        PicHash h;
        rc = hash_planes(d, I, false, h);
        if (rc) break;
        h[3] = (uint64_t)ud;   // (tracks: the user_data of the coding sample in the fourth slot)
        got.push_back(h);
      }
    }
  }
  hipdec_decoder_free(d);
  return rc;
}

struct Barrier {
  std::mutex mu; std::condition_variable cv; int waiting = 0, n = 0; uint64_t gen = 0;
  void wait()
  {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

}  // namespace

int main(int argc, char** argv)
{
  if (argc < 2) { fprintf(stderr, "usage: tsan_host <tests/golden> [threads] [rounds] [seed] [damaged_percent]\n"); return 2; }
  const std::string dir = argv[1];
  const int threads = argc > 2 ? atoi(argv[2]) : 8, rounds = argc > 3 ? atoi(argv[3]) : 6;
  const unsigned seed = argc > 4 ? (unsigned)atoi(argv[4]) : 1u;
  const int damaged_pct = argc > 5 ? atoi(argv[5]) : 15;
  const bool cold = argc > 6 && atoi(argv[6]) != 0;
  if (!cold && hipdec_init(0)) { fprintf(stderr, "hipdec_init: %s\n", hipdec_last_error()); return 2; }

  std::vector<Input> inputs;
  if (DIR* dp = opendir(dir.c_str())) {
    while (dirent* e = readdir(dp)) {
      const std::string n = e->d_name;
      const bool still = n.size() > 5 && n.substr(n.size() - 5) == ".hevc", track = n.size() > 6 && n.substr(n.size() - 6) == ".hevcs";
      if ((!still && !track) || n.find("reject") != std::string::npos) continue;
      Bytes blob;
      if (!read_file(dir + "/" + n, blob) || blob.size() > 100000) continue;   // (the two 1280x854 photos take the emulator a while)
      Input in;
      in.name = n; in.track = track;
      if (still) in.samples.push_back(blob);
      else for (size_t p = 0; p + 4 <= blob.size();) {
        const size_t len = ((size_t)blob[p] << 24) | ((size_t)blob[p + 1] << 16) | ((size_t)blob[p + 2] << 8) | blob[p + 3];
        if (len > blob.size() - p - 4) break;
        in.samples.emplace_back(blob.begin() + (long)p + 4, blob.begin() + (long)(p + 4 + len));
        p += 4 + len;
      }
      inputs.push_back(std::move(in));
    }
    closedir(dp);
  }
  if (inputs.empty()) { fprintf(stderr, "no inputs under %s\n", dir.c_str()); return 2; }
  {   // two grid photos out of 8-bit 4:2:0 stills (one tile in every cell): 2 x 3 and 3 x 4 tiles, sharded over the emulated devices (HIPEMU_DEVICES)
    int made = 0;
    const size_t n0 = inputs.size();
    for (size_t i = 0; i < n0 && made < 2; i++) {
      if (inputs[i].track) continue;
      hipdec_image_info P{};
      if (hipdec_probe(inputs[i].samples[0].data(), inputs[i].samples[0].size(), 0, &P) || P.chroma_format_idc != 1 || P.bit_depth_luma != 8 || (P.width & 1) || (P.height & 1) ||
          P.width != P.coded_width || P.height != P.coded_height || P.width < 64)
        continue;
      Input gin;
      gin.name = "grid of " + inputs[i].name; gin.samples = inputs[i].samples;
      gin.grid_rows = 2 + made; gin.grid_cols = 3 + made; gin.tile_w = P.width; gin.tile_h = P.height;
      inputs.push_back(std::move(gin));
      made++;
    }
  }

  auto serial_pass = [&]() -> bool {
  {   // three batches of 8-bit 4:2:0 stills for the hipdec_batch_* interface
    std::vector<size_t> pool;
    for (size_t i = 0; i < inputs.size(); i++) {
      hipdec_image_info P{};
      if (!inputs[i].track && !inputs[i].grid_rows && !hipdec_probe(inputs[i].samples[0].data(), inputs[i].samples[0].size(), 0, &P) && P.chroma_format_idc == 1 && P.bit_depth_luma == 8)
        pool.push_back(i);
    }
    for (int k = 0; k < 3 && pool.size() >= 3; k++) {
      Input bin;
      bin.name = "batch " + std::to_string(k);
      for (size_t j = 0; j < 3 + (size_t)k && j < pool.size(); j++) {
        const size_t i = pool[((size_t)k * 5 + j * 3) % pool.size()];
        bin.batch_of.push_back(i);
        bin.samples.push_back(inputs[i].samples[0]);
      }
      inputs.push_back(std::move(bin));
    }
  }
  // ---- the serial pass: what every concurrent decode must reproduce (the Python tiers hold these pictures to the oracle) ----
  hipdec_set_sequence_lookahead(32);
  size_t n_pics = 0;
  for (Input& in : inputs) {
    const int rc = play(in, in.samples, false, in.expected);
    if (rc) { fprintf(stderr, "serial pass: %s: %d %s\n", in.name.c_str(), rc, hipdec_last_error()); return false; }
    if (!in.track && !in.grid_rows && in.batch_of.empty()) {   // the tracked form adds the RGB hash
      std::vector<PicHash> t;
      if (play(in, in.samples, true, t) || t.size() != 1 || t[0][0] != in.expected[0][0]) { fprintf(stderr, "serial pass (tracked): %s\n", in.name.c_str()); return false; }
      in.expected = t;
    }
    n_pics += in.expected.size();
  }
  printf("serial pass: %zu inputs (%zu pictures)\n", inputs.size(), n_pics);

    return true;
  };
  if (!cold && !serial_pass()) return 1;
  std::atomic<long> ok{0}, failed{0}, damaged_runs{0}, damaged_errors{0};
  struct Deferred { size_t input; bool tracked; int rc; std::vector<PicHash> got; };
  std::vector<std::vector<Deferred>> deferred((size_t)threads);   // cold mode: compared once the serial pass has run
  Barrier bar;
  bar.n = threads;
  const int lookaheads[5] = {32, 0, 3, 8, 1};
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++)
    th.emplace_back([&, t] {
      std::mt19937 rng(seed * 7919u + (unsigned)t);
      for (int r = 0; r < rounds; r++) {
        if (t == 0) hipdec_set_sequence_lookahead(lookaheads[r % 5]);
        bar.wait();   // everybody asks at once: launch sets are shared
        // even rounds: everybody decodes the same kind (stills or tracks) so that the coalescers gather; odd rounds: a mix
        for (int k = 0; k < 3; k++) {
          size_t pick = rng() % inputs.size();
          if (r % 2 == 0) for (int tries = 0; tries < 64 && inputs[pick].track != (r % 4 == 0); tries++) pick = rng() % inputs.size();
          const Input& in = inputs[pick];
          if (rng() % 8 == 0) {   // an instance that is created, fed and given up without a decode (a leader may be waiting for it to join)
            hipdec_decoder* a = nullptr;
            if (!hipdec_decoder_new(&a, 0, 0)) { if (in.batch_of.empty()) (void)hipdec_decoder_push_data(a, in.samples[0].data(), in.samples[0].size()); hipdec_decoder_free(a); }
          }
          if (t == 1 && rng() % 3 == 0) {   // a host that wants memory back while the others decode: the registry and the pools are emptied under their feet
            hipdec_forget_resident_planes();
            (void)hipdec_set_arena_cache_bytes(0);
            (void)hipdec_set_arena_cache_bytes(size_t(8) << 30);
            uint64_t a0 = 0, a1 = 0, a2 = 0;
            hipdec_decoder_coalesce_stats(&a0, &a1, &a2); hipdec_decoder_chain_stats(&a0, &a1, &a2);
          }
          const bool damage = (int)(rng() % 100) < damaged_pct;
          std::vector<PicHash> got;
          if (damage) {
            std::vector<Bytes> s = in.samples;
            Bytes& victim = s[rng() % s.size()];
            for (int z = 0; z < 3 && victim.size() > 40; z++) victim[40 + rng() % (victim.size() - 40)] ^= (uint8_t)(1u << (rng() % 8));
            damaged_runs++;
            if (play(in, s, !in.track && (rng() & 1), got)) damaged_errors++;
            continue;   // (whatever came out: the point is that the others' pictures are right and nothing crashes)
          }
          const bool tracked = !in.track && (rng() & 1);
          const int rc = play(in, in.samples, tracked, got);
          if (cold) { deferred[(size_t)t].push_back(Deferred{pick, tracked, rc, got}); continue; }
          bool good = rc == 0 && got.size() == in.expected.size();
          for (size_t i = 0; good && i < got.size(); i++)
            for (int c = 0; c < 4; c++)
              if (c < 3 || in.track || (tracked && !in.grid_rows && in.batch_of.empty())) good = good && got[i][(size_t)c] == in.expected[i][(size_t)c];
          if (good) ok++;
          else { failed++; fprintf(stderr, "MISMATCH %s (round %d, thread %d): rc %d %s, %zu of %zu pictures\n", in.name.c_str(), r, t, rc, rc ? hipdec_last_error() : "", got.size(), in.expected.size()); }
        }
      }
    });
  for (auto& x : th) x.join();
  if (cold) {
    if (!serial_pass()) return 1;
    for (auto& v : deferred)
      for (const Deferred& d : v) {
        const Input& in = inputs[d.input];
        bool good = d.rc == 0 && d.got.size() == in.expected.size();
        for (size_t i = 0; good && i < d.got.size(); i++)
          for (int c = 0; c < 4; c++)
            if (c < 3 || in.track || (d.tracked && !in.grid_rows && in.batch_of.empty())) good = good && d.got[i][(size_t)c] == in.expected[i][(size_t)c];
        if (good) ok++;
        else { failed++; fprintf(stderr, "MISMATCH %s (cold): rc %d, %zu of %zu pictures\n", in.name.c_str(), d.rc, d.got.size(), in.expected.size()); }
      }
  }
  uint64_t rq = 0, ls = 0, sh = 0, ch = 0, cls = 0, csh = 0;
  hipdec_decoder_coalesce_stats(&rq, &ls, &sh);
  hipdec_decoder_chain_stats(&ch, &cls, &csh);
  printf("%d threads x %d rounds: %ld decodes identical to the serial pass, %ld MISMATCHES; %ld damaged inputs (%ld reported an error)\n", threads, rounds, ok.load(), failed.load(),
         damaged_runs.load(), damaged_errors.load());
  printf("still requests %llu in %llu launch sets (%llu shared one); chains %llu in %llu launch sets (%llu shared by several tracks)\n", (unsigned long long)rq,
         (unsigned long long)ls, (unsigned long long)sh, (unsigned long long)ch, (unsigned long long)cls, (unsigned long long)csh);
  hipdec_shutdown();
  {   // the library comes up again by itself after a shutdown (streams, pools, registry were given back): one input of every kind once more
    long again = 0;
    for (const Input& in : inputs) {
      if (again >= 4 && !in.grid_rows && in.batch_of.empty()) continue;
      std::vector<PicHash> got;
      const int rc = play(in, in.samples, false, got);
      bool good = rc == 0 && got.size() == in.expected.size();
      for (size_t i = 0; good && i < got.size(); i++)
        for (int c = 0; c < 3; c++) good = good && got[i][(size_t)c] == in.expected[i][(size_t)c];
      if (!good) { failed++; fprintf(stderr, "MISMATCH %s after shutdown + re-initialisation: rc %d %s\n", in.name.c_str(), rc, rc ? hipdec_last_error() : ""); }
      again++;
    }
    printf("after hipdec_shutdown(): %ld inputs decoded again by a library that re-initialised itself\n", again);
    hipdec_shutdown();
  }
  return failed.load() ? 1 : 0;
}
