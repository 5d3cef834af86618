// CPU-TEST-ONLY: a stateful random walk over the decoder object's C ABI (include/heif_hipdec.h) for the AddressSanitizer build of the whole library on
// the emulator (tools/emu_api_fuzz.sh).  The sweeps and stress tools drive a decoder the way libheif does - push, poll, flush, in that order.  An
// application (or a libheif that reacts to an error) may do anything: poll before pushing, push after a flush, mix the samples of two tracks, push
// random bytes between good samples, decode() in the middle of next_picture() polling, read planes at any time, drop the instance half way.  Every step is
// a legal call; the only requirement is that the library answers - with a picture or with an error code - and never crashes, hangs or touches memory it
// does not own.  usage: api_fuzz <tests/golden> <seed> <walks> [steps per walk] [threads]
#include "heif_hipdec.h"
#include <dirent.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef std::vector<uint8_t> Bytes;

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

std::vector<std::vector<Bytes>> g_sources;   // stills: one sample; tracks: their access units in decoding order
std::atomic<long> g_calls{0}, g_pictures{0}, g_errors{0};

void read_everything(hipdec_decoder* d, const hipdec_image_info& I, std::mt19937& rng, Bytes& scratch)
{
  const size_t es = I.bit_depth_luma > 8 ? 2 : 1;
  for (int c = 0; c < 3; c++) {
    const int w = c ? I.chroma_width : I.width, h = c ? I.chroma_height : I.height;
    if (w <= 0 || h <= 0 || w > 8192 || h > 8192) continue;
    const size_t stride = (size_t)w * es + (rng() % 3) * 16;
    scratch.assign(stride * (size_t)h, 0xAB);
    const int rc = (rng() & 1) ? hipdec_decoder_read_plane(d, c, scratch.data(), stride) : hipdec_decoder_read_plane_tracked(d, c, scratch.data(), stride);
    g_calls++;
    if (rc) g_errors++;
  }
  const void* p = nullptr; size_t st = 0;
  (void)hipdec_decoder_device_plane(d, (int)(rng() % 3), &p, &st);
  g_calls++;
}

void walk(unsigned seed, int steps)
{
  std::mt19937 rng(seed);
  hipdec_decoder* d = nullptr;
  Bytes scratch;
  size_t src = rng() % g_sources.size(), next_sample = 0;
  hipdec_image_info I{};
  bool have_info = false;
  for (int s = 0; s < steps; s++) {
    if (!d) {
      if (hipdec_decoder_new(&d, (int)(rng() & 1), (rng() % 16 == 0) ? 5000 : 0)) { d = nullptr; g_errors++; continue; }
      src = rng() % g_sources.size(); next_sample = 0; have_info = false;
      g_calls++;
      continue;
    }
    const unsigned op = rng() % 100;
    if (op < 34) {   // push: the next sample of the current source, mostly
      const std::vector<Bytes>& S = g_sources[src];
      Bytes data;
      const unsigned kind = rng() % 20;
      if (kind == 0) { data.resize(rng() % 300); for (auto& b : data) b = (uint8_t)rng(); }                              // random bytes
      else if (kind == 1) { const auto& o = g_sources[rng() % g_sources.size()]; data = o[rng() % o.size()]; }            // a sample of another stream
      else if (kind == 2 && !S.empty()) { data = S[rng() % S.size()]; }                                                    // an arbitrary sample of this one (repeats, jumps back)
      else if (kind == 3) { data = S[next_sample % S.size()]; if (data.size() > 48) data[40 + rng() % (data.size() - 40)] ^= (uint8_t)(1u << (rng() % 8)); next_sample++; }   // damaged
      else if (kind == 4) { data = S[next_sample % S.size()]; data.resize(data.size() - (data.size() > 8 ? rng() % 8 : 0)); next_sample++; }               // truncated framing
      else if (kind == 5) { /* nothing at all */ }
      else { data = S[next_sample % S.size()]; next_sample++; }
      if (hipdec_decoder_push_data(d, data.empty() ? (const void*)"" : data.data(), data.size())) g_errors++;
      if (rng() & 1) hipdec_decoder_set_user_data(d, (uintptr_t)(s + 1));
      g_calls++;
    } else if (op < 64) {   // poll in output order
      int have = 0; uintptr_t ud = 0;
      const int rc = hipdec_decoder_next_picture(d, (int)(rng() % 5 == 0), &I, &have, &ud);
      g_calls++;
      if (rc) g_errors++;
      else if (have) { g_pictures++; have_info = true; if (rng() % 3) read_everything(d, I, rng, scratch); }
    } else if (op < 76) {   // decode in decoding order
      const bool with_info = rng() & 1;
      const int rc = hipdec_decoder_decode(d, with_info ? &I : nullptr);
      g_calls++;
      if (rc) g_errors++;
      else { g_pictures++; have_info = with_info; }   // (the planes now served are this picture's: sizes a caller may rely on come from ITS info only)
    } else if (op < 86) {
      if (have_info) read_everything(d, I, rng, scratch);
      else { scratch.assign(size_t(4) << 20, 0); (void)hipdec_decoder_read_plane(d, (int)(rng() % 4) - 1, scratch.data(), 4096); g_calls++; }   // (room for whatever decode() may have left: the golden pictures are below 1024 rows of 2048 samples)
    } else if (op < 90) {
      hipdec_set_sequence_lookahead((int)(rng() % 40) - 2);
    } else if (op < 93) {
      hipdec_forget_resident_planes();
    } else if (op < 96) {
      hipdec_decoder_set_strict(d, (int)(rng() & 1));
    } else {
      hipdec_decoder_free(d); d = nullptr; g_calls++;
    }
  }
  if (d) hipdec_decoder_free(d);
}

// The same for the batch and the grid objects: create / run / status / read / convert / recycle / free in any order, items damaged or of mixed formats
void walk_batches(unsigned seed, int steps)
{
  std::mt19937 rng(seed);
  hipdec_batch* b = nullptr;
  hipdec_grid* g = nullptr;
  hipdec_image_info GI{};
  Bytes scratch(size_t(8) << 20, 0);
  void* dev = hipdec_malloc(size_t(8) << 20);
  auto still = [&]() -> Bytes {
    for (;;) { const auto& S = g_sources[rng() % g_sources.size()]; if (S.size() == 1) return S[0]; }
  };
  auto maybe_damage = [&](Bytes& d) { if (rng() % 6 == 0 && d.size() > 48) d[40 + rng() % (d.size() - 40)] ^= (uint8_t)(1u << (rng() % 8)); };
  for (int s = 0; s < steps; s++) {
    const unsigned op = rng() % 100;
    g_calls++;
    if (op < 12) {
      const int n = 1 + (int)(rng() % 5);
      std::vector<Bytes> items;
      const Bytes one = still();
      for (int i = 0; i < n; i++) { items.push_back(rng() % 3 ? one : still()); maybe_damage(items.back()); }   // (mixed formats are refused as a whole)
      std::vector<const void*> ptrs; std::vector<size_t> sizes;
      for (auto& it : items) { ptrs.push_back(it.data()); sizes.push_back(it.size()); }
      hipdec_batch* nb = nullptr;
      const int rc = (b && (rng() & 1)) ? hipdec_batch_create_recycling(&nb, n, ptrs.data(), sizes.data(), 0, b) : hipdec_batch_create(&nb, n, ptrs.data(), sizes.data(), (rng() % 10 == 0) ? 3000 : 0);
      if (rc) g_errors++;
      else { if (b) hipdec_batch_free(b); b = nb; }
    } else if (op < 24 && b) { if (hipdec_batch_run(b, nullptr)) g_errors++; }
    else if (op < 34 && b) { if (hipdec_batch_status(b)) g_errors++; }
    else if (op < 46 && b) {
      hipdec_image_info I{};
      const int i = (int)(rng() % 6) - 1;
      if (hipdec_batch_info(b, i, &I)) { g_errors++; continue; }
      if ((size_t)I.width * 2 > 4096 || I.height > 1024) continue;
      if (hipdec_batch_read_plane(b, i, (int)(rng() % 4) - 1, scratch.data(), 4096)) g_errors++; else g_pictures++;
    } else if (op < 54 && b) {
      hipdec_image_info I{};
      const int i = (int)(rng() % 5);
      if (hipdec_batch_info(b, i, &I) || (size_t)I.width * 8 > 8192 || I.height > 1000) continue;
      const int chroma[5] = {10, 11, 12, 14, 99};
      if (hipdec_batch_to_rgb(b, i, chroma[rng() % 5], dev, 8192, nullptr)) g_errors++;
      (void)hipdec_stream_synchronize(nullptr);
    } else if (op < 58 && b) { hipdec_batch_free(b); b = nullptr; }
    else if (op < 68) {
      if (g) { hipdec_grid_free(g); g = nullptr; }
      const int rows = 1 + (int)(rng() % 3), cols = 1 + (int)(rng() % 3);
      const Bytes one = still();
      hipdec_image_info P{};
      if (hipdec_probe(one.data(), one.size(), 0, &P)) { g_errors++; continue; }
      std::vector<Bytes> tiles((size_t)(rows * cols), one);
      for (auto& t : tiles) { if (rng() % 8 == 0) t = still(); maybe_damage(t); }
      std::vector<const void*> ptrs; std::vector<size_t> sizes;
      for (auto& t : tiles) { ptrs.push_back(t.data()); sizes.push_back(t.size()); }
      const int ow = cols * P.width - (int)(rng() % 3) * 2, oh = rows * P.height - (int)(rng() % 3) * 2;
      if (hipdec_grid_create(&g, rows, cols, ow, oh, ptrs.data(), sizes.data(), nullptr, 0, 0)) { g = nullptr; g_errors++; }
      else { int shards = 0; if (hipdec_grid_info(g, &GI, &shards)) g_errors++; }
    } else if (op < 76 && g) { if (hipdec_grid_decode(g)) g_errors++; }
    else if (op < 82 && g) { if (hipdec_grid_wait(g)) g_errors++; }
    else if (op < 92 && g) {
      if ((size_t)GI.width * 2 > 4096 || GI.height > 2048) continue;
      if (hipdec_grid_read_plane(g, (int)(rng() % 4) - 1, scratch.data(), 4096)) g_errors++; else g_pictures++;
    } else if (op < 97 && g) {
      if ((size_t)GI.width * 3 > 8192 || GI.height > 1000) continue;
      if (hipdec_grid_to_rgb(g, 10, 1 + (int)(rng() & 1), 0, scratch.data(), 8192, 0)) g_errors++;
    } else if (g) { hipdec_grid_free(g); g = nullptr; }
  }
  if (b) hipdec_batch_free(b);
  if (g) hipdec_grid_free(g);
  hipdec_free(dev);
}

}  // namespace

int main(int argc, char** argv)
{
  if (argc < 4) { fprintf(stderr, "usage: api_fuzz <tests/golden> <seed> <walks> [steps per walk] [threads]\n"); return 2; }
  const std::string dir = argv[1];
  const unsigned seed = (unsigned)atoi(argv[2]);
  const int walks = atoi(argv[3]), steps = argc > 4 ? atoi(argv[4]) : 60, threads = argc > 5 ? atoi(argv[5]) : 1;
  if (DIR* dp = opendir(dir.c_str())) {
    while (dirent* e = readdir(dp)) {
      const std::string n = e->d_name;
      const bool still = n.size() > 5 && n.substr(n.size() - 5) == ".hevc", track = n.size() > 6 && n.substr(n.size() - 6) == ".hevcs";
      Bytes blob;
      if ((!still && !track) || !read_file(dir + "/" + n, blob) || blob.size() > 100000) continue;   // (the "reject" streams of the reference's fuzz corpus are welcome here)
      std::vector<Bytes> samples;
      if (still) samples.push_back(blob);
      else for (size_t p = 0; p + 4 <= blob.size();) {
        const size_t len = ((size_t)blob[p] << 24) | ((size_t)blob[p + 1] << 16) | ((size_t)blob[p + 2] << 8) | blob[p + 3];
        if (len > blob.size() - p - 4) break;
        samples.emplace_back(blob.begin() + (long)p + 4, blob.begin() + (long)(p + 4 + len));
        p += 4 + len;
      }
      if (!samples.empty()) g_sources.push_back(std::move(samples));
    }
    closedir(dp);
  }
  if (g_sources.empty()) { fprintf(stderr, "no inputs under %s\n", dir.c_str()); return 2; }
  if (getenv("API_FUZZ_NO_CACHE")) (void)hipdec_set_arena_cache_bytes(0);   // every launch set allocates and frees: with HIPEMU_FAIL_ALLOC the failures land everywhere
  std::vector<std::thread> th;
  std::atomic<int> next{0};
  for (int t = 0; t < threads; t++)
    th.emplace_back([&] { for (int w = next++; w < walks; w = next++) { if (w % 3 == 2) walk_batches(seed * 1000003u + (unsigned)w, steps); else walk(seed * 1000003u + (unsigned)w, steps); } });
  for (auto& x : th) x.join();
  hipdec_shutdown();
  printf("seed %u: %d walks x %d steps on %d thread(s): %ld calls, %ld pictures, %ld error returns, no crash\n", seed, walks, steps, threads, g_calls.load(), g_pictures.load(),
         g_errors.load());
  return 0;
}
