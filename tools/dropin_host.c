/* dropin_host.c — T application threads, each calling heif_decode_image() on HEIC files in host memory through an UNMODIFIED libheif
 * with libheifhip.so as decoder plugin (the usage of /root/reference/tests/test-race.go:73-106).  The few libheif entry points used are
 * declared here by prototype (public C API, /root/reference/libheif/api/libheif/heif_*.h) so that the file compiles where the headers are
 * not installed; the library itself is linked at run time with dlopen.
 *
 *   gcc -O2 -pthread tools/dropin_host.c -ldl -o build/dropin_host
 *   build/dropin_host <libheif.so> <libheifhip.so> <threads> <seconds> <rgb 0|1> file0.heic file1.heic ...
 * prints: decodes seconds mpixel_s requests launch_sets failed mean_ms p95_ms   (the last two: wall time of a heif_decode_image() call inside the measured window)
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

struct heif_error { int code; int subcode; const char* message; };
typedef struct heif_error (*load_plugin_fn)(const char*, const void**);
typedef void* (*ctx_alloc_fn)(void);
typedef void (*ctx_free_fn)(void*);
typedef struct heif_error (*read_mem_fn)(void*, const void*, size_t, const void*);
typedef struct heif_error (*primary_fn)(void*, void**);
typedef struct heif_error (*decode_fn)(void*, void**, int, int, const void*);
typedef void (*release_fn)(void*);
typedef int (*dim_fn)(void*);
typedef void* (*get_track_fn)(void*, uint32_t);
typedef struct heif_error (*track_next_fn)(void*, void**, int, int, const void*);

static ctx_alloc_fn ctx_alloc; static ctx_free_fn ctx_free; static read_mem_fn read_mem; static primary_fn primary; static decode_fn decode;
static release_fn image_release, handle_release; static dim_fn handle_w, handle_h;
static get_track_fn get_track; static track_next_fn track_next; static release_fn track_release; static dim_fn has_sequence, image_w, image_h;   /* image sequences (moov / trak) */

struct file { uint8_t* data; size_t size; };
static struct file* files; static int n_files; static int n_threads; static int want_rgb;
static int stop_flag; static int tolerate; static long tolerated; static pthread_barrier_t start_barrier;
struct worker { pthread_t th; int k; long decodes; double px; int failed; };   /* decodes / px: the owner adds, main reads (relaxed atomics) */
static long ld_long(const long* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
static double ld_double(const double* p) { uint64_t u = __atomic_load_n((const uint64_t*)p, __ATOMIC_RELAXED); double d; memcpy(&d, &u, 8); return d; }
static void add_double(double* p, double v) { double d = ld_double(p) + v; uint64_t u; memcpy(&u, &d, 8); __atomic_store_n((uint64_t*)p, u, __ATOMIC_RELAXED); }

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* mode 2 ("direct"): the same threads drive the plugin's C ABI themselves (hipdec_decoder_new -> push_data -> decode -> read_plane into
 * buffers each thread allocated once): what is left when libheif's container parsing and per-image plane allocation are taken away.  The
 * files then are the plugin-framed HEVC streams. */
typedef int (*dnew_fn)(void**, int, uint64_t); typedef void (*dfree_fn)(void*); typedef int (*dpush_fn)(void*, const void*, size_t);
typedef int (*ddec_fn)(void*, void*); typedef int (*dread_fn)(void*, int, void*, size_t);
static dnew_fn d_new; static dfree_fn d_free; static dpush_fn d_push; static ddec_fn d_decode; static dread_fn d_read;
static int direct_mode;
static int decode_direct(const struct file* f, double* px, uint8_t* buf)
{
  void* d = NULL;
  int info[32];
  if (d_new(&d, 0, 0)) return 1;
  int rc = d_push(d, f->data, f->size);
  if (!rc) rc = d_decode(d, info);
  if (!rc) {
    const int w = info[0], h = info[1];     /* hipdec_image_info starts with width, height */
    rc = d_read(d, 0, buf, (size_t)w);
    if (!rc) rc = d_read(d, 1, buf + (size_t)w * h, (size_t)w / 2);
    if (!rc) rc = d_read(d, 2, buf + (size_t)w * h * 5 / 4, (size_t)w / 2);
    add_double(px, (double)w * h);
  }
  d_free(d);
  return rc;
}

static uint64_t decode_us_total, decode_calls_total;   /* time inside heif_decode_image, all threads */
#define LAT_CAP (1 << 20)
static uint32_t* lat_us; static uint32_t lat_n; static int lat_on;   /* per-call latencies of the measured window (relaxed atomics) */
static int cmp_u32(const void* a, const void* b) { const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }
static int decode_one(const struct file* f, double* px)
{
  void* ctx = ctx_alloc();
  struct heif_error e = read_mem(ctx, f->data, f->size, NULL);
  void* h = NULL; void* img = NULL;
  if (!e.code && has_sequence && has_sequence(ctx)) {
    /* an image-sequence file: every picture of its first visual track, the way an application plays it (libheif's Track_Visual pushes the samples into
     * the plugin and polls it, sequences/track_visual.cc:175-330) */
    void* track = get_track(ctx, 0);
    int n = 0;
    while (track) {
      void* im = NULL;
      e = track_next(track, &im, 99 /* heif_colorspace_undefined */, 99 /* heif_chroma_undefined */, NULL);
      if (e.code) { if (e.code == 13 /* heif_error_End_of_sequence */) e.code = 0; break; }
      n++;
      add_double(px, image_w && image_h ? (double)image_w(im) * image_h(im) : 1.0);
      image_release(im);
    }
    if (track) track_release(track);
    if (!track || (!e.code && n == 0)) { fprintf(stderr, "sequence file: no picture came out\n"); e.code = 1; }
    else if (e.code) fprintf(stderr, "track decode failed: %d.%d %s\n", e.code, e.subcode, e.message ? e.message : "");
    ctx_free(ctx);
    return e.code;
  }
  if (!e.code) e = primary(ctx, &h);
  const double t0 = now();
  if (!e.code) e = decode(h, &img, want_rgb ? 1 /* heif_colorspace_RGB */ : 0 /* YCbCr */, want_rgb ? 10 /* interleaved RGB */ : 1 /* 4:2:0 */, NULL);
  const uint64_t us = (uint64_t)((now() - t0) * 1e6);
  __atomic_fetch_add(&decode_us_total, us, __ATOMIC_RELAXED);
  __atomic_fetch_add(&decode_calls_total, 1, __ATOMIC_RELAXED);
  if (__atomic_load_n(&lat_on, __ATOMIC_RELAXED) && lat_us) {
    const uint32_t k = __atomic_fetch_add(&lat_n, 1, __ATOMIC_RELAXED);
    if (k < LAT_CAP) lat_us[k] = (uint32_t)us;
  }
  if (!e.code) add_double(px, (double)handle_w(h) * handle_h(h));
  else if (!tolerate) fprintf(stderr, "decode failed: %d.%d %s\n", e.code, e.subcode, e.message ? e.message : "");
  if (img) image_release(img);
  if (h) handle_release(h);
  ctx_free(ctx);
  return e.code;
}

static void* run(void* arg)
{
  struct worker* w = (struct worker*)arg;
  uint8_t* buf = direct_mode ? (uint8_t*)malloc((size_t)3840 * 2160 * 2) : NULL;
  if (buf) memset(buf, 1, (size_t)3840 * 2160 * 2);
  pthread_barrier_wait(&start_barrier);
  for (int i = w->k; !__atomic_load_n(&stop_flag, __ATOMIC_RELAXED); i += n_threads) {
    if (direct_mode ? decode_direct(&files[i % n_files], &w->px, buf) : decode_one(&files[i % n_files], &w->px)) {
      if (tolerate) { __atomic_fetch_add(&tolerated, 1, __ATOMIC_RELAXED); continue; }   /* DROPIN_TOLERATE=1: damaged files are part of the mix */
      w->failed = 1; break;
    }
    __atomic_store_n(&w->decodes, w->decodes + 1, __ATOMIC_RELAXED);
  }
  free(buf);
  return NULL;
}

int main(int argc, char** argv)
{
  if (argc < 7) { fprintf(stderr, "usage: %s libheif.so libheifhip.so threads seconds rgb files...\n", argv[0]); return 2; }
  void* L = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
  if (!L) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  load_plugin_fn load_plugin = (load_plugin_fn)dlsym(L, "heif_load_plugin");
  ctx_alloc = (ctx_alloc_fn)dlsym(L, "heif_context_alloc"); ctx_free = (ctx_free_fn)dlsym(L, "heif_context_free");
  read_mem = (read_mem_fn)dlsym(L, "heif_context_read_from_memory_without_copy"); primary = (primary_fn)dlsym(L, "heif_context_get_primary_image_handle");
  decode = (decode_fn)dlsym(L, "heif_decode_image"); image_release = (release_fn)dlsym(L, "heif_image_release");
  handle_release = (release_fn)dlsym(L, "heif_image_handle_release");
  handle_w = (dim_fn)dlsym(L, "heif_image_handle_get_width"); handle_h = (dim_fn)dlsym(L, "heif_image_handle_get_height");
  get_track = (get_track_fn)dlsym(L, "heif_context_get_track"); track_next = (track_next_fn)dlsym(L, "heif_track_decode_next_image");
  track_release = (release_fn)dlsym(L, "heif_track_release"); has_sequence = (dim_fn)dlsym(L, "heif_context_has_sequence");
  image_w = (dim_fn)dlsym(L, "heif_image_get_primary_width"); image_h = (dim_fn)dlsym(L, "heif_image_get_primary_height");
  if (!get_track || !track_next || !track_release) has_sequence = NULL;   /* a libheif without sequence support */
  const void* info = NULL;
  struct heif_error e = load_plugin(argv[2], &info);
  if (e.code) { fprintf(stderr, "heif_load_plugin: %s\n", e.message); return 1; }
  n_threads = atoi(argv[3]); const double seconds = atof(argv[4]); want_rgb = atoi(argv[5]);
  if (want_rgb == 2) {
    direct_mode = 1; want_rgb = 0;
    void* hp = dlopen(argv[2], RTLD_NOW | RTLD_NOLOAD);
    if (!hp) hp = dlopen(argv[2], RTLD_NOW);
    d_new = (dnew_fn)dlsym(hp, "hipdec_decoder_new"); d_free = (dfree_fn)dlsym(hp, "hipdec_decoder_free"); d_push = (dpush_fn)dlsym(hp, "hipdec_decoder_push_data");
    d_decode = (ddec_fn)dlsym(hp, "hipdec_decoder_decode"); d_read = (dread_fn)dlsym(hp, "hipdec_decoder_read_plane");
    if (!d_new || !d_read) { fprintf(stderr, "plugin C ABI not found\n"); return 1; }
  }
  n_files = argc - 6; files = (struct file*)calloc((size_t)n_files, sizeof(struct file));
  for (int i = 0; i < n_files; i++) {
    FILE* f = fopen(argv[6 + i], "rb"); if (!f) { perror(argv[6 + i]); return 1; }
    fseek(f, 0, SEEK_END); files[i].size = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    files[i].data = (uint8_t*)malloc(files[i].size);
    if (fread(files[i].data, 1, files[i].size, f) != files[i].size) return 1;
    fclose(f);
  }
  tolerate = getenv("DROPIN_TOLERATE") && atoi(getenv("DROPIN_TOLERATE"));
  double px0 = 0;   /* warm-up: HIP runtime, code objects, arena pool (DROPIN_COLD=1: none - the threads make the library's first calls, side by side) */
  const int cold = getenv("DROPIN_COLD") && atoi(getenv("DROPIN_COLD"));
  if (cold) { }
  else if (direct_mode) { uint8_t* b0 = (uint8_t*)malloc((size_t)3840 * 2160 * 2); if (decode_direct(&files[0], &px0, b0)) return 1; free(b0); }
  else if (decode_one(&files[0], &px0) && !tolerate) return 1;
  void* hip = dlopen(argv[2], RTLD_NOW | RTLD_NOLOAD);
  void (*stats)(uint64_t*, uint64_t*, uint64_t*) = hip ? (void (*)(uint64_t*, uint64_t*, uint64_t*))dlsym(hip, "hipdec_decoder_coalesce_stats") : NULL;
  uint64_t r0 = 0, s0 = 0, x0 = 0, r1 = 0, s1 = 0, x1 = 0;
  struct worker* ws = (struct worker*)calloc((size_t)n_threads, sizeof(struct worker));
  pthread_barrier_init(&start_barrier, NULL, (unsigned)n_threads + 1);
  pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstacksize(&at, 1 << 20);
  for (int k = 0; k < n_threads; k++) { ws[k].k = k; pthread_create(&ws[k].th, &at, run, &ws[k]); }
  if (stats) stats(&r0, &s0, &x0);
  pthread_barrier_wait(&start_barrier);
  /* warm-up inside the run: the first launch sets allocate what later ones reuse (device arenas, pinned staging); the counters are read at
   * the end of the warm-up and at the end of the measurement, both while all threads keep decoding (steady state) */
  const double warm = getenv("DROPIN_WARMUP_S") ? atof(getenv("DROPIN_WARMUP_S")) : 3.0;
  usleep((useconds_t)(warm * 1e6));
  long n0 = 0; double p0 = 0;
  for (int k = 0; k < n_threads; k++) { n0 += ld_long(&ws[k].decodes); p0 += ld_double(&ws[k].px); }
  if (stats) stats(&r0, &s0, &x0);
  lat_us = (uint32_t*)malloc(sizeof(uint32_t) * LAT_CAP);
  __atomic_store_n(&lat_on, 1, __ATOMIC_RELAXED);
  const double t0 = now();
  usleep((useconds_t)(seconds * 1e6));
  __atomic_store_n(&lat_on, 0, __ATOMIC_RELAXED);
  long n = 0; double px = 0; int failed = 0;
  for (int k = 0; k < n_threads; k++) { n += ld_long(&ws[k].decodes); px += ld_double(&ws[k].px); }
  const double dt = now() - t0;
  if (stats) stats(&r1, &s1, &x1);
  n -= n0; px -= p0;
  __atomic_store_n(&stop_flag, 1, __ATOMIC_RELAXED);
  for (int k = 0; k < n_threads; k++) { pthread_join(ws[k].th, NULL); failed |= ws[k].failed; }
  if (decode_calls_total) fprintf(stderr, "[dropin_host] %d threads: heif_decode_image took %.1f ms on average over %llu calls\n", n_threads,
                                  decode_us_total / 1e3 / decode_calls_total, (unsigned long long)decode_calls_total);
  if (tolerate) fprintf(stderr, "[dropin_host] %ld decodes reported an error and were tolerated\n", tolerated);
  double mean_ms = 0, p95_ms = 0;
  {
    uint32_t m = __atomic_load_n(&lat_n, __ATOMIC_RELAXED);
    if (m > LAT_CAP) m = LAT_CAP;
    if (m && lat_us) {
      qsort(lat_us, m, sizeof(uint32_t), cmp_u32);
      double sum = 0;
      for (uint32_t i = 0; i < m; i++) sum += lat_us[i];
      mean_ms = sum / m / 1e3; p95_ms = lat_us[(uint32_t)((m - 1) * 0.95)] / 1e3;
    }
  }
  printf("%ld %.3f %.1f %llu %llu %d %.2f %.2f\n", n, dt, px / dt / 1e6, (unsigned long long)(r1 - r0), (unsigned long long)(s1 - s0), failed, mean_ms, p95_ms);
  return failed;
}
