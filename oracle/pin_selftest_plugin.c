/*
 * pin_selftest_plugin.c — TEST INFRASTRUCTURE, and NOT a pin: a heif_decoder_plugin (id "oraclepin") around the CPU oracle, built only
 * into oracle/_ref/plugins_selftest/.  Its one job is to prove that tests/test_reference_decoder_pin.py really activates when a second
 * HEVC decoder plugin is loadable by the reference libheif: the harness must find it, select it by decoder_id, decode every golden
 * stream through heif_decode_image() and compare plane hashes.  The moment libheif's libde265 plugin (oracle/Makefile.ref `pin`) or
 * any other independent HEVC decoder plugin is present, the same harness runs against THAT and the parity status flips from
 * "unpinned" to pinned.  Function table as libheif/plugins/decoder_libde265.cc:497-517; plane hand-over as :97-171.
 */
#include <stdlib.h>
#include <string.h>
#include "libheif/heif.h"
#include "libheif/heif_plugin.h"
#include "hevc_oracle.h"

/* Samples are decoded in the order they were pushed (one access unit per push for a track, everything at once for a still); pictures are handed out
 * once the host has flushed, by increasing PicOrderCnt inside a coded video sequence - with the user_data of the sample that coded each one - so that
 * image-sequence files with B pictures come out in output order (what libheif's Track_Visual expects of a decoder, sequences/track_visual.cc:200-330). */
typedef struct { uint8_t* data; size_t size; uintptr_t user; } sample_t;
typedef struct { hevc_oracle_picture pic; uintptr_t user; int cvs; } outpic_t;
typedef struct {
  hevc_oracle_seq* q;
  sample_t* samples; int n_samples, cap_samples;
  outpic_t* outs; int n_outs, cap_outs;
  int flushed, cvs;
  char msg[256];
} dec_t;
static const struct heif_error ok = {heif_error_Ok, heif_suberror_Unspecified, "Success"};

static const char* name(void) { return "CPU oracle behind the plugin ABI (harness self-test, not a pin)"; }
static void init(void) {}
static void deinit(void) {}
static int supports(enum heif_compression_format f) { return f == heif_compression_HEVC ? 1 : 0; }   /* lowest priority: only ever chosen by id */
static int supports2(const struct heif_decoder_plugin_compressed_format_description* d) { return supports(d->format); }
static struct heif_error new2(void** out, const struct heif_decoder_plugin_options* o) { (void)o; *out = calloc(1, sizeof(dec_t)); return ok; }
static struct heif_error new1(void** out) { return new2(out, NULL); }
static void free_dec(void* p) {
  dec_t* d = (dec_t*)p;
  if (!d) return;
  for (int i = 0; i < d->n_samples; i++) free(d->samples[i].data);
  for (int i = 0; i < d->n_outs; i++) hevc_oracle_free_picture(&d->outs[i].pic);
  free(d->samples); free(d->outs);
  if (d->q) hevc_oracle_seq_free(d->q);
  free(d);
}
static void set_strict(void* p, int s) { (void)p; (void)s; }
static struct heif_error push2(void* p, const void* data, size_t size, uintptr_t user) {
  dec_t* d = (dec_t*)p;
  if (d->n_samples == d->cap_samples) { d->cap_samples = d->cap_samples * 2 + 8; d->samples = (sample_t*)realloc(d->samples, sizeof(sample_t) * (size_t)d->cap_samples); }
  sample_t* s = &d->samples[d->n_samples++];
  s->data = (uint8_t*)malloc(size ? size : 1); memcpy(s->data, data, size); s->size = size; s->user = user;
  return ok;
}
static struct heif_error push1(void* p, const void* data, size_t size) { return push2(p, data, size, 0); }
static struct heif_error flush(void* p) { ((dec_t*)p)->flushed = 1; return ok; }
static struct heif_error next2(void* p, struct heif_image** out, uintptr_t* user, const struct heif_security_limits* limits) {
  dec_t* d = (dec_t*)p; (void)limits;
  *out = NULL; if (user) *user = 0;
  if (!d->q) d->q = hevc_oracle_seq_new();
  for (int i = 0; i < d->n_samples; i++) {      /* what was pushed since the last poll, in decoding order */
    hevc_oracle_picture pic; memset(&pic, 0, sizeof(pic));
    const int rc = hevc_oracle_seq_decode(d->q, d->samples[i].data, d->samples[i].size, 0, &pic, d->msg, sizeof(d->msg));
    const uintptr_t u = d->samples[i].user;
    free(d->samples[i].data);
    if (rc) {
      for (int k = i + 1; k < d->n_samples; k++) free(d->samples[k].data);
      d->n_samples = 0;
      struct heif_error e = {heif_error_Decoder_plugin_error, heif_suberror_Unspecified, d->msg}; return e;
    }
    if (pic.poc == 0 && d->n_outs + d->cvs > 0) d->cvs++;      /* PicOrderCnt starts over: a new coded video sequence (the generator's tracks: IDR pictures only) */
    if (d->n_outs == d->cap_outs) { d->cap_outs = d->cap_outs * 2 + 8; d->outs = (outpic_t*)realloc(d->outs, sizeof(outpic_t) * (size_t)d->cap_outs); }
    d->outs[d->n_outs].pic = pic; d->outs[d->n_outs].user = u; d->outs[d->n_outs].cvs = d->cvs; d->n_outs++;
  }
  d->n_samples = 0;
  if (!d->flushed || !d->n_outs) return ok;      /* no image yet: libheif pushes the next sample, or flushes */
  int first = 0;
  for (int i = 1; i < d->n_outs; i++)
    if (d->outs[i].cvs < d->outs[first].cvs || (d->outs[i].cvs == d->outs[first].cvs && d->outs[i].pic.poc < d->outs[first].pic.poc)) first = i;
  hevc_oracle_picture pic = d->outs[first].pic;
  if (user) *user = d->outs[first].user;
  d->outs[first] = d->outs[--d->n_outs];
  struct heif_image* img = NULL;
  struct heif_error e = heif_image_create(pic.width, pic.height, pic.chroma_format_idc ? heif_colorspace_YCbCr : heif_colorspace_monochrome,
                                          (enum heif_chroma)pic.chroma_format_idc, &img);
  if (e.code) { hevc_oracle_free_picture(&pic); return e; }
  static const enum heif_channel ch[3] = {heif_channel_Y, heif_channel_Cb, heif_channel_Cr};
  for (int c = 0; c < (pic.chroma_format_idc ? 3 : 1); c++) {
    const int w = c ? pic.cwidth : pic.width, h = c ? pic.cheight : pic.height, bd = c ? pic.bit_depth_chroma : pic.bit_depth_luma;
    e = heif_image_add_plane(img, ch[c], w, h, bd);
    if (e.code) { heif_image_release(img); hevc_oracle_free_picture(&pic); return e; }
    size_t stride = 0;
    uint8_t* dst = heif_image_get_plane2(img, ch[c], &stride);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const uint16_t v = pic.plane[c][(size_t)y * w + x];
        if (bd > 8) ((uint16_t*)(dst + (size_t)y * stride))[x] = v; else dst[(size_t)y * stride + x] = (uint8_t)v;
      }
  }
  struct heif_color_profile_nclx* nclx = heif_nclx_color_profile_alloc();
  if (nclx) {
    (void)heif_nclx_color_profile_set_color_primaries(nclx, (uint16_t)pic.colour_primaries);
    (void)heif_nclx_color_profile_set_transfer_characteristics(nclx, (uint16_t)pic.transfer_characteristics);
    (void)heif_nclx_color_profile_set_matrix_coefficients(nclx, (uint16_t)pic.matrix_coeffs);
    nclx->full_range_flag = (uint8_t)pic.full_range_flag;
    (void)heif_image_set_nclx_color_profile(img, nclx);
    heif_nclx_color_profile_free(nclx);
  }
  hevc_oracle_free_picture(&pic);
  *out = img;
  return ok;
}
static struct heif_error next1(void* p, struct heif_image** out, const struct heif_security_limits* l) { return next2(p, out, NULL, l); }
static struct heif_error decode_image(void* p, struct heif_image** out) { return next2(p, out, NULL, NULL); }

static const struct heif_decoder_plugin plugin = {
  5, name, init, deinit, supports, new1, free_dec, push1, decode_image, set_strict, "oraclepin", next1,
  LIBHEIF_MAKE_VERSION(1, 21, 0), supports2, new2, push2, flush, next2
};
#if defined(__GNUC__)
__attribute__((visibility("default")))
#endif
struct heif_plugin_info plugin_info = {1, heif_plugin_type_decoder, &plugin};
