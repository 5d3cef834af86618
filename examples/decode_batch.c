/* decode_batch.c — a throughput host over the C ABI (include/heif_hipdec.h): a stream of batches of independent HEVC-intra items, each
 * item in libheif's plugin framing ([4-byte big-endian length][NAL unit]..., parameter sets first — what push_data2 receives,
 * libheif/plugins/decoder_libde265.cc:322-368).  Every batch takes over its predecessor's arena, so the host work of batch k + 1 (header parsing,
 * staging, upload) overlaps the kernels of batch k; the decoded planes and the interleaved RGB stay in HBM until they are read.
 *
 *   cc -I include examples/decode_batch.c -L libheif_amd -lheifhip -Wl,-rpath,$PWD/libheif_amd -o decode_batch
 *   ./decode_batch item0.hevc item1.hevc ...          (files as written by tools/streamgen.py, or dumped from a HEIC's hvcC + item data)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "heif_hipdec.h"

static void* slurp(const char* path, size_t* size)
{
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(n > 0 ? (size_t)n : 1);
  if (p && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
  fclose(f);
  *size = (size_t)n;
  return p;
}

int main(int argc, char** argv)
{
  if (argc < 2) { fprintf(stderr, "usage: %s item.hevc [item.hevc ...]\n", argv[0]); return 2; }
  const int n = argc - 1;
  const void** data = (const void**)calloc((size_t)n, sizeof(void*));
  size_t* sizes = (size_t*)calloc((size_t)n, sizeof(size_t));
  for (int i = 0; i < n; i++) {
    data[i] = slurp(argv[1 + i], &sizes[i]);
    if (!data[i]) { fprintf(stderr, "cannot read %s\n", argv[1 + i]); return 2; }
    hipdec_image_info info;
    int rc = hipdec_probe(data[i], sizes[i], 0, &info);            /* host-only: headers, limits, unsupported tools */
    if (rc) { fprintf(stderr, "%s: %s\n", argv[1 + i], hipdec_last_error()); return 1; }
  }
  hipdec_set_arena_cache_bytes((size_t)64 << 30);                  /* keep two large arenas parked instead of hipFree()ing them */
  hipdec_batch* prev = NULL;
  for (int step = 0; step < 3; step++) {                           /* the same items three times: a stand-in for a stream of batches */
    hipdec_batch* b = NULL;
    int rc = hipdec_batch_create_recycling(&b, n, data, sizes, 0, prev);   /* parses + stages + uploads; overlaps prev's kernels */
    if (!rc) rc = hipdec_batch_run(b, NULL);                       /* asynchronous: CABAC, residual, reconstruction, deblock, SAO */
    if (rc) { fprintf(stderr, "batch %d: %s\n", step, hipdec_last_error()); return 1; }
    if (prev) {
      if (hipdec_batch_status(prev)) { fprintf(stderr, "batch %d failed on the device: %s\n", step - 1, hipdec_last_error()); return 1; }
      hipdec_batch_free(prev);
    }
    prev = b;
  }
  if (hipdec_batch_status(prev)) { fprintf(stderr, "%s\n", hipdec_last_error()); return 1; }
  for (int i = 0; i < n; i++) {
    hipdec_image_info info;
    hipdec_batch_info(prev, i, &info);
    const size_t es = info.bit_depth_luma > 8 ? 2 : 1;
    uint8_t* y = (uint8_t*)malloc((size_t)info.width * info.height * es);
    if (hipdec_batch_read_plane(prev, i, 0, y, (size_t)info.width * es)) { fprintf(stderr, "%s\n", hipdec_last_error()); return 1; }
    unsigned long long sum = 0;
    for (size_t k = 0; k < (size_t)info.width * info.height * es; k++) sum += y[k];
    printf("%s: %dx%d, %d bit, luma byte sum %llu\n", argv[1 + i], info.width, info.height, info.bit_depth_luma, sum);
    free(y);
  }
  hipdec_batch_free(prev);
  hipdec_shutdown();
  return 0;
}
