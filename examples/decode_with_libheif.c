/* decode_with_libheif.c — an application on an UNMODIFIED libheif that decodes HEIC files through the MI355X plugin: the plugin is a shared
 * object with the symbol `plugin_info` (libheif/plugins/decoder_libde265.cc:528-534), loaded like any other libheif plugin.
 *
 *   cc examples/decode_with_libheif.c $(pkg-config --cflags --libs libheif) -o decode_with_libheif
 *   ./decode_with_libheif /path/to/libheif_amd/libheifhip.so photo.heic
 *
 * (or no code at all: copy libheifhip.so into LIBHEIF_PLUGIN_PATH and every libheif application picks it up — INTEGRATION.md section 1)
 */
#include <libheif/heif.h>
#include <stdio.h>

int main(int argc, char** argv)
{
  if (argc < 3) { fprintf(stderr, "usage: %s libheifhip.so file.heic\n", argv[0]); return 2; }
  heif_init(NULL);
  const struct heif_plugin_info* info = NULL;
  struct heif_error e = heif_load_plugin(argv[1], &info);                      /* dlopen + dlsym("plugin_info") + registration */
  if (e.code) { fprintf(stderr, "heif_load_plugin: %s\n", e.message); return 1; }
  struct heif_context* ctx = heif_context_alloc();
  e = heif_context_read_from_file(ctx, argv[2], NULL);
  if (e.code) { fprintf(stderr, "%s\n", e.message); return 1; }
  struct heif_image_handle* h = NULL;
  e = heif_context_get_primary_image_handle(ctx, &h);
  if (e.code) { fprintf(stderr, "%s\n", e.message); return 1; }
  struct heif_decoding_options* opt = heif_decoding_options_alloc();
  opt->decoder_id = "hipdec";                                                   /* exact match on the plugin's id_name; without it the
                                                                                   highest does_support_format() priority wins */
  struct heif_image* img = NULL;
  e = heif_decode_image(h, &img, heif_colorspace_RGB, heif_chroma_interleaved_RGB, opt);
  if (e.code) { fprintf(stderr, "heif_decode_image: %s\n", e.message); return 1; }
  size_t stride = 0;
  const uint8_t* rgb = heif_image_get_plane_readonly2(img, heif_channel_interleaved, &stride);
  printf("%dx%d RGB, stride %zu, first pixel %u %u %u\n", heif_image_get_width(img, heif_channel_interleaved),
         heif_image_get_height(img, heif_channel_interleaved), stride, rgb[0], rgb[1], rgb[2]);
  heif_image_release(img);
  heif_decoding_options_free(opt);
  heif_image_handle_release(h);
  heif_context_free(ctx);
  heif_deinit();
  return 0;
}
