/*
 * heif_plugin_abi.h — the slice of libheif's plugin ABI that libheifhip.so implements and calls.
 *
 * libheif is not a build dependency of this repository (the plugin is dlopen'ed by an unmodified
 * libheif, libheif/plugins_unix.cc:103-118), so the handful of C types that cross the boundary are
 * restated here.  They must stay layout-compatible with:
 *   heif_error, codes             libheif/api/libheif/heif_error.h:37-58, :82-245, :291-301
 *   heif_plugin_info, type enum   libheif/api/libheif/heif_library.h:155-167
 *   heif_decoder_plugin(+options) libheif/api/libheif/heif_plugin.h:70-82, :85-169
 *   heif_security_limits (head)   libheif/api/libheif/heif_security.h:37-46
 *   heif_color_profile_nclx (head) libheif/api/libheif/heif_color.h:195-204
 *   enum values                   heif_context.h:46-52, heif_image.h:55-66, :86-101, :117-119
 * tests/test_plugin_dropin.py::test_plugin_abi_matches_reference_headers checks sizes / offsets against the real headers when /root/reference exists.
 */
#ifndef HEIF_PLUGIN_ABI_H
#define HEIF_PLUGIN_ABI_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hp_error {   /* == heif_error */
  int code;
  int subcode;
  const char* message;      /* never NULL; must outlive the call */
} hp_error;

enum { HP_ERR_OK = 0, HP_ERR_INVALID_INPUT = 2, HP_ERR_UNSUPPORTED_FEATURE = 4, HP_ERR_MEMORY = 6, HP_ERR_DECODER_PLUGIN = 7 };
enum { HP_SUB_UNSPECIFIED = 0, HP_SUB_END_OF_DATA = 100, HP_SUB_INVALID_IMAGE_SIZE = 129, HP_SUB_SECURITY_LIMIT = 1000, HP_SUB_UNSUPPORTED_CODEC = 3000 };
enum { HP_COMPRESSION_HEVC = 1 };
enum { HP_COLORSPACE_YCBCR = 0, HP_COLORSPACE_MONOCHROME = 2 };
enum { HP_CHROMA_MONOCHROME = 0, HP_CHROMA_420 = 1 };
enum { HP_CHANNEL_Y = 0, HP_CHANNEL_CB = 1, HP_CHANNEL_CR = 2 };
enum { HP_PLUGIN_TYPE_ENCODER = 0, HP_PLUGIN_TYPE_DECODER = 1 };

typedef struct hp_security_limits_head {  /* leading fields of heif_security_limits */
  uint8_t version;
  uint64_t max_image_size_pixels;
} hp_security_limits_head;

typedef struct hp_nclx_head {             /* leading fields of heif_color_profile_nclx */
  uint8_t version;
  int color_primaries, transfer_characteristics, matrix_coefficients;
  uint8_t full_range_flag;
} hp_nclx_head;

typedef struct hp_image hp_image;         /* opaque heif_image */

typedef struct hp_format_description { int format; } hp_format_description;

typedef struct hp_decoder_options {       /* == heif_decoder_plugin_options */
  int format;
  int strict_decoding;
  int num_threads;
  const void* limits;                     /* heif_security_limits*, plugin_api_version >= 6 */
} hp_decoder_options;

typedef struct hp_decoder_plugin {        /* == heif_decoder_plugin, plugin_api_version 6 */
  int plugin_api_version;
  const char* (*get_plugin_name)(void);
  void (*init_plugin)(void);
  void (*deinit_plugin)(void);
  int (*does_support_format)(int format);
  hp_error (*new_decoder)(void** decoder);
  void (*free_decoder)(void* decoder);
  hp_error (*push_data)(void* decoder, const void* data, size_t size);
  hp_error (*decode_image)(void* decoder, hp_image** out_img);
  void (*set_strict_decoding)(void* decoder, int flag);
  const char* id_name;
  hp_error (*decode_next_image)(void* decoder, hp_image** out_img, const void* limits);
  uint32_t minimum_required_libheif_version;
  int (*does_support_format2)(const hp_format_description* format);
  hp_error (*new_decoder2)(void** decoder, const hp_decoder_options* options);
  hp_error (*push_data2)(void* decoder, const void* data, size_t size, uintptr_t user_data);
  hp_error (*flush_data)(void* decoder);
  hp_error (*decode_next_image2)(void* decoder, hp_image** out_img, uintptr_t* out_user_data, const void* limits);
} hp_decoder_plugin;

typedef struct hp_plugin_info {           /* == heif_plugin_info */
  int version;
  int type;
  const void* plugin;
  void* internal_handle;
} hp_plugin_info;

#ifdef __cplusplus
}
#endif
#endif
