/*
 * hevc_oracle.c — CPU ORACLE (test infrastructure, NOT product code; see hevc_oracle.h).
 *
 * Spec-literal, scalar restatement of ITU-T H.265 intra-picture decoding.  Clause numbers in the
 * comments refer to ITU-T H.265 (v1/v2 numbering).  Stands in for libde265's de265_decode()
 * (reference call site libheif/plugins/decoder_libde265.cc:402); NAL framing follows
 * libheif/plugins/decoder_libde265.cc:322-368; emulation prevention as in
 * libheif/codecs/hevc_boxes.cc:572-590; conformance crop arithmetic as in
 * libheif/codecs/hevc_boxes.cc:688-716.
 *
 * No optimisation on purpose: one bit per CABAC renormalisation step, z-scan availability through
 * the MinTbAddrZs table, whole-picture deblock/SAO passes.
 */
#include "hevc_oracle.h"
#include "hevc_oracle_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <setjmp.h>
#include <stdarg.h>

#define MAXCTX CTX_COUNT
#define Clip3(lo, hi, v) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
#define Min(a, b) ((a) < (b) ? (a) : (b))
#define Max(a, b) ((a) > (b) ? (a) : (b))
#define Abs(a) ((a) < 0 ? -(a) : (a))

/* ------------------------------------------------------------------------------------------ */
/* tables                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Table 9-46 rangeTabLps */
const uint8_t hevc_cabac_range_lps[64][4] = {
  {128,176,208,240},{128,167,197,227},{128,158,187,216},{123,150,178,205},
  {116,142,169,195},{111,135,160,185},{105,128,152,175},{100,122,144,166},
  { 95,116,137,158},{ 90,110,130,150},{ 85,104,123,142},{ 81, 99,117,135},
  { 77, 94,111,128},{ 73, 89,105,122},{ 69, 85,100,116},{ 66, 80, 95,110},
  { 62, 76, 90,104},{ 59, 72, 86, 99},{ 56, 69, 81, 94},{ 53, 65, 77, 89},
  { 51, 62, 73, 85},{ 48, 59, 69, 80},{ 46, 56, 66, 76},{ 43, 53, 63, 72},
  { 41, 50, 59, 69},{ 39, 48, 56, 65},{ 37, 45, 54, 62},{ 35, 43, 51, 59},
  { 33, 41, 48, 56},{ 32, 39, 46, 53},{ 30, 37, 43, 50},{ 29, 35, 41, 48},
  { 27, 33, 39, 45},{ 26, 31, 37, 43},{ 24, 30, 35, 41},{ 23, 28, 33, 39},
  { 22, 27, 32, 37},{ 21, 26, 30, 35},{ 20, 24, 29, 33},{ 19, 23, 27, 31},
  { 18, 22, 26, 30},{ 17, 21, 25, 28},{ 16, 20, 23, 27},{ 15, 19, 22, 25},
  { 14, 18, 21, 24},{ 14, 17, 20, 23},{ 13, 16, 19, 22},{ 12, 15, 18, 21},
  { 12, 14, 17, 20},{ 11, 14, 16, 19},{ 11, 13, 15, 18},{ 10, 12, 15, 17},
  { 10, 12, 14, 16},{  9, 11, 13, 15},{  9, 11, 12, 14},{  8, 10, 12, 14},
  {  8,  9, 11, 13},{  7,  9, 11, 12},{  7,  9, 10, 12},{  7,  8, 10, 11},
  {  6,  8,  9, 11},{  6,  7,  9, 10},{  6,  7,  8,  9},{  2,  2,  2,  2}};

/* Table 9-47 transIdxLps / transIdxMps */
const uint8_t hevc_cabac_next_lps[64] = {
   0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9,11,11,12,
  13,13,15,15,16,16,18,18,19,19,21,21,22,22,23,24,
  24,25,26,26,27,27,28,29,29,30,30,30,31,32,32,33,
  33,33,34,34,35,35,35,36,36,36,37,37,37,38,38,63};
const uint8_t hevc_cabac_next_mps[64] = {
   1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,
  17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,
  33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,
  49,50,51,52,53,54,55,56,57,58,59,60,61,62,62,63};

/* initValue for initType 0 (I slices), Tables 9-5 .. 9-37 */
const uint8_t hevc_cabac_init_I[CTX_COUNT] = {
  /* sao_merge_*_flag */ 153,
  /* sao_type_idx_* */ 200,
  /* split_cu_flag */ 139, 141, 157,
  /* cu_transquant_bypass_flag */ 154,
  /* part_mode */ 184,
  /* prev_intra_luma_pred_flag */ 184,
  /* intra_chroma_pred_mode */ 63,
  /* split_transform_flag */ 153, 138, 138,
  /* cbf_luma */ 111, 141,
  /* cbf_cb, cbf_cr */ 94, 138, 182, 154,
  /* cu_qp_delta_abs */ 154, 154,
  /* transform_skip_flag luma, chroma */ 139, 139,
  /* last_sig_coeff_x_prefix */
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
  /* last_sig_coeff_y_prefix */
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
  /* coded_sub_block_flag */ 91, 171, 134, 141,
  /* sig_coeff_flag */
  111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179,
  153, 125, 107, 125, 141, 179, 153, 125, 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139,
  111, 136, 139, 111,
  /* coeff_abs_level_greater1_flag */
  140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166,
  182, 140, 227, 122, 197,
  /* coeff_abs_level_greater2_flag */ 138, 153, 136, 167, 152, 152,
  /* cbf_cb, cbf_cr ctxInc 4 */ 154,
  /* (contexts of P / B slices: unused in I slices) */ 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154};

/* initValue for initType 1 and 2 (P and B slices: a P slice takes initType 1, a B slice 2; cabac_init_flag swaps them), Tables 9-5 .. 9-37 */
const uint8_t hevc_cabac_init_P[2][CTX_COUNT] = {
 {/* sao_merge */ 153, /* sao_type_idx */ 185, /* split_cu_flag */ 107, 139, 126, /* cu_transquant_bypass_flag */ 154,
  /* part_mode bin 0 */ 154, /* prev_intra_luma_pred_flag */ 154, /* intra_chroma_pred_mode */ 152,
  /* split_transform_flag */ 124, 138, 94, /* cbf_luma */ 153, 111, /* cbf_cb, cbf_cr */ 149, 107, 167, 154,
  /* cu_qp_delta_abs */ 154, 154, /* transform_skip_flag */ 139, 139,
  /* last_sig_coeff_x_prefix */ 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
  /* last_sig_coeff_y_prefix */ 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
  /* coded_sub_block_flag */ 121, 140, 61, 154,
  /* sig_coeff_flag */
  155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136,
  153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183,
  140, 151, 183, 140,
  /* coeff_abs_level_greater1_flag */
  154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166,
  167, 154, 167, 137, 182,
  /* coeff_abs_level_greater2_flag */ 107, 167, 91, 122, 107, 167,
  /* cbf_cb, cbf_cr ctxInc 4 */ 154,
  /* cu_skip_flag */ 197, 185, 201, /* pred_mode_flag */ 149, /* part_mode bins 1, 2 (min CB), 2 (AMP) */ 139, 154, 154,
  /* merge_flag */ 110, /* merge_idx */ 122, /* ref_idx */ 153, 153, /* abs_mvd_greater0_flag */ 140, /* abs_mvd_greater1_flag */ 198,
  /* mvp_flag */ 168, /* rqt_root_cbf */ 79, /* inter_pred_idc */ 95, 79, 63, 31, 31},
 {/* sao_merge */ 153, /* sao_type_idx */ 160, /* split_cu_flag */ 107, 139, 126, /* cu_transquant_bypass_flag */ 154,
  /* part_mode bin 0 */ 154, /* prev_intra_luma_pred_flag */ 183, /* intra_chroma_pred_mode */ 152,
  /* split_transform_flag */ 224, 167, 122, /* cbf_luma */ 153, 111, /* cbf_cb, cbf_cr */ 149, 92, 167, 154,
  /* cu_qp_delta_abs */ 154, 154, /* transform_skip_flag */ 139, 139,
  /* last_sig_coeff_x_prefix */ 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
  /* last_sig_coeff_y_prefix */ 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
  /* coded_sub_block_flag */ 121, 140, 61, 154,
  /* sig_coeff_flag */
  170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136,
  153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183,
  140, 151, 183, 140,
  /* coeff_abs_level_greater1_flag */
  154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166,
  167, 154, 152, 167, 182,
  /* coeff_abs_level_greater2_flag */ 107, 167, 91, 107, 107, 167,
  /* cbf_cb, cbf_cr ctxInc 4 */ 154,
  /* cu_skip_flag */ 197, 185, 201, /* pred_mode_flag */ 134, /* part_mode bins 1, 2 (min CB), 2 (AMP) */ 139, 154, 154,
  /* merge_flag */ 154, /* merge_idx */ 137, /* ref_idx */ 153, 153, /* abs_mvd_greater0_flag */ 169, /* abs_mvd_greater1_flag */ 198,
  /* mvp_flag */ 168, /* rqt_root_cbf */ 79, /* inter_pred_idc */ 95, 79, 63, 31, 31}};

static const int8_t intraPredAngle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13,
  -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
static const int16_t invAngleTab[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315,
  -390, -482, -630, -910, -1638, -4096}; /* modes 11..25 */

static const uint8_t betaTable[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9,
  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48,
  50, 52, 54, 56, 58, 60, 62, 64};
static const uint8_t tcTable[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1,
  1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18,
  20, 22, 24};

/* Table 7-6 default 8x8 scaling lists (symmetric, so raster == diagonal-scan indexing) */
static const uint8_t defaultScalingIntra[64] = {
  16,16,16,16,17,18,21,24, 16,16,16,16,17,19,22,25, 16,16,17,18,20,22,25,29, 16,16,18,21,24,27,31,36,
  17,17,20,24,30,35,41,47, 18,19,22,27,35,44,54,65, 21,22,25,31,41,54,70,88, 24,25,29,36,47,65,88,115};
static const uint8_t defaultScalingInter[64] = {
  16,16,16,16,17,18,20,24, 16,16,16,17,18,20,24,25, 16,16,17,18,20,24,25,28, 16,17,18,20,24,25,28,33,
  17,18,20,24,25,28,33,41, 18,20,24,25,28,33,41,54, 20,24,25,28,33,41,54,71, 24,25,28,33,41,54,71,91};

int hevc_chroma_qp_420(int qPi) /* Table 8-10, ChromaArrayType == 1 */
{
  static const uint8_t t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
  if (qPi < 30) return qPi;
  if (qPi >= 44) return qPi - 6;
  return t[qPi - 30];
}

/* 6.5.3 - 6.5.5 scan orders for 2x2, 4x4, 8x8 blocks; entry = x | (y << 4) */
static uint8_t g_scan[4][3][64];
static int g_scan_init = 0;
static void init_scans(void)
{
  if (g_scan_init) return;
  for (int l = 1; l <= 3; l++) {
    int blk = 1 << l;
    /* up-right diagonal 6.5.3 */
    int i = 0, x = 0, y = 0, stop = 0;
    while (!stop) {
      while (y >= 0) {
        if (x < blk && y < blk) { g_scan[l][0][i] = (uint8_t)(x | (y << 4)); i++; }
        y--; x++;
      }
      y = x; x = 0;
      if (i >= blk * blk) stop = 1;
    }
    /* horizontal 6.5.4 */
    i = 0;
    for (y = 0; y < blk; y++) for (x = 0; x < blk; x++) g_scan[l][1][i++] = (uint8_t)(x | (y << 4));
    /* vertical 6.5.5 */
    i = 0;
    for (x = 0; x < blk; x++) for (y = 0; y < blk; y++) g_scan[l][2][i++] = (uint8_t)(x | (y << 4));
  }
  g_scan_init = 1;
}
const uint8_t* hevc_scan_order(int log2_size, int scan_idx)
{
  init_scans();
  return g_scan[log2_size][scan_idx];
}

/* 8.6.4.2 DCT basis: transMatrix[m][n] derived from the 32 distinct magnitudes */
static int16_t g_dct[32][32];
static int g_dct_init = 0;
static void init_dct(void)
{
  static const int8_t C[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
    61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
  if (g_dct_init) return;
  for (int m = 0; m < 32; m++)
    for (int n = 0; n < 32; n++) {
      int k = ((2 * n + 1) * m) % 128; /* angle in units of pi/64 */
      int v;
      if (k > 64) k = 128 - k;
      if (k <= 32) v = C[k]; else v = -C[64 - k];
      g_dct[m][n] = (int16_t)v;
    }
  g_dct_init = 1;
}
static const int8_t g_dst[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};

/* ------------------------------------------------------------------------------------------ */
/* error handling                                                                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct Dec Dec;
static void fail(Dec* d, const char* fmt, ...);

/* ------------------------------------------------------------------------------------------ */
/* bit reader over an RBSP                                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  const uint8_t* p;
  size_t nbits;
  size_t pos;
  Dec* d;
} BR;

static unsigned br_u(BR* b, int n)
{
  unsigned v = 0;
  for (int i = 0; i < n; i++) {
    if (b->pos >= b->nbits) fail(b->d, "read past end of RBSP");
    v = (v << 1) | ((b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1);
    b->pos++;
  }
  return v;
}
static unsigned br_ue(BR* b) /* 9.2 */
{
  int lz = 0;
  while (br_u(b, 1) == 0) { lz++; if (lz > 32) fail(b->d, "bad exp-golomb code"); }
  if (lz == 0) return 0;
  return (1u << lz) - 1 + br_u(b, lz);
}
static int br_se(BR* b)
{
  unsigned k = br_ue(b);
  return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1);
}
static void br_skip(BR* b, size_t n) { b->pos += n; if (b->pos > b->nbits) fail(b->d, "skip past end"); }

/* ------------------------------------------------------------------------------------------ */
/* parameter sets                                                                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  uint8_t ScalingFactor4[6][16];    /* [matrixId][y*4+x]  */
  uint8_t ScalingFactor8[6][64];
  uint8_t ScalingFactor16[6][64];   /* 8x8 base, upsampled by 2 */
  uint8_t ScalingFactor32[6][64];   /* 8x8 base, upsampled by 4; matrixId 0 and 3 */
  uint8_t dc16[6], dc32[6];
} ScalingList;

typedef struct {
  int valid;
  int chroma_format_idc, separate_colour_plane_flag;
  int pic_width, pic_height;
  int conf_win_left, conf_win_right, conf_win_top, conf_win_bottom;
  int bit_depth_luma, bit_depth_chroma;
  int log2_max_poc_lsb;
  int log2_min_cb, log2_ctb, log2_min_tb, log2_max_tb;
  int max_transform_hierarchy_depth_inter, max_transform_hierarchy_depth_intra;
  int scaling_list_enabled_flag;
  ScalingList sl;
  int amp_enabled_flag, sao_enabled_flag, pcm_enabled_flag;
  int pcm_bit_depth_luma, pcm_bit_depth_chroma, log2_min_pcm_cb, log2_max_pcm_cb;
  int pcm_loop_filter_disabled_flag;
  int num_short_term_ref_pic_sets;
  int NumDeltaPocs[65], NumNegativePics[65], NumPositivePics[65];
  int DeltaPocS0[65][17], DeltaPocS1[65][17];
  uint8_t UsedS0[65][17], UsedS1[65][17];   /* used_by_curr_pic_s0 / s1_flag */
  int long_term_ref_pics_present_flag, num_long_term_ref_pics_sps;
  int lt_ref_pic_poc_lsb_sps[32]; uint8_t used_by_curr_pic_lt_sps_flag[32];
  int sps_temporal_mvp_enabled_flag, strong_intra_smoothing_enabled_flag;
  int colour_primaries, transfer_characteristics, matrix_coeffs, video_full_range_flag;
  /* derived */
  int PicWidthInCtbsY, PicHeightInCtbsY;
} SPS;

typedef struct {
  int valid;
  int sps_id;
  int dependent_slice_segments_enabled_flag, output_flag_present_flag, num_extra_slice_header_bits;
  int sign_data_hiding_enabled_flag, cabac_init_present_flag;
  int init_qp_minus26, constrained_intra_pred_flag, transform_skip_enabled_flag;
  int cu_qp_delta_enabled_flag, diff_cu_qp_delta_depth;
  int pps_cb_qp_offset, pps_cr_qp_offset, pps_slice_chroma_qp_offsets_present_flag;
  int transquant_bypass_enabled_flag, tiles_enabled_flag, entropy_coding_sync_enabled_flag;
  int num_tile_columns, num_tile_rows, uniform_spacing_flag;
  int column_width[64], row_height[64];
  int loop_filter_across_tiles_enabled_flag, pps_loop_filter_across_slices_enabled_flag;
  int deblocking_filter_control_present_flag, deblocking_filter_override_enabled_flag;
  int pps_deblocking_filter_disabled_flag, pps_beta_offset_div2, pps_tc_offset_div2;
  int pps_scaling_list_data_present_flag;
  ScalingList sl;
  int lists_modification_present_flag, log2_parallel_merge_level;
  int num_ref_idx_l0_default_active, num_ref_idx_l1_default_active, weighted_pred_flag, weighted_bipred_flag;
  int slice_segment_header_extension_present_flag;
} PPS;

typedef struct {
  int first_slice_segment_in_pic_flag, dependent_slice_segment_flag, slice_segment_address;
  int slice_type, slice_sao_luma_flag, slice_sao_chroma_flag;
  int slice_qp_delta, slice_cb_qp_offset, slice_cr_qp_offset;
  int slice_deblocking_filter_disabled_flag, slice_beta_offset_div2, slice_tc_offset_div2;
  int slice_loop_filter_across_slices_enabled_flag;
  int num_entry_point_offsets;
  uint32_t* entry_point_offset; /* in NAL bytes (emulation prevention bytes counted) */
  int SliceAddrRs;
  int SliceQpY;
  /* P / B slices */
  int num_ref_idx_l0_active, num_ref_idx_l1_active, max_num_merge_cand, cabac_init_flag;
  int mvd_l1_zero_flag, slice_temporal_mvp, collocated_from_l0, collocated_ref_idx;
  int8_t ref_list[2][16];   /* RefPicList0 / 1: indices into Dec::dpb */
  int32_t ref_poc[2][16];
  uint8_t ref_is_lt[2][16]; /* the entry is a long-term reference picture */
  /* pred_weight_table (7.3.6.3): weighted = the explicit process applies to this slice */
  int weighted, luma_log2_wd, chroma_log2_wd;
  int16_t wp_weight[2][16][3], wp_offset[2][16][3];   /* [list][refIdx][cIdx]; offsets before the bit-depth scaling */
} SliceHdr;

/* ------------------------------------------------------------------------------------------ */
/* CABAC engine 9.3                                                                           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  const uint8_t* data; /* RBSP of the slice segment NAL (after the 2-byte NAL header) */
  size_t nbits;
  size_t pos;          /* bit position of the next bit to read */
  uint32_t range, offset;
  uint8_t ctx[MAXCTX]; /* (pStateIdx << 1) | valMps */
} Cabac;

/* ------------------------------------------------------------------------------------------ */
/* decoder state                                                                              */
/* ------------------------------------------------------------------------------------------ */
#define MAX_DPB 17
typedef struct {   /* a decoded picture (after deblocking and SAO, coded size) and the motion it was predicted with (for TMVP) */
  uint16_t* plane[3]; int poc; int valid;
  int is_lt;                     /* marked "used for long-term reference" (8.3.2) */
  uint8_t* m_pred; int16_t* mf_mv; int8_t* mf_ref; int32_t* mf_poc;
  uint8_t* mf_lt;                /* per unit and list: the reference picture was a long-term one when this picture was decoded (8.5.3.2.9 LongTermRefPic) */
} RefPic;

struct Dec {
  jmp_buf jb;
  char err[256];

  SPS sps[16];
  PPS pps[64];
  const SPS* s;
  const PPS* p;

  /* picture buffers (coded size) */
  int W, H, Wc, Hc;
  int subw, subh;   /* SubWidthC, SubHeightC (6.2): 2,2 for 4:2:0; 2,1 for 4:2:2; 1,1 for 4:4:4 (and, unused, 4:0:0) */
  uint16_t* rec[3];
  int32_t* coeff[3];
  int have_picture;
  int keep_taps;

  /* 4x4-unit maps */
  int mw, mh;
  uint8_t *m_log2_tb, *m_log2_cb, *m_ipm, *m_ipmc, *m_flags, *m_ctdepth;
  int8_t* m_qp;
  uint8_t* m_decoded; /* unit has been parsed (guards misuse) */

  /* CTB-level */
  int ctbW, ctbH, nCtb;
  int* CtbAddrRsToTs; int* CtbAddrTsToRs; int* TileId; /* TileId indexed by Ts */
  int* colBd; int* rowBd;
  int* MinTbAddrZs; int minTbW, minTbH;
  int* ctb_slice_addr;   /* SliceAddrRs per ctb (Rs), -1 = not decoded */
  int* ctb_slice_idx;    /* index into slices[] per ctb */
  SliceHdr* slices; int nslices, capslices;
  uint8_t* sao_type; uint8_t* sao_bc; int16_t* sao_off;

  /* slice decoding state */
  SliceHdr* sh;
  int sh_idx;
  Cabac c;
  uint8_t ctx_wpp[MAXCTX];
  uint8_t ctx_ds[MAXCTX];
  int ctx_ds_valid;
  int CtbAddrInRs, CtbAddrInTs;
  int IsCuQpDeltaCoded, CuQpDeltaVal;
  int qPY_PRED, last_qp_y, cur_qp_y;
  int cu_transquant_bypass_flag;
  /* entry-point verification */
  const size_t* epb_pos; int n_epb; size_t slice_data_rbsp_byte0; size_t nal_hdr_rbsp_bytes;

  uint64_t n_bins_ctx, n_bins_bypass;
  int n_substreams;

  /* ---- inter prediction (hevc_oracle_inter.c): decoded picture buffer, picture order count, the picture's motion field ---- */
  RefPic dpb[MAX_DPB]; int n_dpb;
  int first_picture;             /* no picture decoded yet in this sequence */
  int poc, prev_tid0_lsb, prev_tid0_msb;
  int st_curr_before[16], st_curr_after[16], n_st_curr_before, n_st_curr_after;   /* RefPicSetStCurrBefore / After as dpb indices */
  int lt_curr[16], n_lt_curr;    /* RefPicSetLtCurr as dpb indices */
  uint8_t* mf_lt;                /* per 4x4 unit and list: the unit's reference picture is a long-term one */
  uint8_t* m_pred;               /* per 4x4 unit: 0 MODE_INTRA, 1 MODE_INTER, 2 MODE_SKIP; NULL in a single intra picture */
  int16_t* mf_mv; int8_t* mf_ref; int32_t* mf_poc;   /* per 4x4 unit: mvL0, mvL1 (4 values), refIdxL0 / L1 (-1: list not used) and the POCs of
                                                        those reference pictures */
  int cu_pred_inter;             /* CuPredMode of the coding unit being decoded != MODE_INTRA */
  int seq_mode;                  /* hevc_oracle_seq: several pictures, P / B slices allowed */
};

static void fail(Dec* d, const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(d->err, sizeof(d->err), fmt, ap);
  va_end(ap);
  longjmp(d->jb, 1);
}

static void* xcalloc(Dec* d, size_t n, size_t sz)
{
  void* p = calloc(n ? n : 1, sz);
  if (!p) fail(d, "out of memory");
  return p;
}

/* ------------------------------------------------------------------------------------------ */
/* 7.3.1.1 / 7.4.2 NAL -> RBSP (emulation prevention removal); records positions of removed bytes */
/* ------------------------------------------------------------------------------------------ */
static uint8_t* nal_to_rbsp(Dec* d, const uint8_t* nal, size_t n, size_t* out_n, size_t** epb, int* n_epb)
{
  uint8_t* r = (uint8_t*)xcalloc(d, n + 8, 1);
  size_t* e = (size_t*)xcalloc(d, n / 3 + 1, sizeof(size_t));
  size_t j = 0; int ne = 0;
  for (size_t i = 0; i < n; i++) {
    if (i >= 2 && nal[i] == 3 && nal[i - 1] == 0 && nal[i - 2] == 0 &&
        !(ne > 0 && e[ne - 1] == i - 1) /* cannot be adjacent EPBs */) {
      /* 00 00 03 -> drop the 03.  (A preceding removed 03 is not a data zero.) */
      e[ne++] = i;
      continue;
    }
    r[j++] = nal[i];
  }
  *out_n = j; *epb = e; *n_epb = ne;
  return r;
}

/* ------------------------------------------------------------------------------------------ */
/* 7.3.3 profile_tier_level                                                                    */
/* ------------------------------------------------------------------------------------------ */
static void parse_ptl(BR* b, int maxNumSubLayersMinus1)
{
  br_skip(b, 88); /* general profile space .. reserved */
  br_skip(b, 8);  /* general_level_idc */
  int prof[8], lev[8];
  for (int i = 0; i < maxNumSubLayersMinus1; i++) { prof[i] = br_u(b, 1); lev[i] = br_u(b, 1); }
  if (maxNumSubLayersMinus1 > 0) for (int i = maxNumSubLayersMinus1; i < 8; i++) br_skip(b, 2);
  for (int i = 0; i < maxNumSubLayersMinus1; i++) {
    if (prof[i]) br_skip(b, 88);
    if (lev[i]) br_skip(b, 8);
  }
}

/* 7.3.4 scaling_list_data + 7.4.5 derivation of ScalingFactor */
static void scaling_list_default(ScalingList* sl)
{
  for (int m = 0; m < 6; m++) {
    memset(sl->ScalingFactor4[m], 16, 16);
    const uint8_t* def = m < 3 ? defaultScalingIntra : defaultScalingInter;
    memcpy(sl->ScalingFactor8[m], def, 64);
    memcpy(sl->ScalingFactor16[m], def, 64);
    memcpy(sl->ScalingFactor32[m], def, 64);
    sl->dc16[m] = 16; sl->dc32[m] = 16;
  }
}
static void parse_scaling_list_data(Dec* d, BR* b, ScalingList* sl)
{
  init_scans();
  for (int sizeId = 0; sizeId < 4; sizeId++)
    for (int matrixId = 0; matrixId < 6; matrixId += (sizeId == 3) ? 3 : 1) {
      uint8_t* dst = sizeId == 0 ? sl->ScalingFactor4[matrixId] : sizeId == 1 ? sl->ScalingFactor8[matrixId]
                   : sizeId == 2 ? sl->ScalingFactor16[matrixId] : sl->ScalingFactor32[matrixId];
      int n = sizeId == 0 ? 16 : 64;
      int pred_mode_flag = br_u(b, 1);
      if (!pred_mode_flag) {
        int delta = (int)br_ue(b);
        if (delta == 0) { /* default list */
          if (sizeId == 0) memset(dst, 16, 16);
          else memcpy(dst, matrixId < 3 ? defaultScalingIntra : defaultScalingInter, 64);
          if (sizeId == 2) sl->dc16[matrixId] = 16;
          if (sizeId == 3) sl->dc32[matrixId] = 16;
        } else {
          int ref = matrixId - delta * (sizeId == 3 ? 3 : 1);
          if (ref < 0) fail(d, "scaling_list_pred_matrix_id_delta out of range");
          const uint8_t* src = sizeId == 0 ? sl->ScalingFactor4[ref] : sizeId == 1 ? sl->ScalingFactor8[ref]
                             : sizeId == 2 ? sl->ScalingFactor16[ref] : sl->ScalingFactor32[ref];
          memcpy(dst, src, n);
          if (sizeId == 2) sl->dc16[matrixId] = sl->dc16[ref];
          if (sizeId == 3) sl->dc32[matrixId] = sl->dc32[ref];
        }
      } else {
        int nextCoef = 8;
        if (sizeId > 1) {
          int dc = br_se(b);
          nextCoef = dc + 8;
          if (sizeId == 2) sl->dc16[matrixId] = (uint8_t)nextCoef; else sl->dc32[matrixId] = (uint8_t)nextCoef;
        }
        const uint8_t* scan = g_scan[sizeId == 0 ? 2 : 3][0];
        for (int i = 0; i < n; i++) {
          int delta = br_se(b);
          nextCoef = (nextCoef + delta + 256) % 256;
          int x = scan[i] & 15, y = scan[i] >> 4;
          dst[y * (sizeId == 0 ? 4 : 8) + x] = (uint8_t)nextCoef;
        }
      }
    }
}

/* E.2.2 hrd_parameters (skipped, parsed only to stay aligned) */
static void parse_sub_layer_hrd(BR* b, int cpb_cnt, int sub_pic)
{
  for (int i = 0; i <= cpb_cnt; i++) {
    br_ue(b); br_ue(b);
    if (sub_pic) { br_ue(b); br_ue(b); }
    br_u(b, 1);
  }
}
static void parse_hrd(BR* b, int common, int maxSub)
{
  int nal = 0, vcl = 0, sub_pic = 0;
  if (common) {
    nal = br_u(b, 1); vcl = br_u(b, 1);
    if (nal || vcl) {
      sub_pic = br_u(b, 1);
      if (sub_pic) { br_u(b, 8); br_u(b, 5); br_u(b, 1); br_u(b, 5); }
      br_u(b, 4); br_u(b, 4);
      if (sub_pic) br_u(b, 4);
      br_u(b, 5); br_u(b, 5); br_u(b, 5);
    }
  }
  for (int i = 0; i <= maxSub; i++) {
    int fixed_general = br_u(b, 1), fixed_cvs = 1, low_delay = 0, cpb_cnt = 0;
    if (!fixed_general) fixed_cvs = br_u(b, 1);
    if (fixed_cvs) br_ue(b); else low_delay = br_u(b, 1);
    if (!low_delay) cpb_cnt = (int)br_ue(b);
    if (nal) parse_sub_layer_hrd(b, cpb_cnt, sub_pic);
    if (vcl) parse_sub_layer_hrd(b, cpb_cnt, sub_pic);
  }
}

/* 7.3.7 st_ref_pic_set */
static void parse_st_rps(Dec* d, BR* b, SPS* s, int idx, int num_sets)
{
  int inter = 0;
  if (idx != 0) inter = br_u(b, 1);
  if (inter) {
    int delta_idx_minus1 = 0;
    if (idx == num_sets) delta_idx_minus1 = (int)br_ue(b);
    int RefRpsIdx = idx - (delta_idx_minus1 + 1);
    if (RefRpsIdx < 0) fail(d, "bad RefRpsIdx");
    int sign = br_u(b, 1);
    int abs_m1 = (int)br_ue(b);
    int deltaRps = (1 - 2 * sign) * (abs_m1 + 1);
    int used[34], use_delta[34];
    for (int j = 0; j <= s->NumDeltaPocs[RefRpsIdx]; j++) {
      used[j] = br_u(b, 1);
      use_delta[j] = 1;
      if (!used[j]) use_delta[j] = br_u(b, 1);
    }
    int i = 0;   /* 7.4.8 (7-61), (7-62): the used_by_curr_pic flag of an entry is the one parsed for the entry it derives from */
    for (int j = s->NumPositivePics[RefRpsIdx] - 1; j >= 0; j--) {
      int dPoc = s->DeltaPocS1[RefRpsIdx][j] + deltaRps;
      if (dPoc < 0 && use_delta[s->NumNegativePics[RefRpsIdx] + j]) { s->UsedS0[idx][i] = (uint8_t)used[s->NumNegativePics[RefRpsIdx] + j]; s->DeltaPocS0[idx][i++] = dPoc; }
    }
    if (deltaRps < 0 && use_delta[s->NumDeltaPocs[RefRpsIdx]]) { s->UsedS0[idx][i] = (uint8_t)used[s->NumDeltaPocs[RefRpsIdx]]; s->DeltaPocS0[idx][i++] = deltaRps; }
    for (int j = 0; j < s->NumNegativePics[RefRpsIdx]; j++) {
      int dPoc = s->DeltaPocS0[RefRpsIdx][j] + deltaRps;
      if (dPoc < 0 && use_delta[j]) { s->UsedS0[idx][i] = (uint8_t)used[j]; s->DeltaPocS0[idx][i++] = dPoc; }
    }
    s->NumNegativePics[idx] = i;
    i = 0;
    for (int j = s->NumNegativePics[RefRpsIdx] - 1; j >= 0; j--) {
      int dPoc = s->DeltaPocS0[RefRpsIdx][j] + deltaRps;
      if (dPoc > 0 && use_delta[j]) { s->UsedS1[idx][i] = (uint8_t)used[j]; s->DeltaPocS1[idx][i++] = dPoc; }
    }
    if (deltaRps > 0 && use_delta[s->NumDeltaPocs[RefRpsIdx]]) { s->UsedS1[idx][i] = (uint8_t)used[s->NumDeltaPocs[RefRpsIdx]]; s->DeltaPocS1[idx][i++] = deltaRps; }
    for (int j = 0; j < s->NumPositivePics[RefRpsIdx]; j++) {
      int dPoc = s->DeltaPocS1[RefRpsIdx][j] + deltaRps;
      if (dPoc > 0 && use_delta[s->NumNegativePics[RefRpsIdx] + j]) { s->UsedS1[idx][i] = (uint8_t)used[s->NumNegativePics[RefRpsIdx] + j]; s->DeltaPocS1[idx][i++] = dPoc; }
    }
    s->NumPositivePics[idx] = i;
  } else {
    int nn = (int)br_ue(b), np = (int)br_ue(b);
    if (nn > 16 || np > 16) fail(d, "too many pictures in RPS");
    int poc = 0;
    for (int i = 0; i < nn; i++) { poc -= (int)br_ue(b) + 1; s->UsedS0[idx][i] = (uint8_t)br_u(b, 1); s->DeltaPocS0[idx][i] = poc; }
    poc = 0;
    for (int i = 0; i < np; i++) { poc += (int)br_ue(b) + 1; s->UsedS1[idx][i] = (uint8_t)br_u(b, 1); s->DeltaPocS1[idx][i] = poc; }
    s->NumNegativePics[idx] = nn; s->NumPositivePics[idx] = np;
  }
  s->NumDeltaPocs[idx] = s->NumNegativePics[idx] + s->NumPositivePics[idx];
  if (s->NumDeltaPocs[idx] > 16) fail(d, "RPS too large");
}

/* 7.3.2.2 seq_parameter_set_rbsp */
static void parse_sps(Dec* d, const uint8_t* rbsp, size_t n)
{
  BR b = {rbsp, n * 8, 0, d};
  SPS tmp; memset(&tmp, 0, sizeof(tmp));
  SPS* s = &tmp;
  br_u(&b, 4);
  int max_sub_layers_minus1 = br_u(&b, 3);
  br_u(&b, 1);
  parse_ptl(&b, max_sub_layers_minus1);
  int id = (int)br_ue(&b);
  if (id > 15) fail(d, "sps id out of range");
  s->chroma_format_idc = (int)br_ue(&b);
  if (s->chroma_format_idc == 3) s->separate_colour_plane_flag = br_u(&b, 1);
  s->pic_width = (int)br_ue(&b);
  s->pic_height = (int)br_ue(&b);
  if (br_u(&b, 1)) {
    s->conf_win_left = (int)br_ue(&b); s->conf_win_right = (int)br_ue(&b);
    s->conf_win_top = (int)br_ue(&b); s->conf_win_bottom = (int)br_ue(&b);
  }
  s->bit_depth_luma = (int)br_ue(&b) + 8;
  s->bit_depth_chroma = (int)br_ue(&b) + 8;
  s->log2_max_poc_lsb = (int)br_ue(&b) + 4;
  int sub_layer_ordering_info_present = br_u(&b, 1);
  for (int i = sub_layer_ordering_info_present ? 0 : max_sub_layers_minus1; i <= max_sub_layers_minus1; i++) {
    br_ue(&b); br_ue(&b); br_ue(&b);
  }
  s->log2_min_cb = (int)br_ue(&b) + 3;
  s->log2_ctb = s->log2_min_cb + (int)br_ue(&b);
  s->log2_min_tb = (int)br_ue(&b) + 2;
  s->log2_max_tb = s->log2_min_tb + (int)br_ue(&b);
  s->max_transform_hierarchy_depth_inter = (int)br_ue(&b);
  s->max_transform_hierarchy_depth_intra = (int)br_ue(&b);
  s->scaling_list_enabled_flag = br_u(&b, 1);
  scaling_list_default(&s->sl);
  if (s->scaling_list_enabled_flag) {
    if (br_u(&b, 1)) parse_scaling_list_data(d, &b, &s->sl);
  }
  s->amp_enabled_flag = br_u(&b, 1);
  s->sao_enabled_flag = br_u(&b, 1);
  s->pcm_enabled_flag = br_u(&b, 1);
  if (s->pcm_enabled_flag) {
    s->pcm_bit_depth_luma = br_u(&b, 4) + 1;
    s->pcm_bit_depth_chroma = br_u(&b, 4) + 1;
    s->log2_min_pcm_cb = (int)br_ue(&b) + 3;
    s->log2_max_pcm_cb = s->log2_min_pcm_cb + (int)br_ue(&b);
    s->pcm_loop_filter_disabled_flag = br_u(&b, 1);
  }
  s->num_short_term_ref_pic_sets = (int)br_ue(&b);
  if (s->num_short_term_ref_pic_sets > 64) fail(d, "too many short-term RPS");
  for (int i = 0; i < s->num_short_term_ref_pic_sets; i++) parse_st_rps(d, &b, s, i, s->num_short_term_ref_pic_sets);
  s->long_term_ref_pics_present_flag = br_u(&b, 1);
  if (s->long_term_ref_pics_present_flag) {
    s->num_long_term_ref_pics_sps = (int)br_ue(&b);
    if (s->num_long_term_ref_pics_sps > 32) fail(d, "num_long_term_ref_pics_sps out of range");
    for (int i = 0; i < s->num_long_term_ref_pics_sps; i++) { s->lt_ref_pic_poc_lsb_sps[i] = (int)br_u(&b, s->log2_max_poc_lsb); s->used_by_curr_pic_lt_sps_flag[i] = (uint8_t)br_u(&b, 1); }
  }
  s->sps_temporal_mvp_enabled_flag = br_u(&b, 1);
  s->strong_intra_smoothing_enabled_flag = br_u(&b, 1);
  /* Annex E defaults when the VUI / colour description is absent */
  s->colour_primaries = 2; s->transfer_characteristics = 2; s->matrix_coeffs = 2; s->video_full_range_flag = 0;
  int vui_present = br_u(&b, 1);
  if (vui_present) { /* E.2.1 */
    if (br_u(&b, 1)) { int idc = br_u(&b, 8); if (idc == 255) { br_u(&b, 16); br_u(&b, 16); } }
    if (br_u(&b, 1)) br_u(&b, 1);
    if (br_u(&b, 1)) {
      br_u(&b, 3);
      s->video_full_range_flag = br_u(&b, 1);
      if (br_u(&b, 1)) {
        s->colour_primaries = br_u(&b, 8);
        s->transfer_characteristics = br_u(&b, 8);
        s->matrix_coeffs = br_u(&b, 8);
      }
    }
    if (br_u(&b, 1)) { br_ue(&b); br_ue(&b); }
    br_u(&b, 1); br_u(&b, 1); br_u(&b, 1);
    if (br_u(&b, 1)) { br_ue(&b); br_ue(&b); br_ue(&b); br_ue(&b); }
    if (br_u(&b, 1)) {
      br_u(&b, 32); br_u(&b, 32);
      if (br_u(&b, 1)) br_ue(&b);
      if (br_u(&b, 1)) parse_hrd(&b, 1, max_sub_layers_minus1);
    }
    if (br_u(&b, 1)) { br_u(&b, 1); br_u(&b, 1); br_u(&b, 1); br_ue(&b); br_ue(&b); br_ue(&b); br_ue(&b); br_ue(&b); }
  }
  if (br_u(&b, 1)) { /* sps_extension_present_flag */
    int range_ext = br_u(&b, 1);
    int other = br_u(&b, 7);
    (void)other;
    if (range_ext) {
      /* 7.3.2.2.2: any enabled RExt coding tool is outside this oracle's scope */
      int any = 0;
      for (int i = 0; i < 9; i++) any |= br_u(&b, 1);
      if (any) fail(d, "unsupported: HEVC range-extension coding tools enabled in SPS");
    }
  }
  /* constraints */
  if (s->chroma_format_idc > 3 || (s->chroma_format_idc == 3 && s->separate_colour_plane_flag))
    fail(d, "unsupported: chroma_format_idc %d%s", s->chroma_format_idc, s->separate_colour_plane_flag ? " with separate_colour_plane_flag" : "");
  if (s->bit_depth_luma > 16 || s->bit_depth_chroma > 16) fail(d, "bit depth out of range");
  if (s->log2_ctb > 6 || s->log2_ctb < 4) fail(d, "CTB size out of range");
  if (s->log2_max_tb > 5 || s->log2_max_tb > s->log2_ctb) fail(d, "bad max TB size");
  if (s->log2_min_tb >= s->log2_min_cb) fail(d, "bad min TB size");
  if (s->pic_width <= 0 || s->pic_height <= 0 || (s->pic_width & ((1 << s->log2_min_cb) - 1)) ||
      (s->pic_height & ((1 << s->log2_min_cb) - 1)))
    fail(d, "picture size is not a multiple of the minimum coding block size");
  s->PicWidthInCtbsY = (s->pic_width + (1 << s->log2_ctb) - 1) >> s->log2_ctb;
  s->PicHeightInCtbsY = (s->pic_height + (1 << s->log2_ctb) - 1) >> s->log2_ctb;
  s->valid = 1;
  d->sps[id] = *s;
}

/* 7.3.2.3 pic_parameter_set_rbsp */
static void parse_pps(Dec* d, const uint8_t* rbsp, size_t n)
{
  BR b = {rbsp, n * 8, 0, d};
  PPS tmp; memset(&tmp, 0, sizeof(tmp));
  PPS* p = &tmp;
  int id = (int)br_ue(&b);
  if (id > 63) fail(d, "pps id out of range");
  p->sps_id = (int)br_ue(&b);
  if (p->sps_id > 15) fail(d, "sps id out of range in PPS");
  p->dependent_slice_segments_enabled_flag = br_u(&b, 1);
  p->output_flag_present_flag = br_u(&b, 1);
  p->num_extra_slice_header_bits = br_u(&b, 3);
  p->sign_data_hiding_enabled_flag = br_u(&b, 1);
  p->cabac_init_present_flag = br_u(&b, 1);
  p->num_ref_idx_l0_default_active = (int)br_ue(&b) + 1;
  p->num_ref_idx_l1_default_active = (int)br_ue(&b) + 1;
  if (p->num_ref_idx_l0_default_active > 15 || p->num_ref_idx_l1_default_active > 15) fail(d, "num_ref_idx_lX_default_active_minus1 out of range");
  p->init_qp_minus26 = br_se(&b);
  p->constrained_intra_pred_flag = br_u(&b, 1);
  p->transform_skip_enabled_flag = br_u(&b, 1);
  p->cu_qp_delta_enabled_flag = br_u(&b, 1);
  if (p->cu_qp_delta_enabled_flag) p->diff_cu_qp_delta_depth = (int)br_ue(&b);
  p->pps_cb_qp_offset = br_se(&b);
  p->pps_cr_qp_offset = br_se(&b);
  p->pps_slice_chroma_qp_offsets_present_flag = br_u(&b, 1);
  p->weighted_pred_flag = br_u(&b, 1); p->weighted_bipred_flag = br_u(&b, 1);
  p->transquant_bypass_enabled_flag = br_u(&b, 1);
  p->tiles_enabled_flag = br_u(&b, 1);
  p->entropy_coding_sync_enabled_flag = br_u(&b, 1);
  p->num_tile_columns = 1; p->num_tile_rows = 1; p->uniform_spacing_flag = 1;
  p->loop_filter_across_tiles_enabled_flag = 1;
  if (p->tiles_enabled_flag) {
    p->num_tile_columns = (int)br_ue(&b) + 1;
    p->num_tile_rows = (int)br_ue(&b) + 1;
    if (p->num_tile_columns > 20 || p->num_tile_rows > 22) fail(d, "too many tiles");
    p->uniform_spacing_flag = br_u(&b, 1);
    if (!p->uniform_spacing_flag) {
      for (int i = 0; i < p->num_tile_columns - 1; i++) p->column_width[i] = (int)br_ue(&b) + 1;
      for (int i = 0; i < p->num_tile_rows - 1; i++) p->row_height[i] = (int)br_ue(&b) + 1;
    }
    p->loop_filter_across_tiles_enabled_flag = br_u(&b, 1);
  }
  p->pps_loop_filter_across_slices_enabled_flag = br_u(&b, 1);
  p->deblocking_filter_control_present_flag = br_u(&b, 1);
  if (p->deblocking_filter_control_present_flag) {
    p->deblocking_filter_override_enabled_flag = br_u(&b, 1);
    p->pps_deblocking_filter_disabled_flag = br_u(&b, 1);
    if (!p->pps_deblocking_filter_disabled_flag) {
      p->pps_beta_offset_div2 = br_se(&b);
      p->pps_tc_offset_div2 = br_se(&b);
    }
  }
  p->pps_scaling_list_data_present_flag = br_u(&b, 1);
  scaling_list_default(&p->sl);
  if (p->pps_scaling_list_data_present_flag) parse_scaling_list_data(d, &b, &p->sl);
  p->lists_modification_present_flag = br_u(&b, 1);
  p->log2_parallel_merge_level = (int)br_ue(&b) + 2;
  p->slice_segment_header_extension_present_flag = br_u(&b, 1);
  if (br_u(&b, 1)) { /* pps_extension_present_flag */
    int range_ext = br_u(&b, 1);
    br_u(&b, 7);
    if (range_ext) {   /* 7.3.2.3.2: accepted when it enables nothing */
      if (p->transform_skip_enabled_flag && br_ue(&b) != 0) fail(d, "unsupported: transform skip blocks larger than 4x4");
      if (br_u(&b, 1)) fail(d, "unsupported: cross-component prediction");
      if (br_u(&b, 1)) fail(d, "unsupported: chroma QP offset lists");
      if (br_ue(&b) != 0 || br_ue(&b) != 0) fail(d, "unsupported: SAO offset scaling");
    }
  }
  p->valid = 1;
  d->pps[id] = *p;
}

/* ------------------------------------------------------------------------------------------ */
/* picture set-up: 6.5.1 CTB raster/tile scan conversion, 6.5.2 z-scan order array            */
/* ------------------------------------------------------------------------------------------ */
static void setup_picture(Dec* d)
{
  const SPS* s = d->s; const PPS* p = d->p;
  d->W = s->pic_width; d->H = s->pic_height;
  d->subw = (s->chroma_format_idc == 1 || s->chroma_format_idc == 2) ? 2 : 1;
  d->subh = s->chroma_format_idc == 1 ? 2 : 1;
  d->Wc = s->chroma_format_idc ? d->W / d->subw : 0;
  d->Hc = s->chroma_format_idc ? d->H / d->subh : 0;
  for (int c = 0; c < 3; c++) {
    size_t n = c ? (size_t)d->Wc * d->Hc : (size_t)d->W * d->H;
    d->rec[c] = (uint16_t*)xcalloc(d, n, sizeof(uint16_t));
    d->coeff[c] = (int32_t*)xcalloc(d, n, sizeof(int32_t));
  }
  d->mw = (d->W + 3) / 4; d->mh = (d->H + 3) / 4;
  size_t mn = (size_t)d->mw * d->mh;
  d->m_log2_tb = (uint8_t*)xcalloc(d, mn, 1); d->m_log2_cb = (uint8_t*)xcalloc(d, mn, 1);
  d->m_ipm = (uint8_t*)xcalloc(d, mn, 1); d->m_ipmc = (uint8_t*)xcalloc(d, mn, 1);
  d->m_flags = (uint8_t*)xcalloc(d, mn, 1); d->m_ctdepth = (uint8_t*)xcalloc(d, mn, 1);
  d->m_qp = (int8_t*)xcalloc(d, mn, 1); d->m_decoded = (uint8_t*)xcalloc(d, mn, 1);
  if (d->seq_mode) {
    d->m_pred = (uint8_t*)xcalloc(d, mn, 1); d->mf_mv = (int16_t*)xcalloc(d, mn * 4, sizeof(int16_t));
    d->mf_ref = (int8_t*)xcalloc(d, mn * 2, 1); d->mf_poc = (int32_t*)xcalloc(d, mn * 2, sizeof(int32_t));
    d->mf_lt = (uint8_t*)xcalloc(d, mn * 2, 1);
    memset(d->mf_ref, -1, mn * 2);
  }
  d->ctbW = s->PicWidthInCtbsY; d->ctbH = s->PicHeightInCtbsY; d->nCtb = d->ctbW * d->ctbH;
  d->CtbAddrRsToTs = (int*)xcalloc(d, d->nCtb, sizeof(int));
  d->CtbAddrTsToRs = (int*)xcalloc(d, d->nCtb, sizeof(int));
  d->TileId = (int*)xcalloc(d, d->nCtb, sizeof(int));
  d->ctb_slice_addr = (int*)xcalloc(d, d->nCtb, sizeof(int));
  d->ctb_slice_idx = (int*)xcalloc(d, d->nCtb, sizeof(int));
  for (int i = 0; i < d->nCtb; i++) d->ctb_slice_addr[i] = -1;
  d->sao_type = (uint8_t*)xcalloc(d, (size_t)d->nCtb * 3, 1);
  d->sao_bc = (uint8_t*)xcalloc(d, (size_t)d->nCtb * 3, 1);
  d->sao_off = (int16_t*)xcalloc(d, (size_t)d->nCtb * 12, sizeof(int16_t));

  /* 6.5.1 */
  int nc = p->num_tile_columns, nr = p->num_tile_rows;
  int colWidth[64], rowHeight[64];
  if (p->uniform_spacing_flag) {
    for (int i = 0; i < nc; i++) colWidth[i] = ((i + 1) * d->ctbW) / nc - (i * d->ctbW) / nc;
    for (int j = 0; j < nr; j++) rowHeight[j] = ((j + 1) * d->ctbH) / nr - (j * d->ctbH) / nr;
  } else {
    int acc = 0;
    for (int i = 0; i < nc - 1; i++) { colWidth[i] = p->column_width[i]; acc += colWidth[i]; }
    colWidth[nc - 1] = d->ctbW - acc;
    acc = 0;
    for (int j = 0; j < nr - 1; j++) { rowHeight[j] = p->row_height[j]; acc += rowHeight[j]; }
    rowHeight[nr - 1] = d->ctbH - acc;
    if (colWidth[nc - 1] <= 0 || rowHeight[nr - 1] <= 0) fail(d, "bad tile sizes");
  }
  d->colBd = (int*)xcalloc(d, nc + 1, sizeof(int));
  d->rowBd = (int*)xcalloc(d, nr + 1, sizeof(int));
  for (int i = 0; i < nc; i++) d->colBd[i + 1] = d->colBd[i] + colWidth[i];
  for (int j = 0; j < nr; j++) d->rowBd[j + 1] = d->rowBd[j] + rowHeight[j];
  for (int ctbAddrRs = 0; ctbAddrRs < d->nCtb; ctbAddrRs++) {
    int tbX = ctbAddrRs % d->ctbW, tbY = ctbAddrRs / d->ctbW, tileX = 0, tileY = 0;
    for (int i = 0; i < nc; i++) if (tbX >= d->colBd[i]) tileX = i;
    for (int j = 0; j < nr; j++) if (tbY >= d->rowBd[j]) tileY = j;
    int v = 0;
    for (int i = 0; i < tileX; i++) v += rowHeight[tileY] * colWidth[i];
    for (int j = 0; j < tileY; j++) v += d->ctbW * rowHeight[j];
    v += (tbY - d->rowBd[tileY]) * colWidth[tileX] + tbX - d->colBd[tileX];
    d->CtbAddrRsToTs[ctbAddrRs] = v;
    d->CtbAddrTsToRs[v] = ctbAddrRs;
  }
  for (int j = 0, tileIdx = 0; j < nr; j++)
    for (int i = 0; i < nc; i++, tileIdx++)
      for (int y = d->rowBd[j]; y < d->rowBd[j + 1]; y++)
        for (int x = d->colBd[i]; x < d->colBd[i + 1]; x++)
          d->TileId[d->CtbAddrRsToTs[y * d->ctbW + x]] = tileIdx;
  /* 6.5.2 */
  int lm = s->log2_min_tb, lc = s->log2_ctb;
  d->minTbW = d->ctbW << (lc - lm); d->minTbH = d->ctbH << (lc - lm);
  d->MinTbAddrZs = (int*)xcalloc(d, (size_t)d->minTbW * d->minTbH, sizeof(int));
  for (int y = 0; y < d->minTbH; y++)
    for (int x = 0; x < d->minTbW; x++) {
      int tbX = (x << lm) >> lc, tbY = (y << lm) >> lc;
      int ctbAddrRs = d->ctbW * tbY + tbX;
      int v = d->CtbAddrRsToTs[ctbAddrRs] << ((lc - lm) * 2);
      for (int i = 0; i < (lc - lm); i++) {
        int m = 1 << i;
        v += ((m & x) ? m * m : 0) + ((m & y) ? 2 * m * m : 0);
      }
      d->MinTbAddrZs[y * d->minTbW + x] = v;
    }
  d->have_picture = 1;
}

/* 6.4.1 derivation process for z-scan order block availability */
static int available_z(Dec* d, int xCurr, int yCurr, int xNbY, int yNbY)
{
  if (xNbY < 0 || yNbY < 0 || xNbY >= d->W || yNbY >= d->H) return 0;
  int lm = d->s->log2_min_tb, lc = d->s->log2_ctb;
  int zN = d->MinTbAddrZs[(yNbY >> lm) * d->minTbW + (xNbY >> lm)];
  int zC = d->MinTbAddrZs[(yCurr >> lm) * d->minTbW + (xCurr >> lm)];
  if (zN > zC) return 0;
  int ctbN = (yNbY >> lc) * d->ctbW + (xNbY >> lc);
  int ctbC = (yCurr >> lc) * d->ctbW + (xCurr >> lc);
  if (d->ctb_slice_addr[ctbN] < 0) return 0; /* not yet decoded (other slice missing) */
  if (d->ctb_slice_addr[ctbN] != d->ctb_slice_addr[ctbC]) return 0;
  if (d->TileId[d->CtbAddrRsToTs[ctbN]] != d->TileId[d->CtbAddrRsToTs[ctbC]]) return 0;
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* CABAC 9.3.2 initialisation, 9.3.4.3 decoding                                               */
/* ------------------------------------------------------------------------------------------ */
static void cabac_init_contexts(Dec* d)
{
  int qp = Clip3(0, 51, d->sh->SliceQpY);
  /* 9.3.2.2: initType 0 for I, 1 for P and 2 for B slices; cabac_init_flag swaps the latter two */
  const uint8_t* tab = d->sh->slice_type == 2 ? hevc_cabac_init_I
                     : hevc_cabac_init_P[(d->sh->slice_type == 1) == (d->sh->cabac_init_flag != 0) ? 1 : 0];
  for (int i = 0; i < MAXCTX; i++) {
    int initValue = tab[i];
    int slopeIdx = initValue >> 4, offsetIdx = initValue & 15;
    int m = slopeIdx * 5 - 45, n = (offsetIdx << 3) - 16;
    int preCtxState = Clip3(1, 126, ((m * qp) >> 4) + n);
    int valMps = preCtxState <= 63 ? 0 : 1;
    int pStateIdx = valMps ? (preCtxState - 64) : (63 - preCtxState);
    d->c.ctx[i] = (uint8_t)((pStateIdx << 1) | valMps);
  }
}
static unsigned cabac_read_bit(Dec* d)
{
  Cabac* c = &d->c;
  /* 9.3.2.5 allows reading past the end only as zero bits for conformant streams; we allow up to
     two bytes of slack (cabac look-ahead never needs more) */
  if (c->pos >= c->nbits) {
    if (c->pos >= c->nbits + 16) fail(d, "CABAC read past end of slice data");
    c->pos++;
    return 0;
  }
  unsigned v = (c->data[c->pos >> 3] >> (7 - (c->pos & 7))) & 1;
  c->pos++;
  return v;
}
static void cabac_init_engine(Dec* d) /* 9.3.2.5 */
{
  Cabac* c = &d->c;
  if (c->pos & 7) fail(d, "CABAC initialisation at unaligned position");
  c->range = 510;
  c->offset = 0;
  for (int i = 0; i < 9; i++) c->offset = (c->offset << 1) | cabac_read_bit(d);
  if (c->offset == 510 || c->offset == 511) fail(d, "illegal CABAC ivlOffset at initialisation");
  d->n_substreams++;
}
static int decode_decision(Dec* d, int ctxIdx) /* 9.3.4.3.2 */
{
  Cabac* c = &d->c;
  int st = c->ctx[ctxIdx];
  int pStateIdx = st >> 1, valMps = st & 1, binVal;
  int qRangeIdx = (c->range >> 6) & 3;
  unsigned ivlLpsRange = hevc_cabac_range_lps[pStateIdx][qRangeIdx];
  c->range -= ivlLpsRange;
  if (c->offset >= c->range) {
    binVal = !valMps;
    c->offset -= c->range;
    c->range = ivlLpsRange;
    if (pStateIdx == 0) valMps = 1 - valMps;
    pStateIdx = hevc_cabac_next_lps[pStateIdx];
  } else {
    binVal = valMps;
    pStateIdx = hevc_cabac_next_mps[pStateIdx];
  }
  c->ctx[ctxIdx] = (uint8_t)((pStateIdx << 1) | valMps);
  while (c->range < 256) { /* 9.3.4.3.3 */
    c->range <<= 1;
    c->offset = (c->offset << 1) | cabac_read_bit(d);
  }
  d->n_bins_ctx++;
  return binVal;
}
static int decode_bypass(Dec* d) /* 9.3.4.3.4 */
{
  Cabac* c = &d->c;
  c->offset = (c->offset << 1) | cabac_read_bit(d);
  d->n_bins_bypass++;
  if (c->offset >= c->range) { c->offset -= c->range; return 1; }
  return 0;
}
static int decode_terminate(Dec* d) /* 9.3.4.3.5 */
{
  Cabac* c = &d->c;
  c->range -= 2;
  if (c->offset >= c->range) return 1;
  while (c->range < 256) {
    c->range <<= 1;
    c->offset = (c->offset << 1) | cabac_read_bit(d);
  }
  return 0;
}
/* After a terminating bin equal to 1 the 9-bit window of the arithmetic decoder ends on the
   stop/alignment '1' bit; what follows up to the next byte boundary must be zero bits. */
static void cabac_finish_and_align(Dec* d)
{
  Cabac* c = &d->c;
  size_t last = c->pos - 1;
  if (last < c->nbits) {
    unsigned bit = (c->data[last >> 3] >> (7 - (last & 7))) & 1;
    if (!bit) fail(d, "CABAC termination: stop bit is not 1 (substream desynchronised)");
    unsigned rest = c->data[last >> 3] & ((1u << (7 - (last & 7))) - 1);
    if (rest) fail(d, "CABAC termination: alignment bits are not zero (substream desynchronised)");
  } else {
    fail(d, "CABAC termination beyond the end of data");
  }
  c->pos = (c->pos + 7) & ~(size_t)7;
}
static int decode_bypass_bits(Dec* d, int n)
{
  int v = 0;
  while (n--) v = (v << 1) | decode_bypass(d);
  return v;
}

/* ------------------------------------------------------------------------------------------ */
/* 8.4.4.2 intra sample prediction                                                            */
/* ------------------------------------------------------------------------------------------ */
void hevc_intra_predict(uint16_t* dst, int dst_stride, int nTbS, int cIdx, int mode,
                        const uint16_t* ref_left, const uint16_t* ref_top, int bit_depth,
                        int strong_intra_smoothing, int chroma_format_idc)
{
  /* p[-1][y] = L[y+1], p[x][-1] = T[x+1], p[-1][-1] = L[0] = T[0] */
  uint16_t Lb[2 * 32 + 1], Tb[2 * 32 + 1];
  const uint16_t *L = ref_left, *T = ref_top;
  int n2 = 2 * nTbS;
  int maxv = (1 << bit_depth) - 1;
  /* 8.4.4.2.3 filtering process of neighbouring samples */
  if (cIdx == 0 || chroma_format_idc == 3) {
    int filterFlag = 0;
    if (mode == 1 || nTbS == 4) filterFlag = 0;
    else {
      int minDistVerHor = Min(Abs(mode - 26), Abs(mode - 10));
      int thres = nTbS == 8 ? 7 : nTbS == 16 ? 1 : 0;
      filterFlag = minDistVerHor > thres;
    }
    if (filterFlag) {
      int biIntFlag = strong_intra_smoothing && cIdx == 0 && nTbS == 32 &&
                      Abs(L[0] + T[n2] - 2 * T[nTbS]) < (1 << (bit_depth - 5)) &&
                      Abs(L[0] + L[n2] - 2 * L[nTbS]) < (1 << (bit_depth - 5));
      if (biIntFlag) {
        Lb[0] = Tb[0] = L[0];
        for (int y = 0; y <= 62; y++) Lb[y + 1] = (uint16_t)(((63 - y) * L[0] + (y + 1) * L[64] + 32) >> 6);
        Lb[64] = L[64];
        for (int x = 0; x <= 62; x++) Tb[x + 1] = (uint16_t)(((63 - x) * T[0] + (x + 1) * T[64] + 32) >> 6);
        Tb[64] = T[64];
      } else {
        Lb[0] = Tb[0] = (uint16_t)((L[1] + 2 * L[0] + T[1] + 2) >> 2);
        for (int y = 0; y <= n2 - 2; y++) Lb[y + 1] = (uint16_t)((L[y + 2] + 2 * L[y + 1] + L[y] + 2) >> 2);
        Lb[n2] = L[n2];
        for (int x = 0; x <= n2 - 2; x++) Tb[x + 1] = (uint16_t)((T[x] + 2 * T[x + 1] + T[x + 2] + 2) >> 2);
        Tb[n2] = T[n2];
      }
      L = Lb; T = Tb;
    }
  }
  if (mode == 0) { /* 8.4.4.2.4 planar */
    int sh = 0; while ((1 << sh) < nTbS) sh++;
    for (int y = 0; y < nTbS; y++)
      for (int x = 0; x < nTbS; x++)
        dst[y * dst_stride + x] = (uint16_t)(((nTbS - 1 - x) * L[y + 1] + (x + 1) * T[nTbS + 1] +
                                              (nTbS - 1 - y) * T[x + 1] + (y + 1) * L[nTbS + 1] + nTbS) >> (sh + 1));
  } else if (mode == 1) { /* 8.4.4.2.5 DC */
    int k = 0; while ((1 << k) < nTbS) k++;
    int sum = nTbS;
    for (int i = 0; i < nTbS; i++) sum += T[i + 1] + L[i + 1];
    int dcVal = sum >> (k + 1);
    for (int y = 0; y < nTbS; y++) for (int x = 0; x < nTbS; x++) dst[y * dst_stride + x] = (uint16_t)dcVal;
    if (cIdx == 0 && nTbS < 32) {
      dst[0] = (uint16_t)((L[1] + 2 * dcVal + T[1] + 2) >> 2);
      for (int x = 1; x < nTbS; x++) dst[x] = (uint16_t)((T[x + 1] + 3 * dcVal + 2) >> 2);
      for (int y = 1; y < nTbS; y++) dst[y * dst_stride] = (uint16_t)((L[y + 1] + 3 * dcVal + 2) >> 2);
    }
  } else { /* 8.4.4.2.6 angular */
    int angle = intraPredAngle[mode];
    int refbuf[3 * 32 + 2];
    int* ref = refbuf + 32; /* ref[-32 .. 64] */
    if (mode >= 18) {
      for (int x = 0; x <= nTbS; x++) ref[x] = T[x]; /* p[-1+x][-1] */
      if (angle < 0) {
        int last = (nTbS * angle) >> 5;
        if (last < -1) {
          int invAngle = invAngleTab[mode - 11];
          for (int x = last; x <= -1; x++) ref[x] = L[((x * invAngle + 128) >> 8)]; /* p[-1][-1+..] */
        }
      } else {
        for (int x = nTbS + 1; x <= n2; x++) ref[x] = T[x];
      }
      for (int y = 0; y < nTbS; y++) {
        int iIdx = ((y + 1) * angle) >> 5, iFact = ((y + 1) * angle) & 31;
        for (int x = 0; x < nTbS; x++) {
          int v;
          if (iFact) v = ((32 - iFact) * ref[x + iIdx + 1] + iFact * ref[x + iIdx + 2] + 16) >> 5;
          else v = ref[x + iIdx + 1];
          dst[y * dst_stride + x] = (uint16_t)v;
        }
      }
      if (mode == 26 && cIdx == 0 && nTbS < 32)
        for (int y = 0; y < nTbS; y++) {
          int v = T[1] + ((L[y + 1] - L[0]) >> 1);
          dst[y * dst_stride] = (uint16_t)Clip3(0, maxv, v);
        }
    } else {
      for (int x = 0; x <= nTbS; x++) ref[x] = L[x]; /* p[-1][-1+x] */
      if (angle < 0) {
        int last = (nTbS * angle) >> 5;
        if (last < -1) {
          int invAngle = invAngleTab[mode - 11];
          for (int x = last; x <= -1; x++) ref[x] = T[((x * invAngle + 128) >> 8)]; /* p[-1+..][-1] */
        }
      } else {
        for (int x = nTbS + 1; x <= n2; x++) ref[x] = L[x];
      }
      for (int x = 0; x < nTbS; x++) {
        int iIdx = ((x + 1) * angle) >> 5, iFact = ((x + 1) * angle) & 31;
        for (int y = 0; y < nTbS; y++) {
          int v;
          if (iFact) v = ((32 - iFact) * ref[y + iIdx + 1] + iFact * ref[y + iIdx + 2] + 16) >> 5;
          else v = ref[y + iIdx + 1];
          dst[y * dst_stride + x] = (uint16_t)v;
        }
      }
      if (mode == 10 && cIdx == 0 && nTbS < 32)
        for (int x = 0; x < nTbS; x++) {
          int v = L[1] + ((T[x + 1] - T[0]) >> 1);
          dst[x] = (uint16_t)Clip3(0, maxv, v);
        }
    }
  }
}

/* 8.4.4.2.2 reference sample availability marking + substitution, then prediction */
static void intra_predict_block(Dec* d, int x0c, int y0c, int log2n, int cIdx, int mode)
{
  /* (x0c,y0c) in component samples */
  const SPS* s = d->s;
  int nTbS = 1 << log2n, n2 = 2 * nTbS;
  int subw = cIdx ? d->subw : 1, subh = cIdx ? d->subh : 1;
  int stride = cIdx ? d->Wc : d->W;
  uint16_t* rec = d->rec[cIdx];
  int bit_depth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
  int xTbY = x0c * subw, yTbY = y0c * subh;
  uint16_t L[65], T[65];
  uint8_t aL[65], aT[65]; /* availability; index 0 = corner */
  int any = 0;
  for (int i = 0; i <= n2; i++) {
    /* left column: p[-1][i-1] */
    int xN = x0c - 1, yN = y0c + i - 1;
    int av = available_z(d, xTbY, yTbY, xN * subw, yN * subh);
    /* 8.4.4.2.2: with constrained_intra_pred_flag a sample of a unit that is not intra coded is marked "not available for intra prediction" */
    if (av && d->p->constrained_intra_pred_flag && d->m_pred && d->m_pred[((yN * subh) >> 2) * d->mw + ((xN * subw) >> 2)] != 0) av = 0;
    aL[i] = (uint8_t)av;
    if (av) { L[i] = rec[yN * stride + xN]; any = 1; }
    /* top row: p[i-1][-1] */
    xN = x0c + i - 1; yN = y0c - 1;
    if (i == 0) { aT[0] = aL[0]; T[0] = L[0]; continue; }
    av = available_z(d, xTbY, yTbY, xN * subw, yN * subh);
    if (av && d->p->constrained_intra_pred_flag && d->m_pred && d->m_pred[((yN * subh) >> 2) * d->mw + ((xN * subw) >> 2)] != 0) av = 0;
    aT[i] = (uint8_t)av;
    if (av) { T[i] = rec[yN * stride + xN]; any = 1; }
  }
  if (!any) {
    for (int i = 0; i <= n2; i++) L[i] = T[i] = (uint16_t)(1 << (bit_depth - 1));
  } else {
    /* search order: p[-1][2n-1] .. p[-1][-1], then p[0][-1] .. p[2n-1][-1] */
    if (!aL[n2]) {
      int found = 0; uint16_t v = 0;
      for (int i = n2 - 1; i >= 0 && !found; i--) if (aL[i]) { v = L[i]; found = 1; }
      for (int i = 1; i <= n2 && !found; i++) if (aT[i]) { v = T[i]; found = 1; }
      L[n2] = v; aL[n2] = 1;
    }
    for (int i = n2 - 1; i >= 0; i--) if (!aL[i]) { L[i] = L[i + 1]; aL[i] = 1; }
    T[0] = L[0];
    for (int i = 1; i <= n2; i++) if (!aT[i]) { T[i] = T[i - 1]; aT[i] = 1; }
  }
  hevc_intra_predict(rec + y0c * stride + x0c, stride, nTbS, cIdx, mode, L, T, bit_depth,
                     s->strong_intra_smoothing_enabled_flag, s->chroma_format_idc);
}

/* ------------------------------------------------------------------------------------------ */
/* 8.6.2 - 8.6.4 scaling and transformation                                                    */
/* ------------------------------------------------------------------------------------------ */
void hevc_scale_and_transform(int32_t* res, const int32_t* coeff, int nTbS, int qP, int bit_depth,
                              const uint8_t* m, int transform_skip, int trType)
{
  static const int levelScale[6] = {40, 45, 51, 57, 64, 72};
  int log2n = 0; while ((1 << log2n) < nTbS) log2n++;
  int32_t dq[32 * 32];
  init_dct();
  /* 8.6.4.2 */
  int bdShift = bit_depth + log2n - 5;
  for (int i = 0; i < nTbS * nTbS; i++) {
    int mm = m ? m[i] : 16;
    int64_t v = ((int64_t)coeff[i] * mm * levelScale[qP % 6]) << (qP / 6);
    v = (v + ((int64_t)1 << (bdShift - 1))) >> bdShift;
    dq[i] = (int32_t)Clip3(-32768, 32767, v);
  }
  int bdShift2 = 20 - bit_depth;
  if (transform_skip) { /* 8.6.4.2 residual modification for transform skip: r = d << 7 */
    for (int i = 0; i < nTbS * nTbS; i++) {
      int32_t r = dq[i] * 128;
      res[i] = (r + (1 << (bdShift2 - 1))) >> bdShift2;
    }
    return;
  }
  /* 8.6.4.2: first stage = columns, intermediate clip, second stage = rows */
  int32_t e[32 * 32], g[32 * 32];
  for (int x = 0; x < nTbS; x++)
    for (int i = 0; i < nTbS; i++) {
      int64_t sum = 0;
      for (int j = 0; j < nTbS; j++) {
        int c = trType ? g_dst[j][i] : g_dct[j * (32 / nTbS)][i];
        sum += (int64_t)c * dq[j * nTbS + x];
      }
      e[i * nTbS + x] = (int32_t)sum;
    }
  for (int i = 0; i < nTbS * nTbS; i++) g[i] = Clip3(-32768, 32767, (e[i] + 64) >> 7);
  for (int y = 0; y < nTbS; y++)
    for (int i = 0; i < nTbS; i++) {
      int64_t sum = 0;
      for (int j = 0; j < nTbS; j++) {
        int c = trType ? g_dst[j][i] : g_dct[j * (32 / nTbS)][i];
        sum += (int64_t)c * g[y * nTbS + j];
      }
      res[y * nTbS + i] = (int32_t)((sum + (1 << (bdShift2 - 1))) >> bdShift2);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* 7.3.8.11 residual_coding + 9.3.4.2 context selection                                        */
/* ------------------------------------------------------------------------------------------ */
static int decode_coeff_abs_level_remaining(Dec* d, int cRiceParam) /* 9.3.3.11 */
{
  int prefix = 0;
  while (prefix < 32 && decode_bypass(d)) prefix++;
  if (prefix == 32) fail(d, "coeff_abs_level_remaining prefix too long");
  if (prefix <= 3) {
    return (prefix << cRiceParam) + decode_bypass_bits(d, cRiceParam);
  }
  int suffix_len = prefix - 3 + cRiceParam;
  if (suffix_len > 30) fail(d, "coeff_abs_level_remaining suffix too long");
  return (((1 << (prefix - 3)) + 3 - 1) << cRiceParam) + decode_bypass_bits(d, suffix_len);
}

static void residual_coding(Dec* d, int x0, int y0, int log2TrafoSize, int cIdx, int predModeIntra,
                            int32_t* TransCoeffLevel /* nTbS*nTbS, raster */, int* transform_skip_flag)
{
  const PPS* p = d->p;
  int nTbS = 1 << log2TrafoSize;
  (void)x0; (void)y0;
  memset(TransCoeffLevel, 0, sizeof(int32_t) * nTbS * nTbS);
  *transform_skip_flag = 0;
  if (p->transform_skip_enabled_flag && !d->cu_transquant_bypass_flag && log2TrafoSize <= 2)
    *transform_skip_flag = decode_decision(d, CTX_TRANSFORM_SKIP + (cIdx ? 1 : 0));

  /* last significant coefficient position: 9.3.4.2.3 */
  int ctxOffset, ctxShift;
  if (cIdx == 0) { ctxOffset = 3 * (log2TrafoSize - 2) + ((log2TrafoSize - 1) >> 2); ctxShift = (log2TrafoSize + 1) >> 2; }
  else { ctxOffset = 15; ctxShift = log2TrafoSize - 2; }
  int cMax = (log2TrafoSize << 1) - 1;
  int last_x_prefix = 0, last_y_prefix = 0;
  while (last_x_prefix < cMax && decode_decision(d, CTX_LAST_X + ctxOffset + (last_x_prefix >> ctxShift))) last_x_prefix++;
  while (last_y_prefix < cMax && decode_decision(d, CTX_LAST_Y + ctxOffset + (last_y_prefix >> ctxShift))) last_y_prefix++;
  int LastX = last_x_prefix, LastY = last_y_prefix;
  if (last_x_prefix > 3) {
    int suf = decode_bypass_bits(d, (last_x_prefix >> 1) - 1);
    LastX = (1 << ((last_x_prefix >> 1) - 1)) * (2 + (last_x_prefix & 1)) + suf;
  }
  if (last_y_prefix > 3) {
    int suf = decode_bypass_bits(d, (last_y_prefix >> 1) - 1);
    LastY = (1 << ((last_y_prefix >> 1) - 1)) * (2 + (last_y_prefix & 1)) + suf;
  }
  /* 7.4.9.11 scanIdx */
  int scanIdx = 0;
  if (log2TrafoSize == 2 || (log2TrafoSize == 3 && (cIdx == 0 || d->s->chroma_format_idc == 3))) {
    if (predModeIntra >= 6 && predModeIntra <= 14) scanIdx = 2;
    else if (predModeIntra >= 22 && predModeIntra <= 30) scanIdx = 1;
  }
  if (scanIdx == 2) { int t = LastX; LastX = LastY; LastY = t; }
  if (LastX >= nTbS || LastY >= nTbS) fail(d, "last significant coefficient outside the block");

  const uint8_t* scanSB = log2TrafoSize > 2 ? g_scan[log2TrafoSize - 2][scanIdx] : NULL;
  const uint8_t* scanPos = g_scan[2][scanIdx];
  int lastScanPos = 16;
  int lastSubBlock = (1 << (log2TrafoSize - 2)) * (1 << (log2TrafoSize - 2)) - 1;
  int xC, yC, xS, yS;
  do {
    if (lastScanPos == 0) { lastScanPos = 16; lastSubBlock--; }
    lastScanPos--;
    xS = scanSB ? (scanSB[lastSubBlock] & 15) : 0; yS = scanSB ? (scanSB[lastSubBlock] >> 4) : 0;
    xC = (xS << 2) + (scanPos[lastScanPos] & 15);
    yC = (yS << 2) + (scanPos[lastScanPos] >> 4);
  } while (xC != LastX || yC != LastY);

  uint8_t coded_sub_block_flag[8][8];
  memset(coded_sub_block_flag, 0, sizeof(coded_sub_block_flag));
  int sbw = 1 << (log2TrafoSize - 2);
  int greater1Ctx_carry = 1; /* 9.3.4.2.6: state carried between sub-blocks */
  int first_subblock_with_g1 = 1;

  for (int i = lastSubBlock; i >= 0; i--) {
    xS = scanSB ? (scanSB[i] & 15) : 0; yS = scanSB ? (scanSB[i] >> 4) : 0;
    int inferSbDcSigCoeffFlag = 0;
    int csbf;
    if (i < lastSubBlock && i > 0) {
      /* 9.3.4.2.4 */
      int csbfCtx = 0;
      if (xS < sbw - 1) csbfCtx += coded_sub_block_flag[xS + 1][yS];
      if (yS < sbw - 1) csbfCtx += coded_sub_block_flag[xS][yS + 1];
      int ctxInc = Min(csbfCtx, 1) + (cIdx ? 2 : 0);
      csbf = decode_decision(d, CTX_CODED_SUB_BLOCK + ctxInc);
      inferSbDcSigCoeffFlag = 1;
    } else csbf = 1;
    coded_sub_block_flag[xS][yS] = (uint8_t)csbf;

    uint8_t sig[16];
    memset(sig, 0, 16);
    int nStart = (i == lastSubBlock) ? lastScanPos - 1 : 15;
    if (i == lastSubBlock) sig[lastScanPos] = 1;
    /* prevCsbf for sig_coeff_flag context 9.3.4.2.5 */
    int prevCsbf = 0;
    if (xS < sbw - 1) prevCsbf += coded_sub_block_flag[xS + 1][yS];
    if (yS < sbw - 1) prevCsbf += 2 * coded_sub_block_flag[xS][yS + 1];
    for (int n = nStart; n >= 0; n--) {
      int xP = scanPos[n] & 15, yP = scanPos[n] >> 4;
      xC = (xS << 2) + xP; yC = (yS << 2) + yP;
      if (csbf && (n > 0 || !inferSbDcSigCoeffFlag)) {
        int sigCtx;
        if (log2TrafoSize == 2) {
          static const uint8_t ctxIdxMap[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
          sigCtx = ctxIdxMap[(yC << 2) + xC];
        } else if (xC + yC == 0) sigCtx = 0;
        else {
          if (prevCsbf == 0) sigCtx = (xP + yP == 0) ? 2 : (xP + yP < 3) ? 1 : 0;
          else if (prevCsbf == 1) sigCtx = (yP == 0) ? 2 : (yP == 1) ? 1 : 0;
          else if (prevCsbf == 2) sigCtx = (xP == 0) ? 2 : (xP == 1) ? 1 : 0;
          else sigCtx = 2;
          if (cIdx == 0) {
            if (xS > 0 || yS > 0) sigCtx += 3;
            if (log2TrafoSize == 3) sigCtx += (scanIdx == 0) ? 9 : 15; else sigCtx += 21;
          } else {
            if (log2TrafoSize == 3) sigCtx += 9; else sigCtx += 12;
          }
        }
        int ctxInc = cIdx == 0 ? sigCtx : 27 + sigCtx;
        sig[n] = (uint8_t)decode_decision(d, CTX_SIG_COEFF + ctxInc);
        if (sig[n]) inferSbDcSigCoeffFlag = 0;
      } else if (csbf && n == 0 && inferSbDcSigCoeffFlag) {
        sig[0] = 1; /* inferred: the sub-block is coded but no other coefficient was significant */
      }
    }

    int firstSigScanPos = 16, lastSigScanPos = -1, numGreater1Flag = 0, lastGreater1ScanPos = -1;
    uint8_t g1[16], g2[16];
    memset(g1, 0, 16); memset(g2, 0, 16);
    int ctxSet = 0, greater1Ctx = 1, first_in_sb = 1;
    for (int n = 15; n >= 0; n--) {
      if (!sig[n]) continue;
      if (numGreater1Flag < 8) {
        if (first_in_sb) { /* 9.3.4.2.6 */
          ctxSet = (i == 0 || cIdx > 0) ? 0 : 2;
          if (!first_subblock_with_g1 && greater1Ctx_carry == 0) ctxSet++;
          greater1Ctx = 1;
          first_in_sb = 0;
          first_subblock_with_g1 = 0;
        }
        int ctxInc = ctxSet * 4 + Min(3, greater1Ctx) + (cIdx ? 16 : 0);
        g1[n] = (uint8_t)decode_decision(d, CTX_GREATER1 + ctxInc);
        if (g1[n]) greater1Ctx = 0; else if (greater1Ctx > 0) greater1Ctx++;
        greater1Ctx_carry = greater1Ctx;
        numGreater1Flag++;
        if (g1[n] && lastGreater1ScanPos == -1) lastGreater1ScanPos = n;
      }
      if (lastSigScanPos == -1) lastSigScanPos = n;
      firstSigScanPos = n;
    }
    int signHidden = d->cu_transquant_bypass_flag ? 0 : (lastSigScanPos - firstSigScanPos > 3);
    if (lastGreater1ScanPos != -1) {
      int ctxInc = ctxSet + (cIdx ? 4 : 0);
      g2[lastGreater1ScanPos] = (uint8_t)decode_decision(d, CTX_GREATER2 + ctxInc);
    }
    uint8_t sign[16];
    memset(sign, 0, 16);
    for (int n = 15; n >= 0; n--)
      if (sig[n] && (!p->sign_data_hiding_enabled_flag || !signHidden || n != firstSigScanPos))
        sign[n] = (uint8_t)decode_bypass(d);
    int numSigCoeff = 0, sumAbsLevel = 0;
    int cRiceParam = 0, have_prev = 0; (void)have_prev;
    for (int n = 15; n >= 0; n--) {
      if (!sig[n]) continue;
      int baseLevel = 1 + g1[n] + g2[n];
      int rem = 0;
      if (baseLevel == ((numSigCoeff < 8) ? ((n == lastGreater1ScanPos) ? 3 : 2) : 1)) {
        rem = decode_coeff_abs_level_remaining(d, cRiceParam);
        int absLevel = baseLevel + rem;
        if (absLevel > 3 * (1 << cRiceParam)) cRiceParam = Min(cRiceParam + 1, 4);
      }
      int absv = baseLevel + rem;
      int v = absv * (1 - 2 * sign[n]);
      if (p->sign_data_hiding_enabled_flag && signHidden) {
        sumAbsLevel += absv;
        if (n == firstSigScanPos && (sumAbsLevel % 2) == 1) v = -v;
      }
      if (v > 32767 || v < -32768) fail(d, "TransCoeffLevel out of 16-bit range");
      xC = (xS << 2) + (scanPos[n] & 15); yC = (yS << 2) + (scanPos[n] >> 4);
      TransCoeffLevel[yC * nTbS + xC] = v;
      numSigCoeff++;
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* 8.6.1 quantisation parameters                                                              */
/* ------------------------------------------------------------------------------------------ */
static void derive_qp_pred(Dec* d, int xCb, int yCb)
{
  const SPS* s = d->s; const PPS* p = d->p;
  int Log2MinCuQpDeltaSize = s->log2_ctb - p->diff_cu_qp_delta_depth;
  int xQg = xCb - (xCb & ((1 << Log2MinCuQpDeltaSize) - 1));
  int yQg = yCb - (yCb & ((1 << Log2MinCuQpDeltaSize) - 1));
  int qPY_PREV = d->last_qp_y; /* set to SliceQpY at slice / tile / WPP-row start */
  int qPY_A = qPY_PREV, qPY_B = qPY_PREV;
  int ctbCur = (yCb >> s->log2_ctb) * d->ctbW + (xCb >> s->log2_ctb);
  if (available_z(d, xCb, yCb, xQg - 1, yQg)) {
    int ctbA = (yQg >> s->log2_ctb) * d->ctbW + ((xQg - 1) >> s->log2_ctb);
    if (ctbA == ctbCur) qPY_A = d->m_qp[(yQg >> 2) * d->mw + ((xQg - 1) >> 2)];
  }
  if (available_z(d, xCb, yCb, xQg, yQg - 1)) {
    int ctbB = ((yQg - 1) >> s->log2_ctb) * d->ctbW + (xQg >> s->log2_ctb);
    if (ctbB == ctbCur) qPY_B = d->m_qp[((yQg - 1) >> 2) * d->mw + (xQg >> 2)];
  }
  d->qPY_PRED = (qPY_A + qPY_B + 1) >> 1;
}
static void set_qp_y(Dec* d)
{
  int QpBdOffsetY = 6 * (d->s->bit_depth_luma - 8);
  d->cur_qp_y = ((d->qPY_PRED + d->CuQpDeltaVal + 52 + 2 * QpBdOffsetY) % (52 + QpBdOffsetY)) - QpBdOffsetY;
}

/* ------------------------------------------------------------------------------------------ */
/* reconstruction of one transform block: prediction + residual                               */
/* ------------------------------------------------------------------------------------------ */
static void reconstruct_tb(Dec* d, int x0c, int y0c, int log2n, int cIdx, int mode, int cbf,
                           const int32_t* coeffs, int transform_skip)
{
  const SPS* s = d->s; const PPS* p = d->p;
  int n = 1 << log2n;
  int stride = cIdx ? d->Wc : d->W;
  uint16_t* rec = d->rec[cIdx];
  int bit_depth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
  if (!d->cu_pred_inter) intra_predict_block(d, x0c, y0c, log2n, cIdx, mode);   /* inter: the prediction samples are in rec already */
  if (!cbf) return;
  if (d->keep_taps)
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++) d->coeff[cIdx][(y0c + y) * stride + x0c + x] = coeffs[y * n + x];
  int32_t res[32 * 32];
  if (d->cu_transquant_bypass_flag) {
    memcpy(res, coeffs, sizeof(int32_t) * n * n);
  } else {
    int qP;
    if (cIdx == 0) qP = d->cur_qp_y + 6 * (s->bit_depth_luma - 8);
    else {
      int QpBdOffsetC = 6 * (s->bit_depth_chroma - 8);
      int off = cIdx == 1 ? p->pps_cb_qp_offset + d->sh->slice_cb_qp_offset : p->pps_cr_qp_offset + d->sh->slice_cr_qp_offset;
      int qPi = Clip3(-QpBdOffsetC, 57, d->cur_qp_y + off);
      qP = (s->chroma_format_idc == 1 ? hevc_chroma_qp_420(qPi) : Min(qPi, 51)) + QpBdOffsetC;   /* 8.6.1: table 8-10 only for ChromaArrayType 1 */
    }
    const uint8_t* m = NULL;
    uint8_t mbuf[32 * 32];
    if (s->scaling_list_enabled_flag && !(transform_skip && n > 4)) {
      const ScalingList* sl = p->pps_scaling_list_data_present_flag ? &p->sl : &s->sl;
      int matrixId = cIdx + (d->cu_pred_inter ? 3 : 0); /* Table 7-4: intra 0..2, inter 3..5 */
      for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
          int v;
          if (n == 4) v = sl->ScalingFactor4[matrixId][y * 4 + x];
          else if (n == 8) v = sl->ScalingFactor8[matrixId][y * 8 + x];
          else if (n == 16) v = (x == 0 && y == 0) ? sl->dc16[matrixId] : sl->ScalingFactor16[matrixId][(y >> 1) * 8 + (x >> 1)];
          else if (cIdx == 0) v = (x == 0 && y == 0) ? sl->dc32[matrixId] : sl->ScalingFactor32[matrixId][(y >> 2) * 8 + (x >> 2)];
          /* 7.4.5: with ChromaArrayType 3 the 32x32 chroma matrices are the 16x16 lists of the component, upsampled by 4, with the 16x16 DC */
          else v = (x == 0 && y == 0) ? sl->dc16[matrixId] : sl->ScalingFactor16[matrixId][(y >> 2) * 8 + (x >> 2)];
          mbuf[y * n + x] = (uint8_t)v;
        }
      m = mbuf;
    }
    int trType = (cIdx == 0 && n == 4 && !d->cu_pred_inter) ? 1 : 0;   /* DST-VII for intra 4x4 luma only (8.6.4.2) */
    hevc_scale_and_transform(res, coeffs, n, qP, bit_depth, m, transform_skip, trType);
  }
  int maxv = (1 << bit_depth) - 1;
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) {
      int v = rec[(y0c + y) * stride + x0c + x] + res[y * n + x];
      rec[(y0c + y) * stride + x0c + x] = (uint16_t)Clip3(0, maxv, v);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* 7.3.8.8 - 7.3.8.10 transform tree / transform unit                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int xCb, yCb, log2CbSize;
  int IntraSplitFlag, MaxTrafoDepth;
  int chroma_mode;
  int inter, PartMode;     /* CuPredMode != MODE_INTRA; its partitioning (interSplitFlag, 7.4.9.8) */
} CuCtx;

static void mark_tu(Dec* d, const CuCtx* cu, int x0, int y0, int log2TrafoSize, int cbfL, int cbfCb, int cbfCr)
{
  /* fills the 4x4-unit maps for a leaf transform unit and records deblocking edges (8.7.2.2/3) */
  const SPS* s = d->s; const PPS* p = d->p;
  int n4 = 1 << (log2TrafoSize - 2);
  int ux = x0 >> 2, uy = y0 >> 2;
  int deblock = !d->sh->slice_deblocking_filter_disabled_flag;
  /* filterEdgeFlag for the left / top edge of this TU */
  int fl = 1, ft = 1;
  if (x0 == 0) fl = 0;
  if (y0 == 0) ft = 0;
  int ctbCur = (y0 >> s->log2_ctb) * d->ctbW + (x0 >> s->log2_ctb);
  if (x0 > 0 && fl) {
    int ctbN = (y0 >> s->log2_ctb) * d->ctbW + ((x0 - 1) >> s->log2_ctb);
    if (ctbN != ctbCur) {
      if (d->ctb_slice_addr[ctbN] != d->ctb_slice_addr[ctbCur] && !d->sh->slice_loop_filter_across_slices_enabled_flag) fl = 0;
      if (d->TileId[d->CtbAddrRsToTs[ctbN]] != d->TileId[d->CtbAddrRsToTs[ctbCur]] && !p->loop_filter_across_tiles_enabled_flag) fl = 0;
    }
  }
  if (y0 > 0 && ft) {
    int ctbN = ((y0 - 1) >> s->log2_ctb) * d->ctbW + (x0 >> s->log2_ctb);
    if (ctbN != ctbCur) {
      if (d->ctb_slice_addr[ctbN] != d->ctb_slice_addr[ctbCur] && !d->sh->slice_loop_filter_across_slices_enabled_flag) ft = 0;
      if (d->TileId[d->CtbAddrRsToTs[ctbN]] != d->TileId[d->CtbAddrRsToTs[ctbCur]] && !p->loop_filter_across_tiles_enabled_flag) ft = 0;
    }
  }
  (void)cu;
  for (int j = 0; j < n4; j++)
    for (int i = 0; i < n4; i++) {
      int idx = (uy + j) * d->mw + ux + i;
      d->m_log2_tb[idx] = (uint8_t)log2TrafoSize;
      uint8_t f = d->m_flags[idx] & 0x18; /* keep bypass / pcm bits set by the CU */
      f |= (uint8_t)((cbfL ? 1 : 0) | (cbfCb ? 2 : 0) | (cbfCr ? 4 : 0));
      if (deblock && i == 0 && fl) f |= 0x20;
      if (deblock && j == 0 && ft) f |= 0x40;
      d->m_flags[idx] = f;
    }
}

/* 7.3.8.14 cu_qp_delta_abs / cu_qp_delta_sign_flag, once per quantisation group, in the first transform unit with a coded block */
static void parse_cu_qp_delta_if_needed(Dec* d, int coded)
{
  const SPS* s = d->s; const PPS* p = d->p;
  if (!coded) return;
  if (p->cu_qp_delta_enabled_flag && !d->IsCuQpDeltaCoded) {
    /* cu_qp_delta_abs 9.3.3.10: prefix TU cMax 5 (ctx 0, then ctx 1), suffix EG0 */
    int v = 0;
    if (decode_decision(d, CTX_CU_QP_DELTA + 0)) {
      v = 1;
      while (v < 5 && decode_decision(d, CTX_CU_QP_DELTA + 1)) v++;
      if (v == 5) {
        int k = 0;
        while (decode_bypass(d)) { v += 1 << k; k++; if (k > 16) fail(d, "cu_qp_delta_abs too large"); }
        while (k--) v += decode_bypass(d) << k;
      }
    }
    int sign = 0;
    if (v) sign = decode_bypass(d);
    d->IsCuQpDeltaCoded = 1;
    d->CuQpDeltaVal = v * (1 - 2 * sign);
    int QpBdOffsetY = 6 * (s->bit_depth_luma - 8);
    if (d->CuQpDeltaVal < -(26 + QpBdOffsetY / 2) || d->CuQpDeltaVal > 25 + QpBdOffsetY / 2)
      fail(d, "CuQpDeltaVal out of range");
    set_qp_y(d);
  }
}


/* 7.3.8.10 for ChromaArrayType 2: the chroma of a transform unit is a block half as wide and as tall as the luma block, coded as TWO square
   blocks one above the other (tIdx 0, 1), each with its own coded-block flag (cbf_cb / cbf_cr: bit tIdx) and transform_skip_flag.
   Order: luma, Cb 0, Cb 1, Cr 0, Cr 1; reconstruction in the same order (block 1 predicts from block 0). */
static void transform_unit_422(Dec* d, CuCtx* cu, int x0, int y0, int xBase, int yBase, int log2TrafoSize, int blkIdx, int cbf_luma,
                               int cbf_cb, int cbf_cr)
{
  int32_t cY[32 * 32];
  int32_t cC[4][16 * 16];   /* Cb 0, Cb 1, Cr 0, Cr 1 */
  int tsY = 0, tsC[4] = {0, 0, 0, 0};
  int lumaMode = d->m_ipm[(y0 >> 2) * d->mw + (x0 >> 2)];
  int chromaMode = cu->chroma_mode;
  int do_chroma = 0, xC0 = 0, yC0 = 0, log2C = 2;
  if (log2TrafoSize > 2) { do_chroma = 1; xC0 = x0 / 2; yC0 = y0; log2C = log2TrafoSize - 1; }
  else if (blkIdx == 3) { do_chroma = 1; xC0 = xBase / 2; yC0 = yBase; log2C = 2; }
  parse_cu_qp_delta_if_needed(d, cbf_luma || cbf_cb || cbf_cr);
  if (cbf_luma) residual_coding(d, x0, y0, log2TrafoSize, 0, lumaMode, cY, &tsY);
  if (do_chroma)
    for (int c = 0; c < 2; c++)
      for (int t = 0; t < 2; t++)
        if (((c ? cbf_cr : cbf_cb) >> t) & 1) residual_coding(d, xC0, yC0 + (t << log2C), log2C, 1 + c, chromaMode, cC[2 * c + t], &tsC[2 * c + t]);
  reconstruct_tb(d, x0, y0, log2TrafoSize, 0, lumaMode, cbf_luma, cY, tsY);
  if (do_chroma)
    for (int c = 0; c < 2; c++)
      for (int t = 0; t < 2; t++)
        reconstruct_tb(d, xC0, yC0 + (t << log2C), log2C, 1 + c, chromaMode, ((c ? cbf_cr : cbf_cb) >> t) & 1, cC[2 * c + t], tsC[2 * c + t]);
  mark_tu(d, cu, x0, y0, log2TrafoSize, cbf_luma, do_chroma ? cbf_cb : 0, do_chroma ? cbf_cr : 0);
}

static void transform_unit(Dec* d, CuCtx* cu, int x0, int y0, int xBase, int yBase, int log2TrafoSize,
                           int trafoDepth, int blkIdx, int cbf_luma, int cbf_cb, int cbf_cr)
{
  const SPS* s = d->s;
  int ChromaArrayType = s->chroma_format_idc;
  int cbfChroma = cbf_cb || cbf_cr; /* for log2TrafoSize == 2 these are the parent's flags; 4:2:2: bit 0 upper block, bit 1 lower block */
  int32_t cY[32 * 32], cCb[32 * 32], cCr[32 * 32];
  int tsY = 0, tsCb = 0, tsCr = 0;
  if (ChromaArrayType == 2) { transform_unit_422(d, cu, x0, y0, xBase, yBase, log2TrafoSize, blkIdx, cbf_luma, cbf_cb, cbf_cr); return; }
  (void)trafoDepth;
  int lumaMode = d->m_ipm[(y0 >> 2) * d->mw + (x0 >> 2)];
  int chromaMode = cu->chroma_mode;
  int do_chroma = 0, xC0 = 0, yC0 = 0, log2C = 0;
  if (ChromaArrayType == 3) {   /* chroma blocks coincide with the luma blocks; the mode is the one of the block's partition */
    do_chroma = 1; xC0 = x0; yC0 = y0; log2C = log2TrafoSize;
    chromaMode = d->m_ipmc[(y0 >> 2) * d->mw + (x0 >> 2)];
  } else if (ChromaArrayType) {
    if (log2TrafoSize > 2) { do_chroma = 1; xC0 = x0 / 2; yC0 = y0 / 2; log2C = log2TrafoSize - 1; }
    else if (blkIdx == 3) { do_chroma = 1; xC0 = xBase / 2; yC0 = yBase / 2; log2C = 2; }
  }
  /* 7.3.8.10: cbfChroma uses the parent's chroma cbf for 4x4 luma blocks regardless of blkIdx */
  parse_cu_qp_delta_if_needed(d, cbf_luma || cbfChroma);
  /* parse residuals (all parsing of the TU precedes its reconstruction; the order of
     reconstruction between colour components is irrelevant in v1) */
  if (cbf_luma) residual_coding(d, x0, y0, log2TrafoSize, 0, lumaMode, cY, &tsY);
  if (do_chroma) {
    if (cbf_cb) residual_coding(d, xC0, yC0, log2C, 1, chromaMode, cCb, &tsCb);
    if (cbf_cr) residual_coding(d, xC0, yC0, log2C, 2, chromaMode, cCr, &tsCr);
  }
  reconstruct_tb(d, x0, y0, log2TrafoSize, 0, lumaMode, cbf_luma, cY, tsY);
  if (do_chroma) {
    reconstruct_tb(d, xC0, yC0, log2C, 1, chromaMode, cbf_cb, cCb, tsCb);
    reconstruct_tb(d, xC0, yC0, log2C, 2, chromaMode, cbf_cr, cCr, tsCr);
  }
  mark_tu(d, cu, x0, y0, log2TrafoSize, cbf_luma, do_chroma ? cbf_cb : 0, do_chroma ? cbf_cr : 0);
}

static void transform_tree(Dec* d, CuCtx* cu, int x0, int y0, int xBase, int yBase, int log2TrafoSize,
                           int trafoDepth, int blkIdx, int parent_cbf_cb, int parent_cbf_cr)
{
  const SPS* s = d->s;
  int ChromaArrayType = s->chroma_format_idc;
  int split;
  /* 7.4.9.8: interSplitFlag - an inter coding unit that is partitioned and may not split its transform tree by syntax splits it once anyway */
  int interSplit = cu->inter && s->max_transform_hierarchy_depth_inter == 0 && cu->PartMode != PART_2Nx2N && trafoDepth == 0;
  if (log2TrafoSize <= s->log2_max_tb && log2TrafoSize > s->log2_min_tb && trafoDepth < cu->MaxTrafoDepth &&
      !(cu->IntraSplitFlag && trafoDepth == 0))
    split = decode_decision(d, CTX_SPLIT_TRANSFORM + 5 - log2TrafoSize);
  else
    split = (log2TrafoSize > s->log2_max_tb || (cu->IntraSplitFlag && trafoDepth == 0) || interSplit) ? 1 : 0;
  int cbf_cb = 0, cbf_cr = 0;
  if ((log2TrafoSize > 2 && ChromaArrayType != 0) || ChromaArrayType == 3) {
    int cc = trafoDepth == 4 ? CTX_CBF_CHROMA4 : CTX_CBF_CHROMA + trafoDepth;
    /* ChromaArrayType 2: a second flag for the lower chroma block where the chroma is coded (a leaf, or the 8x8 node above four 4x4 leaves);
       the parent's flag that gates the parsing is the one at (xBase, yBase): its first */
    int two = ChromaArrayType == 2 && (!split || log2TrafoSize == 3);
    if (trafoDepth == 0 || (parent_cbf_cb & 1)) { cbf_cb = decode_decision(d, cc); if (two) cbf_cb |= decode_decision(d, cc) << 1; }
    if (trafoDepth == 0 || (parent_cbf_cr & 1)) { cbf_cr = decode_decision(d, cc); if (two) cbf_cr |= decode_decision(d, cc) << 1; }
  } else if (ChromaArrayType != 0 && trafoDepth > 0 && log2TrafoSize == 2) {
    cbf_cb = parent_cbf_cb; cbf_cr = parent_cbf_cr; /* 7.4.9.8 inference */
  }
  if (split) {
    int x1 = x0 + (1 << (log2TrafoSize - 1)), y1 = y0 + (1 << (log2TrafoSize - 1));
    transform_tree(d, cu, x0, y0, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 0, cbf_cb, cbf_cr);
    transform_tree(d, cu, x1, y0, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 1, cbf_cb, cbf_cr);
    transform_tree(d, cu, x0, y1, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 2, cbf_cb, cbf_cr);
    transform_tree(d, cu, x1, y1, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 3, cbf_cb, cbf_cr);
  } else {
    /* cbf_luma is present for intra coding units, below the root, or when a chroma block is coded; else inferred 1 (7.3.8.8) */
    int cbf_luma = 1;
    if (!cu->inter || trafoDepth != 0 || cbf_cb || cbf_cr) cbf_luma = decode_decision(d, CTX_CBF_LUMA + (trafoDepth == 0 ? 1 : 0));
    transform_unit(d, cu, x0, y0, xBase, yBase, log2TrafoSize, trafoDepth, blkIdx, cbf_luma, cbf_cb, cbf_cr);
  }
}

#include "hevc_oracle_inter.c"

static void transform_tree(Dec* d, CuCtx* cu, int x0, int y0, int xBase, int yBase, int log2TrafoSize,
                           int trafoDepth, int blkIdx, int parent_cbf_cb, int parent_cbf_cr);

/* 7.3.8.5 for CuPredMode MODE_INTER / MODE_SKIP: part_mode, the prediction units, rqt_root_cbf, the transform tree */
static void inter_coding_unit(Dec* d, CuCtx* cu, int x0, int y0, int log2CbSize, int cqtDepth, int cu_skip)
{
  const SPS* s = d->s;
  int nCbS = 1 << log2CbSize;
  int u0x = x0 >> 2, u0y = y0 >> 2, nu = nCbS >> 2;
  d->cu_pred_inter = 1;
  set_qp_y(d);
  for (int j = 0; j < nu; j++)
    for (int i = 0; i < nu; i++) {
      int idx = (u0y + j) * d->mw + u0x + i;
      d->m_log2_cb[idx] = (uint8_t)log2CbSize;
      d->m_ctdepth[idx] = (uint8_t)cqtDepth;
      d->m_flags[idx] = (uint8_t)(d->cu_transquant_bypass_flag ? 0x08 : 0);
      d->m_decoded[idx] = 1;
      d->m_ipm[idx] = 1; d->m_ipmc[idx] = 1;     /* a neighbour that is not intra coded counts as INTRA_DC (8.4.2) */
      d->m_pred[idx] = (uint8_t)(cu_skip ? 2 : 1);
    }
  int PartMode = PART_2Nx2N;
  if (!cu_skip) {
    PartMode = parse_part_mode_inter(d, log2CbSize);
    if (PartMode == PART_NxN && log2CbSize == 3) fail(d, "inter NxN partition of an 8x8 coding unit");
  }
  PbGeom g;
  int nParts = part_geometry(PartMode, x0, y0, nCbS, 0, &g), merge0 = 0;
  for (int k = 0; k < nParts; k++) {
    part_geometry(PartMode, x0, y0, nCbS, k, &g);
    int mf = prediction_unit(d, &g, PartMode, cu_skip, cqtDepth);
    if (k == 0) merge0 = mf;
  }
  int rqt_root_cbf = 0;
  if (!cu_skip) {
    rqt_root_cbf = 1;
    if (!(PartMode == PART_2Nx2N && merge0)) rqt_root_cbf = decode_decision(d, CTX_RQT_ROOT_CBF);
  }
  cu->inter = 1; cu->PartMode = PartMode; cu->chroma_mode = 1;
  if (rqt_root_cbf) {
    cu->IntraSplitFlag = 0;
    cu->MaxTrafoDepth = s->max_transform_hierarchy_depth_inter;
    transform_tree(d, cu, x0, y0, x0, y0, log2CbSize, 0, 0, 0, 0);
  } else mark_cu_no_residual(d, cu, x0, y0, log2CbSize);
  mark_pu_edges(d, x0, y0, nCbS, PartMode);
  set_qp_y(d);
  for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) d->m_qp[(u0y + j) * d->mw + u0x + i] = (int8_t)d->cur_qp_y;
  d->last_qp_y = d->cur_qp_y;
  d->cu_pred_inter = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* 7.3.8.5 coding_unit, 7.3.8.7 pcm_sample, 8.4.2 luma intra prediction mode                   */
/* ------------------------------------------------------------------------------------------ */
static int cand_mode(Dec* d, int xPb, int yPb, int xN, int yN, int isB)
{
  if (!available_z(d, xPb, yPb, xN, yN)) return 1;
  int idx = (yN >> 2) * d->mw + (xN >> 2);
  if (d->m_flags[idx] & 0x10) return 1; /* pcm_flag */
  if (d->m_pred && d->m_pred[idx]) return 1; /* CuPredMode != MODE_INTRA */
  if (isB && yN < ((yPb >> d->s->log2_ctb) << d->s->log2_ctb)) return 1;
  return d->m_ipm[idx];
}

static void coding_unit(Dec* d, int x0, int y0, int log2CbSize, int cqtDepth)
{
  const SPS* s = d->s; const PPS* p = d->p;
  int nCbS = 1 << log2CbSize;
  CuCtx cu; memset(&cu, 0, sizeof(cu));
  cu.xCb = x0; cu.yCb = y0; cu.log2CbSize = log2CbSize;
  d->cu_transquant_bypass_flag = 0;
  d->cu_pred_inter = 0;
  if (p->transquant_bypass_enabled_flag) d->cu_transquant_bypass_flag = decode_decision(d, CTX_CU_TQ_BYPASS);
  if (d->sh->slice_type != 2) {   /* 7.3.8.5 in a P / B slice: cu_skip_flag, pred_mode_flag */
    int ctxInc = 0;   /* 9.3.4.2.2: the left / above neighbours' cu_skip_flag */
    if (available_z(d, x0, y0, x0 - 1, y0) && d->m_pred[(y0 >> 2) * d->mw + ((x0 - 1) >> 2)] == 2) ctxInc++;
    if (available_z(d, x0, y0, x0, y0 - 1) && d->m_pred[((y0 - 1) >> 2) * d->mw + (x0 >> 2)] == 2) ctxInc++;
    int cu_skip = decode_decision(d, CTX_SKIP_FLAG + ctxInc);
    int inter = 1;
    if (!cu_skip) inter = decode_decision(d, CTX_PRED_MODE) ? 0 : 1;   /* pred_mode_flag 1 = MODE_INTRA */
    if (inter) { inter_coding_unit(d, &cu, x0, y0, log2CbSize, cqtDepth, cu_skip); return; }
  }
  int PartMode = 0; /* 0 = 2Nx2N, 1 = NxN */
  if (log2CbSize == s->log2_min_cb) PartMode = decode_decision(d, CTX_PART_MODE) ? 0 : 1;
  if (PartMode == 1 && log2CbSize == 3 && s->log2_min_tb > 2) fail(d, "NxN partition with 8x8 CU needs 4x4 transforms");
  int pcm_flag = 0;
  if (PartMode == 0 && s->pcm_enabled_flag && log2CbSize >= s->log2_min_pcm_cb && log2CbSize <= s->log2_max_pcm_cb)
    pcm_flag = decode_terminate(d);

  /* QP prediction for this CU (8.6.1): qPY_PRED is per quantisation group */
  set_qp_y(d);

  int u0x = x0 >> 2, u0y = y0 >> 2, nu = nCbS >> 2;
  for (int j = 0; j < nu; j++)
    for (int i = 0; i < nu; i++) {
      int idx = (u0y + j) * d->mw + u0x + i;
      d->m_log2_cb[idx] = (uint8_t)log2CbSize;
      d->m_ctdepth[idx] = (uint8_t)cqtDepth;
      d->m_flags[idx] = (uint8_t)((d->cu_transquant_bypass_flag ? 0x08 : 0) | (pcm_flag ? 0x10 : 0));
      d->m_decoded[idx] = 1;
      d->m_ipm[idx] = 1;
      if (d->m_pred) { d->m_pred[idx] = 0; d->mf_ref[2 * idx] = d->mf_ref[2 * idx + 1] = -1; d->mf_poc[2 * idx] = d->mf_poc[2 * idx + 1] = 0; memset(d->mf_mv + 4 * idx, 0, 4 * sizeof(int16_t)); }
    }

  if (pcm_flag) {
    /* 7.3.8.7: pcm_alignment_zero_bits then raw samples */
    cabac_finish_and_align(d);
    Cabac* c = &d->c;
    for (int cIdx = 0; cIdx < (s->chroma_format_idc ? 3 : 1); cIdx++) {
      int csw = cIdx ? d->subw : 1, csh = cIdx ? d->subh : 1;
      int n = nCbS / csw, nh = nCbS / csh, xs = x0 / csw, ys = y0 / csh;
      int depth = cIdx ? s->pcm_bit_depth_chroma : s->pcm_bit_depth_luma;
      int bd = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
      int stride = cIdx ? d->Wc : d->W;
      for (int y = 0; y < nh; y++)
        for (int x = 0; x < n; x++) {
          unsigned v = 0;
          for (int b = 0; b < depth; b++) {
            if (c->pos >= c->nbits) fail(d, "pcm samples past end of data");
            v = (v << 1) | ((c->data[c->pos >> 3] >> (7 - (c->pos & 7))) & 1);
            c->pos++;
          }
          d->rec[cIdx][(ys + y) * stride + xs + x] = (uint16_t)(v << (bd - depth));
        }
    }
    d->n_substreams--; /* re-initialisation after PCM is not a new substream */
    cabac_init_engine(d);
    /* deblocking edges: a PCM CU is one transform block of CU size for edge purposes */
    mark_tu(d, &cu, x0, y0, log2CbSize, 0, 0, 0);
    /* mark_tu writes log2_tb = log2CbSize which may exceed 5; cap for the maps */
    for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) {
      int idx = (u0y + j) * d->mw + u0x + i;
      d->m_ipmc[idx] = 1;
      d->m_qp[idx] = (int8_t)d->cur_qp_y;
    }
    d->last_qp_y = d->cur_qp_y;
    return;
  }

  /* intra prediction modes */
  int pbOffset = PartMode == 1 ? nCbS / 2 : nCbS;
  int nPart = PartMode == 1 ? 2 : 1;
  int prev_flag[4], mpm_idx[4] = {0, 0, 0, 0}, rem_mode[4] = {0, 0, 0, 0};
  for (int j = 0; j < nPart; j++) for (int i = 0; i < nPart; i++) prev_flag[j * 2 + i] = decode_decision(d, CTX_PREV_INTRA_LUMA);
  for (int j = 0; j < nPart; j++)
    for (int i = 0; i < nPart; i++) {
      int k = j * 2 + i;
      if (prev_flag[k]) { /* mpm_idx: TR cMax 2, bypass */
        int v = 0;
        if (decode_bypass(d)) { v = 1; if (decode_bypass(d)) v = 2; }
        mpm_idx[k] = v;
      } else rem_mode[k] = decode_bypass_bits(d, 5);
      /* 8.4.2 derivation: must happen in order because later PBs use earlier ones as neighbours */
      int xPb = x0 + i * pbOffset, yPb = y0 + j * pbOffset;
      int candA = cand_mode(d, xPb, yPb, xPb - 1, yPb, 0);
      int candB = cand_mode(d, xPb, yPb, xPb, yPb - 1, 1);
      int cl[3];
      if (candA == candB) {
        if (candA < 2) { cl[0] = 0; cl[1] = 1; cl[2] = 26; }
        else { cl[0] = candA; cl[1] = 2 + ((candA + 29) % 32); cl[2] = 2 + ((candA - 2 + 1) % 32); }
      } else {
        cl[0] = candA; cl[1] = candB;
        if (candA != 0 && candB != 0) cl[2] = 0;
        else if (candA != 1 && candB != 1) cl[2] = 1;
        else cl[2] = 26;
      }
      int mode;
      if (prev_flag[k]) mode = cl[mpm_idx[k]];
      else {
        int t;
        if (cl[0] > cl[1]) { t = cl[0]; cl[0] = cl[1]; cl[1] = t; }
        if (cl[0] > cl[2]) { t = cl[0]; cl[0] = cl[2]; cl[2] = t; }
        if (cl[1] > cl[2]) { t = cl[1]; cl[1] = cl[2]; cl[2] = t; }
        mode = rem_mode[k];
        for (int q = 0; q < 3; q++) if (mode >= cl[q]) mode++;
      }
      int pu = pbOffset >> 2;
      for (int jj = 0; jj < pu; jj++) for (int ii = 0; ii < pu; ii++)
        d->m_ipm[((yPb >> 2) + jj) * d->mw + (xPb >> 2) + ii] = (uint8_t)mode;
    }
  /* NOTE on ordering: the syntax codes all prev_intra_luma_pred_flags first, then the
     mpm_idx/rem_intra_luma_pred_mode of each partition (7.3.8.5); the derivation above interleaves
     nothing that reads the bitstream out of order, because the flags were all read beforehand. */
  int chroma_mode = 1;
  if (s->chroma_format_idc) {
    /* 7.3.8.5: one intra_chroma_pred_mode per coding unit, or one per partition when ChromaArrayType is 3 and the CU is split NxN */
    static const uint8_t tab[4] = {0, 26, 10, 1};
    int nc = (s->chroma_format_idc == 3 && PartMode == 1) ? 2 : 1;
    int cpb = nc == 2 ? nCbS / 2 : nCbS;
    for (int j = 0; j < nc; j++)
      for (int i = 0; i < nc; i++) {
        int icpm;
        if (!decode_decision(d, CTX_INTRA_CHROMA)) icpm = 4;
        else icpm = decode_bypass_bits(d, 2);
        int xP = x0 + i * cpb, yP = y0 + j * cpb;
        int lm = d->m_ipm[(yP >> 2) * d->mw + (xP >> 2)];
        int m = icpm == 4 ? lm : ((tab[icpm] == lm) ? 34 : tab[icpm]);
        if (s->chroma_format_idc == 2) {   /* 8.4.3: the 4:2:2 sampling grid is not square, Table 8-3 maps the direction */
          static const uint8_t map422[35] = {0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 11, 13, 15, 16, 18, 19, 20, 21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27,
                                             28, 28, 29, 29, 30, 31};
          m = map422[m];
        }
        if (i == 0 && j == 0) chroma_mode = m;
        for (int jj = 0; jj < (cpb >> 2); jj++) for (int ii = 0; ii < (cpb >> 2); ii++)
          d->m_ipmc[((yP >> 2) + jj) * d->mw + (xP >> 2) + ii] = (uint8_t)m;
      }
  } else {
    for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) d->m_ipmc[(u0y + j) * d->mw + u0x + i] = 1;
  }
  cu.chroma_mode = chroma_mode;

  cu.IntraSplitFlag = PartMode == 1;
  cu.MaxTrafoDepth = s->max_transform_hierarchy_depth_intra + cu.IntraSplitFlag;
  transform_tree(d, &cu, x0, y0, x0, y0, log2CbSize, 0, 0, 0, 0);

  /* QpY of the CU (8.6.1): prediction plus the delta that is in force once the CU is parsed */
  set_qp_y(d);
  for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) d->m_qp[(u0y + j) * d->mw + u0x + i] = (int8_t)d->cur_qp_y;
  d->last_qp_y = d->cur_qp_y;
}

/* 7.3.8.4 coding_quadtree */
static void coding_quadtree(Dec* d, int x0, int y0, int log2CbSize, int cqtDepth)
{
  const SPS* s = d->s; const PPS* p = d->p;
  int split;
  if (x0 + (1 << log2CbSize) <= d->W && y0 + (1 << log2CbSize) <= d->H && log2CbSize > s->log2_min_cb) {
    /* 9.3.4.2.2 */
    int ctxInc = 0;
    if (available_z(d, x0, y0, x0 - 1, y0) && d->m_ctdepth[(y0 >> 2) * d->mw + ((x0 - 1) >> 2)] > cqtDepth) ctxInc++;
    if (available_z(d, x0, y0, x0, y0 - 1) && d->m_ctdepth[((y0 - 1) >> 2) * d->mw + (x0 >> 2)] > cqtDepth) ctxInc++;
    split = decode_decision(d, CTX_SPLIT_CU + ctxInc);
  } else split = log2CbSize > s->log2_min_cb;
  if (p->cu_qp_delta_enabled_flag && log2CbSize >= s->log2_ctb - p->diff_cu_qp_delta_depth) {
    d->IsCuQpDeltaCoded = 0;
    d->CuQpDeltaVal = 0;
    derive_qp_pred(d, x0, y0); /* start of a quantisation group */
  }
  if (split) {
    int x1 = x0 + (1 << (log2CbSize - 1)), y1 = y0 + (1 << (log2CbSize - 1));
    coding_quadtree(d, x0, y0, log2CbSize - 1, cqtDepth + 1);
    if (x1 < d->W) coding_quadtree(d, x1, y0, log2CbSize - 1, cqtDepth + 1);
    if (y1 < d->H) coding_quadtree(d, x0, y1, log2CbSize - 1, cqtDepth + 1);
    if (x1 < d->W && y1 < d->H) coding_quadtree(d, x1, y1, log2CbSize - 1, cqtDepth + 1);
  } else coding_unit(d, x0, y0, log2CbSize, cqtDepth);
}

/* 7.3.8.3 sao */
static void parse_sao(Dec* d, int rx, int ry)
{
  const SPS* s = d->s;
  int ctb = ry * d->ctbW + rx;
  int merge_left = 0, merge_up = 0;
  if (rx > 0) {
    int leftCtbInSliceSeg = d->CtbAddrInRs > d->sh->SliceAddrRs;
    int leftCtbInTile = d->TileId[d->CtbAddrInTs] == d->TileId[d->CtbAddrRsToTs[d->CtbAddrInRs - 1]];
    if (leftCtbInSliceSeg && leftCtbInTile) merge_left = decode_decision(d, CTX_SAO_MERGE);
  }
  if (ry > 0 && !merge_left) {
    int upCtbInSliceSeg = (d->CtbAddrInRs - d->ctbW) >= d->sh->SliceAddrRs;
    int upCtbInTile = d->TileId[d->CtbAddrInTs] == d->TileId[d->CtbAddrRsToTs[d->CtbAddrInRs - d->ctbW]];
    if (upCtbInSliceSeg && upCtbInTile) merge_up = decode_decision(d, CTX_SAO_MERGE);
  }
  if (merge_left || merge_up) {
    int src = merge_left ? ctb - 1 : ctb - d->ctbW;
    memcpy(&d->sao_type[ctb * 3], &d->sao_type[src * 3], 3);
    memcpy(&d->sao_bc[ctb * 3], &d->sao_bc[src * 3], 3);
    memcpy(&d->sao_off[ctb * 12], &d->sao_off[src * 12], 12 * sizeof(int16_t));
    return;
  }
  for (int cIdx = 0; cIdx < (s->chroma_format_idc ? 3 : 1); cIdx++) {
    int on = cIdx == 0 ? d->sh->slice_sao_luma_flag : d->sh->slice_sao_chroma_flag;
    d->sao_type[ctb * 3 + cIdx] = 0;
    d->sao_bc[ctb * 3 + cIdx] = 0;
    for (int i = 0; i < 4; i++) d->sao_off[(ctb * 3 + cIdx) * 4 + i] = 0;
    if (!on) continue;
    int type;
    if (cIdx == 2) type = d->sao_type[ctb * 3 + 1];
    else {
      type = 0;
      if (decode_decision(d, CTX_SAO_TYPE)) type = decode_bypass(d) ? 2 : 1;
    }
    d->sao_type[ctb * 3 + cIdx] = (uint8_t)type;
    if (!type) continue;
    int bitDepth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
    int cMax = (1 << (Min(bitDepth, 10) - 5)) - 1;
    int absv[4], sign[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) { int v = 0; while (v < cMax && decode_bypass(d)) v++; absv[i] = v; }
    if (type == 1) {
      for (int i = 0; i < 4; i++) if (absv[i]) sign[i] = decode_bypass(d);
      d->sao_bc[ctb * 3 + cIdx] = (uint8_t)decode_bypass_bits(d, 5);
    } else {
      if (cIdx == 0) d->sao_bc[ctb * 3] = (uint8_t)decode_bypass_bits(d, 2);
      else if (cIdx == 1) d->sao_bc[ctb * 3 + 1] = (uint8_t)decode_bypass_bits(d, 2);
      else d->sao_bc[ctb * 3 + 2] = d->sao_bc[ctb * 3 + 1];
      sign[0] = sign[1] = 0; sign[2] = sign[3] = 1;
    }
    int log2OffsetScale = bitDepth - Min(bitDepth, 10);
    for (int i = 0; i < 4; i++)
      d->sao_off[(ctb * 3 + cIdx) * 4 + i] = (int16_t)((sign[i] ? -absv[i] : absv[i]) * (1 << log2OffsetScale));
  }
}

/* ------------------------------------------------------------------------------------------ */
/* 7.3.6.1 slice_segment_header + 7.3.8.1 slice_segment_data                                   */
/* ------------------------------------------------------------------------------------------ */
/* escaped (NAL payload) index of RBSP byte r: every removed 0x03 whose following byte has
   RBSP index <= r lies before it */
static size_t escaped_pos(const size_t* epb, int n_epb, size_t r)
{
  size_t e = r;
  for (int i = 0; i < n_epb; i++) if (epb[i] - (size_t)i <= r) e = r + (size_t)i + 1;
  return e;
}
static int ceil_log2(int v) { int n = 0; while ((1 << n) < v) n++; return n; }

static void decode_slice(Dec* d, int nal_type, const uint8_t* nal, size_t nal_len)
{
  size_t rn; size_t* epb; int n_epb;
  uint8_t* rbsp = nal_to_rbsp(d, nal + 2, nal_len - 2, &rn, &epb, &n_epb);
  BR b = {rbsp, rn * 8, 0, d};
  SliceHdr hdr; memset(&hdr, 0, sizeof(hdr));
  hdr.first_slice_segment_in_pic_flag = br_u(&b, 1);
  if (nal_type >= 16 && nal_type <= 23) br_u(&b, 1); /* no_output_of_prior_pics_flag */
  int pps_id = (int)br_ue(&b);
  if (pps_id > 63 || !d->pps[pps_id].valid) fail(d, "slice refers to a missing PPS");
  const PPS* p = &d->pps[pps_id];
  if (!d->sps[p->sps_id].valid) fail(d, "PPS refers to a missing SPS");
  const SPS* s = &d->sps[p->sps_id];
  if (hdr.first_slice_segment_in_pic_flag) {
    if (d->have_picture) fail(d, "more than one picture in one access unit");
    d->s = s; d->p = p;
    setup_picture(d);
  } else {
    if (!d->have_picture) fail(d, "slice segment without a first slice segment");
    if (p != d->p) fail(d, "PPS changes inside a picture");
  }
  if (!hdr.first_slice_segment_in_pic_flag) {
    if (p->dependent_slice_segments_enabled_flag) hdr.dependent_slice_segment_flag = br_u(&b, 1);
    hdr.slice_segment_address = br_u(&b, ceil_log2(d->nCtb));
    if (hdr.slice_segment_address >= d->nCtb) fail(d, "slice_segment_address out of range");
  }
  if (hdr.dependent_slice_segment_flag) {
    if (d->nslices == 0) fail(d, "dependent slice segment without a preceding slice");
    SliceHdr prev = d->slices[d->nslices - 1];
    int addr = hdr.slice_segment_address;
    hdr = prev;
    hdr.first_slice_segment_in_pic_flag = 0;
    hdr.dependent_slice_segment_flag = 1;
    hdr.slice_segment_address = addr;
    hdr.entry_point_offset = NULL; hdr.num_entry_point_offsets = 0;
  } else {
    for (int i = 0; i < p->num_extra_slice_header_bits; i++) br_u(&b, 1);
    hdr.slice_type = (int)br_ue(&b);
    if (hdr.slice_type > 2) fail(d, "slice_type out of range");
    if (hdr.slice_type != 2 && !d->seq_mode) fail(d, "unsupported: slice_type %d (a single picture must be intra coded)", hdr.slice_type);
    if (p->output_flag_present_flag) br_u(&b, 1);
    if (s->separate_colour_plane_flag) br_u(&b, 2);
    int poc_lsb = 0, slice_temporal_mvp = 0;
    StRps rps; memset(&rps, 0, sizeof(rps));
    if (nal_type != 19 && nal_type != 20) {
      poc_lsb = (int)br_u(&b, s->log2_max_poc_lsb);
      int st_sps_flag = br_u(&b, 1);
      int ridx = 0;
      SPS scratch = *s; /* the slice-level RPS is parsed as entry num_short_term_ref_pic_sets of a copy */
      if (!st_sps_flag) {
        ridx = s->num_short_term_ref_pic_sets;
        parse_st_rps(d, &b, &scratch, ridx, s->num_short_term_ref_pic_sets);
      } else {
        if (s->num_short_term_ref_pic_sets == 0) fail(d, "short_term_ref_pic_set_sps_flag without an RPS in the SPS");
        if (s->num_short_term_ref_pic_sets > 1) ridx = (int)br_u(&b, ceil_log2(s->num_short_term_ref_pic_sets));
        if (ridx >= s->num_short_term_ref_pic_sets) fail(d, "short_term_ref_pic_set_idx out of range");
      }
      rps.num_neg = scratch.NumNegativePics[ridx]; rps.num_pos = scratch.NumPositivePics[ridx];
      for (int i = 0; i < rps.num_neg; i++) { rps.delta_s0[i] = scratch.DeltaPocS0[ridx][i]; rps.used_s0[i] = scratch.UsedS0[ridx][i]; }
      for (int i = 0; i < rps.num_pos; i++) { rps.delta_s1[i] = scratch.DeltaPocS1[ridx][i]; rps.used_s1[i] = scratch.UsedS1[ridx][i]; }
      if (s->long_term_ref_pics_present_flag) {   /* 7.3.6.1 / 7.4.7.1: PocLsbLt, UsedByCurrPicLt, DeltaPocMsbCycleLt (7-52) */
        int num_lt_sps = 0;
        if (s->num_long_term_ref_pics_sps > 0) num_lt_sps = (int)br_ue(&b);
        int num_lt_pics = (int)br_ue(&b);
        if (num_lt_sps > s->num_long_term_ref_pics_sps || num_lt_sps + num_lt_pics > 32) fail(d, "num_long_term_sps / num_long_term_pics out of range");
        rps.num_lt = num_lt_sps + num_lt_pics;
        for (int i = 0; i < rps.num_lt; i++) {
          if (i < num_lt_sps) {
            int lt_idx = s->num_long_term_ref_pics_sps > 1 ? (int)br_u(&b, ceil_log2(s->num_long_term_ref_pics_sps)) : 0;
            if (lt_idx >= s->num_long_term_ref_pics_sps) fail(d, "lt_idx_sps out of range");
            rps.lt_poc_lsb[i] = s->lt_ref_pic_poc_lsb_sps[lt_idx]; rps.lt_used[i] = s->used_by_curr_pic_lt_sps_flag[lt_idx];
          } else { rps.lt_poc_lsb[i] = (int)br_u(&b, s->log2_max_poc_lsb); rps.lt_used[i] = (uint8_t)br_u(&b, 1); }
          rps.lt_msb_present[i] = (uint8_t)br_u(&b, 1);
          int cycle = rps.lt_msb_present[i] ? (int)br_ue(&b) : 0;
          rps.lt_msb_cycle[i] = (i == 0 || i == num_lt_sps) ? cycle : cycle + rps.lt_msb_cycle[i - 1];
        }
      }
      if (s->sps_temporal_mvp_enabled_flag) slice_temporal_mvp = br_u(&b, 1);
    }
    if (hdr.first_slice_segment_in_pic_flag && d->seq_mode) inter_begin_picture(d, nal_type, (nal[1] & 7) - 1, poc_lsb, &rps);
    if (s->sao_enabled_flag) {
      hdr.slice_sao_luma_flag = br_u(&b, 1);
      if (s->chroma_format_idc) hdr.slice_sao_chroma_flag = br_u(&b, 1);
    }
    if (hdr.slice_type != 2) {   /* 7.3.6.1, P / B slice */
      const int is_b = hdr.slice_type == 0;
      hdr.slice_temporal_mvp = slice_temporal_mvp;
      hdr.num_ref_idx_l0_active = p->num_ref_idx_l0_default_active;
      hdr.num_ref_idx_l1_active = is_b ? p->num_ref_idx_l1_default_active : 0;
      if (br_u(&b, 1)) {   /* num_ref_idx_active_override_flag */
        hdr.num_ref_idx_l0_active = (int)br_ue(&b) + 1;
        if (is_b) hdr.num_ref_idx_l1_active = (int)br_ue(&b) + 1;
      }
      if (hdr.num_ref_idx_l0_active > 15 || hdr.num_ref_idx_l1_active > 15) fail(d, "num_ref_idx_lX_active_minus1 out of range");
      int total = d->n_st_curr_before + d->n_st_curr_after + d->n_lt_curr, entries[2][16], modified[2] = {0, 0};
      if (p->lists_modification_present_flag && total > 1)
        for (int X = 0; X < (is_b ? 2 : 1); X++) {
          modified[X] = br_u(&b, 1);   /* ref_pic_list_modification_flag_lX */
          if (modified[X]) for (int i = 0; i < (X ? hdr.num_ref_idx_l1_active : hdr.num_ref_idx_l0_active); i++) entries[X][i] = (int)br_u(&b, ceil_log2(total));
        }
      if (is_b) hdr.mvd_l1_zero_flag = br_u(&b, 1);
      if (p->cabac_init_present_flag) hdr.cabac_init_flag = br_u(&b, 1);
      hdr.collocated_from_l0 = 1; hdr.collocated_ref_idx = 0;
      if (slice_temporal_mvp) {
        if (is_b) hdr.collocated_from_l0 = br_u(&b, 1);
        if ((hdr.collocated_from_l0 && hdr.num_ref_idx_l0_active > 1) || (!hdr.collocated_from_l0 && hdr.num_ref_idx_l1_active > 1))
          hdr.collocated_ref_idx = (int)br_ue(&b);
        if (hdr.collocated_ref_idx >= (hdr.collocated_from_l0 ? hdr.num_ref_idx_l0_active : hdr.num_ref_idx_l1_active)) fail(d, "collocated_ref_idx out of range");
      }
      build_ref_list(d, &hdr, 0, modified[0] ? entries[0] : NULL);
      if (is_b) build_ref_list(d, &hdr, 1, modified[1] ? entries[1] : NULL);
      hdr.weighted = is_b ? p->weighted_bipred_flag : p->weighted_pred_flag;
      if (hdr.weighted) {   /* 7.3.6.3 pred_weight_table */
        int nc = s->chroma_format_idc ? 3 : 1;
        hdr.luma_log2_wd = (int)br_ue(&b);
        if (hdr.luma_log2_wd > 7) fail(d, "luma_log2_weight_denom out of range");
        hdr.chroma_log2_wd = hdr.luma_log2_wd;
        if (nc == 3) { hdr.chroma_log2_wd += br_se(&b); if (hdr.chroma_log2_wd < 0 || hdr.chroma_log2_wd > 7) fail(d, "delta_chroma_log2_weight_denom out of range"); }
        for (int X = 0; X < (is_b ? 2 : 1); X++) {
          int n = X ? hdr.num_ref_idx_l1_active : hdr.num_ref_idx_l0_active;
          uint8_t lf[16], cf[16];
          memset(cf, 0, sizeof(cf));
          /* (the flags are present for every entry: a reference picture of these single-layer streams never has the current picture's POC) */
          for (int i = 0; i < n; i++) lf[i] = (uint8_t)br_u(&b, 1);
          if (nc == 3) for (int i = 0; i < n; i++) cf[i] = (uint8_t)br_u(&b, 1);
          for (int i = 0; i < n; i++) {
            hdr.wp_weight[X][i][0] = (int16_t)(1 << hdr.luma_log2_wd); hdr.wp_offset[X][i][0] = 0;
            hdr.wp_weight[X][i][1] = hdr.wp_weight[X][i][2] = (int16_t)(1 << hdr.chroma_log2_wd); hdr.wp_offset[X][i][1] = hdr.wp_offset[X][i][2] = 0;
            if (lf[i]) {
              int dw = br_se(&b), o = br_se(&b);
              if (dw < -128 || dw > 127 || o < -128 || o > 127) fail(d, "luma weight / offset out of range");
              hdr.wp_weight[X][i][0] = (int16_t)((1 << hdr.luma_log2_wd) + dw); hdr.wp_offset[X][i][0] = (int16_t)o;
            }
            if (cf[i])
              for (int j = 1; j < 3; j++) {
                int dw = br_se(&b), dof = br_se(&b);
                if (dw < -128 || dw > 127 || dof < -512 || dof > 511) fail(d, "chroma weight / offset out of range");
                int wgt = (1 << hdr.chroma_log2_wd) + dw;
                hdr.wp_weight[X][i][j] = (int16_t)wgt;
                hdr.wp_offset[X][i][j] = (int16_t)Clip3(-128, 127, (128 + dof - ((128 * wgt) >> hdr.chroma_log2_wd)));
              }
          }
        }
      }
      hdr.max_num_merge_cand = 5 - (int)br_ue(&b);
      if (hdr.max_num_merge_cand < 1 || hdr.max_num_merge_cand > 5) fail(d, "five_minus_max_num_merge_cand out of range");
    }
    hdr.slice_qp_delta = br_se(&b);
    if (p->pps_slice_chroma_qp_offsets_present_flag) { hdr.slice_cb_qp_offset = br_se(&b); hdr.slice_cr_qp_offset = br_se(&b); }
    int override = 0;
    if (p->deblocking_filter_override_enabled_flag) override = br_u(&b, 1);
    hdr.slice_deblocking_filter_disabled_flag = p->pps_deblocking_filter_disabled_flag;
    hdr.slice_beta_offset_div2 = p->pps_beta_offset_div2;
    hdr.slice_tc_offset_div2 = p->pps_tc_offset_div2;
    if (override) {
      hdr.slice_deblocking_filter_disabled_flag = br_u(&b, 1);
      if (!hdr.slice_deblocking_filter_disabled_flag) { hdr.slice_beta_offset_div2 = br_se(&b); hdr.slice_tc_offset_div2 = br_se(&b); }
    }
    hdr.slice_loop_filter_across_slices_enabled_flag = p->pps_loop_filter_across_slices_enabled_flag;
    if (p->pps_loop_filter_across_slices_enabled_flag &&
        (hdr.slice_sao_luma_flag || hdr.slice_sao_chroma_flag || !hdr.slice_deblocking_filter_disabled_flag))
      hdr.slice_loop_filter_across_slices_enabled_flag = br_u(&b, 1);
    hdr.SliceAddrRs = hdr.slice_segment_address; /* 7.4.7.1 */
    hdr.SliceQpY = 26 + p->init_qp_minus26 + hdr.slice_qp_delta;
  }
  if (p->tiles_enabled_flag || p->entropy_coding_sync_enabled_flag) {
    hdr.num_entry_point_offsets = (int)br_ue(&b);
    if (hdr.num_entry_point_offsets > d->nCtb) fail(d, "too many entry points");
    if (hdr.num_entry_point_offsets > 0) {
      int len = (int)br_ue(&b) + 1;
      if (len > 32) fail(d, "offset_len_minus1 out of range");
      hdr.entry_point_offset = (uint32_t*)xcalloc(d, hdr.num_entry_point_offsets, sizeof(uint32_t));
      for (int i = 0; i < hdr.num_entry_point_offsets; i++) hdr.entry_point_offset[i] = br_u(&b, len) + 1;
    }
  }
  if (p->slice_segment_header_extension_present_flag) {
    int len = (int)br_ue(&b);
    br_skip(&b, (size_t)len * 8);
  }
  /* byte_alignment() 7.3.2.5 */
  if (br_u(&b, 1) != 1) fail(d, "slice header: alignment bit is not 1");
  while (b.pos & 7) if (br_u(&b, 1)) fail(d, "slice header: alignment zero bit is 1");

  if (d->nslices == d->capslices) {
    d->capslices = d->capslices ? d->capslices * 2 : 8;
    d->slices = (SliceHdr*)realloc(d->slices, sizeof(SliceHdr) * d->capslices);
    if (!d->slices) fail(d, "out of memory");
  }
  d->slices[d->nslices] = hdr;
  d->sh = &d->slices[d->nslices];
  d->sh_idx = d->nslices;
  d->nslices++;

  /* ---- slice_segment_data ---- */
  size_t data_byte0 = b.pos >> 3;
  d->c.data = rbsp; d->c.nbits = rn * 8; d->c.pos = b.pos;
  int CtbSizeY = 1 << s->log2_ctb;
  d->CtbAddrInTs = d->CtbAddrRsToTs[d->sh->slice_segment_address];
  d->CtbAddrInRs = d->sh->slice_segment_address;
  int entry_idx = 0;
  size_t substream_start_nal = 0; /* NAL-byte position (relative to slice data start) of current substream */
  /* NAL byte offset of the first slice data byte: rbsp offset + number of EPBs before it */
  size_t data_nal0 = escaped_pos(epb, n_epb, data_byte0);
  int first_ctb_in_segment = 1;
  int end_of_slice_segment_flag = 0;
  do {
    int xCtb = (d->CtbAddrInRs % d->ctbW) << s->log2_ctb;
    int yCtb = (d->CtbAddrInRs / d->ctbW) << s->log2_ctb;
    int tile_first = (d->CtbAddrInTs == 0) || d->TileId[d->CtbAddrInTs] != d->TileId[d->CtbAddrInTs - 1];
    int row_first = 0;
    if (p->entropy_coding_sync_enabled_flag) {
      row_first = (d->CtbAddrInRs % d->ctbW == 0) ||
                  d->TileId[d->CtbAddrInTs] != d->TileId[d->CtbAddrRsToTs[d->CtbAddrInRs - 1]];
    }
    if (d->ctb_slice_addr[d->CtbAddrInRs] >= 0) fail(d, "CTB decoded twice");
    d->ctb_slice_addr[d->CtbAddrInRs] = d->sh->SliceAddrRs;
    d->ctb_slice_idx[d->CtbAddrInRs] = d->sh_idx;

    /* 9.3.1 initialisation / synchronisation */
    if (first_ctb_in_segment || tile_first || row_first) {
      if (first_ctb_in_segment) cabac_init_engine(d);
      if (tile_first) cabac_init_contexts(d);
      else if (row_first) {
        int xNbT = xCtb + CtbSizeY, yNbT = yCtb - CtbSizeY;
        if (available_z(d, xCtb, yCtb, xNbT, yNbT)) memcpy(d->c.ctx, d->ctx_wpp, MAXCTX);
        else cabac_init_contexts(d);
      } else if (d->sh->dependent_slice_segment_flag) {
        if (!d->ctx_ds_valid) fail(d, "dependent slice segment without stored contexts");
        memcpy(d->c.ctx, d->ctx_ds, MAXCTX);
      } else cabac_init_contexts(d);
      /* 8.6.1: first quantisation group in a slice, tile, or CTB row (WPP) */
      if ((first_ctb_in_segment && !d->sh->dependent_slice_segment_flag) || tile_first || row_first)
        d->last_qp_y = d->sh->SliceQpY;
      first_ctb_in_segment = 0;
    }

    /* 7.3.8.2 coding_tree_unit */
    if (d->sh->slice_sao_luma_flag || d->sh->slice_sao_chroma_flag) parse_sao(d, xCtb >> s->log2_ctb, yCtb >> s->log2_ctb);
    if (!p->cu_qp_delta_enabled_flag) { /* QP prediction still needs a quantisation-group start */
      d->IsCuQpDeltaCoded = 0; d->CuQpDeltaVal = 0;
    }
    if (!p->cu_qp_delta_enabled_flag) derive_qp_pred(d, xCtb, yCtb);
    coding_quadtree(d, xCtb, yCtb, s->log2_ctb, 0);

    end_of_slice_segment_flag = decode_terminate(d);
    /* 9.3.2.2 storage for WPP after the second CTB of a row (of a tile) */
    if (p->entropy_coding_sync_enabled_flag) {
      /* general form: the CTB to the left is the first CTB of a row in this tile */
      int left_is_row_first = 0;
      if (d->CtbAddrInRs % d->ctbW >= 1) {
        int leftRs = d->CtbAddrInRs - 1;
        int leftTs = d->CtbAddrRsToTs[leftRs];
        if (d->TileId[leftTs] == d->TileId[d->CtbAddrInTs])
          left_is_row_first = (leftRs % d->ctbW == 0) || d->TileId[leftTs] != d->TileId[d->CtbAddrRsToTs[leftRs - 1]];
      }
      if (left_is_row_first) memcpy(d->ctx_wpp, d->c.ctx, MAXCTX);
      /* a tile (or picture) that is one CTB wide: the spec stores after "CtbAddrInRs % W == 0 ..."
         only in later editions; with v1 semantics nothing is stored and the next row re-initialises
         because its top-right neighbour is unavailable */
    }
    d->CtbAddrInTs++;
    if (!end_of_slice_segment_flag) {
      if (d->CtbAddrInTs >= d->nCtb) fail(d, "slice data continues past the last CTB");
      d->CtbAddrInRs = d->CtbAddrTsToRs[d->CtbAddrInTs];
      int new_tile = p->tiles_enabled_flag && d->TileId[d->CtbAddrInTs] != d->TileId[d->CtbAddrInTs - 1];
      int new_row = p->entropy_coding_sync_enabled_flag &&
                    (d->CtbAddrInRs % d->ctbW == 0 || d->TileId[d->CtbAddrInTs] != d->TileId[d->CtbAddrRsToTs[d->CtbAddrInRs - 1]]);
      if (new_tile || new_row) {
        if (!decode_terminate(d)) fail(d, "end_of_subset_one_bit is not 1");
        cabac_finish_and_align(d);
        /* verify against the signalled entry point (pins the CABAC parse to the encoder's layout) */
        if (entry_idx >= d->sh->num_entry_point_offsets) fail(d, "substream boundary without an entry point");
        size_t rbsp_byte = d->c.pos >> 3;
        size_t nal_byte = escaped_pos(epb, n_epb, rbsp_byte);
        size_t expect = data_nal0 + substream_start_nal + d->sh->entry_point_offset[entry_idx];
        if (nal_byte != expect)
          fail(d, "substream %d ends at NAL byte %zu but entry point says %zu", entry_idx, nal_byte, expect);
        substream_start_nal += d->sh->entry_point_offset[entry_idx];
        entry_idx++;
        cabac_init_engine(d);
      }
    }
  } while (!end_of_slice_segment_flag);
  cabac_finish_and_align(d);
  if (entry_idx != d->sh->num_entry_point_offsets) fail(d, "unused entry points in slice segment");
  /* what remains must be cabac_zero_words (0x0000) only */
  for (size_t i = d->c.pos >> 3; i < rn; i++) if (rbsp[i] != 0) fail(d, "garbage after slice segment data");
  if (p->dependent_slice_segments_enabled_flag) { memcpy(d->ctx_ds, d->c.ctx, MAXCTX); d->ctx_ds_valid = 1; }
  free(rbsp); free(epb);
}

/* ------------------------------------------------------------------------------------------ */
/* 8.7.2 deblocking filter                                                                    */
/* ------------------------------------------------------------------------------------------ */
static void deblock_luma_edge(Dec* d, uint16_t* pix, int xstep, int ystep, int QpP, int QpQ,
                              const SliceHdr* sh, int noP, int noQ, int bS)
{
  /* pix points at q0 of line 0; p_i = pix[-(i+1)*xstep], q_i = pix[i*xstep]; lines advance by ystep */
  int bitDepth = d->s->bit_depth_luma;
  int qPL = (QpQ + QpP + 1) >> 1;
  int Q = Clip3(0, 51, qPL + (sh->slice_beta_offset_div2 << 1));
  int beta = betaTable[Q] * (1 << (bitDepth - 8));
  Q = Clip3(0, 53, qPL + 2 * (bS - 1) + (sh->slice_tc_offset_div2 << 1));
  int tC = tcTable[Q] * (1 << (bitDepth - 8));
#define P(i, k) ((int)pix[-((i) + 1) * xstep + (k) * ystep])
#define QQ(i, k) ((int)pix[(i) * xstep + (k) * ystep])
  int dp0 = Abs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = Abs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
  int dq0 = Abs(QQ(2, 0) - 2 * QQ(1, 0) + QQ(0, 0)), dq3 = Abs(QQ(2, 3) - 2 * QQ(1, 3) + QQ(0, 3));
  int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, dd = dpq0 + dpq3;
  int dE = 0, dEp = 0, dEq = 0;
  if (dd < beta) {
    int dSam0 = (2 * dpq0 < (beta >> 2)) && (Abs(P(3, 0) - P(0, 0)) + Abs(QQ(0, 0) - QQ(3, 0)) < (beta >> 3)) &&
                (Abs(P(0, 0) - QQ(0, 0)) < ((5 * tC + 1) >> 1));
    int dSam3 = (2 * dpq3 < (beta >> 2)) && (Abs(P(3, 3) - P(0, 3)) + Abs(QQ(0, 3) - QQ(3, 3)) < (beta >> 3)) &&
                (Abs(P(0, 3) - QQ(0, 3)) < ((5 * tC + 1) >> 1));
    dE = 1;
    if (dSam0 && dSam3) dE = 2;
    if (dp < ((beta + (beta >> 1)) >> 3)) dEp = 1;
    if (dq < ((beta + (beta >> 1)) >> 3)) dEq = 1;
  }
  if (!dE) return;
  int maxv = (1 << bitDepth) - 1;
  for (int k = 0; k < 4; k++) {
    int p0 = P(0, k), p1 = P(1, k), p2 = P(2, k), p3 = P(3, k);
    int q0 = QQ(0, k), q1 = QQ(1, k), q2 = QQ(2, k), q3 = QQ(3, k);
    uint16_t* l = pix + k * ystep;
    if (dE == 2) {
      if (!noP) {
        l[-1 * xstep] = (uint16_t)Clip3(p0 - 2 * tC, p0 + 2 * tC, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        l[-2 * xstep] = (uint16_t)Clip3(p1 - 2 * tC, p1 + 2 * tC, (p2 + p1 + p0 + q0 + 2) >> 2);
        l[-3 * xstep] = (uint16_t)Clip3(p2 - 2 * tC, p2 + 2 * tC, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      }
      if (!noQ) {
        l[0] = (uint16_t)Clip3(q0 - 2 * tC, q0 + 2 * tC, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        l[1 * xstep] = (uint16_t)Clip3(q1 - 2 * tC, q1 + 2 * tC, (p0 + q0 + q1 + q2 + 2) >> 2);
        l[2 * xstep] = (uint16_t)Clip3(q2 - 2 * tC, q2 + 2 * tC, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
      }
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (Abs(delta) < tC * 10) {
        delta = Clip3(-tC, tC, delta);
        if (!noP) l[-1 * xstep] = (uint16_t)Clip3(0, maxv, p0 + delta);
        if (!noQ) l[0] = (uint16_t)Clip3(0, maxv, q0 - delta);
        if (dEp && !noP) {
          int dP = Clip3(-(tC >> 1), tC >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1);
          l[-2 * xstep] = (uint16_t)Clip3(0, maxv, p1 + dP);
        }
        if (dEq && !noQ) {
          int dQ = Clip3(-(tC >> 1), tC >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1);
          l[1 * xstep] = (uint16_t)Clip3(0, maxv, q1 + dQ);
        }
      }
    }
  }
#undef P
#undef QQ
}

static void deblock_chroma_edge_n(Dec* d, uint16_t* pix, int xstep, int ystep, int QpP, int QpQ,
                                  int cQpPicOffset, const SliceHdr* sh, int noP, int noQ, int nlines);
static void deblock_chroma_edge(Dec* d, uint16_t* pix, int xstep, int ystep, int QpP, int QpQ,
                                int cQpPicOffset, const SliceHdr* sh, int noP, int noQ)
{
  deblock_chroma_edge_n(d, pix, xstep, ystep, QpP, QpQ, cQpPicOffset, sh, noP, noQ, 4);
}
static void deblock_chroma_edge_n(Dec* d, uint16_t* pix, int xstep, int ystep, int QpP, int QpQ,
                                  int cQpPicOffset, const SliceHdr* sh, int noP, int noQ, int nlines)
{
  int bitDepth = d->s->bit_depth_chroma;
  int qPi = ((QpQ + QpP + 1) >> 1) + cQpPicOffset;
  int QpC = d->s->chroma_format_idc == 1 ? hevc_chroma_qp_420(qPi) : Min(qPi, 51);   /* 8.7.2.5.5 */
  int Q = Clip3(0, 53, QpC + 2 * (2 - 1) + (sh->slice_tc_offset_div2 << 1));
  int tC = tcTable[Q] * (1 << (bitDepth - 8));
  int maxv = (1 << bitDepth) - 1;
  for (int k = 0; k < nlines; k++) {
    uint16_t* l = pix + k * ystep;
    int p0 = l[-xstep], p1 = l[-2 * xstep], q0 = l[0], q1 = l[xstep];
    int delta = Clip3(-tC, tC, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
    if (!noP) l[-xstep] = (uint16_t)Clip3(0, maxv, p0 + delta);
    if (!noQ) l[0] = (uint16_t)Clip3(0, maxv, q0 - delta);
  }
}

static int unit_no_filter(Dec* d, int idx)
{
  uint8_t f = d->m_flags[idx];
  if (f & 0x08) return 1;                                        /* cu_transquant_bypass */
  if ((f & 0x10) && d->s->pcm_loop_filter_disabled_flag) return 1; /* pcm + pcm_loop_filter_disabled */
  return 0;
}

static void deblock_picture(Dec* d)
{
  const SPS* s = d->s; const PPS* p = d->p;
  for (int dir = 0; dir < 2; dir++) { /* 0: vertical edges, 1: horizontal edges */
    for (int uy = 0; uy < d->mh; uy++)
      for (int ux = 0; ux < d->mw; ux++) {
        int idx = uy * d->mw + ux;
        int x = ux * 4, y = uy * 4;
        if (dir == 0) { if (!(d->m_flags[idx] & 0x20) || (x & 7)) continue; }
        else { if (!(d->m_flags[idx] & 0x40) || (y & 7)) continue; }
        int idxP = dir == 0 ? idx - 1 : idx - d->mw;
        int QpQ = d->m_qp[idx], QpP = d->m_qp[idxP];
        int ctb = (y >> s->log2_ctb) * d->ctbW + (x >> s->log2_ctb);
        const SliceHdr* sh = &d->slices[d->ctb_slice_idx[ctb]];
        int noP = unit_no_filter(d, idxP), noQ = unit_no_filter(d, idx);
        int bS = edge_bs(d, idxP, idx, x, y, dir);   /* 8.7.2.4: 2 where a side is intra coded (every edge of an intra picture) */
        if (!bS) continue;
        if (dir == 0) deblock_luma_edge(d, d->rec[0] + y * d->W + x, 1, d->W, QpP, QpQ, sh, noP, noQ, bS);
        else deblock_luma_edge(d, d->rec[0] + y * d->W + x, d->W, 1, QpP, QpQ, sh, noP, noQ, bS);
        if (bS != 2) continue;                        /* chroma edges are filtered where bS is 2 only (8.7.2.5.? edge filtering process) */
        if (s->chroma_format_idc == 3) {
          /* 8.7.2: with ChromaArrayType 3 the chroma planes have the luma planes' edges (the 8-sample grid in chroma samples IS the luma
             grid), filtered with the chroma filter */
          for (int c = 1; c < 3; c++)
            deblock_chroma_edge(d, d->rec[c] + y * d->Wc + x, dir == 0 ? 1 : d->Wc, dir == 0 ? d->Wc : 1, QpP, QpQ,
                                c == 1 ? p->pps_cb_qp_offset : p->pps_cr_qp_offset, sh, noP, noQ);
        }
        if (s->chroma_format_idc == 2) {
          /* 4:2:2: the 8x8 chroma sample grid is 16 luma samples wide and 8 tall.  A vertical edge segment of 4 luma rows is 4 chroma rows; a
             horizontal one of 4 luma columns is 2 chroma columns */
          if (dir == 0 ? (x & 15) == 0 : 1)
            for (int c = 1; c < 3; c++)
              deblock_chroma_edge_n(d, d->rec[c] + y * d->Wc + x / 2, dir == 0 ? 1 : d->Wc, dir == 0 ? d->Wc : 1, QpP, QpQ,
                                    c == 1 ? p->pps_cb_qp_offset : p->pps_cr_qp_offset, sh, noP, noQ, dir == 0 ? 4 : 2);
        }
        if (s->chroma_format_idc == 1) {
          /* chroma edges lie on the 8x8 chroma sample grid; a 4-row chroma segment corresponds to
             8 luma rows and takes its bS from the first 4-luma-row segment (8.7.2.5.?) */
          if (dir == 0) {
            if ((x & 15) || (y & 7)) continue;
            for (int c = 1; c < 3; c++)
              deblock_chroma_edge(d, d->rec[c] + (y / 2) * d->Wc + x / 2, 1, d->Wc, QpP, QpQ,
                                  c == 1 ? p->pps_cb_qp_offset : p->pps_cr_qp_offset, sh, noP, noQ);
          } else {
            if ((y & 15) || (x & 7)) continue;
            for (int c = 1; c < 3; c++)
              deblock_chroma_edge(d, d->rec[c] + (y / 2) * d->Wc + x / 2, d->Wc, 1, QpP, QpQ,
                                  c == 1 ? p->pps_cb_qp_offset : p->pps_cr_qp_offset, sh, noP, noQ);
          }
        }
      }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* 8.7.3 sample adaptive offset                                                               */
/* ------------------------------------------------------------------------------------------ */
static void sao_picture(Dec* d, uint16_t* const src[3], uint16_t* dst[3])
{
  const SPS* s = d->s; const PPS* p = d->p;
  int lm = s->log2_min_tb;
  for (int cIdx = 0; cIdx < (s->chroma_format_idc ? 3 : 1); cIdx++) {
    int W = cIdx ? d->Wc : d->W, H = cIdx ? d->Hc : d->H, sub = cIdx ? d->subw : 1, subv = cIdx ? d->subh : 1;
    int bitDepth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
    int maxv = (1 << bitDepth) - 1;
    int ctbSize = (1 << s->log2_ctb) / sub, ctbSizeV = (1 << s->log2_ctb) / subv;
    memcpy(dst[cIdx], src[cIdx], sizeof(uint16_t) * W * H);
    for (int ry = 0; ry < d->ctbH; ry++)
      for (int rx = 0; rx < d->ctbW; rx++) {
        int ctb = ry * d->ctbW + rx;
        int type = d->sao_type[ctb * 3 + cIdx];
        if (!type) continue;
        const int16_t* off = &d->sao_off[(ctb * 3 + cIdx) * 4];
        int bc = d->sao_bc[ctb * 3 + cIdx];
        const SliceHdr* shC = &d->slices[d->ctb_slice_idx[ctb]];
        for (int y = ry * ctbSizeV; y < Min((ry + 1) * ctbSizeV, H); y++)
          for (int x = rx * ctbSize; x < Min((rx + 1) * ctbSize, W); x++) {
            int uidx = ((y * subv) >> 2) * d->mw + ((x * sub) >> 2);
            if (unit_no_filter(d, uidx)) continue;
            int v = src[cIdx][y * W + x];
            if (type == 1) {
              int bandShift = bitDepth - 5;
              int k = ((v >> bandShift) - bc) & 31;
              if (k < 4) dst[cIdx][y * W + x] = (uint16_t)Clip3(0, maxv, v + off[k]);
            } else {
              static const int8_t hPos[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}};
              static const int8_t vPos[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
              int edgeIdx = 2, skip = 0;
              for (int k = 0; k < 2; k++) {
                int xs = x + hPos[bc][k], ys = y + vPos[bc][k];
                if (xs < 0 || ys < 0 || xs >= W || ys >= H) { skip = 1; break; }
                int ctbN = ((ys * subv) >> s->log2_ctb) * d->ctbW + ((xs * sub) >> s->log2_ctb);
                if (ctbN != ctb) {
                  if (d->ctb_slice_addr[ctbN] != d->ctb_slice_addr[ctb]) {
                    int zN = d->MinTbAddrZs[((ys * subv) >> lm) * d->minTbW + ((xs * sub) >> lm)];
                    int zC = d->MinTbAddrZs[((y * subv) >> lm) * d->minTbW + ((x * sub) >> lm)];
                    const SliceHdr* shN = &d->slices[d->ctb_slice_idx[ctbN]];
                    if (zN < zC && !shC->slice_loop_filter_across_slices_enabled_flag) { skip = 1; break; }
                    if (zC < zN && !shN->slice_loop_filter_across_slices_enabled_flag) { skip = 1; break; }
                  }
                  if (!p->loop_filter_across_tiles_enabled_flag &&
                      d->TileId[d->CtbAddrRsToTs[ctbN]] != d->TileId[d->CtbAddrRsToTs[ctb]]) { skip = 1; break; }
                }
                int nv = src[cIdx][ys * W + xs];
                edgeIdx += (v > nv) - (v < nv);
              }
              if (skip) continue;
              if (edgeIdx == 0 || edgeIdx == 1 || edgeIdx == 2) edgeIdx = (edgeIdx == 2) ? 0 : edgeIdx + 1;
              if (edgeIdx) dst[cIdx][y * W + x] = (uint16_t)Clip3(0, maxv, v + off[edgeIdx - 1]);
            }
          }
      }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* top level                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static uint16_t* dup_plane(Dec* d, const uint16_t* src, size_t n)
{
  uint16_t* r = (uint16_t*)xcalloc(d, n, sizeof(uint16_t));
  memcpy(r, src, n * sizeof(uint16_t));
  return r;
}

/* everything that belongs to ONE picture (a sequence decoder keeps the parameter sets, the DPB and the POC state) */
static void release_picture(Dec* d)
{
  for (int c = 0; c < 3; c++) { free(d->rec[c]); free(d->coeff[c]); d->rec[c] = NULL; d->coeff[c] = NULL; }
  free(d->m_log2_tb); free(d->m_log2_cb); free(d->m_ipm); free(d->m_ipmc); free(d->m_flags);
  free(d->m_ctdepth); free(d->m_qp); free(d->m_decoded);
  d->m_log2_tb = d->m_log2_cb = d->m_ipm = d->m_ipmc = d->m_flags = d->m_ctdepth = d->m_decoded = NULL; d->m_qp = NULL;
  free(d->m_pred); free(d->mf_mv); free(d->mf_ref); free(d->mf_poc); free(d->mf_lt);
  d->m_pred = NULL; d->mf_mv = NULL; d->mf_ref = NULL; d->mf_poc = NULL; d->mf_lt = NULL;
  free(d->CtbAddrRsToTs); free(d->CtbAddrTsToRs); free(d->TileId); free(d->colBd); free(d->rowBd);
  free(d->MinTbAddrZs); free(d->ctb_slice_addr); free(d->ctb_slice_idx);
  d->CtbAddrRsToTs = d->CtbAddrTsToRs = d->TileId = d->colBd = d->rowBd = d->MinTbAddrZs = d->ctb_slice_addr = d->ctb_slice_idx = NULL;
  for (int i = 0; i < d->nslices; i++) free(d->slices[i].entry_point_offset);
  free(d->slices); d->slices = NULL; d->nslices = d->capslices = 0; d->sh = NULL;
  free(d->sao_type); free(d->sao_bc); free(d->sao_off);
  d->sao_type = d->sao_bc = NULL; d->sao_off = NULL;
  d->have_picture = 0; d->ctx_ds_valid = 0;
  d->n_bins_ctx = d->n_bins_bypass = 0; d->n_substreams = 0;
}

static void free_dec(Dec* d)
{
  release_picture(d);
  for (int i = 0; i < MAX_DPB; i++) dpb_free_entry(&d->dpb[i]);
  free(d);
}

/* one access unit in libheif's plugin framing -> *out; in sequence mode the decoded picture also enters the DPB.  Errors longjmp to d->jb. */
static void decode_access_unit(Dec* d, const uint8_t* data, size_t size, int keep_taps, hevc_oracle_picture* out)
{
  d->keep_taps = keep_taps;
  /* NAL framing: [u32 BE length][NAL] ... (decoder_libde265.cc:322-368) */
  size_t ptr = 0;
  while (ptr < size) {
    if (size - ptr < 4) fail(d, "truncated NAL length field");
    uint32_t nal_size = ((uint32_t)data[ptr] << 24) | ((uint32_t)data[ptr + 1] << 16) | ((uint32_t)data[ptr + 2] << 8) | data[ptr + 3];
    ptr += 4;
    if (nal_size > size - ptr) fail(d, "NAL size exceeds the data");
    const uint8_t* nal = data + ptr;
    ptr += nal_size;
    if (nal_size < 2) continue;
    int nal_type = (nal[0] >> 1) & 63;
    if (nal_type == 33 || nal_type == 34) {
      size_t rn; size_t* epb; int n_epb;
      uint8_t* rbsp = nal_to_rbsp(d, nal + 2, nal_size - 2, &rn, &epb, &n_epb);
      if (nal_type == 33) parse_sps(d, rbsp, rn); else parse_pps(d, rbsp, rn);
      free(rbsp); free(epb);
    } else if (nal_type <= 21) {
      if (nal_type > 9 && nal_type < 16) continue; /* reserved */
      decode_slice(d, nal_type, nal, nal_size);
    }
    /* VPS (32), AUD, SEI, EOS ...: nothing to do */
  }
  if (!d->have_picture) fail(d, "no picture in the data");
  for (int i = 0; i < d->nCtb; i++) if (d->ctb_slice_addr[i] < 0) fail(d, "picture is incomplete (CTB %d missing)", i);

  const SPS* s = d->s;
  int nc = s->chroma_format_idc ? 3 : 1;
  out->coded_width = d->W; out->coded_height = d->H; out->ccoded_width = d->Wc; out->ccoded_height = d->Hc;
  out->chroma_format_idc = s->chroma_format_idc;
  out->bit_depth_luma = s->bit_depth_luma; out->bit_depth_chroma = s->bit_depth_chroma;
  out->colour_primaries = s->colour_primaries; out->transfer_characteristics = s->transfer_characteristics;
  out->matrix_coeffs = s->matrix_coeffs; out->full_range_flag = s->video_full_range_flag;
  out->n_bins_ctx = d->n_bins_ctx; out->n_bins_bypass = d->n_bins_bypass; out->n_substreams = d->n_substreams;
  out->poc = d->poc;
  if (keep_taps) for (int c = 0; c < nc; c++) out->pre_deblock[c] = dup_plane(d, d->rec[c], c ? (size_t)d->Wc * d->Hc : (size_t)d->W * d->H);
  deblock_picture(d);
  if (keep_taps) for (int c = 0; c < nc; c++) out->post_deblock[c] = dup_plane(d, d->rec[c], c ? (size_t)d->Wc * d->Hc : (size_t)d->W * d->H);
  uint16_t* fin[3] = {0, 0, 0};
  for (int c = 0; c < nc; c++) fin[c] = (uint16_t*)xcalloc(d, c ? (size_t)d->Wc * d->Hc : (size_t)d->W * d->H, sizeof(uint16_t));
  sao_picture(d, d->rec, fin);
  /* conformance window crop (7.4.3.2.1; hevc_boxes.cc:688-716) */
  int sw = s->chroma_format_idc ? d->subw : 1, shh = s->chroma_format_idc ? d->subh : 1;
  int x0 = sw * s->conf_win_left, x1 = d->W - sw * s->conf_win_right;
  int y0 = shh * s->conf_win_top, y1 = d->H - shh * s->conf_win_bottom;
  if (x1 <= x0 || y1 <= y0) fail(d, "empty conformance window");
  out->width = x1 - x0; out->height = y1 - y0;
  out->cwidth = s->chroma_format_idc ? out->width / sw : 0;
  out->cheight = s->chroma_format_idc ? out->height / shh : 0;
  for (int c = 0; c < nc; c++) {
    int w = c ? out->cwidth : out->width, h = c ? out->cheight : out->height;
    int xs = c ? x0 / sw : x0, ys = c ? y0 / shh : y0, st = c ? d->Wc : d->W;
    out->plane[c] = (uint16_t*)xcalloc(d, (size_t)w * h, sizeof(uint16_t));
    for (int y = 0; y < h; y++) memcpy(out->plane[c] + (size_t)y * w, fin[c] + (size_t)(ys + y) * st + xs, sizeof(uint16_t) * w);
  }
  if (d->seq_mode) {   /* C.5.2.3: the current picture enters the DPB (every picture of these sequences is a reference picture) */
    int slot = -1;
    for (int i = 0; i < MAX_DPB; i++) if (!d->dpb[i].valid) { slot = i; break; }
    if (slot < 0) fail(d, "decoded picture buffer is full");
    if (slot >= d->n_dpb) d->n_dpb = slot + 1;
    for (int c = 0; c < nc; c++) d->dpb[slot].plane[c] = dup_plane(d, fin[c], c ? (size_t)d->Wc * d->Hc : (size_t)d->W * d->H);
    d->dpb[slot].poc = d->poc; d->dpb[slot].valid = 1; d->dpb[slot].is_lt = 0;
    dpb_store_motion(d, &d->dpb[slot]);
  }
  if (keep_taps) {
    for (int c = 0; c < nc; c++) { out->final_coded[c] = fin[c]; fin[c] = NULL; out->coeff[c] = d->coeff[c]; d->coeff[c] = NULL; }
    out->map_stride = d->mw; out->map_height = d->mh;
    out->map_log2_tb = d->m_log2_tb; d->m_log2_tb = NULL;
    out->map_log2_cb = d->m_log2_cb; d->m_log2_cb = NULL;
    out->map_intra_luma = d->m_ipm; d->m_ipm = NULL;
    out->map_intra_chroma = d->m_ipmc; d->m_ipmc = NULL;
    out->map_qp_y = d->m_qp; d->m_qp = NULL;
    out->map_flags = d->m_flags; d->m_flags = NULL;
    out->ctb_log2 = s->log2_ctb; out->ctbs_w = d->ctbW; out->ctbs_h = d->ctbH;
    out->sao_type = d->sao_type; d->sao_type = NULL;
    out->sao_band_or_class = d->sao_bc; d->sao_bc = NULL;
    out->sao_offset = d->sao_off; d->sao_off = NULL;
    out->map_pred = d->m_pred; d->m_pred = NULL;
    out->mf_mv = d->mf_mv; d->mf_mv = NULL;
    out->mf_ref = d->mf_ref; d->mf_ref = NULL;
  }
  for (int c = 0; c < 3; c++) free(fin[c]);
}

int hevc_oracle_decode(const uint8_t* data, size_t size, int keep_taps, hevc_oracle_picture* out,
                       char* errbuf, size_t errbuf_len)
{
  Dec* d = (Dec*)calloc(1, sizeof(Dec));
  if (!d) return -1;
  memset(out, 0, sizeof(*out));
  init_scans(); init_dct();
  d->first_picture = 1;
  if (setjmp(d->jb)) {
    if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", d->err);
    hevc_oracle_free_picture(out);
    free_dec(d);
    return -2;
  }
  decode_access_unit(d, data, size, keep_taps, out);
  free_dec(d);
  return 0;
}

/* ---- a sequence of pictures (the samples libheif pushes for a track): parameter sets, POC state and the DPB persist ------------------ */
struct hevc_oracle_seq { Dec* d; };

hevc_oracle_seq* hevc_oracle_seq_new(void)
{
  hevc_oracle_seq* q = (hevc_oracle_seq*)calloc(1, sizeof(*q));
  if (!q) return NULL;
  q->d = (Dec*)calloc(1, sizeof(Dec));
  if (!q->d) { free(q); return NULL; }
  init_scans(); init_dct();
  q->d->seq_mode = 1; q->d->first_picture = 1;
  return q;
}

int hevc_oracle_seq_decode(hevc_oracle_seq* q, const uint8_t* data, size_t size, int keep_taps, hevc_oracle_picture* out,
                           char* errbuf, size_t errbuf_len)
{
  Dec* d = q->d;
  memset(out, 0, sizeof(*out));
  if (setjmp(d->jb)) {
    if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", d->err);
    hevc_oracle_free_picture(out);
    release_picture(d);
    return -2;
  }
  decode_access_unit(d, data, size, keep_taps, out);
  release_picture(d);
  return 0;
}

void hevc_oracle_seq_free(hevc_oracle_seq* q)
{
  if (!q) return;
  free_dec(q->d);
  free(q);
}

void hevc_oracle_free_picture(hevc_oracle_picture* pic)
{
  for (int c = 0; c < 3; c++) {
    free(pic->plane[c]); free(pic->pre_deblock[c]); free(pic->post_deblock[c]);
    free(pic->final_coded[c]); free(pic->coeff[c]);
  }
  free(pic->map_log2_tb); free(pic->map_log2_cb); free(pic->map_intra_luma); free(pic->map_intra_chroma);
  free(pic->map_qp_y); free(pic->map_flags); free(pic->sao_type); free(pic->sao_band_or_class); free(pic->sao_offset);
  free(pic->map_pred); free(pic->mf_mv); free(pic->mf_ref);
  memset(pic, 0, sizeof(*pic));
}
