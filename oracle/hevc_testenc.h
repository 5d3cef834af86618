/* hevc_testenc.h — test-only HEVC stream generator: intra pictures, and sequences with P and B pictures (see hevc_testenc.c). */
#ifndef HEVC_TESTENC_H
#define HEVC_TESTENC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct hevc_testenc_params {
  int width, height;            /* display size; coded size is rounded up to the min CB size      */
  int bit_depth;                /* 8..12 (luma == chroma)                                          */
  int chroma_format_idc;        /* 0 .. 3 (4:0:0, 4:2:0, 4:2:2, 4:4:4)                             */
  int log2_ctb, log2_min_cb, log2_min_tb, log2_max_tb;
  int max_transform_hierarchy_depth_intra;
  int qp;
  int wpp, tile_cols, tile_rows, num_slices;
  int sao, deblock_disable, beta_offset_div2, tc_offset_div2;
  int sign_data_hiding, cu_qp_delta, diff_cu_qp_delta_depth, transform_skip;
  int lossless_pct;             /* >0 enables transquant bypass; % of CUs coded lossless           */
  int pcm_pct, pcm_loop_filter_disabled;
  int strong_intra_smoothing, scaling_list;
  int cb_qp_offset, cr_qp_offset;
  int loop_filter_across_tiles, loop_filter_across_slices;
  int vui_primaries, vui_transfer, vui_matrix, vui_full_range; /* vui_matrix < 0: no VUI           */
  uint32_t seed;
  int stress;                   /* 1: random splits / modes (syntax coverage); 0: SAD-driven       */
  int zero_residual_pct;        /* % of transform blocks forced to cbf = 0                         */
  int dependent_segments;       /* > 1: every slice is split into that many slice segments, all but its first dependent */
  /* ---- sequences (hevc_testenc_encode_seq): frame 0 is an IDR intra picture, the others are P pictures (TRAIL_R) - or, with b_frames, B pictures between P anchors ---- */
  int inter_num_refs;           /* reference pictures a picture may use before it (the previous anchors; 0 = 1)           */
  int inter_skip_pct, inter_intra_pct, inter_merge_pct;   /* % of coding units skipped / intra coded, % of prediction units merged */
  int amp;                      /* asymmetric motion partitions                                                            */
  int max_merge_cand;           /* MaxNumMergeCand 1..5 (0 = 5)                                                            */
  int parallel_merge_level;     /* Log2ParMrgLevel 2..log2_ctb (0 = 2)                                                     */
  int max_transform_hierarchy_depth_inter;
  int cabac_init_present, lists_modification;
  int global_mv_x, global_mv_y; /* motion (quarter luma samples) most vectors are drawn around                             */
  /* ---- B pictures, TMVP, weighted prediction ---- */
  int b_frames;                 /* B pictures between two anchor (I / P) pictures: coded after the later anchor, POC order != coding order */
  int b_ref;                    /* 1: B pictures are reference pictures too (TRAIL_R; a B picture also predicts from the one before it)    */
  int inter_bi_pct;             /* % of the non-merged prediction units of a B picture that are bi-predicted (the rest: list 0 or list 1)   */
  int temporal_mvp;             /* sps_temporal_mvp_enabled_flag, slice_temporal_mvp_enabled_flag in every P / B picture                  */
  int weighted_pred;            /* weighted_pred_flag / weighted_bipred_flag with a random pred_weight_table per slice                     */
  int mvd_l1_zero;              /* mvd_l1_zero_flag in B slices                                                                            */
  int constrained_intra_pred;   /* constrained_intra_pred_flag: intra blocks of P / B pictures predict from intra coded neighbours only    */
  int long_term_ref;            /* > 0: the IDR picture stays in the DPB as a LONG-TERM reference picture of every later picture (behind the short-term
                                   ones in both lists): 1 coded in the slice header by its POC LSBs, 2 with delta_poc_msb_present_flag, 3 as a candidate of the SPS */
  int open_gop;                 /* n > 0 (needs b_frames): the n-th anchor behind the IDR picture is an intra CRA picture (NAL type 21) and the B pictures coded
                                   after it that precede it in output order are its RASL pictures (RASL_R 9 with b_ref, else RASL_N 8: they predict from the anchor
                                   BEFORE the CRA picture); pictures behind the CRA picture in output order reference nothing in front of it.  A decoder that starts
                                   at the CRA picture drops the RASL pictures (8.3.3) and decodes everything else identically */
  int hidden_poc;               /* > 0: output_flag_present_flag = 1 and the picture with this PicOrderCnt carries pic_output_flag = 0: it is decoded and may be
                                   referenced but never output (C.5.2.2); every other picture carries pic_output_flag = 1 */
} hevc_testenc_params;

/* planes: tightly packed uint16 samples at display size (chroma (w+1)/2 x (h+1)/2).
 * Output: malloc'd [u32 BE length][NAL]... stream (VPS, SPS, PPS, slices); free with
 * hevc_testenc_free(). */
int hevc_testenc_encode(const hevc_testenc_params* prm, const uint16_t* const planes[3], uint8_t** out,
                        size_t* out_size, char* errbuf, size_t errbuf_len);
/* n_frames pictures at display size: planes[3 * f + c], f = output (POC) order.  out[k] / out_sizes[k]: one malloc'd access unit per picture
 * in plugin framing, k = DECODING order (equal to f without B pictures; the first one carries VPS, SPS, PPS), what libheif pushes sample by
 * sample for a track (libheif/sequences/track_visual.cc:200-280). */
int hevc_testenc_encode_seq(const hevc_testenc_params* prm, int n_frames, const uint16_t* const* planes, uint8_t** out,
                            size_t* out_sizes, char* errbuf, size_t errbuf_len);
void hevc_testenc_free(uint8_t* p);
#ifdef __cplusplus
}
#endif
#endif
