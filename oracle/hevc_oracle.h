/*
 * hevc_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A scalar, single-threaded, spec-literal restatement of the ITU-T H.265 (v1: Main / Main10 /
 * Main Still Picture, intra pictures only) decoding process.  It stands in for libde265, the
 * third-party decoder that libheif's hot path delegates to
 * (reference call site: libheif/plugins/decoder_libde265.cc:386-457, de265_decode at :402;
 * libde265 itself is un-vendored and absent from /root/reference — SURVEY.md §8c).
 *
 * PARITY STATUS: "parity unpinned" for decoded HEVC pixels — the reference tree holds no
 * decoded-pixel golden vector, MD5 or picture-hash SEI for any HEVC fixture, and libde265 cannot
 * be run here.  The oracle is pinned only structurally: it must decode the reference's real
 * x265-produced fixtures (examples/example.heic, tests/data/*.heic) with every CABAC substream
 * terminating exactly on its entry point, and to the dimensions the reference's own tests assert
 * (tests/component_descriptions.cc:286-323).  See DESIGN.md §oracle.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef HEVC_ORACLE_H
#define HEVC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hevc_oracle_picture {
  int width, height;          /* cropped (conformance window applied) luma size            */
  int chroma_format_idc;      /* 0 = 4:0:0, 1 = 4:2:0                                      */
  int bit_depth_luma, bit_depth_chroma;
  int cwidth, cheight;        /* cropped chroma size                                       */
  uint16_t* plane[3];         /* cropped output planes, stride == width / cwidth, malloc'd */
  /* VUI colour description (H.265 Annex E defaults when absent: 2,2,2,limited)            */
  int colour_primaries, transfer_characteristics, matrix_coeffs, full_range_flag;

  /* ---- intermediate taps (coded size, i.e. before the conformance crop) --------------- */
  int coded_width, coded_height;     /* pic_width/height_in_luma_samples                   */
  int ccoded_width, ccoded_height;
  uint16_t* pre_deblock[3];          /* reconstruction before in-loop filters              */
  uint16_t* post_deblock[3];         /* after deblocking, before SAO                       */
  uint16_t* final_coded[3];          /* after SAO, uncropped                               */
  int32_t*  coeff[3];                /* TransCoeffLevel at spatial position (TU origin+xc) */
  /* per 4x4-luma-unit maps, stride = (coded_width+3)/4                                    */
  int      map_stride, map_height;
  uint8_t* map_log2_tb;              /* log2 size of luma TB covering the unit             */
  uint8_t* map_log2_cb;              /* log2 size of the coding block covering the unit    */
  uint8_t* map_intra_luma;           /* IntraPredModeY                                     */
  uint8_t* map_intra_chroma;         /* IntraPredModeC of the CU                           */
  int8_t*  map_qp_y;                 /* QpY of the CU                                      */
  uint8_t* map_flags;                /* bit0 cbf_luma, bit1 cbf_cb, bit2 cbf_cr (of the TU),
                                        bit3 transquant bypass, bit4 pcm,
                                        bit5 vertical deblock edge at the unit's left side,
                                        bit6 horizontal deblock edge at the unit's top     */
  /* SAO parameters per CTB (raster), 3 components                                          */
  int      ctb_log2, ctbs_w, ctbs_h;
  uint8_t* sao_type;                 /* [ctb*3+c] 0 off, 1 band, 2 edge                    */
  uint8_t* sao_band_or_class;        /* [ctb*3+c] band position or eo class                */
  int16_t* sao_offset;               /* [(ctb*3+c)*4+i] SaoOffsetVal[i+1]                  */
  /* statistics */
  uint64_t n_bins_ctx, n_bins_bypass; /* CABAC bins decoded                                */
  int      n_substreams;
  /* ---- sequences (hevc_oracle_seq_*): picture order count and, with keep_taps, the motion field per 4x4 unit ---- */
  int      poc;
  uint8_t* map_pred;                 /* 0 MODE_INTRA, 1 MODE_INTER, 2 MODE_SKIP (NULL for hevc_oracle_decode)      */
  int16_t* mf_mv;                    /* [unit][list 0 / 1][x, y] motion vectors in quarter luma samples (0 where unused)  */
  int8_t*  mf_ref;                   /* [unit][list 0 / 1] reference indices, -1 where the list is not used / intra units */
} hevc_oracle_picture;

/* Decode one intra picture.  `data` is libheif's plugin framing: a concatenation of
 * [4-byte big-endian length][NAL unit without start code]  (reference contract:
 * libheif/plugins/decoder_libde265.cc:322-368, built by libheif/codecs/decoder.cc:275-308).
 * keep_taps != 0 keeps the intermediate buffers in *out.
 * Returns 0 on success; otherwise a negative code and a message in errbuf. */
int hevc_oracle_decode(const uint8_t* data, size_t size, int keep_taps,
                       hevc_oracle_picture* out, char* errbuf, size_t errbuf_len);

void hevc_oracle_free_picture(hevc_oracle_picture* pic);

/* A sequence of pictures: one access unit per call, in decoding order, the way libheif pushes the samples of a track
 * (libheif/sequences/track_visual.cc:200-280); parameter sets, the POC state and the decoded picture buffer persist between calls.
 * P and B slices are decoded (scope: oracle/hevc_oracle_inter.c); every picture is returned at once, i.e. in DECODING order, with its POC. */
typedef struct hevc_oracle_seq hevc_oracle_seq;
hevc_oracle_seq* hevc_oracle_seq_new(void);
int hevc_oracle_seq_decode(hevc_oracle_seq* q, const uint8_t* data, size_t size, int keep_taps, hevc_oracle_picture* out,
                           char* errbuf, size_t errbuf_len);
void hevc_oracle_seq_free(hevc_oracle_seq* q);

#ifdef __cplusplus
}
#endif
#endif
