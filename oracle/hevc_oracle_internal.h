/* hevc_oracle_internal.h — pieces of the CPU oracle shared with the test-stream generator
 * (oracle/hevc_testenc.c).  Test infrastructure only. */
#ifndef HEVC_ORACLE_INTERNAL_H
#define HEVC_ORACLE_INTERNAL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* H.265 §8.4.4.2: intra sample prediction of one nTbS x nTbS block.
 * ref_left[0..2n]  : p[-1][-1], p[-1][0] .. p[-1][2n-1]   (after substitution, before filtering)
 * ref_top[0..2n]   : p[-1][-1], p[0][-1] .. p[2n-1][-1]
 * filter / edge rules applied inside according to cIdx, mode, nTbS. */
void hevc_intra_predict(uint16_t* dst, int dst_stride, int nTbS, int cIdx, int mode,
                        const uint16_t* ref_left, const uint16_t* ref_top, int bit_depth,
                        int strong_intra_smoothing, int chroma_format_idc);

/* H.265 §8.6.2-8.6.4: scaling + inverse transform of one block into residual r[y*n+x].
 * coeff[y*n+x] = TransCoeffLevel.  m = NULL means flat 16.  trType 1 = DST-VII 4x4. */
void hevc_scale_and_transform(int32_t* res, const int32_t* coeff, int nTbS, int qP, int bit_depth,
                              const uint8_t* scaling_m /* [y*n+x] or NULL */,
                              int transform_skip, int trType);

/* forward-order scan tables: ScanOrder[log2BlockSize 0..5? we use 1..3][scanIdx][pos] -> x | y<<4 */
const uint8_t* hevc_scan_order(int log2_size /*1,2,3*/, int scan_idx /*0 diag,1 hor,2 ver*/);

int hevc_chroma_qp_420(int qpi);

extern const uint8_t hevc_cabac_range_lps[64][4];
extern const uint8_t hevc_cabac_next_lps[64];
extern const uint8_t hevc_cabac_next_mps[64];

/* context index bases (shared numbering for decoder and test encoder) */
enum {
  CTX_SAO_MERGE = 0,
  CTX_SAO_TYPE = 1,
  CTX_SPLIT_CU = 2,          /* 3 */
  CTX_CU_TQ_BYPASS = 5,
  CTX_PART_MODE = 6,
  CTX_PREV_INTRA_LUMA = 7,
  CTX_INTRA_CHROMA = 8,
  CTX_SPLIT_TRANSFORM = 9,   /* 3 */
  CTX_CBF_LUMA = 12,         /* 2 */
  CTX_CBF_CHROMA = 14,       /* 4 */
  CTX_CU_QP_DELTA = 18,      /* 2 */
  CTX_TRANSFORM_SKIP = 20,   /* 2: luma, chroma */
  CTX_LAST_X = 22,           /* 18 */
  CTX_LAST_Y = 40,           /* 18 */
  CTX_CODED_SUB_BLOCK = 58,  /* 4 */
  CTX_SIG_COEFF = 62,        /* 42 */
  CTX_GREATER1 = 104,        /* 24 */
  CTX_GREATER2 = 128,        /* 6 */
  CTX_CBF_CHROMA4 = 134,     /* cbf_cb / cbf_cr at trafoDepth 4: reachable only with ChromaArrayType 3 (Table 9-4 of the 2nd edition on) */
  /* P slices (hevc_oracle_inter.c) */
  CTX_SKIP_FLAG = 135,       /* 3 */
  CTX_PRED_MODE = 138,
  CTX_PART_MODE_INTER = 139, /* 3: part_mode bin 1, bin 2 at the minimum CB size, bin 2 with AMP (bin 0 is CTX_PART_MODE) */
  CTX_MERGE_FLAG = 142,
  CTX_MERGE_IDX = 143,
  CTX_REF_IDX = 144,         /* 2 */
  CTX_MVD_GT0 = 146,
  CTX_MVD_GT1 = 147,
  CTX_MVP_FLAG = 148,
  CTX_RQT_ROOT_CBF = 149,
  CTX_INTER_PRED_IDC = 150,  /* 5: bin 0 by CtDepth 0..3, bin 1 (B slices) */
  CTX_COUNT = 155
};
extern const uint8_t hevc_cabac_init_I[CTX_COUNT];
extern const uint8_t hevc_cabac_init_P[2][CTX_COUNT];   /* initType 1 and 2 (P slice: cabac_init_flag ? 2 : 1) */
enum { PART_2Nx2N = 0, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N };   /* Table 7-10 */

#ifdef __cplusplus
}
#endif
#endif
