// parse_tables.h — HEVC constant tables of the entropy decoder (context initValues for slice_type I, rangeTabLps, transIdxLps, scans,
// sig_coeff_flag context patterns) shared by the two parser kernels: parse_core.h (one wave per substream) and parse_lanes_kernel.hip
// (one LANE per substream).  The including file defines PC_CONST (the storage qualifier of the tables) first.
#pragma once
#include <stdint.h>
#ifndef PC_CONST
#define PC_CONST static const
#endif

namespace hipdec {
namespace pcore {

// ---- context variables: group (VGPR) and lane -------------------------------------------------
// group A
enum : int {
  A_SAO_MERGE = 0, A_SAO_TYPE = 1, A_SPLIT_CU = 2 /*3*/, A_CU_TQ_BYPASS = 5, A_PART_MODE = 6, A_PREV_INTRA_LUMA = 7,
  A_INTRA_CHROMA = 8, A_SPLIT_TRANSFORM = 9 /*3*/, A_CBF_LUMA = 12 /*2*/, A_CBF_CHROMA = 14 /*4*/, A_CU_QP_DELTA = 18 /*2*/,
  A_TRANSFORM_SKIP = 20 /*2*/, A_LAST_X = 22 /*18*/, A_LAST_Y = 40 /*18*/, A_CODED_SUB_BLOCK = 58 /*4*/,
  A_CBF_CHROMA4 = 62 /* cbf_cb / cbf_cr at trafoDepth 4 (ChromaArrayType 3 only; initValue 154 like the padding lanes) */,
  // group B: sig_coeff_flag 0..43 (42 used by version 1), greater2 44..49
  B_SIG_COEFF = 0, B_GREATER2 = 44,
  // group C: greater1 0..23, then the contexts P slices add (parse_core.h with HIPDEC_PARSE_INTER)
  C_GREATER1 = 0,
  C_SKIP_FLAG = 24 /*3*/, C_PRED_MODE = 27, C_PART_MODE_INTER = 28 /*3: part_mode bin 1, bin 2 at the minimum CB size, bin 2 with AMP*/, C_MERGE_FLAG = 31,
  C_MERGE_IDX = 32, C_REF_IDX = 33 /*2*/, C_MVD_GT0 = 35, C_MVD_GT1 = 36, C_MVP_FLAG = 37, C_RQT_ROOT_CBF = 38,
  C_INTER_PRED_IDC = 39 /*5: bin 0 by coding quadtree depth 0..3, bin 1*/
};

// initValue for slice_type I (9.3.2.2, tables 9-5 .. 9-37), laid out per group / lane
PC_CONST uint8_t c_init[3][64] = {
  {153, 200, 139, 141, 157, 154, 184, 184, 63, 153, 138, 138, 111, 141, 94, 138, 182, 154, 154, 154, 139, 139,
   110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
   110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
   91, 171, 134, 141, 154, 154},
  {111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141,
   179, 153, 125, 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111, 141, 111,
   138, 153, 136, 167, 152, 152, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154},
  {140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
   154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
   154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154}};

// initValue for initType 1 and 2 (9.3.2.2: a P slice takes 2 when cabac_init_flag is set, else 1), same lanes
PC_CONST uint8_t c_init_p[2][3][64] = {
 {{153, 185, 107, 139, 126, 154, 154, 154, 152, 124, 138, 94, 153, 111, 149, 107, 167, 154, 154, 154, 139, 139,
   125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
   125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
   121, 140, 61, 154, 154, 154},
  {155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140,
   136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140, 140, 140,
   107, 167, 91, 122, 107, 167, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154},
  {154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
   197, 185, 201, 149, 139, 154, 154, 110, 122, 153, 153, 140, 198, 168, 79, 95,
   79, 63, 31, 31, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154}},
 {{153, 160, 107, 139, 126, 154, 154, 183, 152, 224, 167, 122, 153, 111, 149, 92, 167, 154, 154, 154, 139, 139,
   125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
   125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
   121, 140, 61, 154, 154, 154},
  {170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140,
   136, 153, 154, 170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140, 140, 140,
   107, 167, 91, 107, 107, 167, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154},
  {154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182,
   197, 185, 201, 134, 139, 154, 154, 154, 137, 153, 153, 169, 198, 168, 79, 95,
   79, 63, 31, 31, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154}}};

// Table 8-3: the intra prediction direction of a 4:2:2 chroma block from the mode the 4:2:0 / 4:4:4 derivation yields (8.4.3)
PC_CONST uint8_t c_map422[35] = {0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 11, 13, 15, 16, 18, 19, 20, 21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27, 28, 28, 29, 29, 30, 31};

// lane p: rangeTabLps[p][0..3] packed little-endian (table 9-46)
PC_CONST uint8_t c_range_lps[64 * 4] = {
  128,176,208,240, 128,167,197,227, 128,158,187,216, 123,150,178,205, 116,142,169,195, 111,135,160,185,
  105,128,152,175, 100,122,144,166,  95,116,137,158,  90,110,130,150,  85,104,123,142,  81, 99,117,135,
   77, 94,111,128,  73, 89,105,122,  69, 85,100,116,  66, 80, 95,110,  62, 76, 90,104,  59, 72, 86, 99,
   56, 69, 81, 94,  53, 65, 77, 89,  51, 62, 73, 85,  48, 59, 69, 80,  46, 56, 66, 76,  43, 53, 63, 72,
   41, 50, 59, 69,  39, 48, 56, 65,  37, 45, 54, 62,  35, 43, 51, 59,  33, 41, 48, 56,  32, 39, 46, 53,
   30, 37, 43, 50,  29, 35, 41, 48,  27, 33, 39, 45,  26, 31, 37, 43,  24, 30, 35, 41,  23, 28, 33, 39,
   22, 27, 32, 37,  21, 26, 30, 35,  20, 24, 29, 33,  19, 23, 27, 31,  18, 22, 26, 30,  17, 21, 25, 28,
   16, 20, 23, 27,  15, 19, 22, 25,  14, 18, 21, 24,  14, 17, 20, 23,  13, 16, 19, 22,  12, 15, 18, 21,
   12, 14, 17, 20,  11, 14, 16, 19,  11, 13, 15, 18,  10, 12, 15, 17,  10, 12, 14, 16,   9, 11, 13, 15,
    9, 11, 12, 14,   8, 10, 12, 14,   8,  9, 11, 13,   7,  9, 11, 12,   7,  9, 10, 12,   7,  8, 10, 11,
    6,  8,  9, 11,   6,  7,  9, 10,   6,  7,  8,  9,   2,  2,  2,  2};
// lane p: byte 0 transIdxLps[p] (table 9-47; load_tables adds bit 6 for p = 0, where an LPS flips valMps), byte 1 the p-th position of the up-right diagonal
// scan of an 8x8 array (6.5.3) as x | y << 3, byte 2 the inverse of that scan (lane x | y << 3 -> scan position)
PC_CONST uint8_t c_next_lps[64] = {
   0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9,11,11,12, 13,13,15,15,16,16,18,18,19,19,21,21,22,22,23,24,
  24,25,26,26,27,27,28,29,29,30,30,30,31,32,32,33, 33,33,34,34,35,35,35,36,36,36,37,37,37,38,38,63};
PC_CONST uint8_t c_diag8[64] = {
  0, 8, 1, 16, 9, 2, 24, 17, 10, 3, 32, 25, 18, 11, 4, 40, 33, 26, 19, 12, 5, 48, 41, 34, 27, 20, 13, 6, 56, 49, 42, 35,
  28, 21, 14, 7, 57, 50, 43, 36, 29, 22, 15, 58, 51, 44, 37, 30, 23, 59, 52, 45, 38, 31, 60, 53, 46, 39, 61, 54, 47, 62, 55, 63};

// 4x4 scans: nibble k = raster index (x | y << 2) of the k-th scan position
#define PC_DIAG4 0xFBE7AD369C258140ULL
#define PC_HORZ4 0xFEDCBA9876543210ULL
#define PC_VERT4 0xFB73EA62D951C840ULL
// their inverses: nibble r = scan position of raster index r (constexpr-derived, so they cannot drift from the scans)
constexpr uint64_t pc_invert_scan4(uint64_t scan)
{
  uint64_t inv = 0;
  for (int k = 0; k < 16; k++) inv |= (uint64_t)k << (4 * ((scan >> (4 * k)) & 15u));
  return inv;
}
constexpr uint64_t PC_INV_DIAG4 = pc_invert_scan4(PC_DIAG4), PC_INV_HORZ4 = pc_invert_scan4(PC_HORZ4), PC_INV_VERT4 = pc_invert_scan4(PC_VERT4);
static_assert(PC_INV_HORZ4 == PC_HORZ4 && ((PC_INV_DIAG4 >> (4 * 4)) & 15u) == 1u && ((PC_INV_VERT4 >> (4 * 1)) & 15u) == 4u, "inverse scans");
// sig_coeff_flag ctxIdxMap for 4x4 blocks (9.3.4.2.5), nibble r = ctxIdxMap[raster index r]
#define PC_CTXIDXMAP4 0x8877886654325410ULL
// sigCtx of 9.3.4.2.5 for larger blocks before the size / component offsets, two bits per raster
// position r = x | y << 2 of the 4x4 sub-block, one word per prevCsbf (bit0 right, bit1 below)
//   0: x+y == 0 ? 2 : x+y < 3 ? 1 : 0      1: y == 0 ? 2 : y == 1 ? 1 : 0
//   2: x == 0 ? 2 : x == 1 ? 1 : 0          3: 2
constexpr uint32_t pc_sigpat(int prev_csbf)
{
  uint32_t w = 0;
  for (int r = 0; r < 16; r++) {
    const int x = r & 3, y = r >> 2;
    int v = 2;
    if (prev_csbf == 0) v = (x + y == 0) ? 2 : (x + y < 3) ? 1 : 0;
    else if (prev_csbf == 1) v = (y == 0) ? 2 : (y == 1) ? 1 : 0;
    else if (prev_csbf == 2) v = (x == 0) ? 2 : (x == 1) ? 1 : 0;
    else v = 2;
    w |= (uint32_t)v << (2 * r);
  }
  return w;
}
constexpr uint32_t PC_SIGPAT0 = pc_sigpat(0), PC_SIGPAT1 = pc_sigpat(1), PC_SIGPAT2 = pc_sigpat(2), PC_SIGPAT3 = pc_sigpat(3);
static_assert(PC_SIGPAT0 == 0x00010516u && PC_SIGPAT1 == 0x000055AAu && PC_SIGPAT2 == 0x06060606u && PC_SIGPAT3 == 0xAAAAAAAAu, "sig patterns");

}  // namespace pcore
}  // namespace hipdec
