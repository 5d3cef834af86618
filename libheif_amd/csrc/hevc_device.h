// hevc_device.h — POD descriptors shared by the host front end and the HIP kernels.
//
// Data layout in HBM (per picture, all offsets relative to one arena allocation per batch):
//   bitstream   raw bytes exactly as pushed (length-prefixed NALs, emulation prevention bytes still
//               in place; the CABAC reader skips them on the fly)
//   unit maps   one byte per 4x4 luma unit, stored CTB by CTB (raster CTB order) and in z-scan order
//               inside a CTB, so that every CU / TU occupies a contiguous index range:
//                 u_size  low nibble log2 TB size, high nibble log2 CB size
//                 u_flags bit0 cbf_luma, bit1 cbf_cb, bit2 cbf_cr, bit3 cu_transquant_bypass,
//                         bit4 pcm_flag (the unit's "coefficients" are its samples, see coeff), bit5 vertical deblocking
//                         edge on the unit's left side, bit6 horizontal edge on its top,
//                         bit7 transform_skip (luma)
//                 u_ipm   bits0-5 IntraPredModeY, bit6 transform_skip Cb, bit7 transform_skip Cr
//                 u_ipmc  IntraPredModeC
//                 u_qp    QpY of the coding unit (int8)
//   coeff       int16 TransCoeffLevel, TU-contiguous: a luma TU whose first unit has z-index u owns
//               [ctb*ctbSize^2 + u*16, +n*n) in raster order inside the TU; chroma likewise at
//               [ctb*ctbSize^2/4 + u*4, +n*n/4).  A PCM coding unit is stored as one block of CU size per component whose
//               "levels" are its samples (pcm_sample << (BitDepth - PcmBitDepth)); its coded-block flags stay 0.
//               Other chroma formats: the chroma offsets scale with the samples per luma unit — 4:2:2 at [ctb*ctbSize^2/2 + u*8, ...), its TWO
//               square blocks one after the other (upper, lower); 4:4:4 at [ctb*ctbSize^2 + u*16, +n*n) like luma.
//   4:2:2 flags the LOWER chroma block of a unit has its own cbf_cb / cbf_cr / transform_skip flags: they sit in the map entries of unit u ^ 1
//               (u_flags bits 1-2, u_ipm bits 6-7 of the block's second unit, or of the third unit of a quad of 4x4 luma blocks — bits those
//               entries do not use otherwise; only a block's first unit / a quad's fourth unit is ever read for block flags)
//   rec planes  reconstructed samples (coded size, stride padded to 64 B), deblocked in place
//   out planes  SAO output cropped to the conformance window (what the plugin hands to libheif)
#pragma once
#include <stdint.h>

namespace hipdec {

enum : int { CTX_STORE = 192,      // bytes per saved context table: 3 groups x 64 context variables x 1 byte (parse_core.h)
              HANDOFF_DWORDS = 16, // per-CTB record handed to the CTB below (SAO parameters + bottom-row sizes)
              SAVE_DWORDS = 96 };  // suspended-row state (pool scheduler): 84 dwords used, see parse_core.h "parked row state"

enum : uint8_t {
  UF_CBF_LUMA = 1, UF_CBF_CB = 2, UF_CBF_CR = 4, UF_BYPASS = 8, UF_PCM = 16, UF_VEDGE = 32, UF_HEDGE = 64, UF_TS_LUMA = 128
};

// per-CTB availability bits computed on the host (6.4.1: same slice, same tile, inside the picture)
// AV_EDGE_*: the CTB's left / top boundary may be deblocked (8.7.2: picture edge, slice edge with
// slice_loop_filter_across_slices_enabled_flag = 0, tile edge with loop_filter_across_tiles = 0)
enum : uint8_t { AV_LEFT = 1, AV_UP = 2, AV_UPRIGHT = 4, AV_UPLEFT = 8, AV_EDGE_LEFT = 16, AV_EDGE_UP = 32 };

struct SliceParams {
  int32_t slice_qp_y;
  int8_t cb_qp_offset, cr_qp_offset;        // pps + slice offsets (dequantisation)
  int8_t pps_cb_qp_offset, pps_cr_qp_offset; // cQpPicOffset for chroma deblocking
  int8_t beta_offset_div2, tc_offset_div2;
  uint8_t deblocking_disabled;
  uint8_t sao_luma, sao_chroma;
  uint8_t lf_across_slices;                  // slice_loop_filter_across_slices_enabled_flag
  uint16_t slice_addr_rs;
  // ---- P / B slices (sequence tracks); all 0 for an intra slice
  uint8_t is_p;                              // slice_type != I (a P or a B slice)
  uint8_t num_ref_idx;                       // num_ref_idx_l0_active_minus1 + 1
  uint8_t max_merge_cand;                    // MaxNumMergeCand
  uint8_t init_type;                         // 9.3.2.2 initType: 0 (I), 1 (P) or 2 (B); cabac_init_flag swaps the latter two
  uint8_t ref_slot[16];                      // RefPicList0[i] as an index into the picture's RefFrame table (PicParams::off_reftab)
  uint8_t is_b;                              // slice_type == B
  uint8_t num_ref_idx_l1;                    // num_ref_idx_l1_active_minus1 + 1 (0 in a P slice)
  uint8_t mvd_l1_zero;                       // mvd_l1_zero_flag
  uint8_t tmvp;                              // slice_temporal_mvp_enabled_flag
  uint8_t col_slot;                          // the collocated picture (collocated_from_l0_flag / collocated_ref_idx) as a RefFrame slot
  uint8_t col_from_l0;                       // collocated_from_l0_flag (8.5.3.2.9: which list of a bi-predicted collocated block counts)
  uint8_t no_backward;                       // NoBackwardPredFlag: no reference picture of the slice follows the current picture in output order
  uint8_t weighted;                          // explicit weighted prediction (weighted_pred_flag / weighted_bipred_flag): the table below applies
  uint8_t luma_log2_wd, chroma_log2_wd;      // luma_log2_weight_denom, ChromaLog2WeightDenom
  uint16_t wp_index;                         // the slice's WeightTable in PicParams::off_wp
  uint8_t ref_slot_l1[16];                   // RefPicList1[i]
};
static_assert(sizeof(SliceParams) == 64, "SliceParams layout");

// pred_weight_table of one slice (7.3.6.3) as the weighted sample prediction uses it (8.5.3.3.4.3): [list][refIdx][cIdx]
struct WeightTable { int16_t w[2][16][3]; int16_t o[2][16][3]; };   // o: before the << (BitDepth - 8)
static_assert(sizeof(WeightTable) == 384, "WeightTable layout");

// a reference picture of a P / B picture: absolute device pointers (the planes live in an EARLIER batch's arena: decoded, deblocked, SAO applied,
// coded size), strides in bytes; mf: that picture's motion field (MotionUnit per 4x4 unit, CTB-major z-order like the current picture's; 0 if it
// was an intra picture), read by the temporal candidates
struct RefFrame {
  uint64_t plane[3];
  uint32_t stride[3];
  int32_t poc;
  uint64_t mf;
  uint32_t progress_row;   // != 0: the picture is an earlier item of the SAME launch set (a chain, batch_layout.h) and its motion field is derived by the same
                           // k_motion launch: 1 + its first row in the row-progress table (slot 2 of a row counts the CTBs whose motion is complete)
  uint32_t reserved;
};
static_assert(sizeof(RefFrame) == 56, "RefFrame layout");

// motion of one 4x4 luma unit (the motion field k_motion writes)
struct MotionUnit {
  int16_t mv[2][2];      // [list][x, y], quarter luma samples
  int16_t poc_delta[2];  // PicOrderCnt(this picture) - PicOrderCnt(the list's reference picture): all the temporal candidates of LATER pictures need
  int8_t ref_idx[2];     // refIdxL0 / L1, -1: the list is not used (both -1: the unit is intra coded)
  uint8_t slot_pred[2];  // bits 0..5: the list's reference picture as a RefFrame slot (equal slots <=> the same picture, 8.7.2.4);
                         // [1] bits 6 / 7: the reference picture of list 0 / 1 was a LONG-TERM one when this picture was decoded (LongTermRefPic of 8.5.3.2.9);
                         // [0] bits 6..7: 0 MODE_INTRA, 1 MODE_INTER, 2 MODE_SKIP
};
static_assert(sizeof(MotionUnit) == 16, "MotionUnit layout");

// u_ipmc of a P picture: bits 0..5 IntraPredModeC (1 for units that are not intra coded), bit 6 the unit is inter coded, bit 7 it is skipped
enum : uint8_t { UM_INTER = 64, UM_SKIP = 128 };

// motion syntax of one prediction unit (the parser writes it at the unit index of the PU's top-left 4x4 unit; k_motion turns it into the motion
// field): w0 = bit 0 merge_flag, bits 1..3 merge_idx, bits 4..7 ref_idx_l0, bit 8 mvp_l0_flag, bits 9..11 PartMode, bits 12..13 partIdx,
// bit 15 valid, bits 16..17 inter_pred_idc (0 PRED_L0, 1 PRED_L1, 2 PRED_BI), bits 18..21 ref_idx_l1, bit 22 mvp_l1_flag, bits 24..26 the
// coding quadtree depth (unused by k_motion); mvd[X] = mvd_x (int16) | mvd_y (int16) << 16
struct MotionSyntax { uint32_t w0, mvd[2], pad; };

struct SaoParams {   // per CTB and colour component
  uint8_t type;      // 0 off, 1 band, 2 edge
  uint8_t band_or_class;
  int16_t offset[4]; // SaoOffsetVal[1..4]
  uint16_t pad;      // 12 bytes = 3 dwords (the parse kernel keeps them in three register lanes)
};

struct PicParams {
  // geometry
  int32_t width, height;          // coded luma size
  int32_t cwidth, cheight;        // coded chroma size (0 for 4:0:0)
  int32_t out_width, out_height;  // cropped luma size
  int32_t out_cwidth, out_cheight;
  int32_t crop_x, crop_y;         // luma offset of the conformance window
  int32_t chroma_format_idc, bit_depth_luma, bit_depth_chroma;
  int32_t log2_ctb, log2_min_cb, log2_min_tb, log2_max_tb, max_th_depth_intra;
  int32_t ctb_w, ctb_h, units_per_ctb_log2;  // units_per_ctb = 1 << units_per_ctb_log2
  // coding tools
  uint8_t sao_enabled, sign_data_hiding, transform_skip_enabled, cu_qp_delta_enabled;
  uint8_t transquant_bypass_enabled, strong_intra_smoothing, tiles_enabled, wpp;
  uint8_t lf_across_tiles, pcm_loop_filter_disabled;
  uint8_t sao_free_neighbours;   // 1: no slice / tile boundary restricts the SAO edge neighbours and no lossless CU can occur
  uint8_t scaling_lists;         // scaling_list_enabled_flag: the factor tables at off_scaling apply (8.6.4.2)
  uint8_t pcm_enabled, pcm_bd_luma, pcm_bd_chroma;   // pcm_enabled_flag, PcmBitDepthY / C
  uint8_t pcm_cb_range;          // Log2MinIpcmCbSizeY | Log2MaxIpcmCbSizeY << 4
  int32_t log2_min_cu_qp_delta_size;
  // buffers (byte offsets into the batch arena)
  uint64_t off_bitstream, bitstream_size;
  uint64_t off_ctb_ts_to_rs;      // uint16[ctbs]
  uint64_t off_ctb_info;          // CtbInfo[ctbs] (raster)
  uint64_t off_slices;            // SliceParams[nslices]
  uint64_t off_sao;               // SaoParams[ctbs*3]
  uint64_t off_scaling;           // uint8[2048] ScalingFactor m[y][x], intra matrices: component c at c * 336 (4x4, 8x8 at +16, 16x16 at +80), luma 32x32 at 1008;
                                  // 4:4:4 pictures: uint8[4096], the 32x32 matrices of Cb / Cr at 2048 / 3072
  uint64_t off_handoff;           // uint32[ctbs * HANDOFF_DWORDS]
  uint64_t off_u_size, off_u_flags, off_u_ipm, off_u_ipmc, off_u_qp;  // uint8[ctbs*units_per_ctb]
  uint64_t off_coeff[3];          // int16
  uint64_t off_rec[3];            // Pix (uint8 / uint16), coded size
  uint64_t off_out[3];            // Pix, cropped size
  uint64_t off_line[3];           // bottom sample row of every CTB row (ctb_h rows x rec_stride bytes): recon row hand-off
  uint32_t rec_stride[3];         // bytes
  uint32_t out_stride[3];         // bytes
  uint32_t first_row;             // index of this picture's first CTB row in the batch row table
  uint32_t num_slices;
  // ---- P pictures (0 / unused for an intra picture)
  uint8_t is_inter;               // the picture has P slices: motion syntax / motion field are allocated, k_motion and k_mc run
  uint8_t amp_enabled, max_th_depth_inter, log2_par_mrg_level;
  int32_t poc;                    // PicOrderCntVal
  uint32_t num_refs;              // entries of the RefFrame table
  uint8_t constrained_intra_pred; // constrained_intra_pred_flag in a picture with P / B slices: samples of units that are not intra coded are "not available" for intra prediction (8.4.4.2.2)
  uint8_t pad_inter;
  uint16_t lt_mask;               // bit k: slot k of the RefFrame table is a long-term reference picture (8.5.3.2.7 / 8.5.3.2.9: never scaled, only paired with long-term ones)
  uint64_t off_wp;                // WeightTable per slice with explicit weights (SliceParams::wp_index)
  uint64_t off_reftab;            // RefFrame[16]
  uint64_t off_msyn;              // MotionSyntax[ctbs * units_per_ctb]
  uint64_t off_mf;                // MotionUnit[ctbs * units_per_ctb]
};

struct CtbInfo {
  uint16_t slice_idx;   // index into the picture's SliceParams
  uint8_t avail;        // AV_* bits
  uint8_t tile_id;
};

struct Substream {
  uint32_t pic;
  uint32_t byte_start, byte_end;  // offsets inside the picture's bitstream blob
  uint32_t first_ctb_ts, num_ctbs;
  uint32_t slice_idx;
  int32_t dep_sub;                // substream of the CTB row above when it is a WPP predecessor, else -1
  uint32_t dep_len;               // number of CTBs in dep_sub
  uint8_t wpp_sync;               // 1: initialise contexts from dep_sub's table stored after its 2nd CTB (WPP, top-right available);
                                  // 2: from the table (and QpY) stored at dep_sub's END, which must be complete (dependent slice segment)
  uint8_t has_dependent;          // 1: another substream waits on this one's progress (table stored after the 2nd CTB); 2: ... (stored at the end)
  uint8_t last_in_slice_segment;  // 1: last CTB ends with end_of_slice_segment_flag = 1
  uint8_t pad;
  int32_t dependent;              // the substream whose dep_sub is this one (batch-global index), or -1
  uint32_t pad2;
};

struct ReconWave {  // one reconstruction wavefront: CTB rows first_row, first_row + stride, ... of the luma plane (comp 0) or of the
                    // two chroma planes together (comp 1) of one picture
  uint32_t pic, comp, first_row, stride;
  uint32_t base_row;    // batch row index of the picture's CTB row 0 (progress words are per batch row and component)
  uint32_t start_lag;   // CTBs the row above must be ahead before a row is started (>= 2)
  uint32_t pad0, pad1;
};

struct RowDesc {   // one CTB row of one picture, for the reconstruction wavefront
  uint32_t pic;
  uint32_t row;
};

// device-side error codes written to the batch status word (first error wins)
enum : int32_t {
  DEV_OK = 0,
  DEV_ERR_TERMINATE = 1,    // end_of_slice_segment_flag / end_of_subset_one_bit mismatch (desync)
  DEV_ERR_BITSTREAM_END = 2,
  DEV_ERR_SYNTAX = 3,       // value out of range (last position, cu_qp_delta, ...)
  DEV_ERR_TIMEOUT = 4       // a dependency wait exceeded its bound
};

struct ParseWave {  // one parser wavefront: substreams first, first + stride, ... < end (batch-global indices)
  uint32_t first, stride, end;
  uint32_t start_lag;   // CTBs the predecessor row must be ahead before a WPP row is started (>= 2)
};

struct ParseArgs {
  const PicParams* pics;
  const Substream* subs;
  const ParseWave* waves;
  uint32_t num_waves;
  uint8_t* arena;
  uint32_t* progress;   // per substream: CTBs completed
  uint8_t* ctx_store;   // per substream: CTX_STORE bytes, contexts after the 2nd CTB (WPP)
  uint32_t* ticket;
  int32_t* status;
  // ---- pool scheduler (throughput mode): rows are tasks, any wave runs any ready row ----
  uint32_t pool;         // 0: static assignment through `waves`; 1: work pool
  uint32_t queue_cap;    // power of two >= num_subs
  uint32_t num_subs;
  uint32_t* waitneed;    // per substream: 0, or the predecessor progress a suspended row waits for
  uint32_t* resume_k;    // per substream: CTB index to resume at (0 = fresh)
  uint32_t* queue;       // ready queue: substream index + 1, 0 = empty slot
  uint32_t* qctl;        // [0] head ticket, [1] tail ticket, [2] finished substreams
  uint32_t* saved;       // per substream: SAVE_DWORDS of suspended state
  uint32_t yield_ctbs;   // test knob (0 = off): a row yields after this many CTBs per activation
  uint32_t wake_hyst;    // a parked row is woken when its predecessor is this many CTBs beyond the minimum distance
  uint32_t general_chroma;   // 1: the batch holds a 4:2:2 or 4:4:4 picture (the parser build with the ChromaArrayType 2 / 3 paths is launched)
  uint32_t inter;            // 1: the batch holds a P picture (the parser build with the inter syntax is launched)
#ifdef HIPDEC_POOL_TRACE     // measurement build (tools/ab_variant.sh trace -DHIPDEC_POOL_TRACE): 8 x uint64 per pool wave, see parse_wave
  unsigned long long* trace;
#endif
};

}  // namespace hipdec
