// hevc_headers.h — host front end: NAL framing, parameter sets, slice segment headers and the
// substream / CTB tables the kernels consume.  The pixel path never runs on the host.
#pragma once
#include <cstddef>
#include <cstdint>
#include <deque>
#include <string>
#include <vector>
#include "heif_hipdec.h"
#include "hevc_device.h"

namespace hipdec {

// scaling_list_data() (7.3.4) as ScalingList[sizeId][matrixId] in RASTER order (index y * side + x of the 4x4 / 8x8 base
// list), with the DC values of the 16x16 and 32x32 lists kept apart (7.4.5)
struct ScalingLists {
  uint8_t l4[6][16], l8[6][64], l16[6][64], l32[6][64];
  uint8_t dc16[6], dc32[6];
};
void scaling_lists_default(ScalingLists& sl);   // Table 7-5 (flat 16) and Table 7-6

// one short_term_ref_pic_set() (7.3.7, 7.4.8): DeltaPocS0 / S1 with their used_by_curr_pic flags
struct StRps {
  int num_neg = 0, num_pos = 0;
  int delta_s0[16] = {0}, delta_s1[16] = {0};
  bool used_s0[16] = {false}, used_s1[16] = {false};
  // the long-term part of a slice header's reference picture set (7.3.6.1): PocLsbLt, UsedByCurrPicLt, delta_poc_msb_present_flag, DeltaPocMsbCycleLt (7-52)
  int num_lt = 0;
  int lt_poc_lsb[33] = {0}, lt_msb_cycle[33] = {0};
  bool lt_used[33] = {false}, lt_msb_present[33] = {false};
};

// A decoded picture a later P picture may reference: device pointers of its planes (coded size, deblocked, SAO applied), strides in bytes
// mf: its motion field on the device (0: an intra picture) - the collocated picture of temporal candidates (8.5.3.2.8)
// width .. log2_ctb: the format it was decoded in - a picture only predicts from pictures of its own format (a parameter set change without an IDR
// picture in between is refused: the kernels address references with the current picture's geometry)
// batch_item >= 0: a picture of the SAME launch set (an earlier sample of the chain the batch decodes, batch_layout.h): its planes are addressed once
// the arena is known (layout_batch_fill), and it is complete when the later picture's pixel stages start because those are launched picture by picture
struct RefPicture {
  int poc = 0; uint64_t plane[3] = {0, 0, 0}; uint32_t stride[3] = {0, 0, 0}; uint64_t mf = 0;
  int width = 0, height = 0, chroma_format_idc = 0, bit_depth_luma = 0, bit_depth_chroma = 0, log2_ctb = 0;
  int batch_item = -1;
  bool long_term = false;   // marked "used for long-term reference" (8.3.2): referenced by its POC LSBs / full POC, never scaled (8.5.3.2.7)
};

// What a decoder instance keeps between the samples of a sequence track (libheif/sequences/track_visual.cc:200-280 pushes them one by one):
// the picture order count state (8.3.1) and the decoded picture buffer.  nullptr where a single still is decoded: P slices are refused then.
struct SeqContext {
  bool first_picture = true;
  bool no_rasl_output = false;   // the IRAP picture decoded last had NoRaslOutputFlag = 1 (first picture, IDR, BLA): its RASL pictures are dropped (8.3.3)
  int prev_tid0_lsb = 0, prev_tid0_msb = 0;
  std::vector<RefPicture> dpb;
};

struct Sps {
  bool valid = false;
  int chroma_format_idc = 1, pic_width = 0, pic_height = 0;
  int conf_left = 0, conf_right = 0, conf_top = 0, conf_bottom = 0;
  int bit_depth_luma = 8, bit_depth_chroma = 8, log2_max_poc_lsb = 4;
  int log2_min_cb = 3, log2_ctb = 4, log2_min_tb = 2, log2_max_tb = 5;
  int max_th_depth_inter = 0, max_th_depth_intra = 0;
  bool scaling_list_enabled = false, amp = false, sao = false, pcm = false, strong_intra_smoothing = false;
  int pcm_bit_depth_luma = 8, pcm_bit_depth_chroma = 8, log2_min_pcm_cb = 3, log2_max_pcm_cb = 3;   // valid when pcm
  bool pcm_loop_filter_disabled = false;
  bool long_term_ref_pics_present = false, temporal_mvp = false, separate_colour_plane = false;
  int max_num_reorder = 0, max_dec_pic_buffering = 1;   // of the highest sub-layer (output order: C.5.2.2)
  int num_short_term_ref_pic_sets = 0, num_long_term_ref_pics_sps = 0;
  int lt_poc_lsb_sps[32] = {0};          // lt_ref_pic_poc_lsb_sps / used_by_curr_pic_lt_sps_flag: long-term candidates slice headers name by index
  bool lt_used_sps[32] = {false};
  std::vector<StRps> st_rps;            // the short-term reference picture sets of the SPS
  int colour_primaries = 2, transfer_characteristics = 2, matrix_coeffs = 2, full_range = 0;
  ScalingLists sl{};   // valid when scaling_list_enabled (explicit lists or the defaults)
};

struct Pps {
  bool valid = false;
  int sps_id = 0;
  bool dependent_slice_segments_enabled = false, output_flag_present = false, sign_data_hiding = false;
  bool cabac_init_present = false, constrained_intra_pred = false, transform_skip = false, cu_qp_delta = false;
  bool slice_chroma_qp_offsets_present = false, transquant_bypass = false, tiles = false, wpp = false;
  bool uniform_spacing = true, lf_across_tiles = true, lf_across_slices = false;
  bool deblocking_override_enabled = false, deblocking_disabled = false, scaling_list_data_present = false;
  bool slice_header_extension_present = false;
  int num_extra_slice_header_bits = 0, init_qp = 26, diff_cu_qp_delta_depth = 0;
  int cb_qp_offset = 0, cr_qp_offset = 0, beta_offset_div2 = 0, tc_offset_div2 = 0;
  int num_ref_idx_l0_default = 1, num_ref_idx_l1_default = 1, log2_par_mrg_level = 2;
  bool weighted_pred = false, weighted_bipred = false, lists_modification_present = false;
  int tile_cols = 1, tile_rows = 1;
  std::vector<int> col_width, row_height;  // explicit sizes when !uniform_spacing
  ScalingLists sl{};   // valid when scaling_list_data_present
};

struct ParsedSlice {
  SliceParams sp{};
  int segment_address = 0;
  bool dependent = false;  // dependent_slice_segment_flag: header fields (sp) are those of the preceding slice segment, sp.slice_addr_rs = SliceAddrRs
  size_t data_offset = 0;  // offset in the pushed blob of the first slice_segment_data byte
  size_t nal_end = 0;      // offset one past the slice NAL
  std::vector<uint32_t> entry_point_offsets;  // bytes, escaped domain
  bool ref_lt[2][16] = {{false}, {false}};   // ... it is a long-term reference picture
  int ref_poc[2][16] = {{0}, {0}};   // P / B slice: PicOrderCntVal of RefPicListX[i] (sp.ref_slot / ref_slot_l1 are filled once the picture's reference table is known)
  int col_poc = 0;                   // the collocated picture (slice_temporal_mvp_enabled_flag)
  bool has_weights = false;
  WeightTable weights{};             // valid when has_weights (sp.weighted)
};

struct ParsedPicture {
  Sps sps;
  Pps pps;
  std::vector<ParsedSlice> slices;
  bool uses_end_sync = false;   // some substream continues the contexts of the end of another (dependent slice segments)
  hipdec_image_info info{};
  std::vector<uint16_t> ts_to_rs;
  std::vector<CtbInfo> ctb_info;    // raster
  std::vector<Substream> subs;      // `pic` and dep indices are picture-local until the batch relocates them
  std::vector<SliceParams> slice_params;
  // ScalingFactor m[y][x] of the intra matrices, expanded (8.6.4.2 / 7.4.5): per component c the 4x4 (16 B), 8x8 (64 B) and
  // 16x16 (256 B) factors at c * 336, then the luma 32x32 factors (1024 B) at 1008; empty when scaling lists are off
  std::vector<uint8_t> scaling_tables;
  // ---- sequences: picture order count, the pictures its reference picture set keeps (8.3.2), the ones its P slices predict from
  int poc = 0, poc_lsb = 0, nal_type = 0, temporal_id = 0;
  bool pic_output = true;               // pic_output_flag (7.4.7.1); RASL pictures of a NoRaslOutputFlag IRAP are not output either (and not decoded: `skipped`)
  bool skipped = false;                 // a RASL picture associated with a CRA that started the sequence: its references are unavailable, 8.3.3 drops it
  bool is_inter = false;                // some slice is a P slice
  bool is_idr = false;
  std::vector<int> keep_pocs;           // every picture of the RPS (the DPB drops the others once this picture is decoded)
  std::vector<int> lt_pocs;             // the pictures of RefPicSetLtCurr / LtFoll: marked "used for long-term reference" from this picture on (8.3.2)
  std::vector<RefPicture> refs;         // the reference table of the picture (slots of SliceParams::ref_slot), at most 16
  std::vector<WeightTable> weight_tables;   // of the slices with explicit weights (SliceParams::wp_index)
  int max_num_reorder = 0, max_dec_pic_buffering = 1;   // sps_max_num_reorder_pics / sps_max_dec_pic_buffering_minus1 + 1 of the highest sub-layer
};

// What a decoder instance has been pushed and has not decoded yet, split into access units (7.4.2.4.4: a coded slice segment with
// first_slice_segment_in_pic_flag, or a parameter set / AUD / prefix SEI behind the last slice of a picture, starts the next one).  The first
// access unit is the still-image case (everything pushed before the decode, as decoder_libde265.cc:322-368 takes it); every later one is a
// sample of a sequence track (libheif/sequences/track_visual.cc:200-280 pushes them one by one; only a chunk's first sample carries the parameter
// sets, codecs/decoder.cc:422) and waits in `queue` with the parameter sets known at its push in front.  Pure host logic over untrusted bytes:
// the framing ([u32 BE length][NAL]...) has been validated by the caller; exercised on the CPU by tests/test_frontend_cpu.py and the header fuzzer.
struct SampleQueue {
  struct Sample { std::vector<uint8_t> blob; uintptr_t user_data = 0; bool has_vcl = false; };
  std::vector<uint8_t> first;            // the first access unit, as pushed
  bool first_has_vcl = false, first_closed = false;   // it holds a slice / is complete (a later access unit was pushed, or it was decoded)
  std::vector<uint8_t> param_sets;       // VPS / SPS / PPS seen so far (framed; a repeated one moves to the end: the newest wins when parsed)
  std::deque<Sample> queue;
  uintptr_t pending_user_data = 0, first_user_data = 0;   // push_data2's user_data: of the sample(s) the last push brought (decoder_libde265.cc:360, :417-419)
  size_t last_push_first = 0;            // queue index of the first sample the last push added to
  bool last_push_touched_first = false;
  bool last_open = false;                // the newest queued sample may still be continued by the next push (a picture pushed in pieces)
  void push(const uint8_t* p, size_t size);
  void set_user_data(uintptr_t user_data);
  void drop_front(size_t n);             // the n oldest samples were decoded (or refused)
};

// Parses one coded picture from libheif's plugin framing.  Returns a hipdec_status.
// `seq`: the sequence state of the decoder instance the item belongs to (read for the POC and the reference pictures; the caller commits
// the new POC state / DPB after a successful decode with seq_commit), or nullptr (a still: a P slice is HIPDEC_ERR_UNSUPPORTED).
int parse_picture(const uint8_t* blob, size_t size, uint64_t max_image_size_pixels, ParsedPicture& out, std::string& err, const SeqContext* seq = nullptr);
// after the picture was decoded: POC state, the DPB pruned to the picture's RPS; the caller appends the decoded picture itself
void seq_commit(SeqContext& seq, const ParsedPicture& pic, int nal_type_hint_unused = 0);

}  // namespace hipdec
