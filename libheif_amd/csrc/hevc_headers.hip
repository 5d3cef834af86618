// hevc_headers.hip — host front end (see hevc_headers.h).
//
// What it replaces in the reference: libde265's NAL / parameter-set / slice-header parsing behind
// de265_push_NAL + de265_decode (call sites libheif/plugins/decoder_libde265.cc:360, :402); the NAL
// framing contract is libheif/plugins/decoder_libde265.cc:322-368 and the emulation-prevention rule
// is the one libheif itself applies in libheif/codecs/hevc_boxes.cc:572-590.  Syntax per ITU-T
// H.265 clauses 7.3.1-7.3.6 and Annex E.
#include "hevc_headers.h"
#include <cstring>
#include <stdexcept>

namespace hipdec {
namespace {

struct ParseError : std::runtime_error {
  int code;
  ParseError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] void bad(const std::string& m) { throw ParseError(HIPDEC_ERR_BITSTREAM, m); }
[[noreturn]] void unsupported(const std::string& m) { throw ParseError(HIPDEC_ERR_UNSUPPORTED, "unsupported HEVC feature: " + m); }

// Reads bits from a NAL payload with emulation-prevention bytes skipped on the fly; byte_pos() is the
// position in the ESCAPED payload, which is what entry points and the device reader work with.
class NalReader {
 public:
  NalReader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  unsigned u(int bits)
  {
    unsigned v = 0;
    for (int i = 0; i < bits; i++) v = (v << 1) | bit();
    return v;
  }
  // 9.2: ue(v) with at most 31 leading zeros (values up to 2^32 - 2 never occur in a valid stream)
  uint32_t ue()
  {
    int lz = 0;
    while (bit() == 0) { if (++lz > 31) bad("exp-golomb code too long"); }
    return lz == 0 ? 0u : (uint32_t)(((uint64_t)1 << lz) - 1u + u(lz));
  }
  // ue(v) range-checked while still unsigned: every syntax element that is narrowed or used as an index goes through here
  int ue_max(uint32_t limit, const char* what)
  {
    uint32_t v = ue();
    if (v > limit) bad(std::string(what) + " out of range");
    return (int)v;
  }
  int se()
  {
    uint32_t k = ue();
    return (k & 1) ? (int)((k >> 1) + 1) : -(int)(k >> 1);
  }
  int se_range(int lo, int hi, const char* what)
  {
    int v = se();
    if (v < lo || v > hi) bad(std::string(what) + " out of range");
    return v;
  }
  void skip(size_t bits) { for (size_t i = 0; i < bits; i++) bit(); }
  bool aligned() const { return bitpos_ == 0; }
  size_t byte_pos() const { return pos_; }  // next unread escaped byte (valid when aligned)
 private:
  unsigned bit()
  {
    if (bitpos_ == 0) {
      if (pos_ >= n_) bad("read past the end of a NAL unit");
      uint8_t b = p_[pos_];
      if (zeros_ >= 2 && b == 3) {  // emulation_prevention_three_byte
        pos_++;
        zeros_ = 0;
        if (pos_ >= n_) bad("read past the end of a NAL unit");
        b = p_[pos_];
      }
      zeros_ = b == 0 ? zeros_ + 1 : 0;
      cur_ = b;
      pos_++;
    }
    unsigned v = (cur_ >> (7 - bitpos_)) & 1;
    bitpos_ = (bitpos_ + 1) & 7;
    return v;
  }
  const uint8_t* p_;
  size_t n_, pos_ = 0;
  int bitpos_ = 0, zeros_ = 0;
  uint8_t cur_ = 0;
};

int ceil_log2(int v) { int n = 0; while ((1 << n) < v) n++; return n; }

void skip_profile_tier_level(NalReader& r, int max_sub_layers_minus1)
{
  r.skip(88 + 8);
  bool prof[8], lev[8];
  for (int i = 0; i < max_sub_layers_minus1; i++) { prof[i] = r.u(1); lev[i] = r.u(1); }
  if (max_sub_layers_minus1 > 0) for (int i = max_sub_layers_minus1; i < 8; i++) r.skip(2);
  for (int i = 0; i < max_sub_layers_minus1; i++) { if (prof[i]) r.skip(88); if (lev[i]) r.skip(8); }
}

void skip_sub_layer_hrd(NalReader& r, int cpb_cnt, bool sub_pic)
{
  for (int i = 0; i <= cpb_cnt; i++) { r.ue(); r.ue(); if (sub_pic) { r.ue(); r.ue(); } r.u(1); }
}
void skip_hrd(NalReader& r, bool common, int max_sub)
{
  bool nal = false, vcl = false, sub_pic = false;
  if (common) {
    nal = r.u(1); vcl = r.u(1);
    if (nal || vcl) {
      sub_pic = r.u(1);
      if (sub_pic) r.skip(8 + 5 + 1 + 5);
      r.skip(8);
      if (sub_pic) r.skip(4);
      r.skip(15);
    }
  }
  for (int i = 0; i <= max_sub; i++) {
    bool fixed_general = r.u(1), fixed_cvs = true, low_delay = false;
    int cpb_cnt = 0;
    if (!fixed_general) fixed_cvs = r.u(1);
    if (fixed_cvs) r.ue(); else low_delay = r.u(1);
    if (!low_delay) cpb_cnt = r.ue_max(31, "cpb_cnt_minus1");
    if (nal) skip_sub_layer_hrd(r, cpb_cnt, sub_pic);
    if (vcl) skip_sub_layer_hrd(r, cpb_cnt, sub_pic);
  }
}

// 7.3.7: returns NumDeltaPocs of the parsed set; only the count matters for an intra-only decoder
// Table 7-6: default 8x8 intra lists (matrixId 0..2) and inter lists (3..5); symmetric, so raster == diagonal order
const uint8_t kDefaultIntra8[64] = {16, 16, 16, 16, 17, 18, 21, 24, 16, 16, 16, 16, 17, 19, 22, 25, 16, 16, 17, 18, 20, 22, 25, 29,
                                    16, 16, 18, 21, 24, 27, 31, 36, 17, 17, 20, 24, 30, 35, 41, 47, 18, 19, 22, 27, 35, 44, 54, 65,
                                    21, 22, 25, 31, 41, 54, 70, 88, 24, 25, 29, 36, 47, 65, 88, 115};
const uint8_t kDefaultInter8[64] = {16, 16, 16, 16, 17, 18, 20, 24, 16, 16, 16, 17, 18, 20, 24, 25, 16, 16, 17, 18, 20, 24, 25, 28,
                                    16, 17, 18, 20, 24, 25, 28, 33, 17, 18, 20, 24, 25, 28, 33, 41, 18, 20, 24, 25, 28, 33, 41, 54,
                                    20, 24, 25, 28, 33, 41, 54, 71, 24, 25, 28, 33, 41, 54, 71, 91};

// 6.5.3 up-right diagonal scan of a side x side block: position i -> raster index y * side + x
void diag_scan(int side, uint8_t* raster_of)
{
  int i = 0, x = 0, y = 0;
  while (i < side * side) {
    while (y >= 0) {
      if (x < side && y < side) raster_of[i++] = (uint8_t)(y * side + x);
      y--; x++;
    }
    y = x; x = 0;
  }
}

// 7.3.4 scaling_list_data() / 7.4.5 semantics
void parse_scaling_list_data(NalReader& r, ScalingLists& sl)
{
  uint8_t scan4[16], scan8[64];
  diag_scan(4, scan4);
  diag_scan(8, scan8);
  for (int size_id = 0; size_id < 4; size_id++)
    for (int m = 0; m < 6; m += (size_id == 3 ? 3 : 1)) {
      uint8_t* dst = size_id == 0 ? sl.l4[m] : (size_id == 1 ? sl.l8[m] : (size_id == 2 ? sl.l16[m] : sl.l32[m]));
      const int n = size_id == 0 ? 16 : 64;
      if (!r.u(1)) {                                        // scaling_list_pred_mode_flag = 0: copy
        const unsigned delta = r.ue();                      // scaling_list_pred_matrix_id_delta
        if (delta == 0) {                                   // ... the default list
          if (size_id == 0) memset(dst, 16, 16);
          else memcpy(dst, m < 3 ? kDefaultIntra8 : kDefaultInter8, 64);
          if (size_id == 2) sl.dc16[m] = 16;
          if (size_id == 3) sl.dc32[m] = 16;
        } else {                                            // ... an earlier list of the same size
          const long ref = (long)m - (long)delta * (size_id == 3 ? 3 : 1);
          if (ref < 0) bad("scaling_list_pred_matrix_id_delta out of range");
          const uint8_t* src = size_id == 0 ? sl.l4[ref] : (size_id == 1 ? sl.l8[ref] : (size_id == 2 ? sl.l16[ref] : sl.l32[ref]));
          memcpy(dst, src, (size_t)n);
          if (size_id == 2) sl.dc16[m] = sl.dc16[ref];
          if (size_id == 3) sl.dc32[m] = sl.dc32[ref];
        }
      } else {                                              // explicit coefficients, DPCM in diagonal order
        int next = 8;
        if (size_id > 1) {
          const int dc = r.se();                            // scaling_list_dc_coef_minus8
          if (dc < -7 || dc > 247) bad("scaling_list_dc_coef_minus8 out of range");
          next = dc + 8;
          if (size_id == 2) sl.dc16[m] = (uint8_t)next; else sl.dc32[m] = (uint8_t)next;
        }
        const uint8_t* scan = size_id == 0 ? scan4 : scan8;
        for (int i = 0; i < n; i++) {
          const int d = r.se();                             // scaling_list_delta_coef
          if (d < -128 || d > 127) bad("scaling_list_delta_coef out of range");
          next = (next + d + 256) % 256;
          dst[scan[i]] = (uint8_t)next;
        }
      }
    }
}

// 8.6.4.2 / 7.4.5: the factors m[y][x] a block of the given size and (intra) component uses
// inter: a second block of the same layout behind the first with the matrices of inter coded units (matrixId 3 .. 5, Table 7-4; P / B pictures are
// 4:0:0 / 4:2:0 here, so the block is 2048 bytes)
void build_scaling_tables(const ScalingLists& sl, bool chroma444, std::vector<uint8_t>& out, bool inter = false)
{
  const size_t block = chroma444 ? 4096 : 2048;
  out.assign(block * (inter ? 2 : 1), 16);
  for (int pass = 0; pass < (inter ? 2 : 1); pass++)
    for (int c = 0; c < 3; c++) {
      const int m = c + 3 * pass;
      uint8_t* t = out.data() + block * (size_t)pass + c * 336;
      memcpy(t, sl.l4[m], 16);
      memcpy(t + 16, sl.l8[m], 64);
      for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) t[80 + y * 16 + x] = sl.l16[m][(y >> 1) * 8 + (x >> 1)];
      t[80] = sl.dc16[m];
    }
  for (int pass = 0; pass < (inter ? 2 : 1); pass++) {
    uint8_t* t32 = out.data() + block * (size_t)pass + 1008;
    for (int y = 0; y < 32; y++)
      for (int x = 0; x < 32; x++) t32[y * 32 + x] = sl.l32[3 * pass][(y >> 2) * 8 + (x >> 2)];
    t32[0] = sl.dc32[3 * pass];
  }
  if (chroma444)   // 7.4.5: the 32x32 chroma matrices of a 4:4:4 picture are the component's 16x16 lists upsampled by 4, with the 16x16 DC
    for (int c = 1; c < 3; c++) {
      uint8_t* t = out.data() + 2048 + (c - 1) * 1024;
      for (int y = 0; y < 32; y++)
        for (int x = 0; x < 32; x++) t[y * 32 + x] = sl.l16[c][(y >> 2) * 8 + (x >> 2)];
      t[0] = sl.dc16[c];
    }
}

// short_term_ref_pic_set(idx) (7.3.7) with the derivation of 7.4.8 (7-61), (7-62) for inter-predicted sets.  `sets`: the sets parsed so far
// (the SPS's; a slice header parses set num_sets from them)
StRps parse_short_term_rps(NalReader& r, int idx, int num_sets, const std::vector<StRps>& sets)
{
  StRps out;
  const bool inter = idx != 0 ? r.u(1) : false;
  if (inter) {
    const int delta_idx_minus1 = idx == num_sets ? r.ue_max(63, "delta_idx_minus1") : 0;
    const int ref_idx = idx - (delta_idx_minus1 + 1);
    if (ref_idx < 0 || ref_idx >= (int)sets.size()) bad("short-term RPS refers to a missing set");
    const StRps& ref = sets[(size_t)ref_idx];
    const int sign = r.u(1);
    const int delta_rps = (1 - 2 * sign) * (r.ue_max(32767, "abs_delta_rps_minus1") + 1);
    const int n_ref = ref.num_neg + ref.num_pos;
    bool used[33], use_delta[33];
    for (int j = 0; j <= n_ref; j++) { used[j] = r.u(1); use_delta[j] = true; if (!used[j]) use_delta[j] = r.u(1); }
    int i = 0;
    auto push0 = [&](int d, bool u) { if (i >= 16) bad("short-term RPS holds more than 16 pictures"); out.delta_s0[i] = d; out.used_s0[i] = u; i++; };
    for (int j = ref.num_pos - 1; j >= 0; j--) { const int d = ref.delta_s1[j] + delta_rps; if (d < 0 && use_delta[ref.num_neg + j]) push0(d, used[ref.num_neg + j]); }
    if (delta_rps < 0 && use_delta[n_ref]) push0(delta_rps, used[n_ref]);
    for (int j = 0; j < ref.num_neg; j++) { const int d = ref.delta_s0[j] + delta_rps; if (d < 0 && use_delta[j]) push0(d, used[j]); }
    out.num_neg = i;
    i = 0;
    auto push1 = [&](int d, bool u) { if (i >= 16) bad("short-term RPS holds more than 16 pictures"); out.delta_s1[i] = d; out.used_s1[i] = u; i++; };
    for (int j = ref.num_neg - 1; j >= 0; j--) { const int d = ref.delta_s0[j] + delta_rps; if (d > 0 && use_delta[j]) push1(d, used[j]); }
    if (delta_rps > 0 && use_delta[n_ref]) push1(delta_rps, used[n_ref]);
    for (int j = 0; j < ref.num_pos; j++) { const int d = ref.delta_s1[j] + delta_rps; if (d > 0 && use_delta[ref.num_neg + j]) push1(d, used[ref.num_neg + j]); }
    out.num_pos = i;
  } else {
    out.num_neg = r.ue_max(16, "num_negative_pics"); out.num_pos = r.ue_max(16, "num_positive_pics");
    int poc = 0;
    for (int i = 0; i < out.num_neg; i++) { poc -= r.ue_max(32767, "delta_poc_s0_minus1") + 1; out.delta_s0[i] = poc; out.used_s0[i] = r.u(1); }
    poc = 0;
    for (int i = 0; i < out.num_pos; i++) { poc += r.ue_max(32767, "delta_poc_s1_minus1") + 1; out.delta_s1[i] = poc; out.used_s1[i] = r.u(1); }
  }
  if (out.num_neg + out.num_pos > 16) bad("short-term RPS holds more than 16 pictures");
  return out;
}

void parse_sps(NalReader& r, Sps& s)
{
  r.skip(4);
  int max_sub_layers_minus1 = r.u(3);
  r.skip(1);
  skip_profile_tier_level(r, max_sub_layers_minus1);
  r.ue_max(15, "sps_seq_parameter_set_id");
  s.chroma_format_idc = r.ue_max(3, "chroma_format_idc");
  if (s.chroma_format_idc == 3) s.separate_colour_plane = r.u(1);
  // A.4: no level allows a picture side beyond 8 * sqrt(MaxLumaPs) = 16888; 65535 keeps every later product inside 32 bits
  s.pic_width = r.ue_max(65535, "pic_width_in_luma_samples");
  s.pic_height = r.ue_max(65535, "pic_height_in_luma_samples");
  if (r.u(1)) {
    s.conf_left = r.ue_max(65535, "conf_win_left_offset"); s.conf_right = r.ue_max(65535, "conf_win_right_offset");
    s.conf_top = r.ue_max(65535, "conf_win_top_offset"); s.conf_bottom = r.ue_max(65535, "conf_win_bottom_offset");
  }
  s.bit_depth_luma = r.ue_max(8, "bit_depth_luma_minus8") + 8;
  s.bit_depth_chroma = r.ue_max(8, "bit_depth_chroma_minus8") + 8;
  s.log2_max_poc_lsb = r.ue_max(12, "log2_max_pic_order_cnt_lsb_minus4") + 4;
  bool sub_layer_ordering = r.u(1);
  for (int i = sub_layer_ordering ? 0 : max_sub_layers_minus1; i <= max_sub_layers_minus1; i++) {
    s.max_dec_pic_buffering = r.ue_max(15, "sps_max_dec_pic_buffering_minus1") + 1;
    s.max_num_reorder = r.ue_max(15, "sps_max_num_reorder_pics");
    r.ue();   // sps_max_latency_increase_plus1
  }
  s.log2_min_cb = r.ue_max(3, "log2_min_luma_coding_block_size_minus3") + 3;
  s.log2_ctb = s.log2_min_cb + r.ue_max(3, "log2_diff_max_min_luma_coding_block_size");
  s.log2_min_tb = r.ue_max(3, "log2_min_luma_transform_block_size_minus2") + 2;
  s.log2_max_tb = s.log2_min_tb + r.ue_max(3, "log2_diff_max_min_luma_transform_block_size");
  s.max_th_depth_inter = r.ue_max(4, "max_transform_hierarchy_depth_inter");
  s.max_th_depth_intra = r.ue_max(4, "max_transform_hierarchy_depth_intra");
  s.scaling_list_enabled = r.u(1);
  if (s.scaling_list_enabled) {
    scaling_lists_default(s.sl);                         // sps_infer: Table 7-5 / 7-6 unless lists follow
    if (r.u(1)) parse_scaling_list_data(r, s.sl);        // sps_scaling_list_data_present_flag
  }
  s.amp = r.u(1);
  s.sao = r.u(1);
  s.pcm = r.u(1);
  if (s.pcm) {   // 7.3.2.2.1 / 7.4.3.2.1
    s.pcm_bit_depth_luma = (int)r.u(4) + 1;
    s.pcm_bit_depth_chroma = (int)r.u(4) + 1;
    if (s.pcm_bit_depth_luma > s.bit_depth_luma || s.pcm_bit_depth_chroma > s.bit_depth_chroma) bad("PCM sample bit depth above the picture's bit depth");
    s.log2_min_pcm_cb = r.ue_max(2, "log2_min_pcm_luma_coding_block_size_minus3") + 3;
    s.log2_max_pcm_cb = s.log2_min_pcm_cb + r.ue_max(2, "log2_diff_max_min_pcm_luma_coding_block_size");
    const int lo = s.log2_min_cb < 5 ? s.log2_min_cb : 5, hi = s.log2_ctb < 5 ? s.log2_ctb : 5;
    if (s.log2_min_pcm_cb < lo || s.log2_min_pcm_cb > hi || s.log2_max_pcm_cb > hi) bad("PCM coding block size range outside the coding block sizes");
    s.pcm_loop_filter_disabled = r.u(1) != 0;
  }
  s.num_short_term_ref_pic_sets = r.ue_max(64, "num_short_term_ref_pic_sets");
  s.st_rps.clear();
  for (int i = 0; i < s.num_short_term_ref_pic_sets; i++) s.st_rps.push_back(parse_short_term_rps(r, i, s.num_short_term_ref_pic_sets, s.st_rps));
  s.long_term_ref_pics_present = r.u(1);
  if (s.long_term_ref_pics_present) {
    s.num_long_term_ref_pics_sps = r.ue_max(32, "num_long_term_ref_pics_sps");
    for (int i = 0; i < s.num_long_term_ref_pics_sps; i++) { s.lt_poc_lsb_sps[i] = (int)r.u(s.log2_max_poc_lsb); s.lt_used_sps[i] = r.u(1); }
  }
  s.temporal_mvp = r.u(1);
  s.strong_intra_smoothing = r.u(1);
  if (r.u(1)) {  // vui_parameters(), Annex E.2.1
    if (r.u(1)) { if (r.u(8) == 255) r.skip(32); }
    if (r.u(1)) r.skip(1);
    if (r.u(1)) {
      r.skip(3);
      s.full_range = r.u(1);
      if (r.u(1)) { s.colour_primaries = r.u(8); s.transfer_characteristics = r.u(8); s.matrix_coeffs = r.u(8); }
    }
    if (r.u(1)) { r.ue(); r.ue(); }
    r.skip(3);
    if (r.u(1)) { r.ue(); r.ue(); r.ue(); r.ue(); }
    if (r.u(1)) {
      r.skip(64);
      if (r.u(1)) r.ue();
      if (r.u(1)) skip_hrd(r, true, max_sub_layers_minus1);
    }
    if (r.u(1)) { r.skip(3); r.ue(); r.ue(); r.ue(); r.ue(); r.ue(); }
  }
  if (r.u(1)) {  // sps_extension_present_flag
    bool range_ext = r.u(1);
    r.skip(7);
    if (range_ext) {
      unsigned any = r.u(9);
      if (any) unsupported("range-extension coding tools");
    }
  }
  if (s.chroma_format_idc == 3 && s.separate_colour_plane) unsupported("4:4:4 with separate colour planes");
  if (s.bit_depth_luma > 12 || s.bit_depth_chroma > 12) unsupported("bit depth above 12");
  // 7.4.3.2.1: 3 <= MinCbLog2SizeY <= CtbLog2SizeY, CtbLog2SizeY in 4..6; 2 <= MinTbLog2SizeY < MinCbLog2SizeY;
  // MinTbLog2SizeY <= MaxTbLog2SizeY <= Min(CtbLog2SizeY, 5); max_transform_hierarchy_depth_* <= CtbLog2SizeY - MinTbLog2SizeY
  if (s.log2_ctb < 4 || s.log2_ctb > 6 || s.log2_min_cb < 3 || s.log2_min_cb > s.log2_ctb) bad("coding block sizes out of range");
  if (s.log2_min_tb < 2 || s.log2_min_tb >= s.log2_min_cb || s.log2_max_tb < s.log2_min_tb || s.log2_max_tb > 5 || s.log2_max_tb > s.log2_ctb)
    bad("transform block sizes out of range");
  if (s.max_th_depth_intra > s.log2_ctb - s.log2_min_tb || s.max_th_depth_inter > s.log2_ctb - s.log2_min_tb)
    bad("max_transform_hierarchy_depth out of range");
  if (s.pic_width <= 0 || s.pic_height <= 0 || (s.pic_width & ((1 << s.log2_min_cb) - 1)) || (s.pic_height & ((1 << s.log2_min_cb) - 1)))
    bad("picture size is not a multiple of the minimum coding block size");
  {
    const int64_t sub_w = s.chroma_format_idc == 1 || s.chroma_format_idc == 2 ? 2 : 1, sub_h = s.chroma_format_idc == 1 ? 2 : 1;
    if (sub_w * ((int64_t)s.conf_left + s.conf_right) >= s.pic_width || sub_h * ((int64_t)s.conf_top + s.conf_bottom) >= s.pic_height)
      bad("conformance window larger than the picture");
  }
  s.valid = true;
}

void parse_pps(NalReader& r, Pps& p)
{
  r.ue_max(63, "pps_pic_parameter_set_id");
  p.sps_id = r.ue_max(15, "pps_seq_parameter_set_id");
  p.dependent_slice_segments_enabled = r.u(1);
  p.output_flag_present = r.u(1);
  p.num_extra_slice_header_bits = r.u(3);
  p.sign_data_hiding = r.u(1);
  p.cabac_init_present = r.u(1);
  p.num_ref_idx_l0_default = r.ue_max(14, "num_ref_idx_l0_default_active_minus1") + 1;
  p.num_ref_idx_l1_default = r.ue_max(14, "num_ref_idx_l1_default_active_minus1") + 1;
  p.init_qp = 26 + r.se_range(-(26 + 6 * 8), 25, "init_qp_minus26");
  p.constrained_intra_pred = r.u(1);
  p.transform_skip = r.u(1);
  p.cu_qp_delta = r.u(1);
  if (p.cu_qp_delta) p.diff_cu_qp_delta_depth = r.ue_max(3, "diff_cu_qp_delta_depth");
  p.cb_qp_offset = r.se_range(-12, 12, "pps_cb_qp_offset");
  p.cr_qp_offset = r.se_range(-12, 12, "pps_cr_qp_offset");
  p.slice_chroma_qp_offsets_present = r.u(1);
  p.weighted_pred = r.u(1); p.weighted_bipred = r.u(1);
  p.transquant_bypass = r.u(1);
  p.tiles = r.u(1);
  p.wpp = r.u(1);
  if (p.tiles) {
    p.tile_cols = r.ue_max(19, "num_tile_columns_minus1") + 1;    // A.4.1: at most 20 x 22 tiles at any level
    p.tile_rows = r.ue_max(21, "num_tile_rows_minus1") + 1;
    p.uniform_spacing = r.u(1);
    if (!p.uniform_spacing) {
      // a CTB row / column holds at most 65535 / 16 CTBs; the sums are checked against the picture in build_tiles
      for (int i = 0; i < p.tile_cols - 1; i++) p.col_width.push_back(r.ue_max(4095, "column_width_minus1") + 1);
      for (int i = 0; i < p.tile_rows - 1; i++) p.row_height.push_back(r.ue_max(4095, "row_height_minus1") + 1);
    }
    p.lf_across_tiles = r.u(1);
  }
  p.lf_across_slices = r.u(1);
  if (r.u(1)) {  // deblocking_filter_control_present_flag
    p.deblocking_override_enabled = r.u(1);
    p.deblocking_disabled = r.u(1);
    if (!p.deblocking_disabled) { p.beta_offset_div2 = r.se_range(-6, 6, "pps_beta_offset_div2"); p.tc_offset_div2 = r.se_range(-6, 6, "pps_tc_offset_div2"); }
  }
  p.scaling_list_data_present = r.u(1);
  if (p.scaling_list_data_present) { scaling_lists_default(p.sl); parse_scaling_list_data(r, p.sl); }
  p.lists_modification_present = r.u(1);
  p.log2_par_mrg_level = r.ue_max(4, "log2_parallel_merge_level_minus2") + 2;
  p.slice_header_extension_present = r.u(1);
  if (r.u(1)) {
    bool range_ext = r.u(1);
    r.skip(7);
    if (range_ext) {
      // 7.3.2.3.2 pps_range_extension(): accepted when it switches nothing on (encoders of the format-range-extension profiles write it for
      // 4:2:2 / 4:4:4 streams); every tool it can enable is outside this decoder
      if (p.transform_skip && r.ue_max(3, "log2_max_transform_skip_block_size_minus2") != 0) unsupported("transform skip blocks larger than 4x4");
      if (r.u(1)) unsupported("cross-component prediction");
      if (r.u(1)) unsupported("chroma QP offset lists");
      if (r.ue_max(6, "log2_sao_offset_scale_luma") != 0 || r.ue_max(6, "log2_sao_offset_scale_chroma") != 0) unsupported("SAO offset scaling");
    }
  }
  p.valid = true;
}

struct TileLayout {
  std::vector<int> col_bd, row_bd;
  std::vector<uint16_t> rs_to_ts, ts_to_rs;
  std::vector<uint8_t> tile_id_rs;
};

TileLayout build_tiles(const Sps& s, const Pps& p, int ctb_w, int ctb_h)  // 6.5.1
{
  TileLayout t;
  int nc = p.tile_cols, nr = p.tile_rows;
  if (nc < 1 || nr < 1 || nc > 20 || nr > 22 || nc > ctb_w || nr > ctb_h) bad("more tiles than CTBs");
  if (!p.uniform_spacing && ((int)p.col_width.size() != nc - 1 || (int)p.row_height.size() != nr - 1)) bad("tile size lists are incomplete");
  std::vector<int> cw(nc), rh(nr);
  if (p.uniform_spacing) {
    for (int i = 0; i < nc; i++) cw[i] = ((i + 1) * ctb_w) / nc - (i * ctb_w) / nc;
    for (int j = 0; j < nr; j++) rh[j] = ((j + 1) * ctb_h) / nr - (j * ctb_h) / nr;
  } else {
    int acc = 0;
    for (int i = 0; i < nc - 1; i++) { cw[i] = p.col_width[i]; acc += cw[i]; if (cw[i] < 1 || acc >= ctb_w) bad("tile sizes exceed the picture"); }
    cw[nc - 1] = ctb_w - acc;
    acc = 0;
    for (int j = 0; j < nr - 1; j++) { rh[j] = p.row_height[j]; acc += rh[j]; if (rh[j] < 1 || acc >= ctb_h) bad("tile sizes exceed the picture"); }
    rh[nr - 1] = ctb_h - acc;
    if (cw[nc - 1] <= 0 || rh[nr - 1] <= 0) bad("tile sizes exceed the picture");
  }
  t.col_bd.assign(nc + 1, 0); t.row_bd.assign(nr + 1, 0);
  for (int i = 0; i < nc; i++) t.col_bd[i + 1] = t.col_bd[i] + cw[i];
  for (int j = 0; j < nr; j++) t.row_bd[j + 1] = t.row_bd[j] + rh[j];
  int n = ctb_w * ctb_h;
  t.rs_to_ts.resize(n); t.ts_to_rs.resize(n); t.tile_id_rs.resize(n);
  int ts = 0;
  for (int j = 0; j < nr; j++)
    for (int i = 0; i < nc; i++)
      for (int y = t.row_bd[j]; y < t.row_bd[j + 1]; y++)
        for (int x = t.col_bd[i]; x < t.col_bd[i + 1]; x++) {
          int rs = y * ctb_w + x;
          t.rs_to_ts[rs] = (uint16_t)ts;
          t.ts_to_rs[ts] = (uint16_t)rs;
          t.tile_id_rs[rs] = (uint8_t)(j * nc + i);
          ts++;
        }
  (void)s;
  return t;
}

}  // namespace

void scaling_lists_default(ScalingLists& sl)
{
  for (int m = 0; m < 6; m++) {
    memset(sl.l4[m], 16, 16);
    const uint8_t* def = m < 3 ? kDefaultIntra8 : kDefaultInter8;
    memcpy(sl.l8[m], def, 64); memcpy(sl.l16[m], def, 64); memcpy(sl.l32[m], def, 64);
    sl.dc16[m] = 16; sl.dc32[m] = 16;
  }
}

int parse_picture(const uint8_t* blob, size_t size, uint64_t max_pixels, ParsedPicture& out, std::string& err, const SeqContext* seq)
{
  try {
    Sps sps_tab[16];
    Pps pps_tab[64];
    bool have_picture = false;
    TileLayout tiles;
    int ctb_w = 0, ctb_h = 0, n_ctb = 0;
    std::vector<int> ctb_slice;       // slice index per CTB (raster), -1 = not covered
    std::vector<int> ctb_slice_addr;  // SliceAddrRs per CTB
    StRps pic_rps;                     // the RPS of the picture (every slice segment header repeats it)
    std::vector<int> pic_lt_curr;      // RefPicSetLtCurr as POCs
    int pic_poc_lsb = 0, pic_nal_type = 0, pic_tid = 0;
    out.is_inter = false; out.refs.clear(); out.keep_pocs.clear();
    out.pic_output = true; out.skipped = false;
    size_t ptr = 0;
    while (ptr < size) {
      if (size - ptr < 4) { err = "truncated NAL length field"; return HIPDEC_ERR_END_OF_DATA; }
      uint32_t nal_size = ((uint32_t)blob[ptr] << 24) | ((uint32_t)blob[ptr + 1] << 16) | ((uint32_t)blob[ptr + 2] << 8) | blob[ptr + 3];
      ptr += 4;
      if (nal_size > size - ptr) { err = "NAL size exceeds the pushed data"; return HIPDEC_ERR_END_OF_DATA; }
      const uint8_t* nal = blob + ptr;
      size_t nal_off = ptr;
      ptr += nal_size;
      if (nal_size < 2) continue;
      int type = (nal[0] >> 1) & 63;
      if (type == 33) {
        NalReader r(nal + 2, nal_size - 2);
        Sps s;
        // the id sits behind the profile_tier_level: parse into a temporary, then file it
        NalReader peek(nal + 2, nal_size - 2);
        peek.skip(4); int msl = peek.u(3); peek.skip(1); skip_profile_tier_level(peek, msl);
        unsigned id = peek.ue();
        if (id > 15) bad("sps id out of range");
        parse_sps(r, s);
        sps_tab[id] = s;
      } else if (type == 34) {
        NalReader r(nal + 2, nal_size - 2);
        NalReader peek(nal + 2, nal_size - 2);
        unsigned id = peek.ue();
        if (id > 63) bad("pps id out of range");
        Pps p;
        parse_pps(r, p);
        pps_tab[id] = p;
      } else if (type <= 21 && !(type > 9 && type < 16)) {
        // ---- slice segment header 7.3.6.1 ----
        NalReader r(nal + 2, nal_size - 2);
        bool first = r.u(1);
        if (type >= 16 && type <= 23) r.skip(1);
        const int pps_id = r.ue_max(63, "slice_pic_parameter_set_id");
        if (!pps_tab[pps_id].valid) bad("slice refers to a missing PPS");
        const Pps& p = pps_tab[pps_id];
        if (p.sps_id < 0 || p.sps_id > 15 || !sps_tab[p.sps_id].valid) bad("PPS refers to a missing SPS");
        const Sps& s = sps_tab[p.sps_id];
        if (first) {
          if (have_picture) unsupported("more than one coded picture in an item");
          out.sps = s; out.pps = p;
          uint64_t px = (uint64_t)s.pic_width * (uint64_t)s.pic_height;
          if (max_pixels && px > max_pixels) { err = "coded image size exceeds max_image_size_pixels"; return HIPDEC_ERR_LIMIT; }
          ctb_w = (s.pic_width + (1 << s.log2_ctb) - 1) >> s.log2_ctb;
          ctb_h = (s.pic_height + (1 << s.log2_ctb) - 1) >> s.log2_ctb;
          if ((int64_t)ctb_w * (int64_t)ctb_h > 65535) unsupported("more than 65535 CTBs in one picture");
          n_ctb = ctb_w * ctb_h;
          tiles = build_tiles(s, p, ctb_w, ctb_h);
          ctb_slice.assign(n_ctb, -1); ctb_slice_addr.assign(n_ctb, -1);
          have_picture = true;
        } else if (!have_picture) bad("slice segment before the first slice segment of the picture");
        const Sps& S = out.sps; const Pps& P = out.pps;
        ParsedSlice sl;
        bool dependent = false;
        if (!first) {
          if (P.dependent_slice_segments_enabled) dependent = r.u(1);
          sl.segment_address = (int)r.u(ceil_log2(n_ctb));
          if (sl.segment_address >= n_ctb) bad("slice_segment_address out of range");
        }
        if (dependent) {   // 7.3.6.1: the remaining header fields are inferred from the preceding slice segment (7.4.7.1)
          if (out.slices.empty()) bad("dependent slice segment without a preceding slice segment");
          sl.sp = out.slices.back().sp;
          sl.dependent = true;
        } else {
        r.skip(P.num_extra_slice_header_bits);
        const unsigned slice_type = r.ue_max(2, "slice_type");
        if (slice_type != 2 && !seq) unsupported("non-intra slice (slice_type " + std::to_string(slice_type) + ") outside a sequence");
        const bool pic_output_flag = P.output_flag_present ? r.u(1) : true;
        if (S.separate_colour_plane) r.skip(2);
        int poc_lsb = 0;
        bool slice_tmvp = false;
        StRps rps;
        const bool idr = type == 19 || type == 20;
        if (!idr) {
          poc_lsb = (int)r.u(S.log2_max_poc_lsb);
          const bool st_sps = r.u(1);
          if (!st_sps) rps = parse_short_term_rps(r, S.num_short_term_ref_pic_sets, S.num_short_term_ref_pic_sets, S.st_rps);
          else {
            if (S.num_short_term_ref_pic_sets == 0) bad("short_term_ref_pic_set_sps_flag without a set in the SPS");
            const int idx = S.num_short_term_ref_pic_sets > 1 ? (int)r.u(ceil_log2(S.num_short_term_ref_pic_sets)) : 0;
            if (idx >= S.num_short_term_ref_pic_sets) bad("short_term_ref_pic_set_idx out of range");
            rps = S.st_rps[(size_t)idx];
          }
          if (S.long_term_ref_pics_present) {   // 7.3.6.1 / 7.4.7.1: the long-term pictures of the RPS
            const int lt_sps = S.num_long_term_ref_pics_sps > 0 ? r.ue_max((uint32_t)S.num_long_term_ref_pics_sps, "num_long_term_sps") : 0;
            const int lt_pics = r.ue_max(32, "num_long_term_pics");
            if (lt_sps + lt_pics > 32) bad("more than 32 long-term reference pictures");
            rps.num_lt = lt_sps + lt_pics;
            for (int i = 0; i < rps.num_lt; i++) {
              if (i < lt_sps) {
                const int idx = S.num_long_term_ref_pics_sps > 1 ? (int)r.u(ceil_log2(S.num_long_term_ref_pics_sps)) : 0;
                if (idx >= S.num_long_term_ref_pics_sps) bad("lt_idx_sps out of range");
                rps.lt_poc_lsb[i] = S.lt_poc_lsb_sps[idx]; rps.lt_used[i] = S.lt_used_sps[idx];
              } else { rps.lt_poc_lsb[i] = (int)r.u(S.log2_max_poc_lsb); rps.lt_used[i] = r.u(1); }
              rps.lt_msb_present[i] = r.u(1);
              const int cycle = rps.lt_msb_present[i] ? r.ue_max(1 << 20, "delta_poc_msb_cycle_lt") : 0;
              rps.lt_msb_cycle[i] = (i == 0 || i == lt_sps) ? cycle : cycle + rps.lt_msb_cycle[i - 1];
            }
          }
          if (S.temporal_mvp) slice_tmvp = r.u(1);
        }
        if (first) {   // 8.3.1 picture order count, 8.3.2: the pictures the RPS names (sequence mode)
          out.is_idr = idr;
          out.poc = idr ? 0 : poc_lsb;   // (a first IRAP picture: PicOrderCntMsb = 0, 8.3.1; what a still / the first sample of a track is committed with)
          out.keep_pocs.clear();
          pic_poc_lsb = poc_lsb; pic_nal_type = type; pic_tid = (nal[1] & 7) - 1;
          out.pic_output = pic_output_flag;
          if (seq) {
            const bool irap = type >= 16 && type <= 23;
            // 8.3.3: the RASL pictures of an IRAP picture with NoRaslOutputFlag = 1 (a CRA that starts the sequence, a BLA) reference pictures that
            // precede it in decoding order and are not there: they are neither decoded nor output
            if ((type == 8 || type == 9) && seq->no_rasl_output) { out.skipped = true; out.pic_output = false; out.nal_type = type; return HIPDEC_OK; }
            if (idr || (irap && seq->first_picture)) out.poc = idr ? 0 : poc_lsb;
            else {
              const int max_lsb = 1 << S.log2_max_poc_lsb;
              int msb = seq->prev_tid0_msb;
              if (poc_lsb < seq->prev_tid0_lsb && seq->prev_tid0_lsb - poc_lsb >= max_lsb / 2) msb += max_lsb;
              else if (poc_lsb > seq->prev_tid0_lsb && poc_lsb - seq->prev_tid0_lsb > max_lsb / 2) msb -= max_lsb;
              out.poc = msb + poc_lsb;
            }
            out.lt_pocs.clear();
            if (!idr) {
              // 8.3.2: the long-term subsets first - a candidate is ANY reference picture of the DPB, named by its POC LSBs or (delta_poc_msb_present_flag) its
              // whole POC -, then the short-term ones among the pictures that are not long-term reference pictures
              const int max_lsb = 1 << S.log2_max_poc_lsb;
              pic_lt_curr.clear();
              for (int i = 0; i < rps.num_lt; i++) {
                int poc_lt = rps.lt_poc_lsb[i];
                if (rps.lt_msb_present[i]) poc_lt += out.poc - rps.lt_msb_cycle[i] * max_lsb - (out.poc & (max_lsb - 1));
                const RefPicture* found = nullptr;
                for (const RefPicture& rp : seq->dpb)
                  if (!found && (rps.lt_msb_present[i] ? rp.poc == poc_lt : (rp.poc & (max_lsb - 1)) == poc_lt)) found = &rp;
                if (found) { out.keep_pocs.push_back(found->poc); out.lt_pocs.push_back(found->poc); }
                if (rps.lt_used[i]) {
                  if (!found) bad("long-term reference picture with POC (LSBs) " + std::to_string(poc_lt) + " is missing");
                  pic_lt_curr.push_back(found->poc);
                }
              }
              for (int i = 0; i < rps.num_neg; i++) out.keep_pocs.push_back(out.poc + rps.delta_s0[i]);
              for (int i = 0; i < rps.num_pos; i++) out.keep_pocs.push_back(out.poc + rps.delta_s1[i]);
            }
          }
          pic_rps = rps;
        }
        if (S.sao) { sl.sp.sao_luma = r.u(1); if (S.chroma_format_idc) sl.sp.sao_chroma = r.u(1); }
        if (slice_type != 2) {   // 7.3.6.1, P / B slice
          const bool is_b = slice_type == 0;
          if (S.chroma_format_idc > 1) unsupported("P / B slices of a 4:2:2 / 4:4:4 picture");
          int num_ref[2] = {P.num_ref_idx_l0_default, is_b ? P.num_ref_idx_l1_default : 0};
          if (r.u(1)) {   // num_ref_idx_active_override_flag
            num_ref[0] = r.ue_max(14, "num_ref_idx_l0_active_minus1") + 1;
            if (is_b) num_ref[1] = r.ue_max(14, "num_ref_idx_l1_active_minus1") + 1;
          }
          // RefPicSetStCurrBefore / After / LtCurr of the PICTURE's RPS (8.3.2); RefPicListTemp0 = Before, After, Lt; RefPicListTemp1 = After, Before, Lt (8.3.4)
          auto is_lt = [&](int poc) {   // marked long-term before this picture, or by this picture's RPS
            for (int q : out.lt_pocs) if (q == poc) return true;
            for (const RefPicture& rp : seq->dpb) if (rp.poc == poc) return rp.long_term;
            return false;
          };
          std::vector<int> before, after;
          for (int i = 0; i < pic_rps.num_neg; i++) if (pic_rps.used_s0[i]) before.push_back(out.poc + pic_rps.delta_s0[i]);
          for (int i = 0; i < pic_rps.num_pos; i++) if (pic_rps.used_s1[i]) after.push_back(out.poc + pic_rps.delta_s1[i]);
          const int n_st = (int)(before.size() + after.size());
          const int total = n_st + (int)pic_lt_curr.size();   // NumPicTotalCurr
          if (total == 0) bad("P / B slice without a reference picture");
          for (int k = 0; k < n_st; k++) {
            const int poc = k < (int)before.size() ? before[(size_t)k] : after[(size_t)k - before.size()];
            bool have = false;
            for (const RefPicture& rp : seq->dpb) if (rp.poc == poc) have = true;
            if (have && is_lt(poc)) have = false;   // (a short-term entry never names a long-term reference picture)
            if (!have) bad("reference picture with POC " + std::to_string(poc) + " is missing");
          }
          int entries[2][16];
          bool modified[2] = {false, false};
          if (P.lists_modification_present && total > 1)
            for (int X = 0; X < (is_b ? 2 : 1); X++) {
              modified[X] = r.u(1);
              if (modified[X]) for (int i = 0; i < num_ref[X]; i++) entries[X][i] = (int)r.u(ceil_log2(total));
            }
          for (int X = 0; X < (is_b ? 2 : 1); X++) {
            std::vector<int> temp;
            const std::vector<int>& first = X ? after : before;
            const std::vector<int>& second = X ? before : after;
            temp.insert(temp.end(), first.begin(), first.end());
            temp.insert(temp.end(), second.begin(), second.end());
            temp.insert(temp.end(), pic_lt_curr.begin(), pic_lt_curr.end());
            const int want = num_ref[X] > total ? num_ref[X] : total;
            for (int i = 0; i < num_ref[X]; i++) {
              const int e = modified[X] ? entries[X][i] : i;
              if (e < 0 || e >= want) bad("list_entry_lX out of range");
              sl.ref_poc[X][i] = temp[(size_t)(e % total)];
              sl.ref_lt[X][i] = (e % total) >= n_st;
            }
          }
          if (is_b) sl.sp.mvd_l1_zero = r.u(1);
          bool cabac_init_flag = false;
          if (P.cabac_init_present) cabac_init_flag = r.u(1);
          bool col_from_l0 = true;
          int col_ref_idx = 0;
          if (slice_tmvp) {
            if (is_b) col_from_l0 = r.u(1);
            if ((col_from_l0 && num_ref[0] > 1) || (!col_from_l0 && num_ref[1] > 1)) col_ref_idx = r.ue_max(15, "collocated_ref_idx");
            if (col_ref_idx >= num_ref[col_from_l0 ? 0 : 1]) bad("collocated_ref_idx out of range");
            sl.col_poc = sl.ref_poc[col_from_l0 ? 0 : 1][col_ref_idx];
          }
          if (is_b ? P.weighted_bipred : P.weighted_pred) {   // 7.3.6.3 pred_weight_table
            const int nc = S.chroma_format_idc ? 3 : 1;
            WeightTable& wt = sl.weights;
            sl.has_weights = true;
            const int luma_denom = r.ue_max(7, "luma_log2_weight_denom");
            int chroma_denom = luma_denom;
            if (nc == 3) { chroma_denom += r.se_range(-7, 7, "delta_chroma_log2_weight_denom"); if (chroma_denom < 0 || chroma_denom > 7) bad("ChromaLog2WeightDenom out of range"); }
            sl.sp.luma_log2_wd = (uint8_t)luma_denom; sl.sp.chroma_log2_wd = (uint8_t)chroma_denom;
            for (int X = 0; X < (is_b ? 2 : 1); X++) {
              bool lf[16], cf[16];
              for (int i = 0; i < 16; i++) lf[i] = cf[i] = false;
              // (the flags are present for every entry: a reference picture of these single-layer streams never has the current picture's POC)
              for (int i = 0; i < num_ref[X]; i++) lf[i] = r.u(1);
              if (nc == 3) for (int i = 0; i < num_ref[X]; i++) cf[i] = r.u(1);
              for (int i = 0; i < num_ref[X]; i++) {
                wt.w[X][i][0] = (int16_t)(1 << luma_denom); wt.o[X][i][0] = 0;
                wt.w[X][i][1] = wt.w[X][i][2] = (int16_t)(1 << chroma_denom); wt.o[X][i][1] = wt.o[X][i][2] = 0;
                if (lf[i]) {
                  wt.w[X][i][0] = (int16_t)((1 << luma_denom) + r.se_range(-128, 127, "delta_luma_weight"));
                  wt.o[X][i][0] = (int16_t)r.se_range(-128, 127, "luma_offset");
                }
                if (cf[i])
                  for (int j = 1; j < 3; j++) {
                    const int wgt = (1 << chroma_denom) + r.se_range(-128, 127, "delta_chroma_weight");
                    const int dof = r.se_range(-512, 511, "delta_chroma_offset");
                    int o = 128 + dof - ((128 * wgt) >> chroma_denom);
                    o = o < -128 ? -128 : (o > 127 ? 127 : o);
                    wt.w[X][i][j] = (int16_t)wgt; wt.o[X][i][j] = (int16_t)o;
                  }
              }
            }
          }
          const int max_merge = 5 - r.ue_max(4, "five_minus_max_num_merge_cand");
          sl.sp.is_p = 1; sl.sp.is_b = is_b ? 1 : 0;
          sl.sp.num_ref_idx = (uint8_t)num_ref[0]; sl.sp.num_ref_idx_l1 = (uint8_t)num_ref[1]; sl.sp.max_merge_cand = (uint8_t)max_merge;
          sl.sp.init_type = (uint8_t)((is_b != cabac_init_flag) ? 2 : 1);   // 9.3.2.2: P -> 1, B -> 2, swapped by cabac_init_flag
          sl.sp.tmvp = slice_tmvp ? 1 : 0; sl.sp.col_from_l0 = col_from_l0 ? 1 : 0;
          sl.sp.weighted = sl.has_weights ? 1 : 0;
          bool no_backward = true;
          for (int X = 0; X < 2; X++) for (int i = 0; i < num_ref[X]; i++) if (sl.ref_poc[X][i] > out.poc) no_backward = false;
          sl.sp.no_backward = no_backward ? 1 : 0;
          out.is_inter = true;
        }
        int slice_qp_delta = r.se_range(-128, 128, "slice_qp_delta");
        int s_cb = 0, s_cr = 0;
        if (P.slice_chroma_qp_offsets_present) { s_cb = r.se_range(-12, 12, "slice_cb_qp_offset"); s_cr = r.se_range(-12, 12, "slice_cr_qp_offset"); }
        bool override_flag = P.deblocking_override_enabled ? r.u(1) : false;
        sl.sp.deblocking_disabled = P.deblocking_disabled;
        sl.sp.beta_offset_div2 = (int8_t)P.beta_offset_div2;
        sl.sp.tc_offset_div2 = (int8_t)P.tc_offset_div2;
        if (override_flag) {
          sl.sp.deblocking_disabled = r.u(1);
          if (!sl.sp.deblocking_disabled) { sl.sp.beta_offset_div2 = (int8_t)r.se_range(-6, 6, "slice_beta_offset_div2"); sl.sp.tc_offset_div2 = (int8_t)r.se_range(-6, 6, "slice_tc_offset_div2"); }
        }
        sl.sp.lf_across_slices = P.lf_across_slices;
        if (P.lf_across_slices && (sl.sp.sao_luma || sl.sp.sao_chroma || !sl.sp.deblocking_disabled)) sl.sp.lf_across_slices = r.u(1);
        sl.sp.slice_qp_y = P.init_qp + slice_qp_delta;
        if (sl.sp.slice_qp_y < -6 * (S.bit_depth_luma - 8) || sl.sp.slice_qp_y > 51) bad("SliceQpY out of range");
        if (P.cb_qp_offset + s_cb < -12 || P.cb_qp_offset + s_cb > 12 || P.cr_qp_offset + s_cr < -12 || P.cr_qp_offset + s_cr > 12)
          bad("chroma QP offset out of range");
        sl.sp.cb_qp_offset = (int8_t)(P.cb_qp_offset + s_cb);
        sl.sp.cr_qp_offset = (int8_t)(P.cr_qp_offset + s_cr);
        sl.sp.pps_cb_qp_offset = (int8_t)P.cb_qp_offset;
        sl.sp.pps_cr_qp_offset = (int8_t)P.cr_qp_offset;
        sl.sp.slice_addr_rs = (uint16_t)sl.segment_address;
        }   // !dependent
        if (P.tiles || P.wpp) {
          const int n = r.ue_max((uint32_t)n_ctb, "num_entry_point_offsets");
          if (n > 0) {
            const int len = r.ue_max(31, "offset_len_minus1") + 1;
            sl.entry_point_offsets.reserve((size_t)n);
            for (int i = 0; i < n; i++) {
              const uint64_t off = (uint64_t)r.u(len) + 1;
              if (off > nal_size) bad("entry point beyond the slice NAL");
              sl.entry_point_offsets.push_back((uint32_t)off);
            }
          }
        }
        if (P.slice_header_extension_present) { const int len = r.ue_max(256, "slice_segment_header_extension_length"); r.skip((size_t)len * 8); }
        if (r.u(1) != 1) bad("slice header alignment bit is not 1");
        while (!r.aligned()) if (r.u(1)) bad("slice header alignment bits are not zero");
        sl.data_offset = nal_off + 2 + r.byte_pos();
        sl.nal_end = nal_off + nal_size;
        out.slices.push_back(sl);
      }
      // VPS / AUD / SEI / EOS: nothing to do for an intra still
    }
    if (!have_picture) { err = "no coded picture in the pushed data"; return HIPDEC_ERR_NO_IMAGE; }
    out.poc_lsb = pic_poc_lsb; out.nal_type = pic_nal_type; out.temporal_id = pic_tid < 0 ? 0 : pic_tid;
    out.max_num_reorder = out.sps.max_num_reorder; out.max_dec_pic_buffering = out.sps.max_dec_pic_buffering;
    out.weight_tables.clear();
    if (out.is_inter) {   // the picture's reference table: every picture some P / B slice lists, once; SliceParams::ref_slot(_l1) index it
      auto slot_of = [&](int poc) -> int {
        for (size_t k = 0; k < out.refs.size(); k++) if (out.refs[k].poc == poc) return (int)k;
        for (const RefPicture& rp : seq->dpb)
          if (rp.poc == poc) {
            if (rp.width != out.sps.pic_width || rp.height != out.sps.pic_height || rp.chroma_format_idc != out.sps.chroma_format_idc ||
                rp.bit_depth_luma != out.sps.bit_depth_luma || rp.bit_depth_chroma != out.sps.bit_depth_chroma || rp.log2_ctb != out.sps.log2_ctb)
              bad("reference picture with POC " + std::to_string(poc) + " was decoded in another format (parameter sets changed without an IDR picture)");
            out.refs.push_back(rp);
            for (int q : out.lt_pocs) if (q == poc) out.refs.back().long_term = true;
            return (int)out.refs.size() - 1;
          }
        return -1;
      };
      for (ParsedSlice& sl : out.slices) {
        if (!sl.sp.is_p || sl.dependent) continue;
        for (int X = 0; X < 2; X++)
          for (int i = 0; i < (X ? sl.sp.num_ref_idx_l1 : sl.sp.num_ref_idx); i++) {
            const int slot = slot_of(sl.ref_poc[X][i]);
            if (slot < 0 || slot > 15) bad("reference picture table overflow");
            (X ? sl.sp.ref_slot_l1 : sl.sp.ref_slot)[i] = (uint8_t)slot;
          }
        if (sl.sp.tmvp) {
          const int slot = slot_of(sl.col_poc);
          if (slot < 0 || slot > 15) bad("reference picture table overflow");
          sl.sp.col_slot = (uint8_t)slot;
        }
        if (sl.has_weights) { sl.sp.wp_index = (uint16_t)out.weight_tables.size(); out.weight_tables.push_back(sl.weights); }
      }
      // a dependent slice segment carries its slice's fields: refresh the copies made before the slots were known
      for (size_t si = 1; si < out.slices.size(); si++) if (out.slices[si].dependent) { const uint16_t a = out.slices[si].sp.slice_addr_rs; out.slices[si].sp = out.slices[si - 1].sp; out.slices[si].sp.slice_addr_rs = a; }
    }

    const Sps& S = out.sps; const Pps& P = out.pps;
    // ---- substreams: one CABAC engine start each (slice segment / tile / WPP row) ----
    out.ts_to_rs = tiles.ts_to_rs;
    out.subs.clear();
    out.slice_params.clear();
    out.scaling_tables.clear();
    if (S.scaling_list_enabled) build_scaling_tables(P.scaling_list_data_present ? P.sl : S.sl, S.chroma_format_idc == 3, out.scaling_tables, out.is_inter);
    std::vector<int> slice_head(out.slices.size(), 0);   // index of the (independent) slice segment that starts the slice a segment belongs to:
                                                         // what the kernels compare to tell slices apart (CtbInfo::slice_idx)
    for (size_t si = 0; si < out.slices.size(); si++) {
      const ParsedSlice& sl = out.slices[si];
      slice_head[si] = sl.dependent && si > 0 ? slice_head[si - 1] : (int)si;
      out.slice_params.push_back(sl.sp);
      int ts0 = tiles.rs_to_ts[sl.segment_address];
      // the slice extends to the next slice's start (in tile scan) or the end of the picture
      int ts_end = n_ctb;
      for (size_t sj = 0; sj < out.slices.size(); sj++) {
        int t = tiles.rs_to_ts[out.slices[sj].segment_address];
        if (t > ts0 && t < ts_end) ts_end = t;
      }
      size_t entry = 0;
      uint32_t byte_pos = (uint32_t)sl.data_offset;
      int ts = ts0;
      while (ts < ts_end) {
        Substream sub{};
        sub.first_ctb_ts = (uint32_t)ts;
        sub.slice_idx = (uint32_t)si;
        sub.byte_start = byte_pos;
        sub.dep_sub = -1;
        int t = ts;
        for (;;) {
          int rs = tiles.ts_to_rs[t];
          if (ctb_slice[rs] >= 0) bad("a CTB is covered by two slices");
          ctb_slice[rs] = slice_head[si];
          ctb_slice_addr[rs] = sl.sp.slice_addr_rs;   // SliceAddrRs: dependent slice segments belong to the slice of the segment they continue
          t++;
          if (t >= ts_end) break;
          int nrs = tiles.ts_to_rs[t];
          bool new_tile = P.tiles && tiles.tile_id_rs[nrs] != tiles.tile_id_rs[tiles.ts_to_rs[t - 1]];
          bool new_row = P.wpp && (nrs % ctb_w == 0 || tiles.tile_id_rs[nrs] != tiles.tile_id_rs[nrs - 1]);
          if (new_tile || new_row) break;
        }
        sub.num_ctbs = (uint32_t)(t - ts);
        if (t < ts_end) {
          if (entry >= sl.entry_point_offsets.size()) bad("substream boundary without an entry point");
          byte_pos += sl.entry_point_offsets[entry++];
          if (byte_pos > sl.nal_end) bad("entry point beyond the slice NAL");
          sub.byte_end = byte_pos;
        } else {
          sub.byte_end = (uint32_t)sl.nal_end;
          sub.last_in_slice_segment = 1;
        }
        if (sl.dependent && ts == ts0) {
          // 9.3.1 / 9.3.2.4: the first CTB of a dependent slice segment continues the context variables (and qPY_PREV, 8.6.1) of the END of the
          // preceding slice segment, unless it starts a tile (initialisation) or a CTB row under WPP (the WPP rules apply, linked below)
          const int rs0 = tiles.ts_to_rs[ts0];
          const bool tile_first = ts0 == 0 || tiles.tile_id_rs[rs0] != tiles.tile_id_rs[tiles.ts_to_rs[ts0 - 1]];
          const bool row_first = P.wpp && (rs0 % ctb_w == 0 || tiles.tile_id_rs[rs0] != tiles.tile_id_rs[rs0 - 1]);
          if (!tile_first && !row_first) {
            if (P.wpp) unsupported("dependent slice segment that starts inside a CTB row with entropy_coding_sync enabled");
            if (out.subs.empty()) bad("dependent slice segment without a preceding substream");
            Substream& prev = out.subs.back();
            if ((int)(prev.first_ctb_ts + prev.num_ctbs) != ts0) bad("dependent slice segment does not continue the preceding slice segment");
            sub.dep_sub = (int32_t)out.subs.size() - 1;
            sub.dep_len = prev.num_ctbs;
            sub.wpp_sync = 2;          // synchronise with the tables stored at the end of dep_sub
            prev.has_dependent = 2;    // ... which stores them there
            out.uses_end_sync = true;
          }
        }
        out.subs.push_back(sub);
        ts = t;
      }
      if (entry != sl.entry_point_offsets.size()) bad("unused entry points in a slice segment");
    }
    for (int i = 0; i < n_ctb; i++) if (ctb_slice[i] < 0) bad("picture is incomplete: CTB " + std::to_string(i) + " is not covered by any slice");

    // WPP predecessor links
    if (P.wpp) {
      std::vector<int> sub_of_ctb(n_ctb, -1);
      for (size_t k = 0; k < out.subs.size(); k++)
        for (uint32_t c = 0; c < out.subs[k].num_ctbs; c++) sub_of_ctb[tiles.ts_to_rs[out.subs[k].first_ctb_ts + c]] = (int)k;
      for (size_t k = 0; k < out.subs.size(); k++) {
        Substream& sub = out.subs[k];
        int rs = tiles.ts_to_rs[sub.first_ctb_ts];
        int x = rs % ctb_w, y = rs / ctb_w;
        bool row_start = x == 0 || tiles.tile_id_rs[rs] != tiles.tile_id_rs[rs - 1];
        if (!row_start || y == 0) continue;
        int up = rs - ctb_w;
        if (tiles.tile_id_rs[up] != tiles.tile_id_rs[rs] || ctb_slice_addr[up] != ctb_slice_addr[rs]) continue;
        int d = sub_of_ctb[up];
        // same slice and the row above starts at the same column: aligned predecessor
        if (tiles.ts_to_rs[out.subs[d].first_ctb_ts] != up) continue;
        sub.dep_sub = d;
        sub.dep_len = out.subs[d].num_ctbs;
        out.subs[d].has_dependent = 1;
        // 9.3.1: synchronise when the top-right CTB (x0 + CtbSizeY, y0 - CtbSizeY) is available
        int tr = up + 1;
        bool tr_ok = (x + 1 < ctb_w) && tiles.tile_id_rs[tr] == tiles.tile_id_rs[rs] && ctb_slice_addr[tr] == ctb_slice_addr[rs] &&
                     out.subs[d].num_ctbs >= 2;
        sub.wpp_sync = tr_ok ? 1 : 0;
      }
    }

    // per-CTB availability (6.4.1 restricted to CTB granularity: same slice, same tile)
    out.ctb_info.assign(n_ctb, CtbInfo{});
    for (int rs = 0; rs < n_ctb; rs++) {
      int x = rs % ctb_w, y = rs / ctb_w;
      auto same = [&](int o) { return tiles.tile_id_rs[o] == tiles.tile_id_rs[rs] && ctb_slice_addr[o] == ctb_slice_addr[rs]; };
      uint8_t av = 0;
      if (x > 0 && same(rs - 1)) av |= AV_LEFT;
      if (y > 0 && same(rs - ctb_w)) av |= AV_UP;
      if (y > 0 && x + 1 < ctb_w && same(rs - ctb_w + 1)) av |= AV_UPRIGHT;
      if (y > 0 && x > 0 && same(rs - ctb_w - 1)) av |= AV_UPLEFT;
      auto edge_ok = [&](int o) {
        if (ctb_slice_addr[o] != ctb_slice_addr[rs] && !out.slices[ctb_slice[rs]].sp.lf_across_slices) return false;
        if (tiles.tile_id_rs[o] != tiles.tile_id_rs[rs] && !P.lf_across_tiles) return false;
        return true;
      };
      if (x > 0 && edge_ok(rs - 1)) av |= AV_EDGE_LEFT;
      if (y > 0 && edge_ok(rs - ctb_w)) av |= AV_EDGE_UP;
      out.ctb_info[rs].avail = av;
      out.ctb_info[rs].slice_idx = (uint16_t)ctb_slice[rs];
      out.ctb_info[rs].tile_id = tiles.tile_id_rs[rs];
    }

    hipdec_image_info& I = out.info;
    int sub_w = (S.chroma_format_idc == 1 || S.chroma_format_idc == 2) ? 2 : 1, sub_h = S.chroma_format_idc == 1 ? 2 : 1;   // SubWidthC, SubHeightC
    int x0 = sub_w * S.conf_left, x1 = S.pic_width - sub_w * S.conf_right;
    int y0 = sub_h * S.conf_top, y1 = S.pic_height - sub_h * S.conf_bottom;
    if (x1 <= x0 || y1 <= y0) bad("empty conformance window");
    I.width = x1 - x0; I.height = y1 - y0;
    I.chroma_format_idc = S.chroma_format_idc;
    I.chroma_width = S.chroma_format_idc ? I.width / sub_w : 0;
    I.chroma_height = S.chroma_format_idc ? I.height / sub_h : 0;
    I.bit_depth_luma = S.bit_depth_luma; I.bit_depth_chroma = S.bit_depth_chroma;
    I.colour_primaries = S.colour_primaries; I.transfer_characteristics = S.transfer_characteristics;
    I.matrix_coeffs = S.matrix_coeffs; I.full_range_flag = S.full_range;
    I.coded_width = S.pic_width; I.coded_height = S.pic_height;
    I.bitstream_bytes = size;
    I.num_substreams = (int)out.subs.size();
    return HIPDEC_OK;
  } catch (const ParseError& e) {
    err = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    err = "out of host memory while parsing the bitstream headers";
    return HIPDEC_ERR_MEMORY;
  } catch (const std::exception& e) {
    err = std::string("bitstream header parsing failed: ") + e.what();
    return HIPDEC_ERR_BITSTREAM;
  }
}

// The picture was decoded: its POC becomes prevTid0Pic's where 8.3.1 says so, and the DPB drops every picture its reference picture set does
// not name (8.3.2; an IDR picture names none).  The caller appends the decoded picture itself.
void seq_commit(SeqContext& seq, const ParsedPicture& pic, int)
{
  const bool irap = pic.nal_type >= 16 && pic.nal_type <= 23;
  // NoRaslOutputFlag (8.1.3): IDR and BLA pictures, and a CRA picture that is the first of the sequence
  if (irap) seq.no_rasl_output = pic.is_idr || pic.nal_type <= 18 || seq.first_picture;
  // prevTidOPic (8.3.1): the previous picture with TemporalId = 0 that is not a RASL, RADL or sub-layer non-reference picture
  if (pic.is_idr || (irap && seq.first_picture)) { seq.prev_tid0_lsb = pic.is_idr ? 0 : pic.poc_lsb; seq.prev_tid0_msb = 0; }
  else if (pic.temporal_id == 0 && (irap || (pic.nal_type <= 5 && (pic.nal_type & 1)))) { seq.prev_tid0_lsb = pic.poc_lsb; seq.prev_tid0_msb = pic.poc - pic.poc_lsb; }
  seq.first_picture = false;
  for (RefPicture& rp : seq.dpb) for (int poc : pic.lt_pocs) if (poc == rp.poc) rp.long_term = true;   // 8.3.2: RefPicSetLtCurr / LtFoll
  std::vector<RefPicture> kept;
  for (const RefPicture& rp : seq.dpb) {
    bool keep = false;
    for (int poc : pic.keep_pocs) if (poc == rp.poc) keep = true;
    if (keep) kept.push_back(rp);
  }
  seq.dpb.swap(kept);
}

void SampleQueue::push(const uint8_t* p, size_t size)
{
  size_t touched_from = SIZE_MAX;
  last_push_touched_first = false;
  bool open = !queue.empty() && last_open;
  for (size_t q = 0; q + 4 <= size;) {
    const uint32_t n = ((uint32_t)p[q] << 24) | ((uint32_t)p[q + 1] << 16) | ((uint32_t)p[q + 2] << 8) | p[q + 3];
    if ((size_t)n > size - q - 4) break;   // (the caller validated the framing; never read past the push)
    const uint8_t* nal = p + q + 4;
    const size_t whole = 4 + (size_t)n;
    q += whole;
    if (n < 2) continue;
    const int t = (nal[0] >> 1) & 63;
    const bool vcl = t < 32;
    const bool starts = vcl ? (n >= 3 && (nal[2] & 0x80)) : ((t >= 32 && t <= 35) || t == 39 || (t >= 41 && t <= 44) || (t >= 48 && t <= 55));
    if (t >= 32 && t <= 34) {   // VPS / SPS / PPS: remembered for the samples that follow
      std::vector<uint8_t>& ps = param_sets;
      for (size_t a = 0; a + 4 <= ps.size();) {
        const size_t len = 4 + (((size_t)ps[a] << 24) | ((size_t)ps[a + 1] << 16) | ((size_t)ps[a + 2] << 8) | ps[a + 3]);
        if (len > ps.size() - a) break;
        if (len == whole && !memcmp(ps.data() + a, nal - 4, whole)) { ps.erase(ps.begin() + (long)a, ps.begin() + (long)(a + len)); break; }
        a += len;
      }
      ps.insert(ps.end(), nal - 4, nal + n);
    }
    if (!first_closed) {
      if (starts && first_has_vcl) first_closed = true;   // the still's access unit is complete: what follows are samples of a sequence
      else {
        first.insert(first.end(), nal - 4, nal + n);
        if (vcl) first_has_vcl = true;
        last_push_touched_first = true;
        continue;
      }
    }
    if (!open || (starts && queue.back().has_vcl)) {
      queue.emplace_back();
      touched_from = std::min(touched_from, queue.size() - 1);
      queue.back().user_data = pending_user_data;
      open = true;
      // a parameter set that opens the sample is already in param_sets (appended above): the blob starts with all of them either way
      queue.back().blob = param_sets;
      if (t >= 32 && t <= 34) continue;
    }
    touched_from = std::min(touched_from, queue.size() - 1);
    queue.back().blob.insert(queue.back().blob.end(), nal - 4, nal + n);   // (a parameter set inside an open sample stays in front of its slices like the others)
    if (vcl) queue.back().has_vcl = true;
  }
  last_open = open;
  last_push_first = touched_from == SIZE_MAX ? queue.size() : touched_from;
}

void SampleQueue::set_user_data(uintptr_t user_data)
{
  pending_user_data = user_data;
  if (last_push_touched_first) first_user_data = user_data;
  for (size_t i = last_push_first; i < queue.size(); i++) queue[i].user_data = user_data;
}

void SampleQueue::drop_front(size_t n)
{
  if (n > queue.size()) n = queue.size();
  queue.erase(queue.begin(), queue.begin() + (long)n);
  if (queue.empty()) last_open = false;
  last_push_first = last_push_first > n ? last_push_first - n : 0;
}

}  // namespace hipdec
