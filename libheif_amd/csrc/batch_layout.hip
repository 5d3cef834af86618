// batch_layout.hip — see batch_layout.h.  Plain C++ (also compiled by g++ for the CPU-test emulation).
#include "batch_layout.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace hipdec {
namespace {
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Host worker threads for the per-item work of large batches (header parsing, staging copies): items are independent, so they
// are handed out through one atomic counter.  Small batches (the plugin path: one item, or the tiles of a grid) stay on the
// calling thread.
template <class F> void for_each_item(int n, size_t total_bytes, F&& fn)
{
  unsigned hw = std::thread::hardware_concurrency();
  unsigned nt = hw > 16 ? 16 : hw;
  if (const char* e = getenv("HIPDEC_HOST_THREADS")) nt = (unsigned)std::max(1, atoi(e));
  if (n < 32 || total_bytes < (size_t(8) << 20) || nt <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
  std::atomic<int> next{0};
  auto body = [&]() { for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(i); };
  std::vector<std::thread> th;
  try {
    for (unsigned t = 1; t < nt; t++) th.emplace_back(body);
  } catch (...) {}   // could not start (all of) the workers: the calling thread does the rest
  body();
  for (auto& t : th) t.join();
}
}  // namespace

int layout_batch(BatchLayout& b, int n, const void* const* data, const size_t* sizes, uint64_t max_pixels,
                 std::vector<uint8_t>& host, std::string& err_out, const SeqContext* const* seqs)
{
  int rc = layout_batch_plan(b, n, data, sizes, max_pixels, err_out, seqs);
  if (rc != HIPDEC_OK) return rc;
  host.assign(b.upload_size, 0);
  layout_batch_fill(b, data, sizes, host.data());
  return HIPDEC_OK;
}

namespace {
int layout_core(BatchLayout& b, const size_t* sizes, std::string& err_out);
}

int layout_batch_plan(BatchLayout& b, int n, const void* const* data, const size_t* sizes, uint64_t max_pixels, std::string& err_out,
                      const SeqContext* const* seqs)
{
  b.pics.resize(n);
  {
    size_t total = 0;
    for (int i = 0; i < n; i++) total += sizes[i];
    std::vector<int> rcs((size_t)n, HIPDEC_OK);
    std::vector<std::string> errs((size_t)n);
    for_each_item(n, total, [&](int i) { rcs[i] = parse_picture((const uint8_t*)data[i], sizes[i], max_pixels, b.pics[i], errs[i], seqs ? seqs[i] : nullptr); });
    for (int i = 0; i < n; i++) {
      if (rcs[i] != HIPDEC_OK) { err_out = "item " + std::to_string(i) + ": " + errs[i]; return rcs[i]; }
      if (b.pics[i].skipped) { err_out = "item " + std::to_string(i) + ": a RASL picture whose CRA picture started the sequence is not decoded (8.3.3)"; return HIPDEC_ERR_NO_IMAGE; }
    }
  }
  return layout_core(b, sizes, err_out);
}

RefPicture chain_ref_picture(const BatchLayout& b, int j, uint64_t arena_base)
{
  const PicParams& P = b.params[(size_t)j];
  const BatchLayout::ChainItem& ci = b.chain_items[(size_t)j];
  RefPicture rp;
  rp.poc = b.pics[(size_t)j].poc;
  rp.width = P.width; rp.height = P.height; rp.chroma_format_idc = P.chroma_format_idc; rp.bit_depth_luma = P.bit_depth_luma; rp.bit_depth_chroma = P.bit_depth_chroma;
  rp.log2_ctb = P.log2_ctb;
  for (int c = 0; c < 3; c++) {
    rp.plane[c] = arena_base + (ci.off_full_pic ? (uint64_t)ci.off_full[c] : (uint64_t)P.off_out[c]);
    rp.stride[c] = ci.off_full_pic ? ci.full_stride[c] : P.out_stride[c];
  }
  rp.mf = P.is_inter ? arena_base + (uint64_t)P.off_mf : 0;
  return rp;
}

void chain_resolve(BatchLayout& b, uint64_t arena_base, std::vector<int>& own, int track)
{
  own.clear();
  for (RefPicture& rp : b.tracks[(size_t)track].seq_after.dpb)
    if (rp.batch_item >= 0) {
      const bool lt = rp.long_term;   // (marked on the working copy when a later picture of the chain made it a long-term reference: keep it)
      rp = chain_ref_picture(b, rp.batch_item, arena_base);
      rp.long_term = lt;
      own.push_back(rp.poc);
    }
}

int layout_batch_plan_chain(BatchLayout& b, int n, const void* const* data, const size_t* sizes, uint64_t max_pixels, std::string& err_out,
                            const SeqContext& seq)
{
  const int first = 0;
  const SeqContext* seqs[1] = {&seq};
  return layout_batch_plan_chains(b, 1, &first, &n, data, sizes, max_pixels, err_out, seqs);
}

int layout_batch_plan_chains(BatchLayout& b, int n_tracks, const int* first, const int* count, const void* const* data, const size_t* sizes,
                             uint64_t max_pixels, std::string& err_out, const SeqContext* const* seqs, int* bad_track)
{
  b.chain = true;
  b.pics.clear(); b.src_index.clear(); b.pixel_step_of.clear(); b.motion_step_of.clear();
  b.tracks.assign((size_t)n_tracks, BatchLayout::ChainTrack{});
  if (bad_track) *bad_track = -1;
  // every track on its own first: its samples one after the other against a working copy of its sequence state (a decoded sample becomes a
  // reference picture of the samples behind it, 8.3.2 keeps / drops it through their RPS: addressed as "picture j of this track's chain"), and
  // the track's steps (batch_layout.h)
  struct Planned { ParsedPicture pp; int sample = 0, px = 0, mo = 0; };
  std::vector<std::vector<Planned>> planned((size_t)n_tracks);
  std::vector<int> rcs((size_t)n_tracks, HIPDEC_OK);
  std::vector<std::string> errs((size_t)n_tracks);
  size_t total = 0;
  for (int t = 0; t < n_tracks; t++) for (int i = 0; i < count[t]; i++) total += sizes[first[t] + i];
  for_each_item(n_tracks, n_tracks >= 4 ? std::max(total, size_t(8) << 20) : 0, [&](int t) {
    SeqContext work = *seqs[t];
    std::vector<Planned>& mine = planned[(size_t)t];
    int px_first = 0, mo_first = 0, px = 0, mo = 0;
    for (int i = 0; i < count[t]; i++) {
      Planned pl;
      std::string err;
      const int rc = parse_picture((const uint8_t*)data[first[t] + i], sizes[first[t] + i], max_pixels, pl.pp, err, &work);
      if (rc != HIPDEC_OK) { rcs[(size_t)t] = rc; errs[(size_t)t] = "sample " + std::to_string(i) + " of the chain: " + err; return; }
      if (pl.pp.skipped) continue;
      const int j = (int)mine.size();
      bool px_dep = false, mo_dep = false;
      for (const RefPicture& rp : pl.pp.refs) if (rp.batch_item >= px_first) px_dep = true;
      for (const ParsedSlice& sl : pl.pp.slices)
        if (sl.sp.is_p && sl.sp.tmvp && sl.sp.col_slot < pl.pp.refs.size() && pl.pp.refs[sl.sp.col_slot].batch_item >= mo_first) mo_dep = true;
      if (px_dep) { px++; px_first = j; }
      if (mo_dep) { mo++; mo_first = j; }
      pl.sample = i; pl.px = px; pl.mo = mo;
      seq_commit(work, pl.pp);
      RefPicture rp;
      rp.poc = pl.pp.poc; rp.batch_item = j;
      rp.width = pl.pp.sps.pic_width; rp.height = pl.pp.sps.pic_height; rp.chroma_format_idc = pl.pp.sps.chroma_format_idc;
      rp.bit_depth_luma = pl.pp.sps.bit_depth_luma; rp.bit_depth_chroma = pl.pp.sps.bit_depth_chroma; rp.log2_ctb = pl.pp.sps.log2_ctb;
      work.dpb.push_back(rp);
      mine.push_back(std::move(pl));
    }
    b.tracks[(size_t)t].seq_after = std::move(work);
  });
  for (int t = 0; t < n_tracks; t++)
    if (rcs[(size_t)t] != HIPDEC_OK) {
      err_out = n_tracks > 1 ? "track " + std::to_string(t) + " of the launch set, " + errs[(size_t)t] : errs[(size_t)t];
      if (bad_track) *bad_track = t;
      return rcs[(size_t)t];
    }
  // the batch's items: by (pixel step, track, decoding order); "picture j of the track's chain" becomes "item g of the batch"
  int steps = 0;
  for (const auto& mine : planned) if (!mine.empty()) steps = std::max(steps, mine.back().px + 1);
  std::vector<std::vector<int>> item_of((size_t)n_tracks);
  for (int t = 0; t < n_tracks; t++) item_of[(size_t)t].assign(planned[(size_t)t].size(), -1);
  std::vector<size_t> cursor((size_t)n_tracks, 0);
  std::vector<size_t> item_sizes;
  for (int k = 0; k < steps; k++)
    for (int t = 0; t < n_tracks; t++) {
      std::vector<Planned>& mine = planned[(size_t)t];
      for (size_t& j = cursor[(size_t)t]; j < mine.size() && mine[j].px == k; j++) {
        const int g = (int)b.pics.size();
        item_of[(size_t)t][j] = g;
        b.tracks[(size_t)t].items.push_back(g); b.tracks[(size_t)t].samples.push_back(mine[j].sample);
        b.src_index.push_back(first[t] + mine[j].sample);
        b.pixel_step_of.push_back(k); b.motion_step_of.push_back(mine[j].mo);
        item_sizes.push_back(sizes[first[t] + mine[j].sample]);
        b.pics.push_back(std::move(mine[j].pp));
      }
    }
  for (int t = 0; t < n_tracks; t++) {
    for (int g : b.tracks[(size_t)t].items)
      for (RefPicture& rp : b.pics[(size_t)g].refs) if (rp.batch_item >= 0) rp.batch_item = item_of[(size_t)t][(size_t)rp.batch_item];
    for (RefPicture& rp : b.tracks[(size_t)t].seq_after.dpb) if (rp.batch_item >= 0) rp.batch_item = item_of[(size_t)t][(size_t)rp.batch_item];
  }
  if (b.pics.empty()) return HIPDEC_OK;
  return layout_core(b, item_sizes.data(), err_out);
}

namespace {
int layout_core(BatchLayout& b, const size_t* sizes, std::string& err_out)
{
  const int n = (int)b.pics.size();
  b.wide = b.pics[0].info.bit_depth_luma > 8 || b.pics[0].info.bit_depth_chroma > 8;
  for (int i = 0; i < n; i++) {
    const bool w = b.pics[i].info.bit_depth_luma > 8 || b.pics[i].info.bit_depth_chroma > 8;
    if (w != b.wide) { err_out = "batch mixes 8-bit and >8-bit items"; return HIPDEC_ERR_UNSUPPORTED; }
  }
  const size_t es = b.wide ? 2 : 1;
  // ---- layout: [upload region: descriptors, tables, bitstreams][control words][device-only buffers] ----
  size_t off = 0;
  b.off_pics = off; off = align_up(off + sizeof(PicParams) * n, 256);
  uint32_t nsubs = 0, nrows = 0;
  for (auto& p : b.pics) { nsubs += (uint32_t)p.subs.size(); nrows += (uint32_t)((p.sps.pic_height + (1 << p.sps.log2_ctb) - 1) >> p.sps.log2_ctb); }
  b.num_subs = nsubs; b.num_rows = nrows;
  b.off_subs = off; off = align_up(off + sizeof(Substream) * nsubs, 256);
  b.off_rows = off; off = align_up(off + sizeof(RowDesc) * nrows, 256);
  // Parser wavefronts.  A picture's substreams are dealt round-robin to W waves (wave j takes substreams
  // j, j + W, ...): with WPP a wave that finishes row r continues with row r + W, whose predecessor row
  // r + W - 1 is by then well ahead, so every resident wave stays busy instead of parking one wave per
  // row behind its dependency.  W shrinks as the batch grows (the GPU holds ~7k parser waves; ~16 busy
  // waves per CU saturate its scalar pipe); a lone still keeps one wave per substream (lowest latency).
  std::vector<ParseWave>& waves = b.parse_waves;
  waves.clear();
  {
    const uint32_t target_waves = 4096;
    const char* force = getenv("HIPDEC_WAVES_PER_PICTURE");   // test / tuning override
    uint32_t sub_base = 0;
    for (auto& p : b.pics) {
      const uint32_t ns = (uint32_t)p.subs.size();
      uint32_t w = target_waves / (uint32_t)n;
      if (force && atoi(force) > 0) w = (uint32_t)atoi(force);
      if (w < 1) w = 1;
      if (w > ns) w = ns;
      // WPP rows that follow their predecessor at the minimum 2-CTB distance move as a convoy (every CTB waits
      // for the slowest of the W waves); starting a row only once its predecessor is `lag` CTBs ahead absorbs
      // the CTB-to-CTB cost variance.  The ring closes after W rows, so W * lag must stay below the row length;
      // with one wave per row (latency mode) the minimum distance is kept.
      uint32_t lag = 2;
      if (w < ns && p.pps.wpp) {
        const uint32_t row_len = (uint32_t)((p.sps.pic_width + (1 << p.sps.log2_ctb) - 1) >> p.sps.log2_ctb);
        lag = row_len / (w + 1);
        if (lag < 2) lag = 2;
      }
      const char* force_lag = getenv("HIPDEC_WPP_START_LAG");
      if (force_lag && atoi(force_lag) >= 2) lag = (uint32_t)atoi(force_lag);
      for (uint32_t j = 0; j < w; j++) waves.push_back(ParseWave{sub_base + j, w, sub_base + ns, lag});
      sub_base += ns;
    }
  }
  b.num_waves = (uint32_t)waves.size();
  b.off_waves = off; off = align_up(off + sizeof(ParseWave) * waves.size(), 256);
  // Reconstruction wavefronts: the same dealing of a picture's CTB rows to W waves per colour component (a wave that
  // finishes row r continues with row r + W), so that in large batches the resident waves are mostly busy ones.
  std::vector<ReconWave>& rwaves = b.recon_waves;
  rwaves.clear();
  if (b.chain) b.chain_items.assign((size_t)n, BatchLayout::ChainItem{});
  {
    const uint32_t target = 24576;   // ~8 waves per component of a 4K still at 1024 stills in flight (measured optimum 4-16)
    const char* force = getenv("HIPDEC_RECON_WAVES_PER_PICTURE");
    uint32_t row_base = 0;
    for (int i = 0; i < n; i++) {
      const auto& p = b.pics[i];
      const uint32_t rows = (uint32_t)((p.sps.pic_height + (1 << p.sps.log2_ctb) - 1) >> p.sps.log2_ctb);
      const uint32_t row_len = (uint32_t)((p.sps.pic_width + (1 << p.sps.log2_ctb) - 1) >> p.sps.log2_ctb);
      uint32_t w = target / (3u * (uint32_t)n);
      if (w < 8) w = 8;          // measured at 2048 4K stills: 136 ms with 8 row chains per picture and component, 142 with 4 or 16
      if (force && atoi(force) > 0) w = (uint32_t)atoi(force);
      if (w < 1) w = 1;
      if (w > rows) w = rows;
      uint32_t lag = 2;
      if (w < rows) { lag = row_len / (w + 1); if (lag < 2) lag = 2; }
      if (b.chain) b.chain_items[(size_t)i].first_rwave = (uint32_t)rwaves.size();
      for (uint32_t j = 0; j < w; j++)
        for (uint32_t c = 0; c < (p.sps.chroma_format_idc == 3 ? 3u : 2u); c++)   // 0 = luma, 1 = Cb + Cr in one wave (4:2:0); 4:4:4: one wave per plane
          rwaves.push_back(ReconWave{(uint32_t)i, c, j, w, row_base, lag, 0, 0});
      if (b.chain) b.chain_items[(size_t)i].num_rwaves = (uint32_t)rwaves.size() - b.chain_items[(size_t)i].first_rwave;
      row_base += rows;
    }
  }
  b.num_rwaves = (uint32_t)rwaves.size();
  b.off_rwaves = off; off = align_up(off + sizeof(ReconWave) * rwaves.size(), 256);
  b.params.assign(n, PicParams{});
  uint32_t row_base = 0;
  for (int i = 0; i < n; i++) {
    const ParsedPicture& pp = b.pics[i];
    const Sps& S = pp.sps; const Pps& Pp = pp.pps;
    PicParams& P = b.params[i];
    P.width = S.pic_width; P.height = S.pic_height;
    P.chroma_format_idc = S.chroma_format_idc;
    const int csw = (S.chroma_format_idc == 1 || S.chroma_format_idc == 2) ? 1 : 0, csh = S.chroma_format_idc == 1 ? 1 : 0;   // log2 SubWidthC, log2 SubHeightC (6.2)
    P.cwidth = S.chroma_format_idc ? S.pic_width >> csw : 0; P.cheight = S.chroma_format_idc ? S.pic_height >> csh : 0;
    P.out_width = pp.info.width; P.out_height = pp.info.height; P.out_cwidth = pp.info.chroma_width; P.out_cheight = pp.info.chroma_height;
    P.crop_x = S.conf_left << csw; P.crop_y = S.conf_top << csh;
    P.bit_depth_luma = S.bit_depth_luma; P.bit_depth_chroma = S.bit_depth_chroma;
    P.log2_ctb = S.log2_ctb; P.log2_min_cb = S.log2_min_cb; P.log2_min_tb = S.log2_min_tb; P.log2_max_tb = S.log2_max_tb;
    P.max_th_depth_intra = S.max_th_depth_intra;
    P.ctb_w = (S.pic_width + (1 << S.log2_ctb) - 1) >> S.log2_ctb; P.ctb_h = (S.pic_height + (1 << S.log2_ctb) - 1) >> S.log2_ctb;
    P.units_per_ctb_log2 = 2 * (S.log2_ctb - 2);
    P.sao_enabled = S.sao; P.sign_data_hiding = Pp.sign_data_hiding; P.transform_skip_enabled = Pp.transform_skip;
    P.cu_qp_delta_enabled = Pp.cu_qp_delta; P.transquant_bypass_enabled = Pp.transquant_bypass;
    P.strong_intra_smoothing = S.strong_intra_smoothing; P.tiles_enabled = Pp.tiles; P.wpp = Pp.wpp;
    P.lf_across_tiles = Pp.lf_across_tiles; P.pcm_loop_filter_disabled = S.pcm && S.pcm_loop_filter_disabled ? 1 : 0;
    P.pcm_enabled = S.pcm ? 1 : 0; P.pcm_bd_luma = (uint8_t)S.pcm_bit_depth_luma; P.pcm_bd_chroma = (uint8_t)S.pcm_bit_depth_chroma;
    P.pcm_cb_range = (uint8_t)(S.log2_min_pcm_cb | (S.log2_max_pcm_cb << 4));
    P.log2_min_cu_qp_delta_size = S.log2_ctb - Pp.diff_cu_qp_delta_depth;
    if (Pp.diff_cu_qp_delta_depth > S.log2_ctb - S.log2_min_cb) { err_out = "item " + std::to_string(i) + ": diff_cu_qp_delta_depth out of range"; return HIPDEC_ERR_BITSTREAM; }
    P.first_row = row_base; row_base += (uint32_t)P.ctb_h;
    P.num_slices = (uint32_t)pp.slice_params.size();
    {
      bool free_nb = !Pp.transquant_bypass && !(S.pcm && S.pcm_loop_filter_disabled) && (!Pp.tiles || Pp.lf_across_tiles);
      if (pp.slice_params.size() > 1) for (const auto& sl : pp.slice_params) free_nb = free_nb && sl.lf_across_slices;
      P.sao_free_neighbours = free_nb ? 1 : 0;
    }
    const size_t nctb = (size_t)P.ctb_w * P.ctb_h;
    P.off_ctb_ts_to_rs = off; off = align_up(off + nctb * sizeof(uint16_t), 256);
    P.off_ctb_info = off; off = align_up(off + nctb * sizeof(CtbInfo), 256);
    P.off_slices = off; off = align_up(off + pp.slice_params.size() * sizeof(SliceParams), 256);
    P.scaling_lists = pp.scaling_tables.empty() ? 0 : 1;
    P.off_scaling = off;
    if (P.scaling_lists) off = align_up(off + pp.scaling_tables.size(), 256);
    // P pictures: the reference picture table (absolute device pointers into earlier batches' arenas)
    P.is_inter = pp.is_inter ? 1 : 0; P.poc = pp.poc; P.num_refs = (uint32_t)pp.refs.size();
    P.lt_mask = 0;
    for (size_t k = 0; k < pp.refs.size() && k < 16; k++) if (pp.refs[k].long_term) P.lt_mask = (uint16_t)(P.lt_mask | (1u << k));
    P.constrained_intra_pred = (pp.is_inter && Pp.constrained_intra_pred) ? 1 : 0;   // (an intra picture: every unit is intra coded, the flag changes nothing)
    P.amp_enabled = S.amp ? 1 : 0; P.max_th_depth_inter = (uint8_t)S.max_th_depth_inter; P.log2_par_mrg_level = (uint8_t)Pp.log2_par_mrg_level;
    P.off_wp = off;
    if (P.is_inter && !pp.weight_tables.empty()) off = align_up(off + pp.weight_tables.size() * sizeof(WeightTable), 256);
    P.off_reftab = off;
    if (P.is_inter) { off = align_up(off + 16 * sizeof(RefFrame), 256); b.any_inter = true; }
    P.off_bitstream = off; P.bitstream_size = sizes[i]; off = align_up(off + sizes[i] + 512, 256);
    b.max_w = std::max(b.max_w, P.width); b.max_h = std::max(b.max_h, P.height);
    b.max_ctbs = std::max(b.max_ctbs, P.ctb_w * P.ctb_h);
    b.max_ow = std::max(b.max_ow, P.out_width); b.max_oh = std::max(b.max_oh, P.out_height);
  }
  if (b.chain)
    for (int i = 0; i < n; i++) {
      const PicParams& P = b.params[i];
      if (P.out_width != P.width || P.out_height != P.height || P.crop_x || P.crop_y) { b.chain_items[(size_t)i].off_full_pic = off; off = align_up(off + sizeof(PicParams), 256); }
    }
  b.upload_size = off;
  // control words (zeroed before every run)
  b.off_ctrl = off;
  b.off_progress = off; off = align_up(off + sizeof(uint32_t) * nsubs, 256);
  b.off_row_progress = off; off = align_up(off + sizeof(uint32_t) * nrows * 3, 256);   // per (CTB row, component)
  b.off_waitneed = off; off = align_up(off + sizeof(uint32_t) * nsubs, 256);
  b.off_resume_k = off; off = align_up(off + sizeof(uint32_t) * nsubs, 256);
  b.queue_cap = 1; while (b.queue_cap < nsubs) b.queue_cap <<= 1;
  b.off_queue = off; off = align_up(off + sizeof(uint32_t) * b.queue_cap, 256);
  b.off_qctl = off; off += 256;
  b.off_ticket = off; off += 256;  // [0] parse ticket, [1] recon ticket, [2] motion ticket
  if (b.chain) off = align_up(off + 8 * (size_t)n, 256);   // + per item: recon and motion tickets of the per-picture launches (BatchLayout::chain_ticket)
  b.off_status = off; off += 256;
  b.ctrl_size = off - b.off_ctrl;
  b.off_ctx = off; off = align_up(off + (size_t)CTX_STORE * nsubs, 256);
  b.off_saved = off; off = align_up(off + (size_t)SAVE_DWORDS * 4 * nsubs, 256);
  // throughput mode: rows become tasks of a work pool (parse_core.h); latency mode keeps one wave per substream chain
  {
    const char* e = getenv("HIPDEC_PARSE_POOL");
    b.pool = e ? (uint32_t)atoi(e) : (nsubs >= 2048 ? 1u : 0u);
    const char* w = getenv("HIPDEC_POOL_WAVES");
    b.pool_waves = w ? (uint32_t)atoi(w) : 4096u;
    if (b.pool_waves > nsubs) b.pool_waves = nsubs;
    if (b.pool_waves < 1) b.pool_waves = 1;
  }
  for (int i = 0; i < n; i++) {
    PicParams& P = b.params[i];
    const size_t nctb = (size_t)P.ctb_w * P.ctb_h;
    const size_t nunits = nctb << P.units_per_ctb_log2;
    const size_t ctb2 = (size_t)1 << (2 * P.log2_ctb);
    P.off_sao = off; off = align_up(off + nctb * 3 * sizeof(SaoParams), 256);
    P.off_handoff = off; off = align_up(off + nctb * HANDOFF_DWORDS * sizeof(uint32_t), 256);
    P.off_u_size = off; off = align_up(off + nunits, 256);
    P.off_u_flags = off; off = align_up(off + nunits, 256);
    P.off_u_ipm = off; off = align_up(off + nunits, 256);
    P.off_u_ipmc = off; off = align_up(off + nunits, 256);
    P.off_u_qp = off; off = align_up(off + nunits, 256);
    P.off_msyn = P.off_mf = off;
    if (P.is_inter) {
      P.off_msyn = off; off = align_up(off + nunits * sizeof(MotionSyntax), 256);
      P.off_mf = off; off = align_up(off + nunits * sizeof(MotionUnit), 256);
    }
    P.off_coeff[0] = off; off = align_up(off + nctb * ctb2 * 2, 256);
    const int csw = P.chroma_format_idc == 3 ? 0 : 1, csh = (P.chroma_format_idc == 3 || P.chroma_format_idc == 2) ? 0 : 1;   // log2 chroma subsampling (4:0:0: sized like 4:2:0, never touched)
    P.off_coeff[1] = off; off = align_up(off + ((nctb * ctb2 * 2) >> (csw + csh)), 256);
    P.off_coeff[2] = off; off = align_up(off + ((nctb * ctb2 * 2) >> (csw + csh)), 256);
    const int ctb = 1 << P.log2_ctb;
    for (int c = 0; c < 3; c++) {
      const size_t w = c ? ((size_t)P.ctb_w * ctb) >> csw : (size_t)P.ctb_w * ctb, h = c ? ((size_t)P.ctb_h * ctb) >> csh : (size_t)P.ctb_h * ctb;
      P.rec_stride[c] = (uint32_t)align_up(w * es, 64);
      P.off_rec[c] = off; off = align_up(off + (size_t)P.rec_stride[c] * (h + 1), 256);
      P.off_line[c] = off; off = align_up(off + (size_t)P.rec_stride[c] * (size_t)P.ctb_h + 256, 256);
      const size_t ow = c ? P.out_cwidth : P.out_width, oh = c ? P.out_cheight : P.out_height;
      P.out_stride[c] = (uint32_t)align_up(std::max<size_t>(ow, 1) * es, 64);
      P.off_out[c] = off; off = align_up(off + (size_t)P.out_stride[c] * std::max<size_t>(oh, 1), 256);
    }
    if (b.chain && b.chain_items[(size_t)i].off_full_pic) {   // the whole coded picture behind the filters, for the pictures that predict from it
      BatchLayout::ChainItem& ci = b.chain_items[(size_t)i];
      for (int c = 0; c < 3; c++) {
        const size_t w = c ? (size_t)P.cwidth : (size_t)P.width, h = c ? (size_t)P.cheight : (size_t)P.height;
        ci.full_stride[c] = (uint32_t)align_up(std::max<size_t>(w, 1) * es, 64);
        ci.off_full[c] = off; off = align_up(off + (size_t)ci.full_stride[c] * std::max<size_t>(h, 1) + 256, 256);
      }
    }
  }
  b.arena_size = off;
  if (b.chain) {
    // steps (batch_layout.h): the plan numbered them per item; a pixel step is a range of items, a motion step a range of the RowDesc table
    int n_px = 0, n_mo = 0;
    for (int i = 0; i < n; i++) { n_px = std::max(n_px, b.pixel_step_of[(size_t)i] + 1); n_mo = std::max(n_mo, b.motion_step_of[(size_t)i] + 1); }
    b.pixel_steps.assign((size_t)n_px, BatchLayout::ChainStep{}); b.motion_steps.assign((size_t)n_mo, BatchLayout::ChainStep{});
    auto add = [&](BatchLayout::ChainStep& st, int i) {
      const PicParams& P = b.params[(size_t)i];
      st.count++; st.num_rwaves += b.chain_items[(size_t)i].num_rwaves; st.num_rows += (uint32_t)P.ctb_h;
      st.max_w = std::max(st.max_w, (int)P.width); st.max_h = std::max(st.max_h, (int)P.height);
      st.max_ow = std::max(st.max_ow, (int)P.out_width); st.max_oh = std::max(st.max_oh, (int)P.out_height);
      st.any_inter = st.any_inter || P.is_inter;
    };
    for (int i = 0; i < n; i++) {
      BatchLayout::ChainStep& px = b.pixel_steps[(size_t)b.pixel_step_of[(size_t)i]];
      if (!px.count) { px.first = i; px.first_rwave = b.chain_items[(size_t)i].first_rwave; }
      add(px, i);
      px.motion_need = std::max(px.motion_need, b.motion_step_of[(size_t)i] + 1);
      add(b.motion_steps[(size_t)b.motion_step_of[(size_t)i]], i);
    }
    uint32_t row = 0;
    for (auto& mo : b.motion_steps) { mo.first_row = row; row += mo.num_rows; }
  }
  return HIPDEC_OK;
}
}  // namespace

void layout_batch_fill(BatchLayout& b, const void* const* data, const size_t* sizes, uint8_t* host, uint64_t arena_base)
{
  const int n = (int)b.pics.size();
  // alignment gaps of the descriptor area are zeroed; the gap behind every bitstream (the parser's window loads run up to 512 B
  // past the end) is zeroed with the bitstream copy below
  memset(host, 0, b.params.empty() ? b.upload_size : std::min(b.upload_size, (size_t)b.params[0].off_ctb_ts_to_rs));
  memcpy(host + b.off_pics, b.params.data(), sizeof(PicParams) * n);
  memcpy(host + b.off_waves, b.parse_waves.data(), sizeof(ParseWave) * b.parse_waves.size());
  memcpy(host + b.off_rwaves, b.recon_waves.data(), sizeof(ReconWave) * b.recon_waves.size());
  Substream* subs = (Substream*)(host + b.off_subs);
  RowDesc* rows = (RowDesc*)(host + b.off_rows);
  uint32_t sub_base = 0, r = 0;
  for (int i = 0; i < n; i++) {
    const ParsedPicture& pp = b.pics[i];
    const PicParams& P = b.params[i];
    for (size_t k = 0; k < pp.subs.size(); k++) {
      Substream s = pp.subs[k];
      s.pic = (uint32_t)i;
      if (s.dep_sub >= 0) s.dep_sub += (int32_t)sub_base;
      s.dependent = -1;
      subs[sub_base + k] = s;
    }
    for (size_t k = 0; k < pp.subs.size(); k++)
      if (subs[sub_base + k].dep_sub >= 0) subs[subs[sub_base + k].dep_sub].dependent = (int32_t)(sub_base + k);
    sub_base += (uint32_t)pp.subs.size();
    if (!b.chain) for (int y = 0; y < P.ctb_h; y++) { rows[r].pic = (uint32_t)i; rows[r].row = (uint32_t)y; r++; }
  }
  if (b.chain)   // by (motion step, item): k_motion takes a step's rows as one range (a picture's rows ascending: row r waits for row r - 1)
    for (int m = 0; m < (int)b.motion_steps.size(); m++)
      for (int i = 0; i < n; i++)
        if (b.motion_step_of[(size_t)i] == m)
          for (int y = 0; y < b.params[(size_t)i].ctb_h; y++) { rows[r].pic = (uint32_t)i; rows[r].row = (uint32_t)y; r++; }
  size_t total = 0;
  for (int i = 0; i < n; i++) total += sizes[b.src(i)];
  const size_t items_end = b.chain ? [&]() { for (const auto& ci : b.chain_items) if (ci.off_full_pic) return ci.off_full_pic; return b.upload_size; }() : b.upload_size;
  if (b.chain)
    for (int i = 0; i < n; i++) {
      const BatchLayout::ChainItem& ci = b.chain_items[(size_t)i];
      if (!ci.off_full_pic) continue;
      PicParams F = b.params[i];
      F.crop_x = F.crop_y = 0; F.out_width = F.width; F.out_height = F.height; F.out_cwidth = F.cwidth; F.out_cheight = F.cheight;
      for (int c = 0; c < 3; c++) { F.off_out[c] = ci.off_full[c]; F.out_stride[c] = ci.full_stride[c]; }
      memset(host + ci.off_full_pic, 0, align_up(sizeof(PicParams), 256));
      memcpy(host + ci.off_full_pic, &F, sizeof(F));
    }
  for_each_item(n, total, [&](int i) {
    const ParsedPicture& pp = b.pics[i];
    const PicParams& P = b.params[i];
    const size_t end = i + 1 < n ? (size_t)b.params[i + 1].off_ctb_ts_to_rs : items_end;   // this item's slice of the upload region
    auto put = [&](size_t off, const void* src, size_t bytes, size_t next) {
      memcpy(host + off, src, bytes);
      memset(host + off + bytes, 0, next - off - bytes);
    };
    put(P.off_ctb_ts_to_rs, pp.ts_to_rs.data(), pp.ts_to_rs.size() * sizeof(uint16_t), P.off_ctb_info);
    put(P.off_ctb_info, pp.ctb_info.data(), pp.ctb_info.size() * sizeof(CtbInfo), P.off_slices);
    put(P.off_slices, pp.slice_params.data(), pp.slice_params.size() * sizeof(SliceParams), P.off_scaling);
    if (P.scaling_lists) put(P.off_scaling, pp.scaling_tables.data(), pp.scaling_tables.size(), P.is_inter ? P.off_wp : P.off_bitstream);
    if (P.is_inter) {
      if (!pp.weight_tables.empty()) put(P.off_wp, pp.weight_tables.data(), pp.weight_tables.size() * sizeof(WeightTable), P.off_reftab);
      RefFrame tab[16];
      memset(tab, 0, sizeof(tab));
      for (size_t k = 0; k < pp.refs.size() && k < 16; k++) {
        const RefPicture rp = pp.refs[k].batch_item >= 0 ? chain_ref_picture(b, pp.refs[k].batch_item, arena_base) : pp.refs[k];
        for (int c = 0; c < 3; c++) { tab[k].plane[c] = rp.plane[c]; tab[k].stride[c] = rp.stride[c]; }
        tab[k].poc = rp.poc; tab[k].mf = rp.mf;
        if (pp.refs[k].batch_item >= 0) tab[k].progress_row = b.params[(size_t)pp.refs[k].batch_item].first_row + 1;
      }
      put(P.off_reftab, tab, sizeof(tab), P.off_bitstream);
    }
    put(P.off_bitstream, data[b.src(i)], sizes[b.src(i)], end);
  });
  b.parse_waves.clear(); b.parse_waves.shrink_to_fit();
  b.recon_waves.clear(); b.recon_waves.shrink_to_fit();
}

}  // namespace hipdec
