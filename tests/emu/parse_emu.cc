// parse_emu.cc — CPU-TEST-ONLY build of the device parser (libheif_amd/csrc/parse_core.h) with the
// 64 lanes emulated by loops, plus the pure-host front end and batch layout.  It lets the
// `-m "not gpu"` suite check the kernel's parsing logic (unit maps, coefficient levels, SAO
// parameters, substream termination) against the oracle without a GPU.  NOT part of the product:
// never linked into libheifhip.so, never used by libheif_amd/.
#define HIPDEC_HOST_EMU 1
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "batch_layout.h"
#include "emu_batch.h"
#include "hevc_headers.h"
#include "parse_core.h"

using namespace hipdec;


extern "C" {

EmuBatch* emu_create(int n, const uint8_t* const* data, const size_t* sizes, char* errbuf, size_t errlen)
{
  EmuBatch* b = new EmuBatch();
  std::vector<uint8_t> host;
  std::string err;
  int rc = layout_batch(b->L, n, (const void* const*)data, sizes, 0, host, err);
  if (rc != 0) {
    snprintf(errbuf, errlen, "%d: %s", rc, err.c_str());
    delete b;
    return nullptr;
  }
  b->arena.assign(b->L.arena_size + 1024, 0);
  memcpy(b->arena.data(), host.data(), host.size());
  return b;
}

void emu_free(EmuBatch* b) { delete b; }

// initValue of the device's context tables (parse_tables.h): table 0 = slice_type I, 1 / 2 = initType 1 / 2 of P slices; group 0..2 (A, B, C), lane 0..63
int emu_ctx_init_value(int table, int group, int lane)
{
  if (table < 0 || table > 2 || group < 0 || group > 2 || lane < 0 || lane > 63) return -1;
  return table == 0 ? pcore::c_init[group][lane] : pcore::c_init_p[table - 1][group][lane];
}

// ---- sequences: one access unit per emu_seq_create_picture(); emu_seq_commit() after the pipeline ran (pipeline_emu.cc) -----------------------
EmuSeq* emu_seq_new() { return new EmuSeq(); }
void emu_seq_free(EmuSeq* q)
{
  if (!q) return;
  for (EmuBatch* b : q->alive) delete b;
  for (auto* f : q->full_frames) delete f;
  delete q;
}
EmuBatch* emu_seq_create_picture(EmuSeq* q, const uint8_t* data, size_t size, char* errbuf, size_t errlen)
{
  EmuBatch* b = new EmuBatch();
  std::vector<uint8_t> host;
  std::string err;
  const void* ptrs[1] = {data};
  const size_t sizes[1] = {size};
  const SeqContext* seqs[1] = {&q->ctx};
  int rc = layout_batch(b->L, 1, ptrs, sizes, 0, host, err, seqs);
  if (rc != 0) {
    snprintf(errbuf, errlen, "%d: %s", rc, err.c_str());
    delete b;
    return nullptr;
  }
  b->arena.assign(b->L.arena_size + 1024, 0);
  memcpy(b->arena.data(), host.data(), host.size());
  q->alive.push_back(b);
  return b;
}

// A chain: n consecutive samples of the track parsed against the sequence state (each committed to a working copy before the next), ONE batch whose
// reference tables address the earlier pictures of the same arena (batch_layout.h: layout_batch_plan_chain, as the product's decoder with look-ahead)
EmuBatch* emu_seq_create_chain(EmuSeq* q, int n, const uint8_t* const* data, const size_t* sizes, char* errbuf, size_t errlen)
{
  EmuBatch* b = new EmuBatch();
  std::string err;
  int rc = layout_batch_plan_chain(b->L, n, (const void* const*)data, sizes, 0, err, q->ctx);
  if (rc != 0) {
    snprintf(errbuf, errlen, "%d: %s", rc, err.c_str());
    delete b;
    return nullptr;
  }
  if (!b->L.pics.empty()) {
    b->arena.assign(b->L.arena_size + 1024, 0);
    layout_batch_fill(b->L, (const void* const*)data, sizes, b->arena.data(), (uint64_t)(uintptr_t)b->arena.data());
  }
  q->alive.push_back(b);
  return b;
}
// Several tracks' chains in ONE batch (layout_batch_plan_chains, as the product's chain coalescer builds it): track t = samples [first[t], first[t] +
// count[t]) of data[] with sequence state qs[t].  The batch belongs to the CALLER (emu_free after every sequence that references it was freed).
EmuBatch* emu_seq_create_chains(EmuSeq* const* qs, int n_tracks, const int* first, const int* count, const uint8_t* const* data, const size_t* sizes,
                                char* errbuf, size_t errlen)
{
  EmuBatch* b = new EmuBatch();
  std::string err;
  std::vector<const SeqContext*> seqs;
  int total = 0;
  for (int t = 0; t < n_tracks; t++) { seqs.push_back(&qs[t]->ctx); total = std::max(total, first[t] + count[t]); }
  int bad = -1;
  int rc = layout_batch_plan_chains(b->L, n_tracks, first, count, (const void* const*)data, sizes, 0, err, seqs.data(), &bad);
  if (rc != 0) {
    snprintf(errbuf, errlen, "%d (track %d): %s", rc, bad, err.c_str());
    delete b;
    return nullptr;
  }
  if (!b->L.pics.empty()) {
    b->arena.assign(b->L.arena_size + 1024, 0);
    layout_batch_fill(b->L, (const void* const*)data, sizes, b->arena.data(), (uint64_t)(uintptr_t)b->arena.data());
  }
  return b;
}
// items of track t in decoding order (and which of the track's samples each one is); returns their number
int emu_track_items(EmuBatch* b, int t, int* items, int* samples, int cap)
{
  const BatchLayout::ChainTrack& tr = b->L.tracks[(size_t)t];
  for (size_t k = 0; k < tr.items.size() && (int)k < cap; k++) { items[k] = tr.items[k]; samples[k] = tr.samples[k]; }
  return (int)tr.items.size();
}
int emu_seq_commit_chains(EmuSeq* const* qs, int n_tracks, EmuBatch* b)
{
  for (int t = 0; t < n_tracks; t++) {
    std::vector<int> own;
    chain_resolve(b->L, (uint64_t)(uintptr_t)b->arena.data(), own, t);
    qs[t]->ctx = b->L.tracks[(size_t)t].seq_after;
  }
  return 0;
}
int emu_num_items(EmuBatch* b) { return (int)b->L.pics.size(); }
int emu_item_source(EmuBatch* b, int i) { return b->L.src(i); }
// the chain was decoded: the track's sequence state moves behind its last picture (its pictures addressed absolutely from here on)
int emu_seq_commit_chain(EmuSeq* q, EmuBatch* b)
{
  std::vector<int> own;
  chain_resolve(b->L, (uint64_t)(uintptr_t)b->arena.data(), own);
  q->ctx = b->L.tracks[0].seq_after;
  return 0;
}

// ---- the decoder's sample queue (hevc_headers.h: SampleQueue) on its own: access-unit splitting of pushed data, user_data attribution ----------
void* emu_sq_new() { return new SampleQueue(); }
void emu_sq_free(void* q) { delete (SampleQueue*)q; }
// push as hipdec_decoder_push_data does (framing validated first: -2 = End_of_data as the C ABI reports it); with_user_data: push_data2's argument follows
int emu_sq_push(void* q, const uint8_t* p, size_t size, int with_user_data, uint64_t user_data)
{
  for (size_t ptr = 0; ptr < size;) {
    if (size - ptr < 4) return -2;
    const uint32_t n = ((uint32_t)p[ptr] << 24) | ((uint32_t)p[ptr + 1] << 16) | ((uint32_t)p[ptr + 2] << 8) | p[ptr + 3];
    ptr += 4;
    if (n > size - ptr) return -2;
    ptr += n;
  }
  ((SampleQueue*)q)->push(p, size);
  if (with_user_data) ((SampleQueue*)q)->set_user_data((uintptr_t)user_data);
  return 0;
}
void emu_sq_close_first(void* q) { ((SampleQueue*)q)->first_closed = true; }   // the first picture was decoded
void emu_sq_drop(void* q, size_t n) { ((SampleQueue*)q)->drop_front(n); }
int emu_sq_count(void* q) { return (int)((SampleQueue*)q)->queue.size(); }
// i = -1: the first access unit, i = -2: the parameter sets, else queued sample i.  Returns its size (copied up to cap)
size_t emu_sq_get(void* q, int i, uint8_t* dst, size_t cap, uint64_t* user_data, int* has_vcl)
{
  SampleQueue* s = (SampleQueue*)q;
  const std::vector<uint8_t>* v = nullptr;
  if (i == -1) { v = &s->first; if (user_data) *user_data = s->first_user_data; if (has_vcl) *has_vcl = s->first_has_vcl; }
  else if (i == -2) v = &s->param_sets;
  else if (i >= 0 && i < (int)s->queue.size()) { v = &s->queue[(size_t)i].blob; if (user_data) *user_data = s->queue[(size_t)i].user_data; if (has_vcl) *has_vcl = s->queue[(size_t)i].has_vcl; }
  if (!v) return 0;
  if (dst && cap) memcpy(dst, v->data(), v->size() < cap ? v->size() : cap);
  return v->size();
}

}  // extern "C"

// The product's CABAC launchers (parse_kernel*.hip hold gfx950 assembly) for the host build of the WHOLE library (libheifhip_emu.so: decoder.hip,
// runtime.hip, ... compiled against the shim): the lane-emulated parser over the arguments decoder.hip:launch_all prepared, scheduled as emu_run_parse
// below does it.  One build of parse_core.h serves intra, inter and 4:2:2 / 4:4:4 batches here.
typedef void* hipStream_t;   // (as tests/emu/shim/hip/hip_runtime.h has it; this unit does not include the SIMT shim)
namespace hipdec {
void launch_parse(const ParseArgs& args, hipStream_t)
{
  ParseArgs A = args;
  if (!A.num_waves) return;
  pcore::Lds lds;
  memset(&lds, 0, sizeof(lds));
  if (A.pool) {
    A.num_waves = 1;
    pcore::parse_wave(A, 0, &lds);
    return;
  }
  std::vector<ParseWave> one(A.num_subs);
  for (uint32_t s = 0; s < A.num_subs; s++) one[s] = ParseWave{s, 1, s + 1, 2};
  A.waves = one.data();
  for (uint32_t s = 0; s < A.num_subs; s++) pcore::parse_wave(A, s, &lds);
}
void launch_parse_inter(const ParseArgs& a, hipStream_t s) { launch_parse(a, s); }
void launch_parse_general(const ParseArgs& a, bool, hipStream_t s) { launch_parse(a, s); }
}  // namespace hipdec

extern "C" {

// runs every substream in index order (a WPP predecessor always has a smaller index); returns the device status word
int emu_run_parse(EmuBatch* b)
{
  uint8_t* a = b->arena.data();
  memset(a + b->L.off_ctrl, 0, b->L.ctrl_size);
  ParseArgs A{};
  A.pics = (const PicParams*)(a + b->L.off_pics); A.subs = (const Substream*)(a + b->L.off_subs);
  A.waves = (const ParseWave*)(a + b->L.off_waves); A.num_waves = b->L.num_waves; A.arena = a;
  A.progress = (uint32_t*)(a + b->L.off_progress); A.ctx_store = a + b->L.off_ctx;
  A.ticket = (uint32_t*)(a + b->L.off_ticket); A.status = (int32_t*)(a + b->L.off_status);
  A.yield_ctbs = getenv("HIPDEC_POOL_YIELD") ? (uint32_t)atoi(getenv("HIPDEC_POOL_YIELD")) : 1;   // the product default (decoder.hip)
  A.wake_hyst = 2;
  A.pool = b->L.pool; A.queue_cap = b->L.queue_cap; A.num_subs = b->L.num_subs;
  A.waitneed = (uint32_t*)(a + b->L.off_waitneed); A.resume_k = (uint32_t*)(a + b->L.off_resume_k);
  A.queue = (uint32_t*)(a + b->L.off_queue); A.qctl = (uint32_t*)(a + b->L.off_qctl); A.saved = (uint32_t*)(a + b->L.off_saved);
  A.inter = b->L.any_inter ? 1 : 0;
  pcore::Lds lds;
  memset(&lds, 0, sizeof(lds));
  if (A.pool) {
    // pool mode: one emulated wave drains the ready queue (rows suspend and get re-queued exactly as on the device)
    A.num_waves = 1;
    pcore::parse_wave(A, 0, &lds);
  } else {
    // static mode: the waves one after the other in ticket order; a wave's rows are complete before a later wave needs them
    // only if every substream's predecessor has a smaller index AND is run first, so run substream by substream instead:
    // the per-wave table must cover every substream exactly once, then each wave entry is replayed with stride 1 semantics
    std::vector<uint8_t> covered(b->L.num_subs, 0);
    for (uint32_t w = 0; w < A.num_waves; w++)
      for (uint32_t s = A.waves[w].first; s < A.waves[w].end; s += A.waves[w].stride) covered[s]++;
    for (uint32_t s = 0; s < b->L.num_subs; s++)
      if (covered[s] != 1) { b->status = -1; return -1; }
    std::vector<ParseWave> one(b->L.num_subs);
    for (uint32_t s = 0; s < b->L.num_subs; s++) one[s] = ParseWave{s, 1, s + 1, 2};
    A.waves = one.data();
    for (uint32_t s = 0; s < b->L.num_subs; s++) pcore::parse_wave(A, s, &lds);
  }
  b->status = *(int32_t*)(a + b->L.off_status);
  return b->status;
}

int emu_info(EmuBatch* b, int i, int* out /* width, height, ctb_w, ctb_h, log2_ctb, chroma_format_idc, num_subs */)
{
  const PicParams& P = b->L.params[i];
  out[0] = P.width; out[1] = P.height; out[2] = P.ctb_w; out[3] = P.ctb_h; out[4] = P.log2_ctb; out[5] = P.chroma_format_idc;
  out[6] = (int)b->L.pics[i].subs.size();
  return 0;
}

static inline uint32_t il(uint32_t x, uint32_t y)
{
  x = (x | (x << 2)) & 0x33; x = (x | (x << 1)) & 0x55; y = (y | (y << 2)) & 0x33; y = (y | (y << 1)) & 0x55; return x | (y << 1);
}

// raster (4x4-unit) maps, same layout as hipdec_batch_read_maps
int emu_maps(EmuBatch* b, int i, uint8_t* log2_tb, uint8_t* log2_cb, uint8_t* intra_luma, uint8_t* intra_chroma, int8_t* qp_y, uint8_t* flags)
{
  const PicParams& P = b->L.params[i];
  const uint8_t* a = b->arena.data();
  const int uw = (P.width + 3) / 4, uh = (P.height + 3) / 4;
  const int l = P.log2_ctb - 2, mask = (1 << l) - 1;
  for (int uy = 0; uy < uh; uy++)
    for (int ux = 0; ux < uw; ux++) {
      const size_t idx = (((size_t)(uy >> l) * P.ctb_w + (ux >> l)) << P.units_per_ctb_log2) + il(ux & mask, uy & mask);
      const size_t o = (size_t)uy * uw + ux;
      log2_tb[o] = a[P.off_u_size + idx] & 15; log2_cb[o] = a[P.off_u_size + idx] >> 4;
      intra_luma[o] = a[P.off_u_ipm + idx] & 63; intra_chroma[o] = a[P.off_u_ipmc + idx];
      qp_y[o] = (int8_t)a[P.off_u_qp + idx]; flags[o] = a[P.off_u_flags + idx];
    }
  return 0;
}

// coefficient levels scattered to their spatial position (coded size planes, int32), like the oracle's `coeff` tap
int emu_coeffs(EmuBatch* b, int i, int32_t* y, int32_t* cb, int32_t* cr)
{
  const PicParams& P = b->L.params[i];
  const uint8_t* a = b->arena.data();
  const int ctb = 1 << P.log2_ctb, units = 1 << P.units_per_ctb_log2;
  int32_t* out[3] = {y, cb, cr};
  memset(y, 0, sizeof(int32_t) * (size_t)P.width * P.height);
  if (P.chroma_format_idc) { memset(cb, 0, sizeof(int32_t) * (size_t)P.cwidth * P.cheight); memset(cr, 0, sizeof(int32_t) * (size_t)P.cwidth * P.cheight); }
  for (int cy = 0; cy < P.ctb_h; cy++)
    for (int cx = 0; cx < P.ctb_w; cx++) {
      const int ctb_rs = cy * P.ctb_w + cx;
      const size_t base = (size_t)ctb_rs * units;
      int z = 0;
      while (z < units) {
        const int ux = (int)pcore::compact1by1((uint32_t)z), uy = (int)pcore::compact1by1((uint32_t)z >> 1);
        if (cx * ctb + ux * 4 >= P.width || cy * ctb + uy * 4 >= P.height) { z++; continue; }
        const int t = a[P.off_u_size + base + z] & 15;
        if (t < 2 || t > 5) return -1;
        const int fl = a[P.off_u_flags + base + z];
        if (fl & UF_CBF_LUMA) {
          const int16_t* src = (const int16_t*)(a + P.off_coeff[0]) + (size_t)ctb_rs * ctb * ctb + z * 16;
          const int n = 1 << t;
          for (int yy = 0; yy < n; yy++)
            for (int xx = 0; xx < n; xx++) y[(size_t)(cy * ctb + uy * 4 + yy) * P.width + cx * ctb + ux * 4 + xx] = src[yy * n + xx];
        }
        if (P.chroma_format_idc) {
          // chroma blocks of the unit: 4:4:4 one of the luma block's size; 4:2:0 one of half the size (the 4x4 luma quads: one 4x4 block hanging off
          // the 4th unit); 4:2:2 like 4:2:0 but TWO blocks one above the other, the lower one's flags in unit (z ^ 1)
          const int cfi = P.chroma_format_idc;
          const bool c444 = cfi == 3;
          const int csw = c444 ? 0 : 1, csh = cfi == 1 ? 1 : 0;
          int do_c = 0, zc = z, tc = t - 1;
          if (c444) { do_c = 1; tc = t; }
          else if (t > 2) do_c = 1; else if ((z & 3) == 3) { do_c = 1; zc = z & ~3; tc = 2; }
          if (do_c) {
            const int cux = (int)pcore::compact1by1((uint32_t)zc), cuy = (int)pcore::compact1by1((uint32_t)zc >> 1);
            const int mult = c444 ? 16 : (cfi == 2 ? 8 : 4);
            const int n = 1 << tc;
            for (int c = 1; c < 3; c++)
              for (int low = 0; low < (cfi == 2 ? 2 : 1); low++) {
                const int f = low ? a[P.off_u_flags + base + (z ^ 1)] : fl;
                if (!(f & (c == 1 ? UF_CBF_CB : UF_CBF_CR))) continue;
                const int16_t* src = (const int16_t*)(a + P.off_coeff[c]) + (size_t)ctb_rs * ((ctb * ctb) >> (csw + csh)) + zc * mult + low * n * n;
                for (int yy = 0; yy < n; yy++)
                  for (int xx = 0; xx < n; xx++)
                    out[c][(size_t)(((cy * ctb + cuy * 4) >> csh) + low * n + yy) * P.cwidth + ((cx * ctb + cux * 4) >> csw) + xx] = src[yy * n + xx];
              }
          }
        }
        z += 1 << (2 * (t - 2));
      }
    }
  return 0;
}

// The parser's byte reader (bitstream window + emulation-prevention skipping) on its own: reads bytes [start, end) of `buf`
// (which must be followed by >= 768 readable bytes) into `out` and returns the count.  Every `resume_every` bytes (0 = never)
// the reader's position state is carried into a FRESH reader, the way a suspended WPP row resumes (parse_substream): pos, end
// and the zero run are kept, the window is loaded again.
int emu_read_bytes(const uint8_t* buf, uint32_t start, uint32_t end, uint8_t* out, uint32_t max_out, uint32_t resume_every)
{
  pcore::PS* s = new pcore::PS();
  memset((void*)s, 0, sizeof(*s));
  s->bs = buf;
  s->pos = start; s->end = end; s->zeros = 0; s->win_base = 0xfffff000u; s->fast_limit = 0;
  uint32_t n = 0;
  while (s->pos < s->end && n < max_out) {
    if (resume_every && n && n % resume_every == 0) {
      pcore::PS* t = new pcore::PS();
      memset((void*)t, 0, sizeof(*t));
      t->bs = buf; t->pos = s->pos; t->end = s->end; t->zeros = s->zeros; t->win_base = 0xfffff000u; t->fast_limit = 0;
      delete s; s = t;
    }
    const uint32_t b = pcore::read_byte(*s);
    if (s->pos > s->end) break;      // the byte consumed was an emulation-prevention byte at the very end
    out[n++] = (uint8_t)b;
  }
  delete s;
  return (int)n;
}

// SAO parameters per CTB (raster) and component: type, band_or_class, 4 offsets
int emu_sao(EmuBatch* b, int i, uint8_t* type, uint8_t* cls, int16_t* offsets)
{
  const PicParams& P = b->L.params[i];
  const SaoParams* sp = (const SaoParams*)(b->arena.data() + P.off_sao);
  const int n = P.ctb_w * P.ctb_h * 3;
  for (int k = 0; k < n; k++) { type[k] = sp[k].type; cls[k] = sp[k].band_or_class; for (int j = 0; j < 4; j++) offsets[k * 4 + j] = sp[k].offset[j]; }
  return 0;
}

}  // extern "C"
