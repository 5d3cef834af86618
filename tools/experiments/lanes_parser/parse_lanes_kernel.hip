// parse_lanes_kernel.hip — CABAC entropy decoding + syntax parsing, one LANE per independent substream (throughput mode).
//
// parse_core.h spends a whole wavefront on one substream: the arithmetic decoder is a serial chain, so 63 of the 64 lanes only serve as
// a register file, and the CU-shared scalar pipe is the bottleneck.  Here every lane is a complete parser of its own
// substream — its own arithmetic decoder (range / offset / bit count in its registers), its own byte reader (emulation prevention
// included), its own context variables (a byte column in LDS: ctx[context][lane]) — and the 64 parsers of a wave run the SAME bin-level
// state machine in lockstep:
//
//   every iteration of the wave's loop    1. each lane decodes the ONE bin (or bypass group, terminate bin, coeff_abs_level_remaining)
//                                            its syntax position asked for,
//                                         2. each lane consumes the result in the state it is in (a `switch` over ~40 syntax states,
//                                            ordered so that the frequent transitions are fall-throughs) and files its next request.
//
//   Lanes never wait for each other's syntax: a lane inside a 32x32 block and a lane parsing a CU header both advance by one bin per
//   iteration.  What diverges is only step 2, whose cost is the sum of the bodies of the states that are populated — a few tens of
//   instructions each — against 64 bins decoded.  (tools/ubench/cabac_multistream_ubench.hip measured the ceiling of step 1.)
//
// Substreams are dealt to lanes through a host-built table (ParseArgs::lane_subs), sorted by (row index inside the picture, picture):
// a wave holds the same CTB row of 64 pictures when the batch is large (all lanes busy at the same time), and degenerates to "the rows of
// one picture" for a single still (the WPP chain then runs across the lanes of one wave).  A WPP predecessor is always an earlier table
// entry, i.e. the same wave or one with a smaller ticket, and a lane that waits for its predecessor simply files no request (the wave keeps
// iterating for the others), so there is no intra-wave deadlock; every wait is bounded.
//
// Outputs are those of parse_core.h, written by the lane straight to HBM: unit maps (z-order), TransCoeffLevels (TU-contiguous; a coded
// block is zero-filled, then the non-zero levels are stored one by one), SAO parameters, the per-CTB hand-off record and the WPP context
// snapshot (write-through stores + one relaxed progress store, as in parse_core.h).  Neighbour look-ups (split_cu_flag context, most
// probable modes, QP prediction) read the lane's own earlier stores.
//
// Follows the same syntax sections as parse_core.h: 7.3.8.2-7.3.8.12, 9.3 (references: libde265 slice.cc read_coding_tree_unit ..
// residual_coding, cabac.cc; /root/reference/libheif/plugins/decoder_libde265.cc:257 is the call that runs them).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hevc_device.h"
#include "kernels.h"
#define PC_CONST __constant__
#include "parse_tables.h"

namespace hipdec {
namespace planes {
using namespace pcore;

#if defined(HIPDEC_HOST_EMU)
#define PL_DRAIN() ((void)0)
#define PL_DEV static inline
static inline int pl_clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static inline int pl_popc(uint32_t x) { return __builtin_popcount(x); }
static inline int pl_ffs(uint32_t x) { return __builtin_ffs((int)x); }
#else
#define PL_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PL_DEV __device__ __forceinline__
PL_DEV int pl_clz(uint32_t x) { return __clz((int)x); }
PL_DEV int pl_popc(uint32_t x) { return __popc(x); }
PL_DEV int pl_ffs(uint32_t x) { return __ffs((int)x); }
#endif

enum : int32_t { K_NONE = 0, K_CTX, K_BYP, K_TERM, K_REM };
enum : int32_t {
  S_CTB = 0, S_CQT, S_SPLIT, S_SPLIT_R, S_CU, S_TQB_R, S_PART_R, S_PREV_R, S_IPM, S_MPM1_R, S_MPM2_R, S_REMMODE_R, S_CHROMA_R, S_CHROMA2_R,
  S_TSPLIT_R, S_CBFCB_R, S_CBFCR_R, S_TT, S_CBFL_R, S_QPD0_R, S_QPD1_R, S_QPD_EGP_R, S_QPD_EGS_R, S_QPD_SIGN_R,
  S_TS_R, S_LASTX_R, S_LASTY_R, S_LASTXS_R, S_LASTYS_R, S_CSBF_R, S_SIG_R, S_G1_R, S_G2_R, S_SIGN_R, S_REM_R, S_SB, S_RES,
  S_EOS_R, S_EOS2_R, S_DONE
};
enum : uint32_t { TOOL_SDH = 1, TOOL_TS = 2, TOOL_CUQPD = 4, TOOL_TQBYPASS = 8 };
enum : int { CTX_A = 0, CTX_B = 64, CTX_C = 128, N_CTX = 192 };

struct alignas(16) Q4 { uint32_t v[4]; };

enum : int { LN_LEFT_CB = 0, LN_UP_CB = 1, LN_LEFT_IPM = 2, LN_UP_IPM = 3, LN_LEFT_QP = 4, LN_UP_QP = 5 };

struct Shared {
  uint8_t ctx[N_CTX * 64];   // context variable pStateIdx | valMps << 6 of lane l: ctx[c * 64 + l]
  uint32_t ring[16 * 64];    // 64 bytes of bitstream look-ahead per lane: dword (pos >> 2) & 15 of lane l at ring[slot * 64 + l]
  uint32_t line[6 * 4 * 64]; // neighbour line buffers per lane (LN_*): entries 4j..4j+3 (bytes) of array a of lane l at line[(a * 4 + j) * 64 + l].
                             //   "left" arrays are indexed by the unit row inside the CTB, "up" arrays by the unit column: the value of the block
                             //   decoded last in that row / column, which in z-scan order is the left / upper neighbour of the next one
  uint32_t sao[9 * 64];      // SaoParams dwords of the lane's current CTB (the previous CTB's until parse_sao rewrites them: sao_merge_left)
  uint32_t t_lps[64];        // rangeTabLps[p][0..3] packed
  uint8_t t_next[64];        // transIdxLps[p], | 64 where valMps flips (p = 0)
  uint8_t diag8[64];         // k-th position of the 8x8 up-right diagonal scan, x | y << 3
  uint8_t inv8[64];          // its inverse
};

// the whole parser state of one substream: registers of its lane
struct LS {
  // arithmetic decoder (9.3.4.3, scaled-window formulation of parse_core.h) + byte reader
  uint32_t range, value;
  int32_t bits;
  const uint8_t* bs;
  uint32_t pos, end, cur, fill;   // fill: stream offset (multiple of 16) up to which the ring is loaded
  uint64_t rv;                    // reservoir: next unescaped bytes, first one in bits 63..56
  int32_t rn;                     // bytes in it
  int32_t zeros, err;
  // request to the decode step / syntax state
  int32_t kind, arg, state;
  // substream
  uint32_t sub, num_ctbs, first_ctb_ts, dep_len, k, sflags, waits;
  int32_t dep_sub;
  // picture / slice
  const PicParams* P;
  int32_t width, height, ctb_w, log2_ctb, log2_min_cb, log2_min_tb, log2_max_tb, max_th_depth, chroma, bd_luma, bd_chroma, log2_min_qg;
  uint32_t tools;
  int32_t slice_qp, deblock, sao_luma, sao_chroma;
  uint64_t o_size, o_flags, o_ipm, o_ipmc, o_qp, o_coef0, o_coef1, o_coef2;
  // CTB
  int32_t ctb_rs, x_ctb, y_ctb, avail;
  uint32_t ubase;
  uint64_t up_lo, up_hi;   // size bytes of the bottom unit row of the CTB above (hand-off record dwords 9..12)
  int32_t p, lg;
  // CU
  int32_t zb, log2cb, part_nxn, pk, chroma_mode, tqb;
  uint32_t prev_flags, ipm_pack;
  int32_t mpm_idx;
  int32_t qp_coded, qp_delta, qp_pred, last_qp, cur_qp, qv, qk;
  // transform tree / TU
  int32_t q, t, stage, split, cbf_luma, c, tc, zc, do_chroma, luma_mode;
  uint32_t cbf_cb_bits, cbf_cr_bits, ts_bits;
  // residual_coding
  int32_t lgn, px, py, last_x, last_y, scan_idx, ts;
  int16_t* dst;
  int32_t i, last_sb, last_pos, xs, ys, infer_dc, g1_carry, first_g1_sb, sig_off, kpos;
  uint64_t csbf;
  uint32_t pat, sig, g1, g1_coded, g2, rem_mask;
  int32_t g1_ctx, num_g1, last_g1_pos, ctx_set;
  uint32_t sign_bits, sig_signed, need_rem, emit, parity;
  int32_t n_signs, rice, sign_hidden, first_sig_pos;
};

PL_DEV uint32_t interleave4(uint32_t x, uint32_t y)
{
  x = (x | (x << 2)) & 0x33; x = (x | (x << 1)) & 0x55;
  y = (y | (y << 2)) & 0x33; y = (y | (y << 1)) & 0x55;
  return x | (y << 1);
}
PL_DEV uint32_t compact1by1(uint32_t v)
{
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu;
  return v & 0xffu;
}
PL_DEV void wt_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PL_DEV uint32_t wt_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- byte reader ------------------------------------------------------------------------------------------------------------------
// Two levels of look-ahead per lane, both refilled at ONE place of the wave's loop (parse_lanes_wave) so that the decode paths hold neither
// a global load (its s_waitcnt would also wait for the lane's outstanding stores, and the wave for the lane) nor the emulation-prevention logic:
//   ring       64 raw bytes of the NAL payload in LDS, topped up by all lanes together when one of them is below RING_LOW bytes
//   reservoir  up to 8 bytes with the emulation prevention bytes (7.4.2) already removed, in a register pair; take_byte() is a shift.
// One request consumes at most 6 bytes of a valid stream (coeff_abs_level_remaining: <= 20 + 21 bins); a corrupt one that asks for more gets zero
// bytes and is caught by the position check.  The only longer run, the SAO syntax at a CTB start, refills as it goes.
enum : uint32_t { RING_LOW = 40 };
PL_DEV void ring_fill(LS& L, Shared& S, int lane)
{
  while (L.fill - (L.pos & ~15u) <= 48u) {
    const Q4 v = *(const Q4*)(L.bs + L.fill);
    const uint32_t slot = (L.fill >> 2) & 15u;
    S.ring[(slot + 0) * 64 + lane] = v.v[0]; S.ring[(slot + 1) * 64 + lane] = v.v[1];
    S.ring[(slot + 2) * 64 + lane] = v.v[2]; S.ring[(slot + 3) * 64 + lane] = v.v[3];
    L.fill += 16u;
  }
}
PL_DEV void reservoir_fill(LS& L, Shared& S, int lane)
{
  if (L.pos - (uint32_t)L.rn > L.end + 8u) L.err = DEV_ERR_BITSTREAM_END;   // consumed more than 8 bytes behind the end of the substream
#pragma clang loop unroll(disable)
  while (L.rn < 8) {
    uint32_t b = 0;
    if (L.pos >= L.end) L.pos++;
    else {
      b = (L.cur >> ((L.pos & 3u) * 8u)) & 255u;
      L.pos++;
      if ((L.pos & 3u) == 0) L.cur = S.ring[((L.pos >> 2) & 15u) * 64 + lane];
      if (L.zeros >= 2 && b == 3u && L.pos < L.end) { L.zeros = 0; continue; }   // emulation_prevention_three_byte
      L.zeros = b == 0 ? L.zeros + 1 : 0;
    }
    L.rv |= (uint64_t)b << (56 - 8 * L.rn);
    L.rn++;
  }
}
PL_DEV void reader_start(LS& L, Shared& S, int lane, uint32_t start, uint32_t end)
{
  L.pos = start; L.end = end; L.zeros = 0;
  L.fill = start & ~15u;
  ring_fill(L, S, lane);
  L.cur = S.ring[((start >> 2) & 15u) * 64 + lane];
  L.rv = 0; L.rn = 0;
  reservoir_fill(L, S, lane);
}
PL_DEV uint32_t next_byte(LS& L)
{
  const uint32_t b = (uint32_t)(L.rv >> 56);
  L.rv <<= 8;
  if (L.rn > 0) L.rn--; else L.err = DEV_ERR_SYNTAX;   // more than 8 bytes in one request: no valid stream does that
  return b;
}

// ---- arithmetic decoder ------------------------------------------------------------------------------------------------------------
PL_DEV void cabac_start(LS& L, Shared& S, int lane, uint32_t start, uint32_t end)
{
  reader_start(L, S, lane, start, end);
  L.range = 510u; L.bits = -8;
  const uint32_t b0 = next_byte(L), b1 = next_byte(L);
  L.value = (b0 << 8) | b1;
}
PL_DEV uint32_t dec_ctx(LS& L, Shared& S, int lane, int c)
{
  const uint32_t st = S.ctx[c * 64 + lane];
  const uint32_t row = S.t_lps[st & 63u];
  const uint32_t lps = (row >> ((L.range >> 3) & 24u)) & 255u;
  const uint32_t r_mps = L.range - lps, scaled = r_mps << 7;
  const bool is_lps = L.value >= scaled;
  const uint32_t nb = is_lps ? (uint32_t)pl_clz(lps) - 23u : 1u - (scaled >> 15);
  const uint32_t st_m = st + (((st & 63u) != 62u) ? 1u : 0u);
  const uint32_t st_l = ((uint32_t)S.t_next[st & 63u] & 127u) ^ (st & 64u);
  S.ctx[c * 64 + lane] = (uint8_t)(is_lps ? st_l : st_m);
  L.range = (is_lps ? lps : r_mps) << nb;
  L.value = (is_lps ? L.value - scaled : L.value) << nb;
  L.bits += (int32_t)nb;
  if (L.bits >= 0) { L.value += next_byte(L) << L.bits; L.bits -= 8; }
  return (st >> 6) ^ (is_lps ? 1u : 0u);
}
PL_DEV uint32_t dec_byp1(LS& L, Shared& S, int lane)
{
  L.value <<= 1;
  L.bits += 1;
  if (L.bits >= 0) { L.bits = -8; L.value += next_byte(L); }
  const uint32_t scaled = L.range << 7;
  if (L.value >= scaled) { L.value -= scaled; return 1u; }
  return 0u;
}
// n <= 8 bypass bins at once: n steps of 9.3.4.3.4 are one long division of the scaled window by the scaled range (quotient = the bins, MSB
// first; remainder = the new offset); at most one byte is needed.  value < scaled * 2^n <= 2^24 and scaled < 2^16 are exact in fp32: the
// quotient estimate from one reciprocal is off by at most one, which the remainder check repairs (as decode_bypass_multi of parse_core.h).
PL_DEV uint32_t dec_byp_multi(LS& L, int n)
{
  L.value <<= n;
  L.bits += n;
  if (L.bits >= 0) { L.value += next_byte(L) << L.bits; L.bits -= 8; }
  const uint32_t scaled = L.range << 7;
#if defined(HIPDEC_HOST_EMU)
  uint32_t q = (uint32_t)((float)L.value * (1.0f / (float)scaled));
#else
  uint32_t q = (uint32_t)((float)L.value * __builtin_amdgcn_rcpf((float)scaled));
#endif
  int32_t r = (int32_t)(L.value - q * scaled);
  if (r < 0) { q -= 1u; r += (int32_t)scaled; }
  else if ((uint32_t)r >= scaled) { q += 1u; r -= (int32_t)scaled; }
  const uint32_t qmax = (1u << n) - 1u;
  if (q > qmax) { r += (int32_t)((q - qmax) * scaled); q = qmax; }   // only reachable on a corrupt stream
  L.value = (uint32_t)r;
  return q;
}
PL_DEV uint32_t dec_byp(LS& L, Shared& S, int lane, int n)   // n <= 32 bins, MSB first
{
  uint32_t v = 0;
#pragma clang loop unroll(disable)
  while (n > 0) {
    const int c = n > 8 ? 8 : n;
    v = (v << c) | dec_byp_multi(L, c);
    n -= c;
  }
  return v;
}
PL_DEV uint32_t dec_term(LS& L, Shared& S, int lane)
{
  L.range -= 2u;
  const uint32_t scaled = L.range << 7;
  if (L.value >= scaled) return 1u;
  if (scaled < (256u << 7)) {
    L.range = scaled >> 6;
    L.value <<= 1;
    L.bits += 1;
    if (L.bits == 0) { L.bits = -8; L.value += next_byte(L); }
  }
  return 0u;
}
PL_DEV uint32_t dec_rem(LS& L, Shared& S, int lane, int rice)   // 9.3.3.11 coeff_abs_level_remaining
{
  int prefix = 0;
  while (prefix < 32 && dec_byp1(L, S, lane)) prefix++;
  if (prefix >= 32) { L.err = DEV_ERR_SYNTAX; return 0; }
  if (prefix <= 3) return ((uint32_t)prefix << rice) + dec_byp(L, S, lane, rice);
  return ((((1u << (prefix - 3)) + 3u - 1u) << rice)) + dec_byp(L, S, lane, prefix - 3 + rice);
}

// ---- contexts (9.3.2.2; 9.3.2.4 storage / synchronisation for WPP) ----------------------------------------------------------------------
PL_DEV void init_contexts(LS& L, Shared& S, int lane)
{
  const int qp = L.slice_qp < 0 ? 0 : (L.slice_qp > 51 ? 51 : L.slice_qp);
#pragma clang loop unroll(disable)
  for (int c = 0; c < N_CTX; c++) {
    const int init = c_init[c >> 6][c & 63];
    const int m = (init >> 4) * 5 - 45, n = ((init & 15) << 3) - 16;
    int pre = ((m * qp) >> 4) + n;
    pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
    const int mps = pre <= 63 ? 0 : 1;
    const int p_state = mps ? pre - 64 : 63 - pre;
    S.ctx[c * 64 + lane] = (uint8_t)(p_state | (mps << 6));
  }
}
PL_DEV void load_contexts(Shared& S, int lane, const uint32_t* src)
{
#pragma clang loop unroll(disable)
  for (int j = 0; j < N_CTX / 4; j++) {
    const uint32_t w = wt_load(src + j);
    for (int b = 0; b < 4; b++) S.ctx[(4 * j + b) * 64 + lane] = (uint8_t)(w >> (8 * b));
  }
}
PL_DEV void save_contexts(Shared& S, int lane, uint32_t* dst)
{
#pragma clang loop unroll(disable)
  for (int j = 0; j < N_CTX / 4; j++) {
    uint32_t w = 0;
    for (int b = 0; b < 4; b++) w |= (uint32_t)S.ctx[(4 * j + b) * 64 + lane] << (8 * b);
    wt_store(dst + j, w);
  }
}

// ---- unit maps in HBM -------------------------------------------------------------------------------------------------------------------
PL_DEV void fill_units(uint8_t* p, int n, uint32_t b)   // n = 1, 4, 16, 64, 256 units, p aligned to n
{
  const uint32_t w = b * 0x01010101u;
  if (n >= 16) {
    const Q4 q{{w, w, w, w}};
#pragma clang loop unroll(disable)
    for (int i = 0; i < n; i += 16) *(Q4*)(p + i) = q;
  }
  else if (n == 4) *(uint32_t*)p = w;
  else *p = (uint8_t)b;
}
PL_DEV uint32_t line_get(const Shared& S, int lane, int arr, int i) { return (S.line[(arr * 4 + (i >> 2)) * 64 + lane] >> ((i & 3) * 8)) & 255u; }
PL_DEV void line_set(Shared& S, int lane, int arr, int i0, int n, uint32_t v)   // n = 1, 2, 4, 8, 16 entries from i0 (a multiple of n)
{
  v &= 255u;
  if (n >= 4) {
#pragma clang loop unroll(disable)
    for (int j = 0; j < n; j += 4) S.line[(arr * 4 + ((i0 + j) >> 2)) * 64 + lane] = v * 0x01010101u;
  } else {
    const uint32_t mask = (n == 2 ? 0xffffu : 0xffu) << ((i0 & 3) * 8);
    uint32_t& w = S.line[(arr * 4 + (i0 >> 2)) * 64 + lane];
    w = (w & ~mask) | ((v * 0x01010101u) & mask);
  }
}
// log2 CB size of the unit left of / above unit (ux, uy) of the current CTB, or 0 if unavailable
PL_DEV int left_cb_log2(const LS& L, const Shared& S, int lane, int ux, int uy)
{
  if (ux > 0 || (L.avail & AV_LEFT)) return (int)line_get(S, lane, LN_LEFT_CB, uy);
  return 0;
}
PL_DEV int up_cb_log2(const LS& L, const Shared& S, int lane, int ux, int uy)
{
  if (uy > 0) return (int)line_get(S, lane, LN_UP_CB, ux);
  if (L.avail & AV_UP) {
    const uint64_t lo = L.up_lo, hi = L.up_hi;   // (two 64-bit values, not four dwords: a select chain over adjacent struct fields turns
    const uint64_t w = (ux & 8) ? hi : lo;        //  into an indexed load, which keeps the whole lane state in scratch memory)
    return (int)(((uint32_t)(w >> ((ux & 7) * 8)) & 255u) >> 4);
  }
  return 0;
}
PL_DEV void derive_qp_pred(LS& L, const Shared& S, int lane, int ux, int uy)   // 8.6.1 (qPY_A / qPY_B only count inside the current CTB)
{
  int a = L.last_qp, b = L.last_qp;
  if (ux > 0) a = (int8_t)line_get(S, lane, LN_LEFT_QP, uy);
  if (uy > 0) b = (int8_t)line_get(S, lane, LN_UP_QP, ux);
  L.qp_pred = (a + b + 1) >> 1;
}
PL_DEV void set_qp_y(LS& L)
{
  const int off = 6 * (L.bd_luma - 8);
  L.cur_qp = ((L.qp_pred + L.qp_delta + 52 + 2 * off) % (52 + off)) - off;
}

// ---- 7.3.8.3 sao: once per CTB, plain per-lane code (the lanes that are at a CTB start run it, the others wait) -----------------------
//   SaoParams dwords per component c: 3c+0 type | band_or_class << 8 | offset[0] << 16, 3c+1 offset[1] | offset[2] << 16, 3c+2 offset[3]
PL_DEV void parse_sao(LS& L, Shared& S, int lane, uint8_t* arena, int allow_left, int allow_up)
{
  int merge_left = 0, merge_up = 0;
  if (allow_left) merge_left = (int)dec_ctx(L, S, lane, CTX_A + A_SAO_MERGE);
  if (allow_up && !merge_left) merge_up = (int)dec_ctx(L, S, lane, CTX_A + A_SAO_MERGE);
  if (merge_left) return;   // S.sao still holds the parameters of the CTB to the left
  if (merge_up) {
    const uint32_t* src = (const uint32_t*)(arena + L.P->off_handoff) + (size_t)(L.ctb_rs - L.ctb_w) * HANDOFF_DWORDS;
    for (int j = 0; j < 9; j++) S.sao[j * 64 + lane] = wt_load(src + j);
    return;
  }
  const int ncomp = L.chroma ? 3 : 1;
  int type1 = 0, cls1 = 0;
  for (int c = 0; c < 3; c++) {
    uint32_t d0 = 0, d1 = 0, d2 = 0;
    const int on = c < ncomp && (c == 0 ? +L.sao_luma : +L.sao_chroma);
    if (on) {
      int type;
      if (c == 2) type = type1;
      else { type = 0; if (dec_ctx(L, S, lane, CTX_A + A_SAO_TYPE)) type = dec_byp1(L, S, lane) ? 2 : 1; }
      if (c == 1) type1 = type;
      if (type) {
        const int bd = c ? +L.bd_chroma : +L.bd_luma;
        const int c_max = (1 << ((bd < 10 ? bd : 10) - 5)) - 1;
        uint32_t a = 0, sg = 0xCu;   // |offset| four bytes; sign bits (edge offset: + + - -)
        for (int i = 0; i < 4; i++) { uint32_t v = 0; while ((int)v < c_max && dec_byp1(L, S, lane)) v++; a |= v << (8 * i); reservoir_fill(L, S, lane); }
        int cls;
        if (type == 1) {
          sg = 0;
          for (int i = 0; i < 4; i++) if ((a >> (8 * i)) & 255u) sg |= dec_byp1(L, S, lane) << i;
          cls = (int)dec_byp(L, S, lane, 5);
        } else {
          if (c == 2) cls = cls1; else cls = (int)dec_byp(L, S, lane, 2);
        }
        if (c == 1) cls1 = cls;
        const int sh = bd - (bd < 10 ? bd : 10);
        uint32_t o0, o1, o2, o3;
        {
          const int v0 = (int)(a & 255u), v1 = (int)((a >> 8) & 255u), v2 = (int)((a >> 16) & 255u), v3 = (int)(a >> 24);
          o0 = (uint32_t)(uint16_t)(int16_t)(((sg & 1u) ? -v0 : v0) << sh); o1 = (uint32_t)(uint16_t)(int16_t)(((sg & 2u) ? -v1 : v1) << sh);
          o2 = (uint32_t)(uint16_t)(int16_t)(((sg & 4u) ? -v2 : v2) << sh); o3 = (uint32_t)(uint16_t)(int16_t)(((sg & 8u) ? -v3 : v3) << sh);
        }
        d0 = (uint32_t)type | ((uint32_t)cls << 8) | (o0 << 16); d1 = o1 | (o2 << 16); d2 = o3;
      }
    }
    S.sao[(3 * c + 0) * 64 + lane] = d0; S.sao[(3 * c + 1) * 64 + lane] = d1; S.sao[(3 * c + 2) * 64 + lane] = d2;
    ring_fill(L, S, lane);   // a component is at most 4 * 31 + 9 bins
    reservoir_fill(L, S, lane);
  }
}

// ---- pieces of residual_coding (7.3.8.11) shared by several states ------------------------------------------------------------------------
PL_DEV uint64_t scan4_of(int scan_idx) { return scan_idx == 0 ? PC_DIAG4 : (scan_idx == 1 ? PC_HORZ4 : PC_VERT4); }
PL_DEV int sig_ctx(const LS& L, int k)   // context of sig_coeff_flag at scan position k of the current sub-block (9.3.4.2.5)
{
  const uint32_t r = (uint32_t)(scan4_of(L.scan_idx) >> (k * 4)) & 15u;
  uint32_t c;
  if (L.lgn == 2) c = (uint32_t)L.sig_off + (uint32_t)((PC_CTXIDXMAP4 >> (r * 4)) & 15u);
  else if ((L.xs | L.ys) == 0 && r == 0) c = L.c ? 27u : 0u;
  else c = (uint32_t)L.sig_off + ((L.pat >> (r * 2)) & 3u);
  return CTX_B + B_SIG_COEFF + (int)c;
}
PL_DEV void start_g1(LS& L)   // sig != 0: request the first coeff_abs_level_greater1_flag
{
  L.g1 = 0; L.g1_coded = 0; L.g2 = 0;
  L.ctx_set = (L.i == 0 || L.c > 0) ? 0 : 2;
  if (!L.first_g1_sb && L.g1_carry == 0) L.ctx_set++;
  L.first_g1_sb = 0;
  L.g1_ctx = 1; L.num_g1 = 0; L.last_g1_pos = -1;
  L.rem_mask = L.sig;
  L.kind = K_CTX; L.arg = CTX_C + C_GREATER1 + L.ctx_set * 4 + (L.c ? 16 : 0) + 1; L.state = S_G1_R;
}
PL_DEV void begin_sig(LS& L, int right, int below)   // the sub-block (xs, ys) is coded: first sig_coeff_flag request
{
  if (L.lgn == 2) { L.pat = 0; L.sig_off = L.c ? 27 : 0; }
  else {
    const int prev_csbf = right | (below << 1);
    L.pat = prev_csbf == 0 ? PC_SIGPAT0 : (prev_csbf == 1 ? PC_SIGPAT1 : (prev_csbf == 2 ? PC_SIGPAT2 : PC_SIGPAT3));
    if (L.c == 0) L.sig_off = ((L.xs | L.ys) ? 3 : 0) + ((L.lgn == 3) ? (L.scan_idx == 0 ? 9 : 15) : 21);
    else L.sig_off = 27 + ((L.lgn == 3) ? 9 : 12);
  }
  L.sig = 0;
  int n_start = 15;
  if (L.i == L.last_sb) { L.sig = 1u << L.last_pos; n_start = L.last_pos - 1; }
  if (n_start < 0) { start_g1(L); return; }
  L.kpos = n_start;
  L.kind = K_CTX; L.arg = sig_ctx(L, n_start); L.state = S_SIG_R;
}
PL_DEV void store_level(LS& L, int k, uint32_t a)   // scan position k of the current sub-block gets |level| = a
{
  L.parity ^= a & 1u;
  uint32_t neg = 0;
  if ((L.sig_signed >> k) & 1u) {
    const int rank = pl_popc(L.sig_signed >> (k + 1));   // sign bins decoded before this position's (MSB = highest scan position)
    neg = (L.sign_bits >> (L.n_signs - 1 - rank)) & 1u;
  } else neg = L.parity & 1u;                           // the hidden sign (9.3.4.3 / 7.3.8.11 signHidden): parity of the level sum; this is the last level
  if (a > 32768u || (a == 32768u && !neg)) L.err = DEV_ERR_SYNTAX;
  const uint32_t r = (uint32_t)(scan4_of(L.scan_idx) >> (k * 4)) & 15u;
  const int n = 1 << L.lgn;
  L.dst[((L.ys << 2) + (int)(r >> 2)) * n + (L.xs << 2) + (int)(r & 3u)] = (int16_t)(neg ? -(int32_t)a : (int32_t)a);
}

#define REQ(K, A, ST) do { L.kind = (K); L.arg = (A); L.state = (ST); goto step_done; } while (0)
#define HOP(ST) do { L.kind = K_NONE; L.state = (ST); goto step_done; } while (0)

// one step of the lane's parser: consume the result `r` of the request filed last, run to the next request
PL_DEV void step(LS& L, Shared& S, const int lane, const ParseArgs& A, uint32_t r)
{
  uint8_t* const arena = A.arena;
  switch (L.state) {
  // ======================================================= coding_tree_unit =======================================================
  case S_CTB: {
    if (L.k >= L.num_ctbs) { L.kind = K_NONE; L.state = S_DONE; goto step_done; }
    const PicParams* P = L.P;
    const uint32_t ts = L.first_ctb_ts + L.k;
    L.ctb_rs = ((const uint16_t*)(arena + P->off_ctb_ts_to_rs))[ts];
    // ---- WPP dependency on the CTB row above (parse_core.h: k + 1 finished CTBs, two for the row's first CTB) ----
    if (L.dep_sub >= 0) {
      uint32_t need = L.k == 0 ? 2u : L.k + 1u;
      if (need > L.dep_len) need = L.dep_len;
      // (polled on every 8th iteration only: the load's s_waitcnt stalls the whole wave for a round trip to the coherence point)
      if ((L.waits & 7u) != 0 || wt_load(A.progress + L.dep_sub) < need) {
        L.waits++;
        if (L.waits > (1u << 22) || ((L.waits & 1023u) == 1 && wt_load((const uint32_t*)A.status) != 0)) L.err = DEV_ERR_TIMEOUT;
        L.kind = K_NONE;
        goto step_done;
      }
      L.waits = 0;
    }
    const int cx = L.ctb_rs % L.ctb_w, cy = L.ctb_rs / L.ctb_w;
    const CtbInfo ci = ((const CtbInfo*)(arena + P->off_ctb_info))[L.ctb_rs];
    L.x_ctb = cx << L.log2_ctb; L.y_ctb = cy << L.log2_ctb; L.avail = ci.avail;
    L.ubase = (uint32_t)L.ctb_rs << (2 * (L.log2_ctb - 2));
    if (L.k == 0) {
      if ((L.sflags & 255u) && L.dep_sub >= 0) load_contexts(S, lane, (const uint32_t*)(A.ctx_store + (size_t)L.dep_sub * CTX_STORE));
      else init_contexts(L, S, lane);
    }
    L.up_lo = 0; L.up_hi = 0;
    if (L.avail & AV_UP) {
      const uint32_t* src = (const uint32_t*)(arena + P->off_handoff) + (size_t)(L.ctb_rs - L.ctb_w) * HANDOFF_DWORDS;
      L.up_lo = (uint64_t)wt_load(src + 9) | ((uint64_t)wt_load(src + 10) << 32);
      L.up_hi = (uint64_t)wt_load(src + 11) | ((uint64_t)wt_load(src + 12) << 32);
    }
    if (L.sao_luma || L.sao_chroma) parse_sao(L, S, lane, arena, (L.avail & AV_LEFT) && L.k > 0, (L.avail & AV_UP) ? 1 : 0);
    else for (int j = 0; j < 9; j++) S.sao[j * 64 + lane] = 0;
    {
      uint32_t* sao_dst = (uint32_t*)(arena + P->off_sao) + (size_t)L.ctb_rs * 9;
      for (int j = 0; j < 9; j++) sao_dst[j] = S.sao[j * 64 + lane];
    }
    if (!(L.tools & TOOL_CUQPD)) { L.qp_coded = 0; L.qp_delta = 0; L.qp_pred = L.last_qp; }
    L.p = 0;
  }
  // fall through
  case S_CQT: {   // next node of the coding quadtree, stackless over the z-ordered min-CB index
    const int n_mincb = 1 << (2 * (L.log2_ctb - L.log2_min_cb));
    for (;;) {
      if (L.p >= n_mincb) REQ(K_TERM, 0, S_EOS_R);   // end_of_slice_segment_flag / end_of_subset_one_bit
      int lg;
      if (L.p == 0) lg = L.log2_ctb; else { lg = L.log2_min_cb + ((pl_ffs((uint32_t)L.p) - 1) >> 1); if (lg > L.log2_ctb) lg = L.log2_ctb; }
      const int zb = L.p << (2 * (L.log2_min_cb - 2));
      const int ux = (int)compact1by1((uint32_t)zb), uy = (int)compact1by1((uint32_t)zb >> 1);
      if (L.x_ctb + (ux << 2) >= L.width || L.y_ctb + (uy << 2) >= L.height) { L.p += 1 << (2 * (lg - L.log2_min_cb)); continue; }
      L.lg = lg; L.zb = zb;
      break;
    }
  }
  // fall through
  case S_SPLIT: {
    const int ux = (int)compact1by1((uint32_t)L.zb), uy = (int)compact1by1((uint32_t)L.zb >> 1);
    const int x0 = L.x_ctb + (ux << 2), y0 = L.y_ctb + (uy << 2), size = 1 << L.lg;
    if (x0 + size <= L.width && y0 + size <= L.height && L.lg > L.log2_min_cb) {
      const int depth = L.log2_ctb - L.lg;
      int inc = 0;
      const int l = left_cb_log2(L, S, lane, ux, uy), u = up_cb_log2(L, S, lane, ux, uy);
      if (l && L.log2_ctb - l > depth) inc++;
      if (u && L.log2_ctb - u > depth) inc++;
      REQ(K_CTX, CTX_A + A_SPLIT_CU + inc, S_SPLIT_R);
    }
    r = L.lg > L.log2_min_cb ? 1u : 0u;
  }
  // fall through
  case S_SPLIT_R: {
    if ((L.tools & TOOL_CUQPD) && L.lg >= L.log2_min_qg) {
      L.qp_coded = 0; L.qp_delta = 0;
      derive_qp_pred(L, S, lane, (int)compact1by1((uint32_t)L.zb), (int)compact1by1((uint32_t)L.zb >> 1));
    }
    if (r) { L.lg--; HOP(S_SPLIT); }
    if (!(L.tools & TOOL_CUQPD)) L.qp_pred = L.last_qp;
  }
  // fall through
  // ========================================================= coding_unit ==========================================================
  case S_CU: {
    L.log2cb = L.lg;
    if (L.tools & TOOL_TQBYPASS) REQ(K_CTX, CTX_A + A_CU_TQ_BYPASS, S_TQB_R);
    r = 0;
  }
  // fall through
  case S_TQB_R: {
    L.tqb = (int32_t)r;
    if (L.log2cb == L.log2_min_cb) REQ(K_CTX, CTX_A + A_PART_MODE, S_PART_R);
    r = 1;
  }
  // fall through
  case S_PART_R: {
    L.part_nxn = r ? 0 : 1;
    if (L.part_nxn && L.log2cb == 3 && L.log2_min_tb > 2) { L.err = DEV_ERR_SYNTAX; L.part_nxn = 0; }
    set_qp_y(L);
    const int n_units = 1 << (2 * (L.log2cb - 2));
    fill_units(arena + L.o_size + L.ubase + L.zb, n_units, (uint32_t)(L.log2cb << 4));
    fill_units(arena + L.o_flags + L.ubase + L.zb, n_units, (uint32_t)(L.tqb ? UF_BYPASS : 0));
    fill_units(arena + L.o_ipm + L.ubase + L.zb, n_units, 1u);
    {
      const int cw = 1 << (L.log2cb - 2);
      line_set(S, lane, LN_LEFT_CB, (int)compact1by1((uint32_t)L.zb >> 1), cw, (uint32_t)L.log2cb);
      line_set(S, lane, LN_UP_CB, (int)compact1by1((uint32_t)L.zb), cw, (uint32_t)L.log2cb);
    }
    L.pk = 0; L.prev_flags = 0; L.ipm_pack = 0;
    REQ(K_CTX, CTX_A + A_PREV_INTRA_LUMA, S_PREV_R);
  }
  case S_PREV_R: {
    L.prev_flags |= r << L.pk;
    L.pk++;
    if (L.pk < (L.part_nxn ? 4 : 1)) REQ(K_CTX, CTX_A + A_PREV_INTRA_LUMA, S_PREV_R);
    L.pk = 0;
  }
  // fall through
  case S_IPM: {
    if ((L.prev_flags >> L.pk) & 1u) REQ(K_BYP, 1, S_MPM1_R);
    REQ(K_BYP, 5, S_REMMODE_R);
  }
  case S_MPM1_R:
    if (r) REQ(K_BYP, 1, S_MPM2_R);
    L.mpm_idx = 0;
    goto ipm_derive;
  case S_MPM2_R:
    L.mpm_idx = r ? 2 : 1;
    goto ipm_derive;
  case S_REMMODE_R:
  ipm_derive: {   // 8.4.2
    const int ux0 = (int)compact1by1((uint32_t)L.zb), uy0 = (int)compact1by1((uint32_t)L.zb >> 1);
    const int n_units = 1 << (2 * (L.log2cb - 2));
    const int n_part = L.part_nxn ? 4 : 1;
    const int pu_units = n_units / n_part;
    const int pu_w = 1 << (L.log2cb - 2 - (L.part_nxn ? 1 : 0));
    const int ux = ux0 + (L.pk & 1) * pu_w, uy = uy0 + (L.pk >> 1) * pu_w;
    int cand_a = 1, cand_b = 1;
    if (ux > 0 || (L.avail & AV_LEFT)) cand_a = (int)line_get(S, lane, LN_LEFT_IPM, uy);
    if (uy > 0) cand_b = (int)line_get(S, lane, LN_UP_IPM, ux);   // above the CTB row: INTRA_DC
    int c0, c1, c2;
    if (cand_a == cand_b) {
      if (cand_a < 2) { c0 = 0; c1 = 1; c2 = 26; }
      else { c0 = cand_a; c1 = 2 + ((cand_a + 29) & 31); c2 = 2 + ((cand_a - 2 + 1) & 31); }
    } else {
      c0 = cand_a; c1 = cand_b;
      if (cand_a != 0 && cand_b != 0) c2 = 0; else if (cand_a != 1 && cand_b != 1) c2 = 1; else c2 = 26;
    }
    int mode;
    if ((L.prev_flags >> L.pk) & 1u) mode = L.mpm_idx == 0 ? c0 : (L.mpm_idx == 1 ? c1 : c2);
    else {
      int t;
      if (c0 > c1) { t = c0; c0 = c1; c1 = t; }
      if (c0 > c2) { t = c0; c0 = c2; c2 = t; }
      if (c1 > c2) { t = c1; c1 = c2; c2 = t; }
      mode = (int)r;
      if (mode >= c0) mode++;
      if (mode >= c1) mode++;
      if (mode >= c2) mode++;
    }
    fill_units(arena + L.o_ipm + L.ubase + L.zb + L.pk * pu_units, pu_units, (uint32_t)mode);
    line_set(S, lane, LN_LEFT_IPM, uy, pu_w, (uint32_t)mode);
    line_set(S, lane, LN_UP_IPM, ux, pu_w, (uint32_t)mode);
    L.ipm_pack |= (uint32_t)mode << (8 * L.pk);
    L.pk++;
    if (L.pk < n_part) HOP(S_IPM);
    if (L.chroma) REQ(K_CTX, CTX_A + A_INTRA_CHROMA, S_CHROMA_R);
    L.chroma_mode = 1;
    goto tt_start;
  }
  case S_CHROMA_R:
    if (r) REQ(K_BYP, 2, S_CHROMA2_R);
    r = 4;
  // fall through
  case S_CHROMA2_R: {
    const int icpm = (int)r, lm = (int)(L.ipm_pack & 63u);
    if (icpm == 4) L.chroma_mode = lm;
    else { const int m = icpm == 0 ? 0 : icpm == 1 ? 26 : icpm == 2 ? 10 : 1; L.chroma_mode = (m == lm) ? 34 : m; }
  }
  tt_start: {
    fill_units(arena + L.o_ipmc + L.ubase + L.zb, 1 << (2 * (L.log2cb - 2)), (uint32_t)L.chroma_mode);
    L.q = 0; L.t = L.log2cb; L.stage = 0; L.cbf_cb_bits = 0; L.cbf_cr_bits = 0;
    goto tt;
  }
  // ======================================================= transform_tree =========================================================
  case S_TSPLIT_R:
    L.split = (int32_t)r; L.stage = 1;
    goto tt;
  case S_CBFCB_R: {
    const uint32_t bit = 1u << (L.log2cb - L.t);
    L.cbf_cb_bits = (L.cbf_cb_bits & ~bit) | (r ? bit : 0u); L.stage = 2;
    goto tt;
  }
  case S_CBFCR_R: {
    const uint32_t bit = 1u << (L.log2cb - L.t);
    L.cbf_cr_bits = (L.cbf_cr_bits & ~bit) | (r ? bit : 0u); L.stage = 3;
  }
  // fall through
  case S_TT:
  tt: {
    const int max_trafo_depth = L.max_th_depth + L.part_nxn;
    for (;;) {   // the node of size 1 << t that starts at unit q
      const int depth = L.log2cb - L.t;
      const uint32_t bit = 1u << depth, pbit = depth ? (1u << (depth - 1)) : 0u;
      if (L.stage == 0) {
        if (L.t <= L.log2_max_tb && L.t > L.log2_min_tb && depth < max_trafo_depth && !(L.part_nxn && depth == 0))
          REQ(K_CTX, CTX_A + A_SPLIT_TRANSFORM + 5 - L.t, S_TSPLIT_R);
        L.split = (L.t > L.log2_max_tb || (L.part_nxn && depth == 0)) ? 1 : 0;
        L.stage = 1;
      }
      if (L.stage == 1) {
        if (L.chroma) {
          if (L.t > 2) {
            if (depth == 0 || (L.cbf_cb_bits & pbit)) REQ(K_CTX, CTX_A + A_CBF_CHROMA + depth, S_CBFCB_R);
            L.cbf_cb_bits &= ~bit;
          } else {   // 4x4 luma: the chroma flags are the parent's (7.4.9.8)
            L.cbf_cb_bits = (L.cbf_cb_bits & ~bit) | ((L.cbf_cb_bits & pbit) ? bit : 0u);
            L.cbf_cr_bits = (L.cbf_cr_bits & ~bit) | ((L.cbf_cr_bits & pbit) ? bit : 0u);
            L.stage = 3;
          }
        } else L.stage = 3;
        if (L.stage == 1) L.stage = 2;
      }
      if (L.stage == 2) {
        if (depth == 0 || (L.cbf_cr_bits & pbit)) REQ(K_CTX, CTX_A + A_CBF_CHROMA + depth, S_CBFCR_R);
        L.cbf_cr_bits &= ~bit;
        L.stage = 3;
      }
      if (!L.split) break;
      L.t--; L.stage = 0;
    }
    REQ(K_CTX, CTX_A + A_CBF_LUMA + (L.log2cb == L.t ? 1 : 0), S_CBFL_R);
  }
  // ======================================================== transform_unit ========================================================
  case S_CBFL_R: {
    L.cbf_luma = (int32_t)r;
    const int depth = L.log2cb - L.t;
    const int any = L.cbf_luma | (int)((L.cbf_cb_bits >> depth) & 1u) | (int)((L.cbf_cr_bits >> depth) & 1u);
    if (any && (L.tools & TOOL_CUQPD) && !L.qp_coded) REQ(K_CTX, CTX_A + A_CU_QP_DELTA, S_QPD0_R);
    goto res_start;
  }
  case S_QPD0_R:   // 7.3.8.14 cu_qp_delta_abs: prefix TU(5) context coded, suffix EG0 bypass
    L.qv = 0;
    if (!r) goto qpd_done;
    L.qv = 1;
    REQ(K_CTX, CTX_A + A_CU_QP_DELTA + 1, S_QPD1_R);
  case S_QPD1_R:
    if (r) {
      L.qv++;
      if (L.qv < 5) REQ(K_CTX, CTX_A + A_CU_QP_DELTA + 1, S_QPD1_R);
      L.qk = 0;
      REQ(K_BYP, 1, S_QPD_EGP_R);
    }
    REQ(K_BYP, 1, S_QPD_SIGN_R);
  case S_QPD_EGP_R:
    if (r) {
      L.qv += 1 << L.qk; L.qk++;
      if (L.qk > 16) { L.err = DEV_ERR_SYNTAX; REQ(K_BYP, 1, S_QPD_SIGN_R); }
      REQ(K_BYP, 1, S_QPD_EGP_R);
    }
    if (L.qk > 0) REQ(K_BYP, L.qk, S_QPD_EGS_R);
    REQ(K_BYP, 1, S_QPD_SIGN_R);
  case S_QPD_EGS_R:
    L.qv += (int32_t)r;
    REQ(K_BYP, 1, S_QPD_SIGN_R);
  case S_QPD_SIGN_R:
    if (r) L.qv = -L.qv;
  qpd_done: {
    L.qp_coded = 1;
    L.qp_delta = L.qv;
    const int off = 6 * (L.bd_luma - 8);
    if (L.qp_delta < -(26 + off / 2) || L.qp_delta > 25 + off / 2) L.err = DEV_ERR_SYNTAX;
    set_qp_y(L);
  }
  res_start: {
    L.c = 0; L.ts_bits = 0;
    L.do_chroma = 0; L.zc = L.zb + L.q; L.tc = L.t - 1;
    if (L.chroma) {
      if (L.t > 2) L.do_chroma = 1;
      else if ((L.q & 3) == 3) { L.do_chroma = 1; L.zc = L.zb + (L.q & ~3); L.tc = 2; }
    }
    {   // intra mode of the PU this TU lies in
      const int n_units = 1 << (2 * (L.log2cb - 2));
      const int part = L.part_nxn ? L.q / (n_units >> 2) : 0;
      L.luma_mode = (int)((L.ipm_pack >> (8 * part)) & 63u);
    }
    goto res;
  }
  // ======================================================= residual_coding ========================================================
  case S_TS_R:
    L.ts = (int32_t)r;
    L.px = 0;
    REQ(K_CTX, CTX_A + A_LAST_X + (L.c == 0 ? 3 * (L.lgn - 2) + ((L.lgn - 1) >> 2) : 15), S_LASTX_R);
  case S_LASTX_R: {
    const int ctx_offset = L.c == 0 ? 3 * (L.lgn - 2) + ((L.lgn - 1) >> 2) : 15, ctx_shift = L.c == 0 ? (L.lgn + 1) >> 2 : L.lgn - 2;
    const int c_max = (L.lgn << 1) - 1;
    if (r) { L.px++; if (L.px < c_max) REQ(K_CTX, CTX_A + A_LAST_X + ctx_offset + (L.px >> ctx_shift), S_LASTX_R); }
    L.py = 0;
    REQ(K_CTX, CTX_A + A_LAST_Y + ctx_offset, S_LASTY_R);
  }
  case S_LASTY_R: {
    const int ctx_offset = L.c == 0 ? 3 * (L.lgn - 2) + ((L.lgn - 1) >> 2) : 15, ctx_shift = L.c == 0 ? (L.lgn + 1) >> 2 : L.lgn - 2;
    const int c_max = (L.lgn << 1) - 1;
    if (r) { L.py++; if (L.py < c_max) REQ(K_CTX, CTX_A + A_LAST_Y + ctx_offset + (L.py >> ctx_shift), S_LASTY_R); }
    L.last_x = L.px; L.last_y = L.py;
    if (L.px > 3) REQ(K_BYP, (L.px >> 1) - 1, S_LASTXS_R);
    goto last_y_suffix;
  }
  case S_LASTXS_R:
    L.last_x = (1 << ((L.px >> 1) - 1)) * (2 + (L.px & 1)) + (int32_t)r;
  last_y_suffix:
    if (L.py > 3) REQ(K_BYP, (L.py >> 1) - 1, S_LASTYS_R);
    goto res_setup;
  case S_LASTYS_R:
    L.last_y = (1 << ((L.py >> 1) - 1)) * (2 + (L.py & 1)) + (int32_t)r;
  res_setup: {
    const int pred_mode = L.c == 0 ? +L.luma_mode : +L.chroma_mode;
    L.scan_idx = 0;
    if (L.lgn == 2 || (L.lgn == 3 && L.c == 0)) {
      if (pred_mode >= 6 && pred_mode <= 14) L.scan_idx = 2;
      else if (pred_mode >= 22 && pred_mode <= 30) L.scan_idx = 1;
    }
    if (L.scan_idx == 2) { const int t = L.last_x; L.last_x = L.last_y; L.last_y = t; }
    const int n = 1 << L.lgn;
    if (L.last_x >= n || L.last_y >= n) { L.err = DEV_ERR_SYNTAX; L.last_x = 0; L.last_y = 0; }
    const int lg = L.lgn - 2;
    const int xs_t = L.last_x >> 2, ys_t = L.last_y >> 2;
    const uint32_t r_t = (uint32_t)((L.last_x & 3) | ((L.last_y & 3) << 2));
    const uint64_t inv4 = L.scan_idx == 0 ? PC_INV_DIAG4 : (L.scan_idx == 1 ? PC_INV_HORZ4 : PC_INV_VERT4);
    L.last_pos = (int)((uint32_t)(inv4 >> (r_t * 4u)) & 15u);
    if (lg == 0) L.last_sb = 0;
    else if (lg == 1) L.last_sb = L.scan_idx == 1 ? (xs_t | (ys_t << 1)) : ((xs_t << 1) | ys_t);
    else if (lg == 2) L.last_sb = (int)((uint32_t)(PC_INV_DIAG4 >> ((uint32_t)(xs_t | (ys_t << 2)) * 4u)) & 15u);
    else L.last_sb = (int)S.inv8[xs_t | (ys_t << 3)];
    L.csbf = 0; L.g1_carry = 1; L.first_g1_sb = 1;
    L.i = L.last_sb;
    goto sb_enter;
  }
  case S_CSBF_R: {
    L.infer_dc = 1;
    if (!r) goto sb_dec;
    L.csbf |= 1ull << (L.ys * 8 + L.xs);
    const int sbw = 1 << (L.lgn - 2);
    const int right = (L.xs < sbw - 1) ? (int)((L.csbf >> (L.ys * 8 + L.xs + 1)) & 1) : 0;
    const int below = (L.ys < sbw - 1) ? (int)((L.csbf >> ((L.ys + 1) * 8 + L.xs)) & 1) : 0;
    begin_sig(L, right, below);
    goto step_done;
  }
  case S_SIG_R: {
    L.sig |= r << L.kpos;
    if (L.kpos > 0) {
      L.kpos--;
      if (L.kpos == 0 && L.infer_dc && !L.sig) L.sig = 1u;   // position 0 inferred significant (the sub-block was signalled coded)
      else REQ(K_CTX, sig_ctx(L, L.kpos), S_SIG_R);
    }
    if (!L.sig) goto sb_dec;
    start_g1(L);
    goto step_done;
  }
  case S_G1_R: {
    const int k = 31 - pl_clz(L.rem_mask);
    L.rem_mask &= ~(1u << k);
    L.g1_coded |= 1u << k;
    if (r) { L.g1 |= 1u << k; L.g1_ctx = 0; if (L.last_g1_pos < 0) L.last_g1_pos = k; }
    else if (L.g1_ctx > 0) L.g1_ctx++;
    L.num_g1++;
    if (L.rem_mask && L.num_g1 < 8) REQ(K_CTX, CTX_C + C_GREATER1 + L.ctx_set * 4 + (L.c ? 16 : 0) + (L.g1_ctx > 3 ? 3 : L.g1_ctx), S_G1_R);
    L.g1_carry = L.g1_ctx;
    if (L.last_g1_pos >= 0) REQ(K_CTX, CTX_B + B_GREATER2 + L.ctx_set + (L.c ? 4 : 0), S_G2_R);
    r = 0;
  }
  // fall through
  case S_G2_R: {
    L.g2 = (r && L.last_g1_pos >= 0) ? 1u << L.last_g1_pos : 0u;
    const int last_sig_pos = 31 - pl_clz(L.sig);
    L.first_sig_pos = pl_ffs(L.sig) - 1;
    L.sign_hidden = L.tqb ? 0 : (((L.tools & TOOL_SDH) && (last_sig_pos - L.first_sig_pos > 3)) ? 1 : 0);
    L.sig_signed = L.sign_hidden ? L.sig & ~(1u << L.first_sig_pos) : L.sig;
    L.n_signs = pl_popc(L.sig_signed);
    REQ(K_BYP, L.n_signs, S_SIGN_R);   // coeff_sign_flag of the sub-block in one group
  }
  case S_SIGN_R: {
    L.sign_bits = r;
    const uint32_t first_g1_bit = L.last_g1_pos >= 0 ? 1u << L.last_g1_pos : 0u;
    L.need_rem = (L.g1_coded & L.g1 & ~(first_g1_bit & ~L.g2)) | (L.sig & ~L.g1_coded);
    L.emit = L.sig; L.rice = 0; L.parity = 0;
    goto emit;
  }
  case S_REM_R: {
    const int k = 31 - pl_clz(L.emit);
    const uint32_t a = 1u + ((L.g1 >> k) & 1u) + ((L.g2 >> k) & 1u) + r;
    if (a > 3u * (1u << L.rice)) L.rice = L.rice < 4 ? L.rice + 1 : 4;
    store_level(L, k, a);
    L.emit &= ~(1u << k);
  }
  emit:
    while (L.emit) {   // levels in decreasing scan position; the ones whose base level hit its cap need coeff_abs_level_remaining
      const int k = 31 - pl_clz(L.emit);
      if ((L.need_rem >> k) & 1u) REQ(K_REM, L.rice, S_REM_R);
      store_level(L, k, 1u + ((L.g1 >> k) & 1u) + ((L.g2 >> k) & 1u));
      L.emit &= ~(1u << k);
    }
  sb_dec:
    L.i--;
  // fall through
  case S_SB:
  sb_enter: {
    if (L.i < 0) goto block_done;
    const int lg = L.lgn - 2, i = L.i;
    if (lg == 0) { L.xs = 0; L.ys = 0; }
    else if (lg == 1) { if (L.scan_idx == 1) { L.xs = i & 1; L.ys = i >> 1; } else { L.xs = i >> 1; L.ys = i & 1; } }
    else if (lg == 2) { const uint32_t v = (uint32_t)(PC_DIAG4 >> (i * 4)) & 15u; L.xs = (int)(v & 3u); L.ys = (int)(v >> 2); }
    else { const uint32_t v = S.diag8[i]; L.xs = (int)(v & 7u); L.ys = (int)(v >> 3); }
    const int sbw = 1 << lg;
    const int right = (L.xs < sbw - 1) ? (int)((L.csbf >> (L.ys * 8 + L.xs + 1)) & 1) : 0;
    const int below = (L.ys < sbw - 1) ? (int)((L.csbf >> ((L.ys + 1) * 8 + L.xs)) & 1) : 0;
    if (i < L.last_sb && i > 0) REQ(K_CTX, CTX_A + A_CODED_SUB_BLOCK + ((right | below) ? 1 : 0) + (L.c ? 2 : 0), S_CSBF_R);
    L.infer_dc = 0;
    L.csbf |= 1ull << (L.ys * 8 + L.xs);
    begin_sig(L, right, below);
    goto step_done;
  }
  block_done:
    L.ts_bits |= (uint32_t)L.ts << L.c;
    L.c++;
  // fall through
  case S_RES:
  res: {
    const int depth = L.log2cb - L.t;
    for (; L.c < 3; L.c++) {
      const int coded = L.c == 0 ? +L.cbf_luma : (L.do_chroma && (int)(((L.c == 1 ? +L.cbf_cb_bits : +L.cbf_cr_bits) >> depth) & 1u));
      if (coded) break;
    }
    if (L.c < 3) {
      L.lgn = L.c == 0 ? +L.t : +L.tc;
      const int ctb2_log2 = 2 * L.log2_ctb;
      if (L.c == 0) L.dst = (int16_t*)(arena + L.o_coef0) + ((size_t)L.ctb_rs << ctb2_log2) + (size_t)(L.zb + L.q) * 16;
      else L.dst = (int16_t*)(arena + (L.c == 1 ? +L.o_coef1 : +L.o_coef2)) + ((size_t)L.ctb_rs << (ctb2_log2 - 2)) + (size_t)L.zc * 4;
      {
        const Q4 z{{0, 0, 0, 0}};
        const int n2 = 1 << (2 * L.lgn);
        for (int j = 0; j < n2; j += 8) *(Q4*)(L.dst + j) = z;
      }
      L.ts = 0; L.px = 0;
      if ((L.tools & TOOL_TS) && !L.tqb && L.lgn <= 2) REQ(K_CTX, CTX_A + A_TRANSFORM_SKIP + (L.c ? 1 : 0), S_TS_R);
      REQ(K_CTX, CTX_A + A_LAST_X + (L.c == 0 ? 3 * (L.lgn - 2) + ((L.lgn - 1) >> 2) : 15), S_LASTX_R);
    }
    // ---- the transform unit is complete: size, cbf, transform-skip, deblocking edges (8.7.2.2 / 8.7.2.3) of its units ----
    const int zu = L.zb + L.q;
    const int tu_units = 1 << (2 * (L.t - 2));
    const int cbf_cb = (int)((L.cbf_cb_bits >> depth) & 1u), cbf_cr = (int)((L.cbf_cr_bits >> depth) & 1u);
    const int tux0 = (int)compact1by1((uint32_t)zu), tuy0 = (int)compact1by1((uint32_t)zu >> 1);
    const int edge_l = L.deblock && (tux0 > 0 || (L.avail & AV_EDGE_LEFT));
    const int edge_t = L.deblock && (tuy0 > 0 || (L.avail & AV_EDGE_UP));
    const uint32_t fl = (uint32_t)((L.cbf_luma ? UF_CBF_LUMA : 0) | ((L.do_chroma && cbf_cb) ? UF_CBF_CB : 0) | ((L.do_chroma && cbf_cr) ? UF_CBF_CR : 0) |
                                   (L.tqb ? UF_BYPASS : 0) | ((L.ts_bits & 1u) ? UF_TS_LUMA : 0));
    const uint32_t ipm = (uint32_t)L.luma_mode | ((L.ts_bits & 2u) ? 64u : 0u) | ((L.ts_bits & 4u) ? 128u : 0u);
    const uint32_t szb = (uint32_t)((L.log2cb << 4) | L.t);
    const uint32_t ve = edge_l ? UF_VEDGE : 0u, he = edge_t ? UF_HEDGE : 0u;
    fill_units(arena + L.o_size + L.ubase + zu, tu_units, szb);
    fill_units(arena + L.o_ipm + L.ubase + zu, tu_units, ipm);
    uint8_t* pf = arena + L.o_flags + L.ubase + zu;
    if (tu_units == 1) *pf = (uint8_t)(fl | ve | he);
    else {
#pragma clang loop unroll(disable)
      for (int j = 0; j < tu_units; j += 4) {   // units j .. j+3 in z-order: (x0, y0), (x0+1, y0), (x0, y0+1), (x0+1, y0+1)
        uint32_t wf = fl * 0x01010101u;
        if (((uint32_t)j & 0x55555555u) == 0) wf |= ve * 0x00010001u;   // x == 0: the left column of the quad
        if (((uint32_t)j & 0xAAAAAAAAu) == 0) wf |= he * 0x00000101u;   // y == 0: its top row
        *(uint32_t*)(pf + j) = wf;
      }
    }
    L.q += tu_units;
    const int n_units = 1 << (2 * (L.log2cb - 2));
    if (L.q < n_units) {
      int t = 2 + ((pl_ffs((uint32_t)L.q) - 1) >> 1);
      if (t > L.log2cb) t = L.log2cb;
      L.t = t; L.stage = 0;
      HOP(S_TT);
    }
    // ---- the coding unit is complete ----
    set_qp_y(L);
    fill_units(arena + L.o_qp + L.ubase + L.zb, n_units, (uint32_t)(uint8_t)(int8_t)L.cur_qp);
    L.last_qp = L.cur_qp;
    {
      const int cw = 1 << (L.log2cb - 2);
      line_set(S, lane, LN_LEFT_QP, (int)compact1by1((uint32_t)L.zb >> 1), cw, (uint32_t)(uint8_t)(int8_t)L.cur_qp);
      line_set(S, lane, LN_UP_QP, (int)compact1by1((uint32_t)L.zb), cw, (uint32_t)(uint8_t)(int8_t)L.cur_qp);
    }
    L.p += 1 << (2 * (L.log2cb - L.log2_min_cb));
    HOP(S_CQT);
  }
  // ======================================================= end of the CTB ==========================================================
  case S_EOS_R: {
    const int last = (L.k + 1 == L.num_ctbs);
    if (last) {
      if ((L.sflags >> 16) & 255u) { if (!r) L.err = DEV_ERR_TERMINATE; }      // last CTB of the slice segment: end_of_slice_segment_flag = 1
      else { if (r) L.err = DEV_ERR_TERMINATE; else REQ(K_TERM, 0, S_EOS2_R); }   // end of a substream inside it: end_of_subset_one_bit
    } else if (r) L.err = DEV_ERR_TERMINATE;
    goto publish;
  }
  case S_EOS2_R:
    if (!r) L.err = DEV_ERR_TERMINATE;
  publish: {
    // hand-off record for the CTB below (SAO parameters, CB sizes of the bottom unit row), WPP context snapshot, progress
    const PicParams* P = L.P;
    uint32_t* dst = (uint32_t*)(arena + P->off_handoff) + (size_t)L.ctb_rs * HANDOFF_DWORDS;
    for (int j = 0; j < 9; j++) wt_store(dst + j, S.sao[j * 64 + lane]);
    const int uw = 1 << (L.log2_ctb - 2);
    for (int j = 0; j < (uw + 3) / 4; j++) {   // the up line buffer now holds the CB sizes of the CTB's bottom unit row
      uint32_t w = 0;
      for (int b = 0; b < 4 && 4 * j + b < uw; b++) w |= (line_get(S, lane, LN_UP_CB, 4 * j + b) << 4) << (8 * b);
      wt_store(dst + 9 + j, w);
    }
    const int has_dependent = (int)((L.sflags >> 8) & 255u);
    if (has_dependent && L.k == 1) save_contexts(S, lane, (uint32_t*)(A.ctx_store + (size_t)L.sub * CTX_STORE));
    PL_DRAIN();
    if (has_dependent) wt_store(A.progress + L.sub, L.k + 1u);
    L.k++;
    HOP(S_CTB);
  }
  default: break;
  }
step_done:
  return;
}

#if defined(HIPDEC_HOST_EMU) && defined(HIPDEC_LANES_STATS)
void hipdec_lanes_stats(int lane, int state, int kind, int waiting);   // tests/emu/lanes_stats.cc
#endif

PL_DEV void lane_start(LS& L, Shared& S, int lane, const ParseArgs& A, uint32_t sub)
{
  const Substream* sp = A.subs + sub;
  L.sub = sub;
  L.num_ctbs = sp->num_ctbs; L.first_ctb_ts = sp->first_ctb_ts; L.dep_sub = sp->dep_sub; L.dep_len = sp->dep_len;
  L.sflags = (uint32_t)sp->wpp_sync | ((uint32_t)sp->has_dependent << 8) | ((uint32_t)sp->last_in_slice_segment << 16);
  const PicParams* P = A.pics + sp->pic;
  L.P = P;
  L.width = P->width; L.height = P->height; L.ctb_w = P->ctb_w;
  L.log2_ctb = P->log2_ctb; L.log2_min_cb = P->log2_min_cb; L.log2_min_tb = P->log2_min_tb; L.log2_max_tb = P->log2_max_tb;
  L.max_th_depth = P->max_th_depth_intra; L.chroma = P->chroma_format_idc; L.bd_luma = P->bit_depth_luma; L.bd_chroma = P->bit_depth_chroma;
  L.log2_min_qg = P->log2_min_cu_qp_delta_size;
  L.tools = (P->sign_data_hiding ? TOOL_SDH : 0u) | (P->transform_skip_enabled ? TOOL_TS : 0u) | (P->cu_qp_delta_enabled ? TOOL_CUQPD : 0u) |
            (P->transquant_bypass_enabled ? TOOL_TQBYPASS : 0u);
  const SliceParams* sl = (const SliceParams*)(A.arena + P->off_slices) + sp->slice_idx;
  L.slice_qp = sl->slice_qp_y; L.deblock = sl->deblocking_disabled ? 0 : 1; L.sao_luma = sl->sao_luma; L.sao_chroma = sl->sao_chroma;
  L.o_size = P->off_u_size; L.o_flags = P->off_u_flags; L.o_ipm = P->off_u_ipm; L.o_ipmc = P->off_u_ipmc; L.o_qp = P->off_u_qp;
  L.o_coef0 = P->off_coeff[0]; L.o_coef1 = P->off_coeff[1]; L.o_coef2 = P->off_coeff[2];
  L.bs = A.arena + P->off_bitstream;
  L.qp_coded = 0; L.qp_delta = 0; L.qp_pred = L.slice_qp; L.last_qp = L.slice_qp; L.cur_qp = L.slice_qp;
  cabac_start(L, S, lane, sp->byte_start, sp->byte_end);
  L.state = S_CTB;
}

// the body of the kernel (also what tests/emu runs under the SIMT shim)
PL_DEV void parse_lanes_wave(const ParseArgs& A, uint32_t wave_idx, Shared& S)
{
  const int lane = (int)threadIdx.x;
  S.t_lps[lane] = (uint32_t)c_range_lps[lane * 4] | ((uint32_t)c_range_lps[lane * 4 + 1] << 8) | ((uint32_t)c_range_lps[lane * 4 + 2] << 16) |
                  ((uint32_t)c_range_lps[lane * 4 + 3] << 24);
  S.t_next[lane] = (uint8_t)(c_next_lps[lane] | (lane == 0 ? 64 : 0));
  S.diag8[lane] = c_diag8[lane];
  S.inv8[c_diag8[lane]] = (uint8_t)lane;
  __syncthreads();
  LS L{};
  L.kind = K_NONE; L.state = S_DONE;
  const uint32_t sub = A.lane_subs[(size_t)wave_idx * 64u + (uint32_t)lane];
  if (sub != 0xffffffffu) lane_start(L, S, lane, A, sub);
  for (;;) {
    if (__ballot(L.state != S_DONE && L.fill - L.pos < RING_LOW) != 0) { if (L.state != S_DONE) ring_fill(L, S, lane); }   // top every ring up together
    if (L.state != S_DONE) reservoir_fill(L, S, lane);
    uint32_t r = 0;
    if (L.kind == K_CTX) r = dec_ctx(L, S, lane, L.arg);
    else if (L.kind == K_BYP) r = dec_byp(L, S, lane, L.arg);
    else if (L.kind == K_REM) r = dec_rem(L, S, lane, L.arg);
    else if (L.kind == K_TERM) r = dec_term(L, S, lane);
    if (L.state != S_DONE) {
      step(L, S, lane, A, r);
      if (L.err) {
        atomicCAS((int*)A.status, 0, L.err | (int32_t)(L.sub << 8));
        L.kind = K_NONE; L.state = S_DONE;
      }
    }
#if defined(HIPDEC_HOST_EMU) && defined(HIPDEC_LANES_STATS)
    {   // CPU-test instrumentation: iterations of the wave's loop, populated states and busy lanes per iteration
      hipdec_lanes_stats(lane, L.state, L.kind, L.state == S_CTB && L.waits);
    }
#endif
    if (__ballot(L.state != S_DONE) == 0) break;
    if (__ballot(L.state != S_DONE && !(L.state == S_CTB && L.waits)) == 0) __builtin_amdgcn_s_sleep(32);   // every live lane waits for a row above
  }
}

}  // namespace planes

__global__ __launch_bounds__(64) void k_parse_lanes(ParseArgs A)
{
  __shared__ planes::Shared S;
  const int lane = (int)threadIdx.x;
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(A.ticket, 1u);
  const uint32_t wave_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  if (wave_idx >= A.num_lane_waves) return;
  planes::parse_lanes_wave(A, wave_idx, S);
}

void launch_parse_lanes(const ParseArgs& a, hipStream_t s)
{
  if (!a.num_lane_waves) return;
  hipLaunchKernelGGL(k_parse_lanes, dim3(a.num_lane_waves), dim3(64), 0, s, a);
}

}  // namespace hipdec
