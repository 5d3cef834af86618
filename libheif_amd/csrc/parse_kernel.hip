// parse_kernel.hip — CABAC entropy decoding + syntax parsing, one wavefront per independent substream.
//
// Stands in for libde265's slice-data parser behind de265_decode()
// (reference call site libheif/plugins/decoder_libde265.cc:402).  Syntax and context selection per
// ITU-T H.265 7.3.8 / 9.3 (intra slices).
//
// MI355X mapping
//   * one 64-lane wavefront (= one workgroup) per CABAC substream: a slice segment, a tile, or one
//     CTB row under WPP.  The bin-decoding loop is inherently serial, so every lane runs the same
//     uniform instruction stream (no divergence, no exec-mask traffic); the 64 lanes are used for the
//     data-parallel side jobs: 256-byte coalesced bitstream fetches into an LDS window, context-table
//     initialisation / WPP save-restore, zeroing and flushing the LDS-staged coefficient block with
//     16-byte-per-lane stores, and writing the CTB's unit maps.
//   * context variables, range-LPS table, the bitstream window, the coefficient block and the CTB's
//     unit maps all live in LDS (~4.6 KB per wave, so the 32-waves/CU limit binds, not LDS).
//   * WPP rows wait on the row above through an agent-scope release/acquire progress word
//     (cdna guide, Guideline 16); work is handed out by an atomic ticket so a waiting wave's
//     predecessor is always resident.  Every spin is bounded.
//   * throughput comes from the number of live substreams, not from HBM: the kernel moves
//     ~(bitstream + 3 B/px coefficients + 0.3 B/px maps); its roofline is instruction issue.
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "kernels.h"

namespace hipdec {


namespace {

// ---- tables (copied to LDS at kernel start) ---------------------------------------------------
__constant__ uint8_t c_range_lps[64 * 4] = {
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
__constant__ uint8_t c_next_lps[64] = {
   0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9,11,11,12, 13,13,15,15,16,16,18,18,19,19,21,21,22,22,23,24,
  24,25,26,26,27,27,28,29,29,30,30,30,31,32,32,33, 33,33,34,34,35,35,35,36,36,36,37,37,37,38,38,63};
__constant__ uint8_t c_init_I[CTX_COUNT] = {
  153, 200, 139, 141, 157, 154, 184, 184, 63, 153, 138, 138, 111, 141, 94, 138, 182, 154, 154, 154, 139, 139,
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
  91, 171, 134, 141,
  111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141,
  179, 153, 125, 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111,
  140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
  138, 153, 136, 167, 152, 152};
// up-right diagonal scan of an 8x8 array of sub-blocks (6.5.3), entry = x | y << 3
__constant__ uint8_t c_diag8[64] = {
  0, 8, 1, 16, 9, 2, 24, 17, 10, 3, 32, 25, 18, 11, 4, 40, 33, 26, 19, 12, 5, 48, 41, 34, 27, 20, 13, 6, 56, 49, 42, 35, 28, 21, 14, 7, 57, 50, 43, 36, 29, 22, 15, 58, 51, 44, 37, 30, 23, 59, 52, 45, 38, 31, 60, 53, 46, 39, 61, 54, 47, 62, 55, 63};

// 4x4 scans packed into immediates: nibble n = x | y << 2
#define DIAG4 0xFBE7AD369C258140ULL  /* (0,0)(0,1)(1,0)(0,2)(1,1)(2,0)(0,3)(1,2)(2,1)(3,0)(1,3)(2,2)(3,1)(2,3)(3,2)(3,3) */
// sig_coeff ctxIdxMap for 4x4 blocks (9.3.4.2.5), nibble i = ctxIdxMap[i]
#define CTXIDXMAP4 0x8877886654325410ULL

struct alignas(16) WaveLds {
  uint32_t win[64];       // bitstream window (256 B)
  int16_t coef[32 * 32];  // coefficient block being parsed
  uint8_t ctx[CTX_STORE];
  uint8_t range_lps[256];
  uint8_t next_lps[64];
  uint8_t diag8[64];
  uint8_t m_size[256], m_flags[256], m_ipm[256], m_ipmc[256];
  int8_t m_qp[256];
  uint8_t left_size[16], left_ipm[16], up_size[16];
  SaoParams sao_cur[3], sao_left[3];
};

struct Cabac {
  uint32_t range, value;
  int32_t bits_needed;
  uint32_t pos, end, win_base;
  int32_t zeros;
};

struct State {
  // immutable per substream
  const PicParams* pp;
  uint8_t* arena;
  const uint8_t* bs;     // picture bitstream
  WaveLds* L;
  int lane;
  // slice
  SliceParams sl;
  // CTB
  int ctb_rs, x_ctb, y_ctb, ctb_avail, units_log2;
  // QP
  int is_cu_qp_delta_coded, cu_qp_delta_val, qpy_pred, last_qp_y, cur_qp_y;
  int cu_tq_bypass;
  int err;
  Cabac c;
};

__device__ __forceinline__ uint32_t interleave4(uint32_t x, uint32_t y)  // z-index of unit (x,y), x,y < 16
{
  x = (x | (x << 2)) & 0x33; x = (x | (x << 1)) & 0x55;
  y = (y | (y << 2)) & 0x33; y = (y | (y << 1)) & 0x55;
  return x | (y << 1);
}
__device__ __forceinline__ uint32_t compact1by1(uint32_t v)
{
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu;
  return v;
}

// ---- bitstream + CABAC engine (9.3.4.3, scaled-window formulation) -----------------------------
__device__ __forceinline__ uint32_t fetch_byte(State& s, uint32_t pos)
{
  Cabac& c = s.c;
  if ((pos & ~255u) != c.win_base) {  // wave-uniform
    c.win_base = pos & ~255u;
    __syncthreads();
    s.L->win[s.lane] = *(const uint32_t*)(s.bs + c.win_base + 4u * s.lane);
    __syncthreads();
  }
  return ((const uint8_t*)s.L->win)[pos & 255u];
}
__device__ __forceinline__ uint32_t read_byte(State& s)
{
  Cabac& c = s.c;
  if (c.pos >= c.end) { c.pos++; if (c.pos > c.end + 8) s.err = DEV_ERR_BITSTREAM_END; return 0; }
  uint32_t b = fetch_byte(s, c.pos++);
  if (c.zeros >= 2 && b == 3 && c.pos < c.end) {  // emulation_prevention_three_byte
    b = fetch_byte(s, c.pos++);
    c.zeros = 0;
  }
  c.zeros = b == 0 ? c.zeros + 1 : 0;
  return b;
}
__device__ __forceinline__ void cabac_start(State& s, uint32_t start, uint32_t end)
{
  Cabac& c = s.c;
  c.pos = start; c.end = end; c.zeros = 0; c.win_base = 0xffffffffu;
  c.range = 510; c.bits_needed = -8;
  uint32_t b0 = read_byte(s), b1 = read_byte(s);
  c.value = (b0 << 8) | b1;
}
__device__ __forceinline__ int decode_bin(State& s, int ctx_idx)
{
  Cabac& c = s.c;
  uint32_t st = s.L->ctx[ctx_idx];
  uint32_t p_state = st >> 1, mps = st & 1;
  uint32_t lps = s.L->range_lps[(p_state << 2) | ((c.range >> 6) & 3)];
  c.range -= lps;
  uint32_t scaled = c.range << 7;
  int bin;
  if (c.value < scaled) {
    bin = (int)mps;
    if (p_state < 62) p_state++;
    s.L->ctx[ctx_idx] = (uint8_t)((p_state << 1) | mps);
    if (scaled < (256u << 7)) {
      c.range = scaled >> 6;
      c.value <<= 1;
      if (++c.bits_needed == 0) { c.bits_needed = -8; c.value += read_byte(s); }
    }
  } else {
    bin = (int)(mps ^ 1);
    int num_bits = __clz((int)lps) - 23;
    c.value = (c.value - scaled) << num_bits;
    c.range = lps << num_bits;
    if (p_state == 0) mps ^= 1;
    p_state = s.L->next_lps[p_state];
    s.L->ctx[ctx_idx] = (uint8_t)((p_state << 1) | mps);
    c.bits_needed += num_bits;
    if (c.bits_needed >= 0) { c.value += read_byte(s) << c.bits_needed; c.bits_needed -= 8; }
  }
  return bin;
}
__device__ __forceinline__ int decode_bypass(State& s)
{
  Cabac& c = s.c;
  c.value <<= 1;
  if (++c.bits_needed >= 0) { c.bits_needed = -8; c.value += read_byte(s); }
  uint32_t scaled = c.range << 7;
  if (c.value >= scaled) { c.value -= scaled; return 1; }
  return 0;
}
__device__ __forceinline__ int decode_bypass_bits(State& s, int n)
{
  int v = 0;
  for (int i = 0; i < n; i++) v = (v << 1) | decode_bypass(s);
  return v;
}
__device__ __forceinline__ int decode_terminate(State& s)
{
  Cabac& c = s.c;
  c.range -= 2;
  uint32_t scaled = c.range << 7;
  if (c.value >= scaled) return 1;
  if (scaled < (256u << 7)) {
    c.range = scaled >> 6;
    c.value <<= 1;
    if (++c.bits_needed == 0) { c.bits_needed = -8; c.value += read_byte(s); }
  }
  return 0;
}

// ---- context initialisation 9.3.2.2 (lane-parallel) -------------------------------------------
__device__ __forceinline__ void init_contexts(State& s)
{
  int qp = s.sl.slice_qp_y < 0 ? 0 : (s.sl.slice_qp_y > 51 ? 51 : s.sl.slice_qp_y);
  __syncthreads();
  for (int i = s.lane; i < CTX_COUNT; i += 64) {
    int init = c_init_I[i];
    int m = (init >> 4) * 5 - 45, n = ((init & 15) << 3) - 16;
    int pre = ((m * qp) >> 4) + n;
    pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
    int mps = pre <= 63 ? 0 : 1;
    int p_state = mps ? pre - 64 : 63 - pre;
    s.L->ctx[i] = (uint8_t)((p_state << 1) | mps);
  }
  __syncthreads();
}

// ---- neighbour helpers over the CTB-local z-ordered maps ---------------------------------------
// log2 CB size of the unit left of / above unit (ux, uy) of the current CTB, or 0 if unavailable
__device__ __forceinline__ int left_cb_log2(State& s, int ux, int uy)
{
  if (ux > 0) return s.L->m_size[interleave4(ux - 1, uy)] >> 4;
  if (s.ctb_avail & AV_LEFT) return s.L->left_size[uy] >> 4;
  return 0;
}
__device__ __forceinline__ int up_cb_log2(State& s, int ux, int uy)
{
  if (uy > 0) return s.L->m_size[interleave4(ux, uy - 1)] >> 4;
  if (s.ctb_avail & AV_UP) return s.L->up_size[ux] >> 4;
  return 0;
}

// 8.6.1 (qPY_A / qPY_B only count inside the current CTB)
__device__ __forceinline__ void derive_qp_pred(State& s, int ux, int uy)
{
  int prev = s.last_qp_y;
  int a = prev, b = prev;
  if (ux > 0) a = s.L->m_qp[interleave4(ux - 1, uy)];
  if (uy > 0) b = s.L->m_qp[interleave4(ux, uy - 1)];
  s.qpy_pred = (a + b + 1) >> 1;
}
__device__ __forceinline__ void set_qp_y(State& s)
{
  int off = 6 * (s.pp->bit_depth_luma - 8);
  s.cur_qp_y = ((s.qpy_pred + s.cu_qp_delta_val + 52 + 2 * off) % (52 + off)) - off;
}

// ---- coefficient block staging -----------------------------------------------------------------
__device__ __forceinline__ void flush_coef(State& s, int16_t* dst, int n2)
{
  __syncthreads();
  if (n2 >= 64) {
    for (int i = s.lane * 8; i < n2; i += 512) {
      uint4 v = *(const uint4*)&s.L->coef[i];
      *(uint4*)&dst[i] = v;
      *(uint4*)&s.L->coef[i] = make_uint4(0, 0, 0, 0);
    }
  } else {  // 4x4: 32 bytes
    if (s.lane < 4) {
      uint2 v = *(const uint2*)&s.L->coef[s.lane * 4];
      *(uint2*)&dst[s.lane * 4] = v;
      *(uint2*)&s.L->coef[s.lane * 4] = make_uint2(0, 0);
    }
  }
  __syncthreads();
}

// ---- 7.3.8.11 residual_coding ----------------------------------------------------------------
__device__ __forceinline__ int decode_remaining(State& s, int rice)
{
  int prefix = 0;
  while (prefix < 32 && decode_bypass(s)) prefix++;
  if (prefix >= 32) { s.err = DEV_ERR_SYNTAX; return 0; }
  if (prefix <= 3) return (prefix << rice) + decode_bypass_bits(s, rice);
  return (((1 << (prefix - 3)) + 3 - 1) << rice) + decode_bypass_bits(s, prefix - 3 + rice);
}

__device__ __forceinline__ void scan_pos4(int scan_idx, int n, int& x, int& y)
{
  if (scan_idx == 0) { uint32_t v = (uint32_t)(DIAG4 >> (n * 4)) & 15; x = v & 3; y = v >> 2; }
  else if (scan_idx == 1) { x = n & 3; y = n >> 2; }
  else { x = n >> 2; y = n & 3; }
}
// scan of sub-blocks: log2 of the sub-block grid width (0..3)
__device__ __forceinline__ void scan_sb(State& s, int lg, int scan_idx, int i, int& xs, int& ys)
{
  if (lg == 0) { xs = 0; ys = 0; }
  else if (lg == 1) {
    if (scan_idx == 1) { xs = i & 1; ys = i >> 1; }  // horizontal
    else { xs = i >> 1; ys = i & 1; }                // diagonal and vertical coincide for 2x2
  } else if (lg == 2) { uint32_t v = (uint32_t)(DIAG4 >> (i * 4)) & 15; xs = v & 3; ys = v >> 2; }
  else { uint32_t v = s.L->diag8[i]; xs = v & 7; ys = v >> 3; }
}

// returns transform_skip_flag; coefficients go to L->coef (raster, n x n)
__device__ __forceinline__ int residual_coding(State& s, int log2n, int c_idx, int pred_mode)
{
  const PicParams& P = *s.pp;
  const int n = 1 << log2n;
  int ts = 0;
  if (P.transform_skip_enabled && !s.cu_tq_bypass && log2n <= 2) ts = decode_bin(s, CTX_TRANSFORM_SKIP + (c_idx ? 1 : 0));
  int ctx_offset, ctx_shift;
  if (c_idx == 0) { ctx_offset = 3 * (log2n - 2) + ((log2n - 1) >> 2); ctx_shift = (log2n + 1) >> 2; }
  else { ctx_offset = 15; ctx_shift = log2n - 2; }
  const int c_max = (log2n << 1) - 1;
  int px = 0, py = 0;
  while (px < c_max && decode_bin(s, CTX_LAST_X + ctx_offset + (px >> ctx_shift))) px++;
  while (py < c_max && decode_bin(s, CTX_LAST_Y + ctx_offset + (py >> ctx_shift))) py++;
  int last_x = px, last_y = py;
  if (px > 3) last_x = (1 << ((px >> 1) - 1)) * (2 + (px & 1)) + decode_bypass_bits(s, (px >> 1) - 1);
  if (py > 3) last_y = (1 << ((py >> 1) - 1)) * (2 + (py & 1)) + decode_bypass_bits(s, (py >> 1) - 1);
  int scan_idx = 0;
  if (log2n == 2 || (log2n == 3 && c_idx == 0)) {
    if (pred_mode >= 6 && pred_mode <= 14) scan_idx = 2;
    else if (pred_mode >= 22 && pred_mode <= 30) scan_idx = 1;
  }
  if (scan_idx == 2) { int t = last_x; last_x = last_y; last_y = t; }
  if (last_x >= n || last_y >= n) { s.err = DEV_ERR_SYNTAX; return ts; }

  // locate the last position in scan order: sub-block (last_x>>2, last_y>>2), position inside it
  const int lg = log2n - 2;  // log2 of the sub-block grid width
  int last_sb = 0, last_pos = 0;
  {
    int xs_t = last_x >> 2, ys_t = last_y >> 2, xp_t = last_x & 3, yp_t = last_y & 3;
    int nsb = 1 << (2 * lg);
    for (int i = 0; i < nsb; i++) { int xs, ys; scan_sb(s, lg, scan_idx, i, xs, ys); if (xs == xs_t && ys == ys_t) { last_sb = i; break; } }
    for (int k = 0; k < 16; k++) { int x, y; scan_pos4(scan_idx, k, x, y); if (x == xp_t && y == yp_t) { last_pos = k; break; } }
  }
  uint64_t csbf = 0;  // coded_sub_block_flag bitmap, bit (ys*8 + xs)
  const int sbw = 1 << lg;
  int g1_carry = 1, first_sb_with_g1 = 1;
  for (int i = last_sb; i >= 0; i--) {
    int xs, ys;
    scan_sb(s, lg, scan_idx, i, xs, ys);
    int infer_dc = 0, coded;
    int right = (xs < sbw - 1) ? (int)((csbf >> (ys * 8 + xs + 1)) & 1) : 0;
    int below = (ys < sbw - 1) ? (int)((csbf >> ((ys + 1) * 8 + xs)) & 1) : 0;
    if (i < last_sb && i > 0) {
      coded = decode_bin(s, CTX_CODED_SUB_BLOCK + ((right | below) ? 1 : 0) + (c_idx ? 2 : 0));
      infer_dc = 1;
    } else coded = 1;
    if (coded) csbf |= 1ull << (ys * 8 + xs);
    if (!coded) continue;
    const int prev_csbf = right | (below << 1);
    uint32_t sig = 0;  // bit k = sig_coeff_flag at scan position k
    int n_start = 15;
    if (i == last_sb) { sig = 1u << last_pos; n_start = last_pos - 1; }
    for (int k = n_start; k >= 0; k--) {
      if (k > 0 || !infer_dc) {
        int xp, yp;
        scan_pos4(scan_idx, k, xp, yp);
        int sig_ctx;
        if (log2n == 2) sig_ctx = (int)((CTXIDXMAP4 >> (((yp << 2) + xp) * 4)) & 15);
        else if (((xs | ys) | (xp | yp)) == 0) sig_ctx = 0;
        else {
          if (prev_csbf == 0) sig_ctx = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
          else if (prev_csbf == 1) sig_ctx = (yp == 0) ? 2 : (yp == 1) ? 1 : 0;
          else if (prev_csbf == 2) sig_ctx = (xp == 0) ? 2 : (xp == 1) ? 1 : 0;
          else sig_ctx = 2;
          if (c_idx == 0) {
            if (xs | ys) sig_ctx += 3;
            sig_ctx += (log2n == 3) ? (scan_idx == 0 ? 9 : 15) : 21;
          } else sig_ctx += (log2n == 3) ? 9 : 12;
        }
        int f = decode_bin(s, CTX_SIG_COEFF + (c_idx == 0 ? sig_ctx : 27 + sig_ctx));
        if (f) { sig |= 1u << k; infer_dc = 0; }
      } else sig |= 1u;  // k == 0 inferred significant
    }
    if (!sig) continue;
    // greater1 / greater2 flags
    uint32_t g1 = 0, g2 = 0;
    int ctx_set = (i == 0 || c_idx > 0) ? 0 : 2;
    if (!first_sb_with_g1 && g1_carry == 0) ctx_set++;
    first_sb_with_g1 = 0;
    int g1_ctx = 1, num_g1 = 0, last_g1_pos = -1;
    const int last_sig_pos = 31 - __clz((int)sig), first_sig_pos = __ffs((int)sig) - 1;
    for (int k = last_sig_pos; k >= first_sig_pos && num_g1 < 8; k--) {
      if (!((sig >> k) & 1)) continue;
      int f = decode_bin(s, CTX_GREATER1 + ctx_set * 4 + (g1_ctx > 3 ? 3 : g1_ctx) + (c_idx ? 16 : 0));
      if (f) { g1 |= 1u << k; g1_ctx = 0; if (last_g1_pos < 0) last_g1_pos = k; }
      else if (g1_ctx > 0) g1_ctx++;
      num_g1++;
    }
    g1_carry = g1_ctx;
    const int sign_hidden = s.cu_tq_bypass ? 0 : (last_sig_pos - first_sig_pos > 3);
    if (last_g1_pos >= 0 && decode_bin(s, CTX_GREATER2 + ctx_set + (c_idx ? 4 : 0))) g2 |= 1u << last_g1_pos;
    uint32_t signs = 0;
    for (int k = last_sig_pos; k >= first_sig_pos; k--)
      if (((sig >> k) & 1) && (!P.sign_data_hiding || !sign_hidden || k != first_sig_pos))
        if (decode_bypass(s)) signs |= 1u << k;
    int num_sig = 0, sum_abs = 0, rice = 0;
    for (int k = last_sig_pos; k >= first_sig_pos; k--) {
      if (!((sig >> k) & 1)) continue;
      int base = 1 + ((g1 >> k) & 1) + ((g2 >> k) & 1);
      int abs_level = base;
      if (base == ((num_sig < 8) ? ((k == last_g1_pos) ? 3 : 2) : 1)) {
        abs_level += decode_remaining(s, rice);
        if (abs_level > 3 * (1 << rice)) rice = rice < 4 ? rice + 1 : 4;
      }
      int v = ((signs >> k) & 1) ? -abs_level : abs_level;
      if (P.sign_data_hiding && sign_hidden) {
        sum_abs += abs_level;
        if (k == first_sig_pos && (sum_abs & 1)) v = -v;
      }
      if (v > 32767 || v < -32768) { s.err = DEV_ERR_SYNTAX; v = 0; }
      int xp, yp;
      scan_pos4(scan_idx, k, xp, yp);
      s.L->coef[((ys << 2) + yp) * n + (xs << 2) + xp] = (int16_t)v;
      num_sig++;
    }
  }
  return ts;
}

// ---- 7.3.8.10 transform_unit (parse only) ------------------------------------------------------
__device__ __forceinline__ void parse_cu_qp_delta(State& s)
{
  int v = 0;
  if (decode_bin(s, CTX_CU_QP_DELTA)) {
    v = 1;
    while (v < 5 && decode_bin(s, CTX_CU_QP_DELTA + 1)) v++;
    if (v == 5) {
      int k = 0;
      while (decode_bypass(s)) { v += 1 << k; k++; if (k > 16) { s.err = DEV_ERR_SYNTAX; break; } }
      while (k-- > 0) v += decode_bypass(s) << k;
    }
  }
  int sign = v ? decode_bypass(s) : 0;
  s.is_cu_qp_delta_coded = 1;
  s.cu_qp_delta_val = sign ? -v : v;
  int off = 6 * (s.pp->bit_depth_luma - 8);
  if (s.cu_qp_delta_val < -(26 + off / 2) || s.cu_qp_delta_val > 25 + off / 2) s.err = DEV_ERR_SYNTAX;
  set_qp_y(s);
}

// ---- 7.3.8.5 coding_unit + 7.3.8.8 transform_tree, stackless over the z-ordered unit index -------
__device__ __forceinline__ void coding_unit(State& s, int zb /*unit z-index of the CU inside the CTB*/, int log2cb, int16_t* coef_y, int16_t* coef_cb,
                            int16_t* coef_cr)
{
  const PicParams& P = *s.pp;
  WaveLds& L = *s.L;
  const int ux0 = (int)compact1by1((uint32_t)zb), uy0 = (int)compact1by1((uint32_t)zb >> 1);
  const int n_units = 1 << (2 * (log2cb - 2));
  s.cu_tq_bypass = 0;
  if (P.transquant_bypass_enabled) s.cu_tq_bypass = decode_bin(s, CTX_CU_TQ_BYPASS);
  int part_nxn = 0;
  if (log2cb == P.log2_min_cb) part_nxn = decode_bin(s, CTX_PART_MODE) ? 0 : 1;
  if (part_nxn && log2cb == 3 && P.log2_min_tb > 2) { s.err = DEV_ERR_SYNTAX; part_nxn = 0; }
  set_qp_y(s);
  // CU-level map fill (lane-parallel, contiguous in z-order)
  {
    uint8_t fl = (uint8_t)(s.cu_tq_bypass ? UF_BYPASS : 0);
    __syncthreads();
    for (int i = s.lane; i < n_units; i += 64) { L.m_size[zb + i] = (uint8_t)(log2cb << 4); L.m_flags[zb + i] = fl; L.m_ipm[zb + i] = 1; }
    __syncthreads();
  }
  // intra prediction modes 7.3.8.5 / 8.4.2
  const int n_part = part_nxn ? 4 : 1;
  const int pu_units = n_units / n_part;       // units per PU (contiguous quadrant)
  const int pu_w = 1 << (log2cb - 2 - (part_nxn ? 1 : 0));  // PU width in units
  uint32_t prev_flags = 0;
  for (int k = 0; k < n_part; k++) prev_flags |= (uint32_t)decode_bin(s, CTX_PREV_INTRA_LUMA) << k;
  for (int k = 0; k < n_part; k++) {
    int mpm_idx = 0, rem = 0;
    if ((prev_flags >> k) & 1) { if (decode_bypass(s)) mpm_idx = decode_bypass(s) ? 2 : 1; }
    else rem = decode_bypass_bits(s, 5);
    const int ux = ux0 + (k & 1) * pu_w, uy = uy0 + (k >> 1) * pu_w;
    int cand_a = 1, cand_b = 1;
    if (ux > 0) cand_a = L.m_ipm[interleave4(ux - 1, uy)] & 63;
    else if (s.ctb_avail & AV_LEFT) cand_a = L.left_ipm[uy] & 63;
    if (uy > 0) cand_b = L.m_ipm[interleave4(ux, uy - 1)] & 63;  // above CTB row: INTRA_DC (8.4.2)
    int c0, c1, c2;
    if (cand_a == cand_b) {
      if (cand_a < 2) { c0 = 0; c1 = 1; c2 = 26; }
      else { c0 = cand_a; c1 = 2 + ((cand_a + 29) & 31); c2 = 2 + ((cand_a - 2 + 1) & 31); }
    } else {
      c0 = cand_a; c1 = cand_b;
      if (cand_a != 0 && cand_b != 0) c2 = 0; else if (cand_a != 1 && cand_b != 1) c2 = 1; else c2 = 26;
    }
    int mode;
    if ((prev_flags >> k) & 1) mode = mpm_idx == 0 ? c0 : (mpm_idx == 1 ? c1 : c2);
    else {
      int t;
      if (c0 > c1) { t = c0; c0 = c1; c1 = t; }
      if (c0 > c2) { t = c0; c0 = c2; c2 = t; }
      if (c1 > c2) { t = c1; c1 = c2; c2 = t; }
      mode = rem;
      if (mode >= c0) mode++;
      if (mode >= c1) mode++;
      if (mode >= c2) mode++;
    }
    __syncthreads();
    for (int i = s.lane; i < pu_units; i += 64) L.m_ipm[zb + k * pu_units + i] = (uint8_t)mode;
    __syncthreads();
  }
  int chroma_mode = 1;
  if (P.chroma_format_idc) {
    int icpm = 4;
    if (decode_bin(s, CTX_INTRA_CHROMA)) icpm = decode_bypass_bits(s, 2);
    const int lm = L.m_ipm[zb] & 63;
    if (icpm == 4) chroma_mode = lm;
    else { int m = icpm == 0 ? 0 : icpm == 1 ? 26 : icpm == 2 ? 10 : 1; chroma_mode = (m == lm) ? 34 : m; }
  }
  __syncthreads();
  for (int i = s.lane; i < n_units; i += 64) L.m_ipmc[zb + i] = (uint8_t)chroma_mode;
  __syncthreads();

  // ---- transform tree ----
  const int max_trafo_depth = P.max_th_depth_intra + part_nxn;
  const int deblock = !s.sl.deblocking_disabled;
  uint32_t cbf_cb_bits = 0, cbf_cr_bits = 0;  // bit d = cbf at trafoDepth d along the current path
  int q = 0;
  while (q < n_units && !s.err) {
    int t;  // log2 size of the node that starts at q
    if (q == 0) t = log2cb; else { t = 2 + ((__ffs(q) - 1) >> 1); if (t > log2cb) t = log2cb; }
    for (;;) {
      const int depth = log2cb - t;
      int split;
      if (t <= P.log2_max_tb && t > P.log2_min_tb && depth < max_trafo_depth && !(part_nxn && depth == 0))
        split = decode_bin(s, CTX_SPLIT_TRANSFORM + 5 - t);
      else split = (t > P.log2_max_tb || (part_nxn && depth == 0)) ? 1 : 0;
      if (P.chroma_format_idc) {
        uint32_t bit = 1u << depth, pbit = depth ? (1u << (depth - 1)) : 0;
        if (t > 2) {
          int cb = 0, cr = 0;
          if (depth == 0 || (cbf_cb_bits & pbit)) cb = decode_bin(s, CTX_CBF_CHROMA + depth);
          if (depth == 0 || (cbf_cr_bits & pbit)) cr = decode_bin(s, CTX_CBF_CHROMA + depth);
          cbf_cb_bits = (cbf_cb_bits & ~bit) | (cb ? bit : 0);
          cbf_cr_bits = (cbf_cr_bits & ~bit) | (cr ? bit : 0);
        } else {  // 4x4 luma: inherits the parent's flags (7.4.9.8)
          cbf_cb_bits = (cbf_cb_bits & ~bit) | ((cbf_cb_bits & pbit) ? bit : 0);
          cbf_cr_bits = (cbf_cr_bits & ~bit) | ((cbf_cr_bits & pbit) ? bit : 0);
        }
      }
      if (!split) break;
      t--;
    }
    // leaf transform unit at unit index zb + q, size 1 << t
    const int depth = log2cb - t;
    const int zu = zb + q;
    const int tu_units = 1 << (2 * (t - 2));
    const int cbf_luma = decode_bin(s, CTX_CBF_LUMA + (depth == 0 ? 1 : 0));
    const int cbf_cb = (cbf_cb_bits >> depth) & 1, cbf_cr = (cbf_cr_bits >> depth) & 1;
    if ((cbf_luma | cbf_cb | cbf_cr) && P.cu_qp_delta_enabled && !s.is_cu_qp_delta_coded) parse_cu_qp_delta(s);
    const int luma_mode = L.m_ipm[zu] & 63;
    int ts_y = 0, ts_cb = 0, ts_cr = 0;
    if (cbf_luma) { ts_y = residual_coding(s, t, 0, luma_mode); flush_coef(s, coef_y + zu * 16, 1 << (2 * t)); }
    int do_chroma = 0, zc = zu, tc = t - 1;
    if (P.chroma_format_idc) {
      if (t > 2) do_chroma = 1;
      else if ((q & 3) == 3) { do_chroma = 1; zc = zb + (q & ~3); tc = 2; }
    }
    if (do_chroma) {
      if (cbf_cb) { ts_cb = residual_coding(s, tc, 1, chroma_mode); flush_coef(s, coef_cb + zc * 4, 1 << (2 * tc)); }
      if (cbf_cr) { ts_cr = residual_coding(s, tc, 2, chroma_mode); flush_coef(s, coef_cr + zc * 4, 1 << (2 * tc)); }
    }
    // TU-level map fill: size, cbf, transform-skip, deblocking edges (8.7.2.2 / 8.7.2.3)
    {
      const int tux0 = (int)compact1by1((uint32_t)zu), tuy0 = (int)compact1by1((uint32_t)zu >> 1);
      const int edge_l = deblock && (tux0 > 0 || (s.ctb_avail & AV_EDGE_LEFT));
      const int edge_t = deblock && (tuy0 > 0 || (s.ctb_avail & AV_EDGE_UP));
      uint8_t fl = (uint8_t)((cbf_luma ? UF_CBF_LUMA : 0) | ((do_chroma && cbf_cb) ? UF_CBF_CB : 0) | ((do_chroma && cbf_cr) ? UF_CBF_CR : 0) |
                             (s.cu_tq_bypass ? UF_BYPASS : 0) | (ts_y ? UF_TS_LUMA : 0));
      uint8_t ipm = (uint8_t)(luma_mode | (ts_cb ? 64 : 0) | (ts_cr ? 128 : 0));
      __syncthreads();
      for (int i = s.lane; i < tu_units; i += 64) {
        const int rx = (int)compact1by1((uint32_t)i), ry = (int)compact1by1((uint32_t)i >> 1);
        uint8_t f = fl;
        if (rx == 0 && edge_l) f |= UF_VEDGE;
        if (ry == 0 && edge_t) f |= UF_HEDGE;
        L.m_flags[zu + i] = f;
        L.m_size[zu + i] = (uint8_t)((log2cb << 4) | t);
        L.m_ipm[zu + i] = ipm;
      }
      __syncthreads();
    }
    q += tu_units;
  }
  set_qp_y(s);
  __syncthreads();
  for (int i = s.lane; i < n_units; i += 64) L.m_qp[zb + i] = (int8_t)s.cur_qp_y;
  __syncthreads();
  s.last_qp_y = s.cur_qp_y;
}

// ---- 7.3.8.3 sao ---------------------------------------------------------------------------------
__device__ __forceinline__ void parse_sao(State& s, const SaoParams* sao_up /*global, above CTB*/, int allow_left, int allow_up)
{
  const PicParams& P = *s.pp;
  WaveLds& L = *s.L;
  int merge_left = 0, merge_up = 0;
  if (allow_left) merge_left = decode_bin(s, CTX_SAO_MERGE);
  if (allow_up && !merge_left) merge_up = decode_bin(s, CTX_SAO_MERGE);
  const int ncomp = P.chroma_format_idc ? 3 : 1;
  if (merge_left) { for (int c = 0; c < ncomp; c++) L.sao_cur[c] = L.sao_left[c]; return; }
  if (merge_up) { for (int c = 0; c < ncomp; c++) L.sao_cur[c] = sao_up[c]; return; }
  for (int c = 0; c < ncomp; c++) {
    SaoParams sp;
    sp.type = 0; sp.band_or_class = 0; sp.offset[0] = sp.offset[1] = sp.offset[2] = sp.offset[3] = 0;
    const int on = c == 0 ? s.sl.sao_luma : s.sl.sao_chroma;
    if (on) {
      int type;
      if (c == 2) type = L.sao_cur[1].type;
      else { type = 0; if (decode_bin(s, CTX_SAO_TYPE)) type = decode_bypass(s) ? 2 : 1; }
      sp.type = (uint8_t)type;
      if (type) {
        const int bd = c ? P.bit_depth_chroma : P.bit_depth_luma;
        const int c_max = (1 << ((bd < 10 ? bd : 10) - 5)) - 1;
        int a[4], sg[4] = {0, 0, 1, 1};
        for (int i = 0; i < 4; i++) { int v = 0; while (v < c_max && decode_bypass(s)) v++; a[i] = v; }
        if (type == 1) {
          for (int i = 0; i < 4; i++) sg[i] = a[i] ? decode_bypass(s) : 0;
          sp.band_or_class = (uint8_t)decode_bypass_bits(s, 5);
        } else {
          if (c == 0 || c == 1) sp.band_or_class = (uint8_t)decode_bypass_bits(s, 2);
          else sp.band_or_class = L.sao_cur[1].band_or_class;
        }
        const int sh = bd - (bd < 10 ? bd : 10);
        for (int i = 0; i < 4; i++) sp.offset[i] = (int16_t)((sg[i] ? -a[i] : a[i]) << sh);
      }
    }
    L.sao_cur[c] = sp;
  }
}

}  // namespace

// =================================================================================================
__global__ __launch_bounds__(64) void k_parse(ParseArgs A)
{
  __shared__ WaveLds lds;
  __shared__ uint32_t s_ticket;
  const int lane = threadIdx.x;
  if (lane == 0) s_ticket = atomicAdd(A.ticket, 1u);
  for (int i = lane; i < 256; i += 64) lds.range_lps[i] = c_range_lps[i];
  lds.next_lps[lane] = c_next_lps[lane];
  lds.diag8[lane] = c_diag8[lane];
  for (int i = lane * 8; i < 32 * 32; i += 512) *(uint4*)&lds.coef[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const uint32_t sub_idx = s_ticket;
  if (sub_idx >= A.num_subs) return;
  if (A.debug_level == 1) return;
  const Substream sub = A.subs[sub_idx];
  const PicParams& P = A.pics[sub.pic];

  State s;
  s.pp = &P; s.arena = A.arena; s.bs = A.arena + P.off_bitstream; s.L = &lds; s.lane = lane; s.err = 0;
  s.sl = ((const SliceParams*)(A.arena + P.off_slices))[sub.slice_idx];
  s.units_log2 = P.units_per_ctb_log2;
  s.is_cu_qp_delta_coded = 0; s.cu_qp_delta_val = 0; s.qpy_pred = s.sl.slice_qp_y; s.last_qp_y = s.sl.slice_qp_y; s.cur_qp_y = s.sl.slice_qp_y;
  s.cu_tq_bypass = 0;

  const uint16_t* ts_to_rs = (const uint16_t*)(A.arena + P.off_ctb_ts_to_rs);
  const CtbInfo* ctb_info = (const CtbInfo*)(A.arena + P.off_ctb_info);
  SaoParams* sao_all = (SaoParams*)(A.arena + P.off_sao);
  uint8_t* g_size = A.arena + P.off_u_size; uint8_t* g_flags = A.arena + P.off_u_flags; uint8_t* g_ipm = A.arena + P.off_u_ipm;
  uint8_t* g_ipmc = A.arena + P.off_u_ipmc; uint8_t* g_qp = A.arena + P.off_u_qp;
  const int units = 1 << P.units_per_ctb_log2;
  const int uw = 1 << (P.log2_ctb - 2);  // units per CTB side
  const int ctb_size = 1 << P.log2_ctb;
  const int n_mincb_log2 = 2 * (P.log2_ctb - P.log2_min_cb);

  if (A.debug_level == 2) return;
  cabac_start(s, sub.byte_start, sub.byte_end);
  if (A.debug_level == 3) return;

  for (uint32_t k = 0; k < sub.num_ctbs && !s.err; k++) {
    const int ctb_rs = ts_to_rs[sub.first_ctb_ts + k];
    const int cx = ctb_rs % P.ctb_w, cy = ctb_rs / P.ctb_w;
    const CtbInfo ci = ctb_info[ctb_rs];
    s.ctb_rs = ctb_rs; s.x_ctb = cx << P.log2_ctb; s.y_ctb = cy << P.log2_ctb; s.ctb_avail = ci.avail;

    // ---- WPP dependency on the CTB row above (bounded spin, relaxed poll + one acquire) ----
    if (sub.dep_sub >= 0) {
      uint32_t need = k + 2 < sub.dep_len ? k + 2 : sub.dep_len;
      uint32_t spins = 0;
      while (__hip_atomic_load(&A.progress[sub.dep_sub], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22) || __hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { s.err = DEV_ERR_TIMEOUT; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
      if (s.err) break;
    }
    // ---- context initialisation / synchronisation (9.3.1) ----
    if (k == 0) {
      if (sub.wpp_sync && sub.dep_sub >= 0) {
        __syncthreads();
        for (int i = lane; i < CTX_STORE / 4; i += 64) ((uint32_t*)lds.ctx)[i] = ((const uint32_t*)(A.ctx_store + (size_t)sub.dep_sub * CTX_STORE))[i];
        __syncthreads();
      } else init_contexts(s);
    }
    if (A.debug_level == 4) return;
    // ---- neighbour rows for this CTB ----
    __syncthreads();
    if (lane < uw) {
      uint8_t v = 0;
      if (ci.avail & AV_UP) v = g_size[(size_t)(ctb_rs - P.ctb_w) * units + interleave4(lane, uw - 1)];
      lds.up_size[lane] = v;
    }
    __syncthreads();

    if (A.debug_level == 5) return;
    // ---- coding_tree_unit ----
    if (s.sl.sao_luma || s.sl.sao_chroma) {
      parse_sao(s, sao_all + (size_t)(ctb_rs - P.ctb_w) * 3, (ci.avail & AV_LEFT) && k > 0, (ci.avail & AV_UP) ? 1 : 0);
    } else {
      for (int c = 0; c < 3; c++) { lds.sao_cur[c].type = 0; lds.sao_cur[c].band_or_class = 0; for (int i = 0; i < 4; i++) lds.sao_cur[c].offset[i] = 0; }
    }
    if (!P.cu_qp_delta_enabled) { s.is_cu_qp_delta_coded = 0; s.cu_qp_delta_val = 0; s.qpy_pred = s.last_qp_y; }

    if (A.debug_level == 6) return;
    int16_t* coef_y = (int16_t*)(A.arena + P.off_coeff[0]) + (size_t)ctb_rs * ctb_size * ctb_size;
    int16_t* coef_cb = (int16_t*)(A.arena + P.off_coeff[1]) + (size_t)ctb_rs * (ctb_size * ctb_size / 4);
    int16_t* coef_cr = (int16_t*)(A.arena + P.off_coeff[2]) + (size_t)ctb_rs * (ctb_size * ctb_size / 4);

    // coding quadtree, stackless over the z-ordered min-CB index
    const int n_mincb = 1 << n_mincb_log2;
    const int mincb_units_log2 = 2 * (P.log2_min_cb - 2);
    int p = 0;
    while (p < n_mincb && !s.err) {
      int lg;  // log2 size of the node starting at p
      if (p == 0) lg = P.log2_ctb; else { lg = P.log2_min_cb + ((__ffs(p) - 1) >> 1); if (lg > P.log2_ctb) lg = P.log2_ctb; }
      const int zb = p << mincb_units_log2;
      const int ux = (int)compact1by1((uint32_t)zb), uy = (int)compact1by1((uint32_t)zb >> 1);
      const int x0 = s.x_ctb + (ux << 2), y0 = s.y_ctb + (uy << 2);
      if (x0 >= P.width || y0 >= P.height) { p += 1 << (2 * (lg - P.log2_min_cb)); continue; }
      for (;;) {
        const int size = 1 << lg;
        int split;
        if (x0 + size <= P.width && y0 + size <= P.height && lg > P.log2_min_cb) {
          const int depth = P.log2_ctb - lg;
          int inc = 0;
          const int l = left_cb_log2(s, ux, uy), u = up_cb_log2(s, ux, uy);
          if (l && P.log2_ctb - l > depth) inc++;
          if (u && P.log2_ctb - u > depth) inc++;
          split = decode_bin(s, CTX_SPLIT_CU + inc);
        } else split = lg > P.log2_min_cb;
        if (P.cu_qp_delta_enabled && lg >= P.log2_min_cu_qp_delta_size) {
          s.is_cu_qp_delta_coded = 0; s.cu_qp_delta_val = 0;
          derive_qp_pred(s, ux, uy);
        }
        if (!split) break;
        lg--;
      }
      if (!P.cu_qp_delta_enabled) s.qpy_pred = s.last_qp_y;
      coding_unit(s, zb, lg, coef_y, coef_cb, coef_cr);
      p += 1 << (2 * (lg - P.log2_min_cb));
    }

    if (A.debug_level == 7) return;
    // end_of_slice_segment_flag / end_of_subset_one_bit
    const int last = (k + 1 == sub.num_ctbs);
    const int eos = decode_terminate(s);
    if (last) {
      if (sub.last_in_slice_segment) { if (!eos) s.err = DEV_ERR_TERMINATE; }
      else { if (eos || !decode_terminate(s)) s.err = DEV_ERR_TERMINATE; }
    } else if (eos) s.err = DEV_ERR_TERMINATE;

    // ---- publish the CTB: unit maps, SAO parameters, WPP context table ----
    __syncthreads();
    {
      const size_t base = (size_t)ctb_rs * units;
      for (int i = lane * 4; i < units; i += 256) {
        *(uint32_t*)(g_size + base + i) = *(const uint32_t*)&lds.m_size[i];
        *(uint32_t*)(g_flags + base + i) = *(const uint32_t*)&lds.m_flags[i];
        *(uint32_t*)(g_ipm + base + i) = *(const uint32_t*)&lds.m_ipm[i];
        *(uint32_t*)(g_ipmc + base + i) = *(const uint32_t*)&lds.m_ipmc[i];
        *(uint32_t*)(g_qp + base + i) = *(const uint32_t*)&lds.m_qp[i];
      }
      if (lane < 3) sao_all[(size_t)ctb_rs * 3 + lane] = lds.sao_cur[lane];
      if (sub.has_dependent && k == 1)
        for (int i = lane; i < CTX_STORE / 4; i += 64) ((uint32_t*)(A.ctx_store + (size_t)sub_idx * CTX_STORE))[i] = ((const uint32_t*)lds.ctx)[i];
      // right column / SAO of this CTB become the left neighbour of the next one
      if (lane < uw) { const int z = interleave4(uw - 1, lane); lds.left_size[lane] = lds.m_size[z]; lds.left_ipm[lane] = lds.m_ipm[z]; }
      if (lane < 3) lds.sao_left[lane] = lds.sao_cur[lane];
    }
    if (A.debug_level == 8) return;
    __syncthreads();
    if (sub.has_dependent) {
      if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&A.progress[sub_idx], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (s.err) {
    if (lane == 0) atomicCAS((int*)A.status, 0, s.err | (int)(sub_idx << 8));
    // unblock waiters: they also poll the status word
  }
}

void launch_parse(const ParseArgs& a, hipStream_t s)
{
  if (a.num_subs) hipLaunchKernelGGL(k_parse, dim3(a.num_subs), dim3(64), 0, s, a);
}

}  // namespace hipdec
