// filter_kernels.hip — in-loop filters: deblocking (8.7.2) and sample adaptive offset (8.7.3) with
// the conformance-window crop fused into the SAO store.
//
// Stands in for libde265's deblocking / SAO stages behind de265_decode()
// (reference call site libheif/plugins/decoder_libde265.cc:402) and for the plane hand-over crop of
// convert_libde265_image_to_heif_image (libheif/plugins/decoder_libde265.cc:97-171).
//
// MI355X mapping: both filters are embarrassingly parallel streaming passes (HBM-bound):
//   deblock  one thread per 4-sample edge segment; vertical edges of the whole batch first, then
//            horizontal edges (kernel boundary = the ordering the standard requires); consecutive
//            lanes take consecutive segments of one sample row so each row access of a wave is one
//            contiguous 256-512 B span; in place (segments never overlap: edges are 8 apart, at most
//            3 samples are modified per side).  Algorithmic bytes: 3*s per luma pixel per pass.
//   SAO      one thread per 4 output samples, reads the deblocked picture, writes the cropped output
//            plane.  Algorithmic bytes: 1.5*s in + 1.5*s out per luma pixel.
// grid.y = picture index of the batch, grid.z = colour component where applicable.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hevc_device.h"
#include "kernels.h"
#include "color_device.h"

namespace hipdec {


namespace {

__constant__ uint8_t c_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
                                   26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
__constant__ uint8_t c_tc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
__constant__ uint8_t c_chroma_qp_f[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};

// ---- two 16-bit values per operation (v_pk_*_i16, v_perm_b32): the lean SAO + RGB kernel and the 8-bit luma deblocking filter ----
namespace swar {
#ifndef HIPDEC_HOST_EMU
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#define HIPDEC_SWAR_PK(NAME, INSN)                                                                                                              \
  __device__ __forceinline__ uint32_t NAME(uint32_t a, uint32_t b) { uint32_t r; asm(INSN " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }  \
  __device__ __forceinline__ uint32_t NAME##_c(uint32_t a, uint32_t b) { uint32_t r; asm(INSN " %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }   // b: a wave-uniform constant pair
HIPDEC_SWAR_PK(pk_add, "v_pk_add_i16") HIPDEC_SWAR_PK(pk_sub, "v_pk_sub_i16") HIPDEC_SWAR_PK(pk_max, "v_pk_max_i16") HIPDEC_SWAR_PK(pk_min, "v_pk_min_i16")
#undef HIPDEC_SWAR_PK
// shifts of both halves by the (wave-uniform) amounts in the halves of n, e.g. 0x00030003
__device__ __forceinline__ uint32_t pk_shl_c(uint32_t a, uint32_t n) { uint32_t r; asm("v_pk_lshlrev_b16 %0, %2, %1" : "=v"(r) : "v"(a), "s"(n)); return r; }
__device__ __forceinline__ uint32_t pk_ashr_c(uint32_t a, uint32_t n) { uint32_t r; asm("v_pk_ashrrev_i16 %0, %2, %1" : "=v"(r) : "v"(a), "s"(n)); return r; }
struct __attribute__((packed)) U1 { uint32_t v; };
__device__ __forceinline__ uint32_t lds32u(const uint8_t* p) { return ((const U1*)p)->v; }   // any alignment: one ds_read_b32
#else   // CPU-test build: the same operations spelled out
inline uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)   // v_perm_b32: byte i of the result = byte sel[i] of {hi, lo}; 8 .. 11: the sign of byte 1 / 3 / 5 / 7; 12: 0; >= 13: 0xff
{
  const uint64_t in = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t k = (sel >> (8 * i)) & 255u;
    uint32_t b;
    if (k < 8) b = (uint32_t)(in >> (8 * k)) & 255u;
    else if (k < 12) b = ((in >> (16 * (k - 8) + 15)) & 1u) ? 255u : 0u;
    else b = k == 12 ? 0u : 255u;
    r |= b << (8 * i);
  }
  return r;
}
#define HIPDEC_SWAR_PK(NAME, EXPR)                                                                              \
  inline uint32_t NAME(uint32_t a, uint32_t b)                                                                  \
  {                                                                                                             \
    uint32_t r = 0;                                                                                             \
    for (int h = 0; h < 2; h++) { const int x = (int16_t)(a >> (16 * h)), y = (int16_t)(b >> (16 * h)); r |= ((uint32_t)(EXPR) & 0xffffu) << (16 * h); }  \
    return r;                                                                                                   \
  }                                                                                                             \
  inline uint32_t NAME##_c(uint32_t a, uint32_t b) { return NAME(a, b); }
HIPDEC_SWAR_PK(pk_add, x + y) HIPDEC_SWAR_PK(pk_sub, x - y) HIPDEC_SWAR_PK(pk_max, x > y ? x : y) HIPDEC_SWAR_PK(pk_min, x < y ? x : y)
HIPDEC_SWAR_PK(pk_shl_c, x << (y & 15)) HIPDEC_SWAR_PK(pk_ashr_c, x >> (y & 15))
#undef HIPDEC_SWAR_PK
inline uint32_t lds32u(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
#endif
constexpr uint32_t kEven = 0x0c020c00u, kOdd = 0x0c030c01u;   // bytes 0, 2 / 1, 3 of a dword as two zero-extended 16-bit values
constexpr uint32_t kSextOdd = 0x09030801u;                   // bytes 1, 3 as two SIGN-extended 16-bit values
__device__ __forceinline__ uint32_t pk_neg(uint32_t a) { return pk_sub(0u, a); }
__device__ __forceinline__ uint32_t pk_sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }   // v_bfi_b32
__device__ __forceinline__ uint32_t pk_clip255(uint32_t a) { return pk_min_c(pk_max_c(a, 0u), 0x00ff00ffu); }
}  // namespace swar

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ uint32_t interleave4(uint32_t x, uint32_t y)
{
  x = (x | (x << 2)) & 0x33; x = (x | (x << 1)) & 0x55;
  y = (y | (y << 2)) & 0x33; y = (y | (y << 1)) & 0x55;
  return x | (y << 1);
}
// index of 4x4 unit (ux, uy) (picture coordinates) in the CTB-major, z-ordered unit maps
__device__ __forceinline__ size_t unit_index(const PicParams& P, int ux, int uy, int* ctb_rs)
{
  const int l = P.log2_ctb - 2, mask = (1 << l) - 1;
  const int c = (uy >> l) * P.ctb_w + (ux >> l);
  *ctb_rs = c;
  return ((size_t)c << P.units_per_ctb_log2) + interleave4((uint32_t)(ux & mask), (uint32_t)(uy & mask));
}

// The 8 x 4 window of a luma edge segment (4 lines along the edge, 4 samples on either side) in registers, moved with whole-dword accesses:
// a vertical edge's lines are picture rows (8 contiguous samples each: x is a multiple of 8, so the window starts dword-aligned), a horizontal
// edge's lines are columns (every picture row of the window is 4 contiguous samples).  Windows of different segments never overlap (edges are
// 8 apart), so writing a whole window back - unchanged samples included - races with nobody.  (Byte accesses made this kernel issue 32 loads and
// up to 24 stores per segment.)   p[k][i] / q[k][i]: line k, distance i from the edge
template <typename Pix, int DIR>
struct EdgeWindow {
  static constexpr int ES = (int)sizeof(Pix), NW = ES == 1 ? 8 : 16;
  uint32_t w[NW];   // packed as loaded: the samples are extracted line by line, so the live set stays small (8 waves per SIMD)
  __device__ __forceinline__ void load(const Pix* pix, int stride)
  {
    if (DIR == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t* src = (const uint32_t*)(pix + (size_t)k * stride - 4);
#pragma unroll
        for (int i = 0; i < 2 * ES; i++) w[2 * ES * k + i] = src[i];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; r++) {           // picture rows y - 4 .. y + 3
        const uint32_t* src = (const uint32_t*)(pix + (ptrdiff_t)(r - 4) * stride);
#pragma unroll
        for (int i = 0; i < ES; i++) w[ES * r + i] = src[i];
      }
    }
  }
  __device__ __forceinline__ void store(Pix* pix, int stride) const
  {
    if (DIR == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t* dst = (uint32_t*)(pix + (size_t)k * stride - 4);
#pragma unroll
        for (int i = 0; i < 2 * ES; i++) dst[i] = w[2 * ES * k + i];
      }
    } else {
#pragma unroll
      for (int r = 1; r < 7; r++) {           // rows p2 .. q2: the outermost rows (p3, q3) are never modified
        uint32_t* dst = (uint32_t*)(pix + (ptrdiff_t)(r - 4) * stride);
#pragma unroll
        for (int i = 0; i < ES; i++) dst[i] = w[ES * r + i];
      }
    }
  }
  // sample of line k at position j across the edge: j = 0 .. 3 are p3 .. p0, j = 4 .. 7 are q0 .. q3
  static __device__ __forceinline__ int word(int k, int j) { return DIR == 0 ? (ES == 1 ? 2 * k + (j >> 2) : 4 * k + (j >> 1)) : (ES == 1 ? j : 2 * j + (k >> 1)); }
  static __device__ __forceinline__ int shift(int k, int j) { return DIR == 0 ? (ES == 1 ? 8 * (j & 3) : 16 * (j & 1)) : (ES == 1 ? 8 * k : 16 * (k & 1)); }
  __device__ __forceinline__ int get(int k, int j) const { return (int)((w[word(k, j)] >> shift(k, j)) & (ES == 1 ? 255u : 0xffffu)); }
  __device__ __forceinline__ void set(int k, int j, int v)
  {
    const uint32_t m = (ES == 1 ? 255u : 0xffffu) << shift(k, j);
    w[word(k, j)] = (w[word(k, j)] & ~m) | ((uint32_t)v << shift(k, j));
  }
};

// luma edge segment of 4 lines (8.7.2.5.3 decisions, 8.7.2.5.7 filters); pix -> q0 of line 0.  DIR 0: vertical edge (lines are rows), 1: horizontal
// (W: the segment's window, loaded by the caller together with everything else the segment reads - see k_deblock)
// deblock_luma_regs filters the window in registers and says whether anything may have changed; deblock_luma stores it as well
template <typename Pix, int DIR>
__device__ __forceinline__ bool deblock_luma_regs(EdgeWindow<Pix, DIR>& W, int qp_p, int qp_q, int beta_off2, int tc_off2, int bit_depth, int no_p, int no_q, int bs = 2)
{
  const int qpl = (qp_q + qp_p + 1) >> 1;
  const int beta = c_beta[clip3(0, 51, qpl + (beta_off2 << 1))] * (1 << (bit_depth - 8));
  const int tc = c_tc[clip3(0, 53, qpl + 2 * (bs - 1) + (tc_off2 << 1))] * (1 << (bit_depth - 8));   // 8.7.2.5.3: bS 2 where a side is intra coded
#define WP(k, i) W.get(k, 3 - (i))
#define WQ(k, i) W.get(k, 4 + (i))
  const int dp0 = iabs(WP(0, 2) - 2 * WP(0, 1) + WP(0, 0)), dp3 = iabs(WP(3, 2) - 2 * WP(3, 1) + WP(3, 0));
  const int dq0 = iabs(WQ(0, 2) - 2 * WQ(0, 1) + WQ(0, 0)), dq3 = iabs(WQ(3, 2) - 2 * WQ(3, 1) + WQ(3, 0));
  const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
  if (dpq0 + dpq3 >= beta) return false;
  const int s0 = (2 * dpq0 < (beta >> 2)) && (iabs(WP(0, 3) - WP(0, 0)) + iabs(WQ(0, 0) - WQ(0, 3)) < (beta >> 3)) &&
                 (iabs(WP(0, 0) - WQ(0, 0)) < ((5 * tc + 1) >> 1));
  const int s3 = (2 * dpq3 < (beta >> 2)) && (iabs(WP(3, 3) - WP(3, 0)) + iabs(WQ(3, 0) - WQ(3, 3)) < (beta >> 3)) &&
                 (iabs(WP(3, 0) - WQ(3, 0)) < ((5 * tc + 1) >> 1));
  const int strong = s0 && s3;
  const int dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
  const int maxv = (1 << bit_depth) - 1;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int p0 = WP(k, 0), p1 = WP(k, 1), p2 = WP(k, 2), p3 = WP(k, 3), q0 = WQ(k, 0), q1 = WQ(k, 1), q2 = WQ(k, 2), q3 = WQ(k, 3);
    if (strong) {
      if (!no_p) {
        W.set(k, 3, clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3));
        W.set(k, 2, clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2));
        W.set(k, 1, clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3));
      }
      if (!no_q) {
        W.set(k, 4, clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3));
        W.set(k, 5, clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2));
        W.set(k, 6, clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3));
      }
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (iabs(delta) < tc * 10) {
        delta = clip3(-tc, tc, delta);
        if (!no_p) W.set(k, 3, clip3(0, maxv, p0 + delta));
        if (!no_q) W.set(k, 4, clip3(0, maxv, q0 - delta));
        if (dep && !no_p) W.set(k, 2, clip3(0, maxv, p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1)));
        if (deq && !no_q) W.set(k, 5, clip3(0, maxv, q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1)));
      }
    }
  }
#undef WP
#undef WQ
  return true;
}
template <typename Pix, int DIR>
__device__ __forceinline__ void deblock_luma(EdgeWindow<Pix, DIR>& W, Pix* pix, int stride, int qp_p, int qp_q, int beta_off2, int tc_off2, int bit_depth, int no_p, int no_q, int bs = 2)
{
  if (deblock_luma_regs<Pix, DIR>(W, qp_p, qp_q, beta_off2, tc_off2, bit_depth, no_p, no_q, bs)) W.store(pix, stride);
}

// The same segment for 8-bit luma with both sides filtered (no PCM / bypass side), TWO LINES PER OPERATION: the block of k_deblock_fused is eight rows of
// two dwords; the samples at distance i from the edge of the line pairs (0, 1) and (2, 3) are gathered into the halves of one register each with one
// v_perm_b32, decisions (lines 0 and 3: the low half of the first pair, the high half of the second) and both filters run on v_pk_*_i16 - every
// intermediate of 8.7.2.5.7 fits 16 bits (9 |q0 - p0| + 3 |q1 - p1| + 8 <= 3068) -, the per-line condition |delta| < 10 tc is a mask (sign of a
// difference, v_bfi_b32), and the results go back into the rows with v_perm_b32.  About half the vector instructions of the scalar form above.
// w: the block, row r = picture row yc - 4 + r, dword 0 / 1 = columns xc - 4 .. xc - 1 / xc .. xc + 3.  DIR 0: the vertical edge's rows 4 sgm .. 4 sgm + 3;
// DIR 1: the horizontal edge's columns 4 sgm .. 4 sgm + 3 (dword sgm of every row).
template <int DIR>
__device__ __forceinline__ bool deblock_luma_pk8(uint32_t (&w)[8][2], int sgm, int qp_p, int qp_q, int beta_off2, int tc_off2)
{
  using namespace swar;
  const int qpl = (qp_q + qp_p + 1) >> 1;
  const int beta = c_beta[clip3(0, 51, qpl + (beta_off2 << 1))];
  const int tc = c_tc[clip3(0, 53, qpl + 2 + (tc_off2 << 1))];   // bS 2 (intra pictures)
  uint32_t P[4][2], Q[4][2];   // [distance from the edge][line pair]: line 2 pr in the low half, line 2 pr + 1 in the high half
#pragma unroll
  for (int pr = 0; pr < 2; pr++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (DIR == 0) {
        const int r = 4 * sgm + 2 * pr;
        P[i][pr] = perm(w[r + 1][0], w[r][0], 0x0c000c00u | (uint32_t)(3 - i) | ((uint32_t)(7 - i) << 16));
        Q[i][pr] = perm(w[r + 1][1], w[r][1], 0x0c000c00u | (uint32_t)i | ((uint32_t)(4 + i) << 16));
      } else {
        const uint32_t sel = 0x0c000c00u | (uint32_t)(2 * pr) | ((uint32_t)(2 * pr + 1) << 16);
        P[i][pr] = perm(0u, w[3 - i][sgm], sel);
        Q[i][pr] = perm(0u, w[4 + i][sgm], sel);
      }
    }
  constexpr uint32_t K1 = 0x00010001u, K2 = 0x00020002u, K3 = 0x00030003u, K4 = 0x00040004u, K8 = 0x00080008u, K15 = 0x000f000fu;
  uint32_t dP[2], dQ[2];
#pragma unroll
  for (int pr = 0; pr < 2; pr++) {
    const uint32_t a = pk_sub(pk_add(P[2][pr], P[0][pr]), pk_shl_c(P[1][pr], K1)), b = pk_sub(pk_add(Q[2][pr], Q[0][pr]), pk_shl_c(Q[1][pr], K1));
    dP[pr] = pk_max(a, pk_neg(a)); dQ[pr] = pk_max(b, pk_neg(b));
  }
  const int dp0 = (int)(dP[0] & 0xffffu), dp3 = (int)(dP[1] >> 16), dq0 = (int)(dQ[0] & 0xffffu), dq3 = (int)(dQ[1] >> 16);
  const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
  if (dpq0 + dpq3 >= beta) return false;
  const int p00 = (int)(P[0][0] & 0xffffu), p30 = (int)(P[3][0] & 0xffffu), q00 = (int)(Q[0][0] & 0xffffu), q30 = (int)(Q[3][0] & 0xffffu);   // line 0
  const int p03 = (int)(P[0][1] >> 16), p33 = (int)(P[3][1] >> 16), q03 = (int)(Q[0][1] >> 16), q33 = (int)(Q[3][1] >> 16);                   // line 3
  const int s0 = (2 * dpq0 < (beta >> 2)) && (iabs(p30 - p00) + iabs(q00 - q30) < (beta >> 3)) && (iabs(p00 - q00) < ((5 * tc + 1) >> 1));
  const int s3 = (2 * dpq3 < (beta >> 2)) && (iabs(p33 - p03) + iabs(q03 - q33) < (beta >> 3)) && (iabs(p03 - q03) < ((5 * tc + 1) >> 1));
  const int dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
  if (s0 && s3) {
    const uint32_t T2 = (uint32_t)(2 * tc) * 0x00010001u;
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
      const uint32_t p0 = P[0][pr], p1 = P[1][pr], p2 = P[2][pr], p3 = P[3][pr], q0 = Q[0][pr], q1 = Q[1][pr], q2 = Q[2][pr], q3 = Q[3][pr];
      const uint32_t s = pk_add(p0, q0), t = pk_add(p1, s), u = pk_add(q1, s);
      const uint32_t np0 = pk_ashr_c(pk_add_c(pk_add(pk_add(p2, q1), pk_shl_c(t, K1)), K4), K3);                    // (p2 + 2 p1 + 2 p0 + 2 q0 + q1 + 4) >> 3
      const uint32_t np1 = pk_ashr_c(pk_add_c(pk_add(p2, t), K2), K2);                                              // (p2 + p1 + p0 + q0 + 2) >> 2
      const uint32_t np2 = pk_ashr_c(pk_add_c(pk_add(pk_add(pk_shl_c(pk_add(p3, p2), K1), p2), t), K4), K3);        // (2 p3 + 3 p2 + p1 + p0 + q0 + 4) >> 3
      const uint32_t nq0 = pk_ashr_c(pk_add_c(pk_add(pk_add(p1, q2), pk_shl_c(u, K1)), K4), K3);                    // (p1 + 2 p0 + 2 q0 + 2 q1 + q2 + 4) >> 3
      const uint32_t nq1 = pk_ashr_c(pk_add_c(pk_add(q2, u), K2), K2);                                              // (p0 + q0 + q1 + q2 + 2) >> 2
      const uint32_t nq2 = pk_ashr_c(pk_add_c(pk_add(pk_add(pk_shl_c(pk_add(q3, q2), K1), q2), u), K4), K3);        // (p0 + q0 + q1 + 3 q2 + 2 q3 + 4) >> 3
      P[0][pr] = pk_min(pk_max(np0, pk_sub(p0, T2)), pk_add(p0, T2));
      P[1][pr] = pk_min(pk_max(np1, pk_sub(p1, T2)), pk_add(p1, T2));
      P[2][pr] = pk_min(pk_max(np2, pk_sub(p2, T2)), pk_add(p2, T2));
      Q[0][pr] = pk_min(pk_max(nq0, pk_sub(q0, T2)), pk_add(q0, T2));
      Q[1][pr] = pk_min(pk_max(nq1, pk_sub(q1, T2)), pk_add(q1, T2));
      Q[2][pr] = pk_min(pk_max(nq2, pk_sub(q2, T2)), pk_add(q2, T2));
    }
  } else {
    const uint32_t TC = (uint32_t)tc * 0x00010001u, TC10 = (uint32_t)(10 * tc) * 0x00010001u, TCH = (uint32_t)(tc >> 1) * 0x00010001u;
    const uint32_t nTC = pk_neg(TC), nTCH = pk_neg(TCH);
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
      const uint32_t p0 = P[0][pr], p1 = P[1][pr], p2 = P[2][pr], q0 = Q[0][pr], q1 = Q[1][pr], q2 = Q[2][pr];
      const uint32_t d0 = pk_sub(q0, p0), d1 = pk_sub(q1, p1);
      const uint32_t delta = pk_ashr_c(pk_add_c(pk_sub(pk_add(pk_shl_c(d0, K3), d0), pk_add(pk_shl_c(d1, K1), d1)), K8), K4);   // (9 (q0 - p0) - 3 (q1 - p1) + 8) >> 4
      const uint32_t m = pk_ashr_c(pk_sub(pk_max(delta, pk_neg(delta)), TC10), K15);                                            // 0xffff where |delta| < 10 tc
      const uint32_t dc = pk_min(pk_max(delta, nTC), TC);
      P[0][pr] = pk_sel(m, pk_clip255(pk_add(p0, dc)), p0);
      Q[0][pr] = pk_sel(m, pk_clip255(pk_sub(q0, dc)), q0);
      if (dep) {
        const uint32_t x = pk_ashr_c(pk_add(pk_sub(pk_ashr_c(pk_add_c(pk_add(p2, p0), K1), K1), p1), dc), K1);                  // (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1
        P[1][pr] = pk_sel(m, pk_clip255(pk_add(p1, pk_min(pk_max(x, nTCH), TCH))), p1);
      }
      if (deq) {
        const uint32_t y = pk_ashr_c(pk_sub(pk_sub(pk_ashr_c(pk_add_c(pk_add(q2, q0), K1), K1), q1), dc), K1);                  // (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1
        Q[1][pr] = pk_sel(m, pk_clip255(pk_add(q1, pk_min(pk_max(y, nTCH), TCH))), q1);
      }
    }
  }
  // back into the rows (p3 / q3 never change)
#pragma unroll
  for (int pr = 0; pr < 2; pr++) {
    if (DIR == 0) {
      const int r = 4 * sgm + 2 * pr;
      {   // line 2 pr: the low halves
        const uint32_t xp = perm(P[1][pr], P[2][pr], 0x0c04000cu), yp = perm(P[0][pr], xp, 0x0402010cu);
        w[r][0] = (w[r][0] & 0x000000ffu) | yp;
        const uint32_t xq = perm(Q[1][pr], Q[0][pr], 0x0c0c0400u), yq = perm(Q[2][pr], xq, 0x0c040100u);
        w[r][1] = (w[r][1] & 0xff000000u) | yq;
      }
      {   // line 2 pr + 1: the high halves
        const uint32_t xp = perm(P[1][pr], P[2][pr], 0x0c06020cu), yp = perm(P[0][pr], xp, 0x0602010cu);
        w[r + 1][0] = (w[r + 1][0] & 0x000000ffu) | yp;
        const uint32_t xq = perm(Q[1][pr], Q[0][pr], 0x0c0c0602u), yq = perm(Q[2][pr], xq, 0x0c060100u);
        w[r + 1][1] = (w[r + 1][1] & 0xff000000u) | yq;
      }
    }
  }
  if (DIR == 1) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      w[3 - i][sgm] = perm(P[i][1], P[i][0], 0x06040200u);
      w[4 + i][sgm] = perm(Q[i][1], Q[i][0], 0x06040200u);
    }
  }
  return true;
}

template <typename Pix>
__device__ __forceinline__ void deblock_chroma(Pix* pix, int xs, int ys, int qp_p, int qp_q, int c_qp_pic_offset, int tc_off2, int bit_depth,
                                               int no_p, int no_q, bool c444 = false, int nlines = 4)
{
  const int qpi = ((qp_q + qp_p + 1) >> 1) + c_qp_pic_offset;
  // 8.7.2.5.5: QpC "as specified in table 8-10" for ChromaArrayType 1, Min(qPi, 51) otherwise
  const int qpc = c444 ? (qpi < 51 ? qpi : 51) : (qpi < 30 ? qpi : (qpi >= 44 ? qpi - 6 : c_chroma_qp_f[qpi - 30]));
  const int tc = c_tc[clip3(0, 53, qpc + 2 + (tc_off2 << 1))] * (1 << (bit_depth - 8));
  const int maxv = (1 << bit_depth) - 1;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (k >= nlines) break;
    Pix* l = pix + k * ys;
    const int p0 = l[-xs], p1 = l[-2 * xs], q0 = l[0], q1 = l[xs];
    const int delta = clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
    if (!no_p) l[-xs] = (Pix)clip3(0, maxv, p0 + delta);
    if (!no_q) l[0] = (Pix)clip3(0, maxv, q0 - delta);
  }
}

// The same for a chroma edge segment of 4 lines: the 4 x 4 window (p1 p0 | q0 q1) moved with whole-dword accesses, p0 / q0 replaced in the packed
// words.  A vertical edge's window rows start 2 samples left of a multiple of 8: 2-byte aligned for 8-bit samples (the hardware takes unaligned dwords)
typedef uint32_t u32_a2 __attribute__((aligned(2)));
template <typename Pix, int DIR>
struct ChromaWindow {
  static constexpr int ES = (int)sizeof(Pix);
  uint32_t w[4 * ES];
  __device__ __forceinline__ void load(const Pix* pix, int stride)
  {
#pragma unroll
    for (int r = 0; r < 4; r++) {     // DIR 0: line r (a picture row); DIR 1: picture row y - 2 + r
      const u32_a2* src = (const u32_a2*)(DIR == 0 ? pix + (size_t)r * stride - 2 : pix + (ptrdiff_t)(r - 2) * stride);
#pragma unroll
      for (int i = 0; i < ES; i++) w[ES * r + i] = src[i];
    }
  }
};
template <typename Pix, int DIR>
__device__ __forceinline__ void deblock_chroma4_regs(ChromaWindow<Pix, DIR>& CW, int qp_p, int qp_q, int c_qp_pic_offset, int tc_off2, int bit_depth, int no_p, int no_q, bool not420)
{
  constexpr int ES = (int)sizeof(Pix);
  const int qpi = ((qp_q + qp_p + 1) >> 1) + c_qp_pic_offset;
  const int qpc = not420 ? (qpi < 51 ? qpi : 51) : (qpi < 30 ? qpi : (qpi >= 44 ? qpi - 6 : c_chroma_qp_f[qpi - 30]));   // 8.7.2.5.5
  const int tc = c_tc[clip3(0, 53, qpc + 2 + (tc_off2 << 1))] * (1 << (bit_depth - 8));
  const int maxv = (1 << bit_depth) - 1;
  uint32_t* w = CW.w;
  // word / shift of sample j (0 .. 3 = p1 p0 q0 q1) of line k
  auto word = [](int k, int j) { return DIR == 0 ? (ES == 1 ? k : 2 * k + (j >> 1)) : (ES == 1 ? j : 2 * j + (k >> 1)); };
  auto shift = [](int k, int j) { return DIR == 0 ? (ES == 1 ? 8 * j : 16 * (j & 1)) : (ES == 1 ? 8 * k : 16 * (k & 1)); };
  const uint32_t smask = ES == 1 ? 255u : 0xffffu;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int p1 = (int)((w[word(k, 0)] >> shift(k, 0)) & smask), p0 = (int)((w[word(k, 1)] >> shift(k, 1)) & smask);
    const int q0 = (int)((w[word(k, 2)] >> shift(k, 2)) & smask), q1 = (int)((w[word(k, 3)] >> shift(k, 3)) & smask);
    const int delta = clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
    if (!no_p) w[word(k, 1)] = (w[word(k, 1)] & ~(smask << shift(k, 1))) | ((uint32_t)clip3(0, maxv, p0 + delta) << shift(k, 1));
    if (!no_q) w[word(k, 2)] = (w[word(k, 2)] & ~(smask << shift(k, 2))) | ((uint32_t)clip3(0, maxv, q0 - delta) << shift(k, 2));
  }
}
template <typename Pix, int DIR>
__device__ __forceinline__ void deblock_chroma4(ChromaWindow<Pix, DIR>& CW, Pix* pix, int stride, int qp_p, int qp_q, int c_qp_pic_offset, int tc_off2, int bit_depth, int no_p, int no_q,
                                                bool not420)
{
  constexpr int ES = (int)sizeof(Pix);
  deblock_chroma4_regs<Pix, DIR>(CW, qp_p, qp_q, c_qp_pic_offset, tc_off2, bit_depth, no_p, no_q, not420);
  const uint32_t* w = CW.w;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (DIR == 1 && (r == 0 || r == 3)) continue;     // rows p1 / q1 are never modified
    u32_a2* dst = (u32_a2*)(DIR == 0 ? pix + (size_t)r * stride - 2 : pix + (ptrdiff_t)(r - 2) * stride);
#pragma unroll
    for (int i = 0; i < ES; i++) dst[i] = w[ES * r + i];
  }
}
template <typename Pix, int DIR>
__device__ __forceinline__ void deblock_chroma4(Pix* pix, int stride, int qp_p, int qp_q, int c_qp_pic_offset, int tc_off2, int bit_depth, int no_p, int no_q, bool not420)
{
  ChromaWindow<Pix, DIR> CW;
  CW.load(pix, stride);
  deblock_chroma4<Pix, DIR>(CW, pix, stride, qp_p, qp_q, c_qp_pic_offset, tc_off2, bit_depth, no_p, no_q, not420);
}

}  // namespace

// DIR 0: vertical edges (filtering across x), DIR 1: horizontal edges
// 8.7.2.4, the motion part: the two blocks use different reference PICTURES (RefFrame slots: not indices, not lists) or a different number of
// motion vectors, or vectors that point at the same picture differ by a luma sample or more
__device__ __forceinline__ bool mv_far(const int16_t* a, const int16_t* b) { return iabs(a[0] - b[0]) >= 4 || iabs(a[1] - b[1]) >= 4; }
__device__ __forceinline__ bool motion_differs(const MotionUnit& p, const MotionUnit& q)
{
  const int np = (p.ref_idx[0] >= 0) + (p.ref_idx[1] >= 0), nq = (q.ref_idx[0] >= 0) + (q.ref_idx[1] >= 0);
  if (np != nq) return true;
  const int sp0 = p.slot_pred[0] & 63, sp1 = p.slot_pred[1] & 63, sq0 = q.slot_pred[0] & 63, sq1 = q.slot_pred[1] & 63;
  if (np == 1) {
    const int lp = p.ref_idx[0] >= 0 ? 0 : 1, lq = q.ref_idx[0] >= 0 ? 0 : 1;
    if ((lp ? sp1 : sp0) != (lq ? sq1 : sq0)) return true;
    return mv_far(p.mv[lp], q.mv[lq]);
  }
  if (!((sp0 == sq0 && sp1 == sq1) || (sp0 == sq1 && sp1 == sq0))) return true;
  if (sp0 != sp1) {   // two different reference pictures: compare the vectors that point at the same one
    if (sp0 == sq0) return mv_far(p.mv[0], q.mv[0]) || mv_far(p.mv[1], q.mv[1]);
    return mv_far(p.mv[0], q.mv[1]) || mv_far(p.mv[1], q.mv[0]);
  }
  // both vectors of both blocks point at the same picture: either pairing may match
  return (mv_far(p.mv[0], q.mv[0]) || mv_far(p.mv[1], q.mv[1])) && (mv_far(p.mv[0], q.mv[1]) || mv_far(p.mv[1], q.mv[0]));
}

template <typename Pix, int DIR>
__global__ __launch_bounds__(256) void k_deblock(FilterArgs A)
{
  if (*A.status != 0) return;
  const PicParams& P = A.pics[blockIdx.y];
  const int uw = (P.width + 3) >> 2, uh = (P.height + 3) >> 2;
  // work item: (edge index along the filtered direction on the 8-sample grid, unit index along the edge)
  const int n_edges = DIR == 0 ? (P.width + 7) >> 3 : (P.height + 7) >> 3;
  const int n_along = DIR == 0 ? uh : uw;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_edges * n_along) return;
  int e, a;
  if (DIR == 0) { e = tid % n_edges; a = tid / n_edges; }  // consecutive lanes: consecutive x
  else { a = tid % n_along; e = tid / n_along; }            // consecutive lanes: consecutive x as well
  if (e == 0) return;
  const int x = DIR == 0 ? e * 8 : a * 4, y = DIR == 0 ? a * 4 : e * 8;
  if (x >= P.width || y >= P.height) return;
  const uint8_t* u_flags = A.arena + P.off_u_flags;
  const int8_t* u_qp = (const int8_t*)(A.arena + P.off_u_qp);
  int ctb_q, ctb_p;
  // Everything a segment reads is requested BEFORE the first decision (round 6: the kernel kept 0.12 memory instructions in flight per wave - flags, then
  // the neighbour's flags and QPs, then the slice parameters, then the samples, each behind the answer to the one before: four memory round trips per
  // segment; now one, with the slice parameters of slice 0 - the only slice of most pictures - requested speculatively beside the CTB's slice index).
  // Segments without an edge flag have read a window for nothing: the kernel is far from the HBM roof, it was waiting.
  const size_t iq = unit_index(P, x >> 2, y >> 2, &ctb_q);
  const size_t ip = DIR == 0 ? unit_index(P, (x >> 2) - 1, y >> 2, &ctb_p) : unit_index(P, x >> 2, (y >> 2) - 1, &ctb_p);
  const CtbInfo* ctb_info = (const CtbInfo*)(A.arena + P.off_ctb_info);
  const SliceParams* slices = (const SliceParams*)(A.arena + P.off_slices);
  const uint8_t fq = u_flags[iq], fp = u_flags[ip];
  const int qp_q = u_qp[iq], qp_p = u_qp[ip];
  const uint32_t slice_idx = ctb_info[ctb_q].slice_idx;
  int s_beta = slices[0].beta_offset_div2, s_tc = slices[0].tc_offset_div2, s_cb = slices[0].pps_cb_qp_offset, s_cr = slices[0].pps_cr_qp_offset;
  Pix* const rec_y = (Pix*)(A.arena + P.off_rec[0]);
  const int stride_y = P.rec_stride[0] / sizeof(Pix);
  Pix* const pix_y = rec_y + (size_t)y * stride_y + x;
  EdgeWindow<Pix, DIR> W;
  W.load(pix_y, stride_y);
  // 4:2:0 chroma edges on the 8x8 chroma grid; one 4-sample chroma segment spans 8 luma samples along the edge
  const bool c420_grid = P.chroma_format_idc == 1 && (DIR == 0 ? ((x & 15) == 0 && (y & 7) == 0) : ((y & 15) == 0 && (x & 7) == 0));
  const int stride_c = P.rec_stride[1] / sizeof(Pix);
  Pix* const pix_cb = (Pix*)(A.arena + P.off_rec[1]) + (size_t)(y >> 1) * stride_c + (x >> 1);
  Pix* const pix_cr = (Pix*)(A.arena + P.off_rec[2]) + (size_t)(y >> 1) * stride_c + (x >> 1);
  ChromaWindow<Pix, DIR> WB, WR;
  if (c420_grid) { WB.load(pix_cb, stride_c); WR.load(pix_cr, stride_c); }
  else {
#pragma unroll
    for (int i = 0; i < 4 * (int)sizeof(Pix); i++) { WB.w[i] = 0; WR.w[i] = 0; }
  }
#ifndef HIPDEC_HOST_EMU
  // (the loads above stay above the branches below: a value pinned here cannot be sunk into the block that uses it)
#pragma unroll
  for (int i = 0; i < EdgeWindow<Pix, DIR>::NW; i++) asm volatile("" : "+v"(W.w[i]));
#pragma unroll
  for (int i = 0; i < 4 * (int)sizeof(Pix); i++) asm volatile("" : "+v"(WB.w[i]), "+v"(WR.w[i]));
  asm volatile("" : "+v"(s_beta), "+v"(s_tc), "+v"(s_cb), "+v"(s_cr));
#endif
  if (!(fq & (DIR == 0 ? UF_VEDGE : UF_HEDGE))) return;
  // 8.7.2.5.7: samples of cu_transquant_bypass units, and of PCM units when pcm_loop_filter_disabled_flag = 1, are left unchanged
  const int keep = UF_BYPASS | (P.pcm_loop_filter_disabled ? UF_PCM : 0);
  const int no_q = (fq & keep) != 0, no_p = (fp & keep) != 0;
  if (slice_idx != 0) {
    s_beta = slices[slice_idx].beta_offset_div2; s_tc = slices[slice_idx].tc_offset_div2;
    s_cb = slices[slice_idx].pps_cb_qp_offset; s_cr = slices[slice_idx].pps_cr_qp_offset;
  }
  struct { int beta_offset_div2, tc_offset_div2, pps_cb_qp_offset, pps_cr_qp_offset; } sl = {s_beta, s_tc, s_cb, s_cr};
  int bs = 2;   // every edge of an intra picture
  if (P.is_inter) {
    // 8.7.2.4 in a P picture: 2 where a side is intra coded; else 1 at a transform block edge next to a block with luma coefficients, or where
    // the two sides predict from different pictures / with vectors a sample or more apart; else no filtering.  (Transform blocks are aligned to
    // their size, so the edge is a transform block edge iff the Q block starts there; prediction block edges were flagged by the parser.)
    const MotionUnit* mf = (const MotionUnit*)(A.arena + P.off_mf);
    const MotionUnit mq = mf[iq], mp = mf[ip];
    const bool inter_q = mq.ref_idx[0] >= 0 || mq.ref_idx[1] >= 0, inter_p = mp.ref_idx[0] >= 0 || mp.ref_idx[1] >= 0;
    if (inter_q && inter_p) {
      const int tbq = 1 << (A.arena[P.off_u_size + iq] & 15);
      const bool tu_edge = ((DIR == 0 ? x : y) & (tbq - 1)) == 0;
      if (tu_edge && ((fq | fp) & UF_CBF_LUMA)) bs = 1;
      else if (motion_differs(mp, mq)) bs = 1;
      else return;
    }
  }
  deblock_luma<Pix, DIR>(W, pix_y, stride_y, qp_p, qp_q, sl.beta_offset_div2, sl.tc_offset_div2, P.bit_depth_luma, no_p, no_q, bs);
  if (bs != 2) return;   // chroma edges are filtered where bS is 2 only (8.7.2.5)
  if (P.chroma_format_idc == 3) {
    // 4:4:4: the chroma planes have the luma planes' edges (the 8-sample chroma grid IS the luma grid) and take the chroma filter
    for (int c = 1; c < 3; c++) {
      Pix* rec = (Pix*)(A.arena + P.off_rec[c]);
      const int stride = P.rec_stride[c] / sizeof(Pix);
      Pix* pix = rec + (size_t)y * stride + x;
      const int off = c == 1 ? sl.pps_cb_qp_offset : sl.pps_cr_qp_offset;
      deblock_chroma4<Pix, DIR>(pix, stride, qp_p, qp_q, off, sl.tc_offset_div2, P.bit_depth_chroma, no_p, no_q, true);
    }
  }
  if (P.chroma_format_idc == 2) {
    // 4:2:2: the 8x8 chroma grid is 16 luma samples wide and 8 tall; the 4 luma rows of a vertical edge segment are 4 chroma rows, the 4 luma columns
    // of a horizontal one 2 chroma columns.  QpC = Min(qPi, 51) like 4:4:4
    if (DIR == 1 || (x & 15) == 0) {
      for (int c = 1; c < 3; c++) {
        Pix* rec = (Pix*)(A.arena + P.off_rec[c]);
        const int stride = P.rec_stride[c] / sizeof(Pix);
        Pix* pix = rec + (size_t)y * stride + (x >> 1);
        const int off = c == 1 ? sl.pps_cb_qp_offset : sl.pps_cr_qp_offset;
        if (DIR == 0) deblock_chroma4<Pix, 0>(pix, stride, qp_p, qp_q, off, sl.tc_offset_div2, P.bit_depth_chroma, no_p, no_q, true);
        else deblock_chroma<Pix>(pix, stride, 1, qp_p, qp_q, off, sl.tc_offset_div2, P.bit_depth_chroma, no_p, no_q, true, 2);
      }
    }
  }
  if (c420_grid) {
    deblock_chroma4<Pix, DIR>(WB, pix_cb, stride_c, qp_p, qp_q, sl.pps_cb_qp_offset, sl.tc_offset_div2, P.bit_depth_chroma, no_p, no_q, false);
    deblock_chroma4<Pix, DIR>(WR, pix_cr, stride_c, qp_p, qp_q, sl.pps_cr_qp_offset, sl.tc_offset_div2, P.bit_depth_chroma, no_p, no_q, false);
  }
}


// Tile of a workgroup, XCD-aware.  Workgroups go to the chip's 8 XCDs (each with an L2 of its own) round-robin by their linear id, and a tile's rows
// start one sample left of a 128-byte line and end one sample into the next: with tile = blockIdx.x the two neighbours of every tile ran on other
// XCDs and each of them fetched those two lines from HBM again (measured: 5.2 B/px fetched for 1.5 B/px of planes, profiles/pmc_traffic.json round 5:
// three lines per luma row and two per chroma row instead of one).  Here every XCD takes a contiguous band of the picture's tiles, so that
// horizontal neighbours run on the same XCD at about the same time and share the lines in its L2.  gridDim.x is a multiple of 8.
__device__ __forceinline__ int sao_tile_of_block(int n_tiles)
{
  const int bid = (int)blockIdx.x, chunk = (n_tiles + 7) >> 3;
  const int t = (bid & 7) * chunk + (bid >> 3);
  return ((bid >> 3) < chunk && t < n_tiles) ? t : -1;
}

// ---- both edge directions in ONE pass (intra pictures, 4:0:0 / 4:2:0) -----------------------------------------------------------------------
// The window of a vertical edge segment is [x - 4, x + 4) x 4 rows, that of a horizontal one 4 columns x [y - 4, y + 4), edges sit on the 8-sample
// grid: the 8 x 8 blocks centred on the grid's crossings - [8j - 4, 8j + 4) x [8k - 4, 8k + 4) - tile the plane, and each holds exactly the two segments
// of vertical edge 8j that cross it and the two segments of horizontal edge 8k, whose windows lie inside it.  The horizontal edges must see the
// vertically filtered samples (8.7.2: all vertical edges of the picture first) - of THIS block only, since no other vertical edge touches its
// columns.  So one thread takes one block: 8 rows x 8 samples into registers, the vertical edge's two segments, then the horizontal edge's two on
// the result, one store.  Every sample is read once and written once: 3 s B per luma pixel instead of the 6 s of the two-pass form (k_deblock<0>,
// k_deblock<1>, which stay for P / B pictures - their boundary strength needs the motion field - and for 4:2:2 / 4:4:4 chroma grids).
// CH = 0: the luma plane.  CH = 1: the Cb and Cr planes of a 4:2:0 picture (one thread filters the block of both; chroma edges lie on the 8-sample
// CHROMA grid = 16 luma samples, a 4-sample chroma segment takes the flags and QPs of the luma segment at its first row / column, as k_deblock does).
template <typename Pix, int CH>
__global__ __launch_bounds__(256) void k_deblock_fused(FilterArgs A)
{
  constexpr int ES = (int)sizeof(Pix), RW = 2 * ES;    // dwords per 8-sample row of the block
  if (*A.status != 0) return;
  const PicParams& P = A.pics[blockIdx.y];
  if (CH && P.chroma_format_idc != 1) return;
  const int Wc = CH ? P.cwidth : P.width, Hc = CH ? P.cheight : P.height;
  const int nbx = (Wc >> 3) + 1, nby = (Hc >> 3) + 1;                 // grid crossings 0, 8, ... <= W (the first / last blocks are half outside)
  // XCD-aware, like the SAO tiles: an XCD takes RUNS of consecutive workgroups (256 consecutive blocks each, about half a block row of a 4K picture), so
  // that the unit-map lines of a CTB - read again by each of the 8 block rows that cross it - and the plane lines two neighbouring workgroups share are
  // fetched into few L2s instead of all eight (with workgroup = blockIdx.x: 2.58 B/px fetched for 1.5 B/px of planes + 0.2 of maps)
#ifndef HIPDEC_DBK_XCD_RUN
#define HIPDEC_DBK_XCD_RUN 2   // N > 0: every XCD takes runs of N consecutive workgroups out of each group of 8 N; measurement builds: 0 = one contiguous band of the
                               // picture per XCD (fewest bytes, 3.13 B/px, but +0.5 ms: eight distant regions of HBM at a time), -1 = workgroup = blockIdx.x
#endif
  const int n_wg = (nbx * nby + 255) >> 8;
  int wg;
  if (HIPDEC_DBK_XCD_RUN == 0) wg = sao_tile_of_block(n_wg);
  else if (HIPDEC_DBK_XCD_RUN < 0) wg = (int)blockIdx.x < n_wg ? (int)blockIdx.x : -1;
  else {
    constexpr int R = HIPDEC_DBK_XCD_RUN > 0 ? HIPDEC_DBK_XCD_RUN : 1;
    const int bid = (int)blockIdx.x, g = bid / (8 * R), o = bid % (8 * R);
    wg = g * 8 * R + (o & 7) * R + (o >> 3);
    if (wg >= n_wg) wg = -1;
  }
  if (wg < 0) return;
  const int tid = wg * 256 + (int)threadIdx.x;
  if (tid >= nbx * nby) return;
  const int j = tid % nbx, k = tid / nbx;                            // consecutive lanes: consecutive x
  const int xc = j * 8, yc = k * 8;
  // what exists of the block: left / right half (columns), upper / lower half (rows)
  const bool hl = xc > 0, hr = xc < Wc, vu = yc > 0, vd = yc < Hc;
  // the luma units (4x4) the four segments take their flags and QPs from: Q and P unit of V segment 0 / 1 (upper / lower rows) and H segment 0 / 1
  // (left / right columns).  Luma: the 2 x 2 units around the crossing.  Chroma: the luma segment at the chroma segment's first row / column.
  const int cx = CH ? xc >> 1 : xc >> 2, cy = CH ? yc >> 1 : yc >> 2;   // unit right of / below the crossing
  const int d = CH ? 2 : 1;                                           // a chroma half block is two units
  const int vq_y[2] = {cy - d, cy}, hq_x[2] = {cx - d, cx};
  const uint8_t* u_flags = A.arena + P.off_u_flags;
  const int8_t* u_qp = (const int8_t*)(A.arena + P.off_u_qp);
  const CtbInfo* ctb_info = (const CtbInfo*)(A.arena + P.off_ctb_info);
  const SliceParams* slices = (const SliceParams*)(A.arena + P.off_slices);
  uint8_t fq[4] = {0, 0, 0, 0}, fp[4] = {0, 0, 0, 0};
  int qq[4] = {0, 0, 0, 0}, qp[4] = {0, 0, 0, 0};
  uint32_t sidx[4] = {0, 0, 0, 0};
  const bool seg_on[4] = {hl && hr && vu, hl && hr && vd, vu && vd && hl, vu && vd && hr};   // V0, V1, H0, H1 exist (an edge needs both of its sides)
#pragma unroll
  for (int sgm = 0; sgm < 4; sgm++) {
    if (!seg_on[sgm]) continue;
    const int qx = sgm < 2 ? cx : hq_x[sgm - 2], qy = sgm < 2 ? vq_y[sgm] : cy;
    const int px = sgm < 2 ? cx - 1 : qx, py = sgm < 2 ? qy : cy - 1;
    int ctb_q, ctb_p;
    const size_t iq = unit_index(P, qx, qy, &ctb_q), ip = unit_index(P, px, py, &ctb_p);
    fq[sgm] = u_flags[iq]; fp[sgm] = u_flags[ip]; qq[sgm] = u_qp[iq]; qp[sgm] = u_qp[ip];
    sidx[sgm] = ctb_info[ctb_q].slice_idx;
  }
  // the slice parameters of slice 0 (most pictures have one slice) travel with the first round of loads; other slices are read when they turn up
  const int s0_beta = slices[0].beta_offset_div2, s0_tc = slices[0].tc_offset_div2, s0_cb = slices[0].pps_cb_qp_offset, s0_cr = slices[0].pps_cr_qp_offset;
  // ---- the block(s): row r = picture row yc - 4 + r, dwords [0, ES) the left half, [ES, 2 ES) the right half
  constexpr int NP = CH ? 2 : 1;
  struct __attribute__((packed, aligned(4))) RowT { uint32_t d[RW]; };   // a block row: 8 samples at a 4-sample-aligned address
  const bool interior = hl && hr && vu && vd;
  uint32_t w[NP][8][RW];
  Pix* base[NP];
  int stride;
  {
    stride = (int)(P.rec_stride[CH ? 1 : 0] / sizeof(Pix));
#pragma unroll
    for (int pl = 0; pl < NP; pl++) {
      base[pl] = (Pix*)(A.arena + P.off_rec[CH ? 1 + pl : 0]) + (ptrdiff_t)(yc - 4) * stride + (xc - 4);
      if (interior) {   // all of the block exists (everywhere but at the picture's borders): whole rows, no per-dword predicates
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const RowT row = *(const RowT*)(base[pl] + (ptrdiff_t)r * stride);
#pragma unroll
          for (int i = 0; i < RW; i++) w[pl][r][i] = row.d[i];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const bool row_on = r < 4 ? vu : vd;
          const uint32_t* src = (const uint32_t*)(base[pl] + (ptrdiff_t)r * stride);
#pragma unroll
          for (int i = 0; i < RW; i++) {
            w[pl][r][i] = 0;
            if (row_on && (i < ES ? hl : hr)) w[pl][r][i] = src[i];
          }
        }
      }
    }
  }
  const int keep = UF_BYPASS | (P.pcm_loop_filter_disabled ? UF_PCM : 0);   // 8.7.2.5.7: such units stay as they are
  const int bd = CH ? P.bit_depth_chroma : P.bit_depth_luma;
  bool dirty = false;
  // ---- vertical edge xc: segments of rows 0..3 and 4..7
#pragma unroll
  for (int sgm = 0; sgm < 2; sgm++) {
    if (!seg_on[sgm] || !(fq[sgm] & UF_VEDGE)) continue;
    int beta = s0_beta, tc = s0_tc, ocb = s0_cb, ocr = s0_cr;
    if (sidx[sgm] != 0) { const SliceParams& sl = slices[sidx[sgm]]; beta = sl.beta_offset_div2; tc = sl.tc_offset_div2; ocb = sl.pps_cb_qp_offset; ocr = sl.pps_cr_qp_offset; }
    const int no_q = (fq[sgm] & keep) != 0, no_p = (fp[sgm] & keep) != 0;
    if constexpr (!CH && ES == 1) {
      if (!(no_p | no_q)) {   // (almost always: no PCM / bypass unit on either side) two lines per operation
        if (deblock_luma_pk8<0>(w[0], sgm, qp[sgm], qq[sgm], beta, tc)) dirty = true;
        continue;
      }
    }
    if (!CH) {
      EdgeWindow<Pix, 0> W;
#pragma unroll
      for (int l = 0; l < 4; l++)
#pragma unroll
        for (int i = 0; i < RW; i++) W.w[RW * l + i] = w[0][4 * sgm + l][i];
      if (deblock_luma_regs<Pix, 0>(W, qp[sgm], qq[sgm], beta, tc, bd, no_p, no_q)) {
        dirty = true;
#pragma unroll
        for (int l = 0; l < 4; l++)
#pragma unroll
          for (int i = 0; i < RW; i++) w[0][4 * sgm + l][i] = W.w[RW * l + i];
      }
    } else {
      // the chroma window of a vertical edge: samples xc - 2 .. xc + 1 of 4 rows = the upper half word of the left dword(s) and the lower of the right
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        ChromaWindow<Pix, 0> CW;
#pragma unroll
        for (int l = 0; l < 4; l++) {
          if (ES == 1) CW.w[l] = (w[pl][4 * sgm + l][0] >> 16) | (w[pl][4 * sgm + l][1] << 16);
          else { CW.w[2 * l] = w[pl][4 * sgm + l][1]; CW.w[2 * l + 1] = w[pl][4 * sgm + l][2]; }
        }
        deblock_chroma4_regs<Pix, 0>(CW, qp[sgm], qq[sgm], pl ? ocr : ocb, tc, bd, no_p, no_q, false);
#pragma unroll
        for (int l = 0; l < 4; l++) {
          if (ES == 1) {
            w[pl][4 * sgm + l][0] = (w[pl][4 * sgm + l][0] & 0xffffu) | (CW.w[l] << 16);
            w[pl][4 * sgm + l][1] = (w[pl][4 * sgm + l][1] & 0xffff0000u) | (CW.w[l] >> 16);
          } else { w[pl][4 * sgm + l][1] = CW.w[2 * l]; w[pl][4 * sgm + l][2] = CW.w[2 * l + 1]; }
        }
      }
      dirty = true;
    }
  }
  // ---- horizontal edge yc: segments of columns 0..3 and 4..7, on the vertically filtered block
#pragma unroll
  for (int sgm = 0; sgm < 2; sgm++) {
    if (!seg_on[2 + sgm] || !(fq[2 + sgm] & UF_HEDGE)) continue;
    int beta = s0_beta, tc = s0_tc, ocb = s0_cb, ocr = s0_cr;
    if (sidx[2 + sgm] != 0) { const SliceParams& sl = slices[sidx[2 + sgm]]; beta = sl.beta_offset_div2; tc = sl.tc_offset_div2; ocb = sl.pps_cb_qp_offset; ocr = sl.pps_cr_qp_offset; }
    const int no_q = (fq[2 + sgm] & keep) != 0, no_p = (fp[2 + sgm] & keep) != 0;
    if constexpr (!CH && ES == 1) {
      if (!(no_p | no_q)) {
        if (deblock_luma_pk8<1>(w[0], sgm, qp[2 + sgm], qq[2 + sgm], beta, tc)) dirty = true;
        continue;
      }
    }
    if (!CH) {
      EdgeWindow<Pix, 1> W;
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < ES; i++) W.w[ES * r + i] = w[0][r][ES * sgm + i];
      if (deblock_luma_regs<Pix, 1>(W, qp[2 + sgm], qq[2 + sgm], beta, tc, bd, no_p, no_q)) {
        dirty = true;
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
          for (int i = 0; i < ES; i++) w[0][r][ES * sgm + i] = W.w[ES * r + i];
      }
    } else {
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        ChromaWindow<Pix, 1> CW;   // picture rows yc - 2 .. yc + 1 = block rows 2 .. 5, the segment's 4 columns
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int i = 0; i < ES; i++) CW.w[ES * r + i] = w[pl][2 + r][ES * sgm + i];
        deblock_chroma4_regs<Pix, 1>(CW, qp[2 + sgm], qq[2 + sgm], pl ? ocr : ocb, tc, bd, no_p, no_q, false);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int i = 0; i < ES; i++) w[pl][2 + r][ES * sgm + i] = CW.w[ES * r + i];
      }
      dirty = true;
    }
  }
  if (!dirty) return;
  // ---- store (rows / halves that exist; the outermost luma rows and columns of a block are never modified but travel with their dword)
#pragma unroll
  for (int pl = 0; pl < NP; pl++) {
    if (interior) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        RowT row;
#pragma unroll
        for (int i = 0; i < RW; i++) row.d[i] = w[pl][r][i];
        *(RowT*)(base[pl] + (ptrdiff_t)r * stride) = row;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const bool row_on = r < 4 ? vu : vd;
        uint32_t* dst = (uint32_t*)(base[pl] + (ptrdiff_t)r * stride);
#pragma unroll
        for (int i = 0; i < RW; i++) if (row_on && (i < ES ? hl : hr)) dst[i] = w[pl][r][i];
      }
    }
  }
}

// SAO + conformance crop.  One 256-thread workgroup per 128x16 tile of OUTPUT samples of one component (blockIdx.z) of one
// picture (blockIdx.y): the deblocked source tile plus a one-sample halo is staged in LDS with coalesced dword loads issued
// back to back (memory-level parallelism instead of a dependent load chain per thread), then every thread classifies and
// offsets 4 samples of SAO_RPT rows out of LDS and stores one dword per row.
// a CTB's parameter set in registers (never a struct in memory: dynamic indexing would push it to scratch)
struct SaoRegs { int type, cls, o0, o1, o2, o3; };
__device__ __forceinline__ SaoRegs sao_unpack(uint32_t w0, uint32_t w1, uint32_t w2)
{
  SaoRegs r;
  r.type = (int)(w0 & 255u); r.cls = (int)((w0 >> 8) & 255u);
  r.o0 = (int)(int16_t)(w0 >> 16); r.o1 = (int)(int16_t)(w1 & 0xffffu); r.o2 = (int)(int16_t)(w1 >> 16); r.o3 = (int)(int16_t)(w2 & 0xffffu);
  return r;
}
__device__ __forceinline__ int sao_off(const SaoRegs& sp, int i) { return i == 0 ? sp.o0 : (i == 1 ? sp.o1 : (i == 2 ? sp.o2 : sp.o3)); }
// offset j (0 .. 3) out of four offsets packed as signed bytes: one v_bfe_i32
__device__ __forceinline__ int sao_packed_offset(uint32_t packed, int j) { return (int)(int8_t)(packed >> ((j & 3) * 8)); }

constexpr int SAO_TW = 128, SAO_TH = 32, SAO_RPT = SAO_TH / 8;   // rows per thread

// Workgroup barrier for data that travels through LDS only.  __syncthreads() also drains the wave's GLOBAL memory counter (s_waitcnt vmcnt(0)): in
// the SAO kernels that made every phase wait for the plane stores of the phase before it (nothing in these kernels reads global memory another
// thread of the launch has written; a value loaded for the tile is waited for where it is stored to LDS, by the compiler).
__device__ __forceinline__ void lds_barrier()
{
#ifndef HIPDEC_HOST_EMU
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  __syncthreads();
#endif
}

// What one component of one picture needs for SAO, gathered once per workgroup (wave-uniform)
template <typename Pix>
struct SaoComp {
  const PicParams* P;
  int c, sub, suby, ow, oh, W, H, bit_depth, maxv, rs_bytes, os, lctb, lctby, crop_xc, crop_yc, ctb_w;   // sub / lctb: horizontal, suby / lctby: vertical
  const uint8_t* rec;
  Pix* out;
  const uint8_t* u_flags;
  const SaoParams* sao;
  const CtbInfo* ctb_info;
  const SliceParams* slices;
  bool check_bypass, lf_across_tiles, free_nb;
};
template <typename Pix>
__device__ __forceinline__ SaoComp<Pix> sao_comp(const FilterArgs& A, const PicParams& P, int c, bool may_keep, bool restricted)
{
  SaoComp<Pix> S;
  S.P = &P; S.c = c; S.sub = (c && P.chroma_format_idc != 3) ? 2 : 1; S.suby = (c && P.chroma_format_idc == 1) ? 2 : 1;
  S.ow = c ? P.out_cwidth : P.out_width; S.oh = c ? P.out_cheight : P.out_height;
  S.W = c ? P.cwidth : P.width; S.H = c ? P.cheight : P.height;
  S.bit_depth = c ? P.bit_depth_chroma : P.bit_depth_luma; S.maxv = (1 << S.bit_depth) - 1;
  S.rec = A.arena + P.off_rec[c]; S.rs_bytes = (int)P.rec_stride[c];
  S.out = (Pix*)(A.arena + P.off_out[c]); S.os = P.out_stride[c] / (int)sizeof(Pix);
  S.u_flags = A.arena + P.off_u_flags;
  S.sao = (const SaoParams*)(A.arena + P.off_sao);
  S.ctb_info = (const CtbInfo*)(A.arena + P.off_ctb_info);
  S.slices = (const SliceParams*)(A.arena + P.off_slices);
  S.lctb = P.log2_ctb - (S.sub == 2 ? 1 : 0); S.lctby = P.log2_ctb - (S.suby == 2 ? 1 : 0);   // log2 CTB width / height in component samples
  S.crop_xc = P.crop_x / S.sub; S.crop_yc = P.crop_y / S.suby; S.ctb_w = P.ctb_w;
  // may_keep: compile-time false in the kernel variant for batches without lossless CUs / unfiltered PCM (the common case): the per-sample
  // unit look-ups and the paths behind them leave the kernel, which is larger than the instruction cache
  S.check_bypass = may_keep && (P.transquant_bypass_enabled != 0 || (P.pcm_enabled && P.pcm_loop_filter_disabled));
  S.lf_across_tiles = P.lf_across_tiles != 0;
  // restricted: compile-time false in the variant for batches whose pictures all have sao_free_neighbours (one slice or filtering across slices
  // / tiles allowed): the per-neighbour slice / tile checks leave the kernel
  S.free_nb = restricted ? P.sao_free_neighbours != 0 : true;
  return S;
}
// the SAO parameters (three dwords) of the CTB that holds output sample (ox, oy) — requested before the tile loads so that they travel with them
template <typename Pix>
__device__ __forceinline__ void sao_params_at(const SaoComp<Pix>& S, int ox, int oy, uint32_t spw[3])
{
  int y = oy + S.crop_yc, x = ox + S.crop_xc;
  y = y < S.H ? y : S.H - 1; x = x < S.W ? x : S.W - 1;
  const uint32_t* src = (const uint32_t*)&S.sao[(size_t)((y >> S.lctby) * S.ctb_w + (x >> S.lctb)) * 3 + S.c];
  spw[0] = src[0]; spw[1] = src[1]; spw[2] = src[2];
}
// stages source rows ys0-1 .. ys0+TH, bytes [ab, ...) of each row, into tile[TH + 2][ROW_WORDS]; returns ab.  All threads of the workgroup
// (NT of them) take part; the caller synchronises.
template <typename Pix, int TH, int ROW_WORDS, int NT>
__device__ __forceinline__ int sao_stage(const SaoComp<Pix>& S, uint32_t* tile, int ox_t, int oy_t, int tid)
{
  constexpr int ES = (int)sizeof(Pix);
  const int xs0 = ox_t + S.crop_xc, ys0 = oy_t + S.crop_yc;
  int ab = (xs0 - 1) * ES;
  ab = ab < 0 ? 0 : (ab & ~3);
  for (int i = tid; i < (TH + 2) * ROW_WORDS; i += NT) {
    const int r = i / ROW_WORDS, wi = i - r * ROW_WORDS;
    const int ys = ys0 - 1 + r, bo = ab + wi * 4;
    uint32_t v = 0;
    if (ys >= 0 && ys < S.H && bo < S.rs_bytes) v = *(const uint32_t*)(S.rec + (size_t)ys * S.rs_bytes + bo);
    tile[i] = v;
  }
  return ab;
}
// The same in two halves, so that a kernel can request SEVERAL tiles (luma, Cb, Cr) before it waits for any of them: sao_stage_load issues the loads into
// registers (NL = words per thread), sao_stage_store puts them into LDS.  One memory round trip for all three components instead of one per component
// (round 6: k_sao_rgb kept 0.06 memory instructions in flight per wave, profiles/r05_wait_breakdown_b512.txt).
template <typename Pix, int TH, int ROW_WORDS, int NT, int NL>
__device__ __forceinline__ int sao_stage_load(const SaoComp<Pix>& S, int ox_t, int oy_t, int tid, uint32_t v[NL])
{
  constexpr int ES = (int)sizeof(Pix);
  static_assert(NL * NT >= (TH + 2) * ROW_WORDS, "words per thread");
  const int xs0 = ox_t + S.crop_xc, ys0 = oy_t + S.crop_yc;
  int ab = (xs0 - 1) * ES;
  ab = ab < 0 ? 0 : (ab & ~3);
#pragma unroll
  for (int k = 0; k < NL; k++) {
    const int i = tid + k * NT;
    const int r = i / ROW_WORDS, wi = i - r * ROW_WORDS;
    const int ys = ys0 - 1 + r, bo = ab + wi * 4;
    v[k] = 0;
    if (i < (TH + 2) * ROW_WORDS && ys >= 0 && ys < S.H && bo < S.rs_bytes) v[k] = *(const uint32_t*)(S.rec + (size_t)ys * S.rs_bytes + bo);
  }
  return ab;
}
template <int TH, int ROW_WORDS, int NT, int NL>
__device__ __forceinline__ void sao_stage_store(uint32_t* tile, int tid, const uint32_t v[NL])
{
#pragma unroll
  for (int k = 0; k < NL; k++) { const int i = tid + k * NT; if (i < (TH + 2) * ROW_WORDS) tile[i] = v[k]; }
}
// SAO of the (up to) 4 output samples (ox0 .. ox0+3, oy) out of the staged tile (tile row `lr` holds source row oy + crop); returns the
// number of valid samples (0: outside the output)
template <typename Pix, int ROW_WORDS>
__device__ __forceinline__ int sao_quad(const SaoComp<Pix>& S, const uint32_t* tile, int ab, int ox0, int oy, int lr, const uint32_t spw[3], Pix res[4])
{
  constexpr int ES = (int)sizeof(Pix);
  if (oy >= S.oh || ox0 >= S.ow) return 0;
  const int W = S.W, H = S.H, lctb = S.lctb, lctby = S.lctby, ctb_w = S.ctb_w, bit_depth = S.bit_depth, maxv = S.maxv, sub = S.sub, suby = S.suby, c = S.c;
  const int y = oy + S.crop_yc;
  const int npx = S.ow - ox0 < 4 ? S.ow - ox0 : 4;
  const int xf = ox0 + S.crop_xc, xl = xf + npx - 1;
  const int ctb_first = (y >> lctby) * ctb_w + (xf >> lctb);
  const bool one_ctb = (xf >> lctb) == (xl >> lctb);
  const SaoRegs sp_first = sao_unpack(spw[0], spw[1], spw[2]);
#define SAO_AT(row, x) (((const Pix*)((const uint8_t*)(tile + (row) * ROW_WORDS) + ((x) * ES - ab)))[0])
  // fast paths: the 4 samples share one CTB (one parameter set) and no per-sample lossless check is needed
  const int cmask = (1 << lctb) - 1, cmasky = (1 << lctby) - 1;
  // edge offsets without per-neighbour checks: no neighbour leaves the CTB - or nothing restricts neighbours in other
  // CTBs (one slice or filtering across slices / tiles allowed, no lossless CUs) - and none leaves the picture
  const bool interior = (S.free_nb ? (xf > 0 && y > 0) : ((xf & cmask) > 0 && (xl & cmask) < cmask && (y & cmasky) > 0 && (y & cmasky) < cmasky)) &&
                        xl + 1 < W && y + 1 < H;
  bool done = false;
  if (one_ctb && !S.check_bypass && npx == 4) {
    const SaoRegs sp = sp_first;
    // The two filtering paths below are pure lane arithmetic: signs through clamps, the offset of a band / an edge category through a bit-field
    // extract out of the four offsets packed into one word - compare-and-select chains here compile to exec-mask bookkeeping on the scalar pipe,
    // which is what this kernel is short of (1.1 scalar instructions per pixel, 56 % of the issue peak: profiles/pmc_issue.json).  SaoOffsetVal
    // fits a byte for every bit depth this path takes (|offset| <= 31 << (bitDepth - 10) = 124 at 12 bits).
    const uint32_t packed = ((uint32_t)sp.o0 & 255u) | (((uint32_t)sp.o1 & 255u) << 8) | (((uint32_t)sp.o2 & 255u) << 16) | (((uint32_t)sp.o3 & 255u) << 24);
    if (sp.type == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) res[i] = SAO_AT(lr, xf + i);
      done = true;
    } else if (sp.type == 1 && bit_depth <= 12) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int v = SAO_AT(lr, xf + i);
        const int k = ((v >> (bit_depth - 5)) - sp.cls) & 31;                  // band position relative to sao_band_position: the first four carry offsets
        int off = sao_packed_offset(packed, k & 3);
        off = k < 4 ? off : 0;
        res[i] = (Pix)clip3(0, maxv, v + off);
      }
      done = true;
    } else if (sp.type == 2 && interior && bit_depth <= 12) {
      const int cls = sp.cls;
      const int hx = cls == 1 ? 0 : (cls == 3 ? 1 : -1), hy = cls == 0 ? 0 : -1;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int x = xf + i;
        const int v = SAO_AT(lr, x), a = SAO_AT(lr + hy, x + hx), b = SAO_AT(lr - hy, x - hx);
        const int t = 2 + clip3(-1, 1, v - a) + clip3(-1, 1, v - b);          // Sign(v - a) + Sign(v - b) + 2: 0 .. 4; edgeIdx = {1, 2, 0, 3, 4}[t] (8.7.3.2)
        int off = sao_packed_offset(packed, t - ((t + 1) >> 2));               // t = 0, 1 -> offsets 0, 1; t = 3, 4 -> offsets 2, 3
        off = t == 2 ? 0 : off;
        res[i] = (Pix)clip3(0, maxv, v + off);
      }
      done = true;
    }
  }
  if (!done)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (i >= npx) { res[i] = 0; continue; }
    const int x = xf + i;
    int v = SAO_AT(lr, x);
    const int ctb = one_ctb ? ctb_first : (y >> lctby) * ctb_w + (x >> lctb);
    SaoRegs sp = sp_first;
    if (!one_ctb) { const uint32_t* q = (const uint32_t*)&S.sao[(size_t)ctb * 3 + c]; sp = sao_unpack(q[0], q[1], q[2]); }
    if (sp.type) {
      int ctb_dummy;
      bool keep = false;   // 8.7.3: samples of cu_transquant_bypass units, and of PCM units with pcm_loop_filter_disabled_flag, stay as they are
      if (S.check_bypass) {
        const uint8_t fl = S.u_flags[unit_index(*S.P, (x * sub) >> 2, (y * suby) >> 2, &ctb_dummy)];
        keep = (fl & (UF_BYPASS | (S.P->pcm_loop_filter_disabled ? UF_PCM : 0))) != 0;
      }
      if (!keep) {
        if (sp.type == 1) {
          const int k = ((v >> (bit_depth - 5)) - sp.cls) & 31;
          if (k < 4) v = clip3(0, maxv, v + sao_off(sp, k));
        } else {
          const int cls = sp.cls;
          const int hx = cls == 1 ? 0 : (cls == 3 ? 1 : -1), hy = cls == 0 ? 0 : -1;  // first neighbour; second is the mirror
          int edge_idx = 2, skip = 0;
          for (int k = 0; k < 2; k++) {
            const int dx = k ? -hx : hx, dy = k ? -hy : hy;
            const int xs = x + dx, ys = y + dy;
            if (xs < 0 || ys < 0 || xs >= W || ys >= H) { skip = 1; break; }
            const int ctb_n = (ys >> lctby) * ctb_w + (xs >> lctb);
            if (ctb_n != ctb) {
              const CtbInfo cn = S.ctb_info[ctb_n], cc = S.ctb_info[ctb];
              if (cn.slice_idx != cc.slice_idx) {
                if (cn.slice_idx < cc.slice_idx && !S.slices[cc.slice_idx].lf_across_slices) { skip = 1; break; }
                if (cn.slice_idx > cc.slice_idx && !S.slices[cn.slice_idx].lf_across_slices) { skip = 1; break; }
              }
              if (!S.lf_across_tiles && cn.tile_id != cc.tile_id) { skip = 1; break; }
            }
            const int nv = SAO_AT(lr + dy, xs);
            edge_idx += (v > nv) - (v < nv);
          }
          if (!skip) {
            if (edge_idx <= 2) edge_idx = edge_idx == 2 ? 0 : edge_idx + 1;
            if (edge_idx) v = clip3(0, maxv, v + sao_off(sp, edge_idx - 1));
          }
        }
      }
    }
    res[i] = (Pix)v;
  }
#undef SAO_AT
  return npx;
}
template <typename Pix>
__device__ __forceinline__ void sao_store(const SaoComp<Pix>& S, int ox0, int oy, int npx, const Pix res[4])
{
  constexpr int ES = (int)sizeof(Pix);
  Pix* o = S.out + (size_t)oy * S.os + ox0;
  if (npx == 4 && ES == 1) *(uint32_t*)o = res[0] | (res[1] << 8) | (res[2] << 16) | ((uint32_t)res[3] << 24);
  else if (npx == 4 && ES == 2) *(uint2*)o = make_uint2(res[0] | ((uint32_t)res[1] << 16), res[2] | ((uint32_t)res[3] << 16));
  else {
#pragma unroll
    for (int i = 0; i < 4; i++) if (i < npx) o[i] = res[i];
  }
}

template <typename Pix, bool MAY_KEEP, bool RESTRICTED>
__global__ __launch_bounds__(256) void k_sao(FilterArgs A)
{
  constexpr int ES = (int)sizeof(Pix);
  constexpr int ROW_WORDS = ((SAO_TW + 2) * ES + 3) / 4 + 2;
  __shared__ uint32_t tile[(SAO_TH + 2) * ROW_WORDS];
  if (*A.status != 0) return;
  const PicParams& P = A.pics[blockIdx.y];
  const int c = blockIdx.z;
  if (c > 0 && !P.chroma_format_idc) return;
  const SaoComp<Pix> S = sao_comp<Pix>(A, P, c, MAY_KEEP, RESTRICTED);
  const int tiles_x = (S.ow + SAO_TW - 1) / SAO_TW, tiles_y = (S.oh + SAO_TH - 1) / SAO_TH;
  const int tile_idx = sao_tile_of_block(tiles_x * tiles_y);
  if (tile_idx < 0) return;
  const int ox_t = (tile_idx % tiles_x) * SAO_TW, oy_t = (tile_idx / tiles_x) * SAO_TH;
  const int tid = threadIdx.x;
  const int tx = (tid & 31) * 4, ty = tid >> 5;
  uint32_t spw[SAO_RPT][3];   // SaoParams as three dwords per row (statically indexed: stays in registers)
#pragma unroll
  for (int rr = 0; rr < SAO_RPT; rr++) sao_params_at(S, ox_t + tx, oy_t + ty + rr * 8, spw[rr]);
  const int ab = sao_stage<Pix, SAO_TH, ROW_WORDS, 256>(S, tile, ox_t, oy_t, tid);
  lds_barrier();
#pragma unroll
  for (int rr = 0; rr < SAO_RPT; rr++) {
    const int oy = oy_t + ty + rr * 8, ox0 = ox_t + tx;
    Pix res[4];
    const int npx = sao_quad<Pix, ROW_WORDS>(S, tile, ab, ox0, oy, ty + rr * 8 + 1, spw[rr], res);
    if (npx) sao_store(S, ox0, oy, npx, res);
  }
}

// SAO of all three components of a 128x32 luma tile (8-bit 4:2:0) with the colour stage fused into the store path: the workgroup filters the
// luma tile (results stay in registers), then the 64x16 Cb and Cr tiles (results also go to LDS), writes the three output planes — the
// plugin ABI hands those to libheif — and emits the tile's interleaved RGB24 from registers + LDS.  Against k_sao + k_ycbcr_to_rgb_batch
// this saves the colour pass's 1.5 B/px re-read of the planes and one launch.  Same arithmetic: colordev::convert_px.
constexpr int SAO_CW = SAO_TW / 2, SAO_CH = SAO_TH / 2;
#ifndef HIPDEC_SAO_TPW
#define HIPDEC_SAO_TPW 4
#endif
constexpr int SAO_TPW = HIPDEC_SAO_TPW;   // tiles per workgroup of k_sao_rgb / k_sao_rgb_lean (measurement builds: -DHIPDEC_SAO_TPW=8)
#if !defined(HIPDEC_HOST_EMU) && defined(HIPDEC_SAO_RGB_OCC7)
#define SAO_RGB_OCCUPANCY __attribute__((amdgpu_waves_per_eu(7, 8)))   // <= 72 VGPRs (measurement build: costs 12 - 40 B of scratch per lane)
#else
#define SAO_RGB_OCCUPANCY
#endif
__host__ __device__ inline bool sao_rgb_pic_is_lean(const PicParams& P, const colordev::ColorParams& cp);   // (below, with k_sao_rgb_lean)
// skip_lean: the launch is paired with k_sao_rgb_lean, which takes the pictures that qualify for it
template <bool MAY_KEEP, bool RESTRICTED>
__global__ __launch_bounds__(256) SAO_RGB_OCCUPANCY void k_sao_rgb(FilterArgs A, const colordev::ColorParams* __restrict__ cps, int skip_lean)
{
  typedef uint8_t Pix;
  constexpr int ROW_WORDS = ((SAO_TW + 2) + 3) / 4 + 2;
  constexpr int CROW_WORDS = ((SAO_CW + 2) + 3) / 4 + 2;
  __shared__ uint32_t tile[(SAO_TH + 2) * ROW_WORDS];
  __shared__ uint32_t tile_c[2][(SAO_CH + 2) * CROW_WORDS];
  __shared__ __attribute__((aligned(4))) uint8_t chroma_s[2][SAO_CH][SAO_CW];   // (read back two samples at a time)
  if (*A.status != 0) return;
  const PicParams& P = A.pics[blockIdx.y];
  if (skip_lean && sao_rgb_pic_is_lean(P, cps[blockIdx.y])) return;
  const SaoComp<Pix> SY = sao_comp<Pix>(A, P, 0, MAY_KEEP, RESTRICTED);
  const int tiles_x = (SY.ow + SAO_TW - 1) / SAO_TW, tiles_y = (SY.oh + SAO_TH - 1) / SAO_TH;
  const int n_tiles = tiles_x * tiles_y;
  const int group = sao_tile_of_block((n_tiles + SAO_TPW - 1) / SAO_TPW);   // SAO_TPW consecutive tiles per workgroup (a quarter of the workgroups to dispatch)
  if (group < 0) return;
  const int tid = threadIdx.x;
  for (int tile_idx = group * SAO_TPW; tile_idx < (group + 1) * SAO_TPW && tile_idx < n_tiles; tile_idx++) {
  const int ox_t = (tile_idx % tiles_x) * SAO_TW, oy_t = (tile_idx / tiles_x) * SAO_TH;
  // A wave stays inside ONE CTB column (64 luma samples wide, CTB 64 and an aligned crop: the benchmarked streams): its lanes then share the CTB's SAO
  // parameters, so that of the three paths of sao_quad (off / band / edge) a wave runs the one its CTB takes - with 128 samples per wave row every wave
  // straddled two CTBs and ran whatever both of them needed.  Wave w: luma columns (w & 1) * 64 .., rows (w >> 1) * 16 + lane / 16 + 4 rr; chroma alike.
  const int wv = tid >> 6, ln = tid & 63;
  const int tx = (wv & 1) * 64 + (ln & 15) * 4, ty = (wv >> 1) * 16 + (ln >> 4);
  const int ctx = (wv & 1) * 32 + (ln & 7) * 4, cty = (wv >> 1) * 8 + (ln >> 3);
  constexpr int RSTEP = 4;   // rows between a thread's luma quads
  const SaoComp<Pix> SB = sao_comp<Pix>(A, P, 1, MAY_KEEP, RESTRICTED), SR = sao_comp<Pix>(A, P, 2, MAY_KEEP, RESTRICTED);
  // ---- everything the tile needs from global memory is requested first: the SAO parameters of the thread's rows and the three source tiles
  constexpr int NL_Y = ((SAO_TH + 2) * ROW_WORDS + 255) / 256, NL_C = ((SAO_CH + 2) * CROW_WORDS + 255) / 256;
  uint32_t spw[SAO_RPT][3], spw_b[3], spw_r[3];
#pragma unroll
  for (int rr = 0; rr < SAO_RPT; rr++) sao_params_at(SY, ox_t + tx, oy_t + ty + rr * RSTEP, spw[rr]);
  sao_params_at(SB, ox_t / 2 + ctx, oy_t / 2 + cty, spw_b);
  sao_params_at(SR, ox_t / 2 + ctx, oy_t / 2 + cty, spw_r);
  uint32_t vy[NL_Y], vb[NL_C], vr[NL_C];
  const int ab = sao_stage_load<Pix, SAO_TH, ROW_WORDS, 256, NL_Y>(SY, ox_t, oy_t, tid, vy);
  const int ab_b = sao_stage_load<Pix, SAO_CH, CROW_WORDS, 256, NL_C>(SB, ox_t / 2, oy_t / 2, tid, vb);
  const int ab_r = sao_stage_load<Pix, SAO_CH, CROW_WORDS, 256, NL_C>(SR, ox_t / 2, oy_t / 2, tid, vr);
  sao_stage_store<SAO_TH, ROW_WORDS, 256, NL_Y>(tile, tid, vy);
  sao_stage_store<SAO_CH, CROW_WORDS, 256, NL_C>(tile_c[0], tid, vb);
  sao_stage_store<SAO_CH, CROW_WORDS, 256, NL_C>(tile_c[1], tid, vr);
  lds_barrier();
  // ---- luma
  Pix yres[SAO_RPT][4];
  int ynpx[SAO_RPT];
#pragma unroll
  for (int rr = 0; rr < SAO_RPT; rr++) {
    const int oy = oy_t + ty + rr * RSTEP, ox0 = ox_t + tx;
    ynpx[rr] = sao_quad<Pix, ROW_WORDS>(SY, tile, ab, ox0, oy, ty + rr * RSTEP + 1, spw[rr], yres[rr]);
    if (ynpx[rr]) sao_store(SY, ox0, oy, ynpx[rr], yres[rr]);
  }
  // ---- Cb, Cr: 64 x 16 samples each, one row of 4 samples per thread
#pragma unroll
  for (int c = 1; c < 3; c++) {
    const SaoComp<Pix>& SC = c == 1 ? SB : SR;
    Pix res[4];
    const int ox0 = ox_t / 2 + ctx, oy = oy_t / 2 + cty;
    const int npx = sao_quad<Pix, CROW_WORDS>(SC, tile_c[c - 1], c == 1 ? ab_b : ab_r, ox0, oy, cty + 1, c == 1 ? spw_b : spw_r, res);
    if (npx) sao_store(SC, ox0, oy, npx, res);
#pragma unroll
    for (int i = 0; i < 4; i++) chroma_s[c - 1][cty][ctx + i] = i < npx ? res[i] : (Pix)0;
  }
  lds_barrier();
  // ---- RGB24 of this thread's luma samples (nearest-neighbour chroma: x / 2, y / 2)
  const colordev::ColorParams cp = cps[blockIdx.y];
  const bool int88 = cp.arith == colordev::AR_INT88;   // Op_YCbCr420_to_RGB24's integer arithmetic (the common case): decided once, not per sample
#pragma unroll
  for (int rr = 0; rr < SAO_RPT; rr++) {
    const int npx = ynpx[rr];
    if (!npx) continue;
    const int ly = ty + rr * RSTEP, oy = oy_t + ly, ox0 = ox_t + tx;
    int R[4], G[4], B[4];
    // the two chroma samples under this thread's four luma samples (nearest neighbour: x / 2; tx is a multiple of 4): one 16-bit LDS read each
    const uint32_t cb2 = *(const uint16_t*)&chroma_s[0][ly >> 1][tx >> 1], cr2 = *(const uint16_t*)&chroma_s[1][ly >> 1][tx >> 1];
    if (int88) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int cb = (int)((cb2 >> ((i >> 1) * 8)) & 255u) - 128, cr = (int)((cr2 >> ((i >> 1) * 8)) & 255u) - 128, Y = yres[rr][i];   // yuv2rgb.cc:377-421
        R[i] = colordev::clip_i(Y + ((cp.i_r_cr * cr + 128) >> 8), 255);
        G[i] = colordev::clip_i(Y + ((cp.i_g_cb * cb + cp.i_g_cr * cr + 128) >> 8), 255);
        B[i] = colordev::clip_i(Y + ((cp.i_b_cb * cb + 128) >> 8), 255);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) colordev::convert_px(cp, yres[rr][i], (int)((cb2 >> ((i >> 1) * 8)) & 255u), (int)((cr2 >> ((i >> 1) * 8)) & 255u), R[i], G[i], B[i]);
    }
    // (typed as global memory: through the generic pointer of the parameter block these were FLAT stores, which count in lgkmcnt - every LDS wait of the
    //  next phase waited for the RGB stores of this one)
    HIPDEC_GLOBAL uint8_t* o = (HIPDEC_GLOBAL uint8_t*)cp.o0 + (size_t)oy * cp.os + (size_t)ox0 * 3;
    if (npx == 4 && ((cp.os | (uintptr_t)cp.o0) & 3) == 0) {
      colordev::U3 v;
      v.a = R[0] | (G[0] << 8) | (B[0] << 16) | ((uint32_t)R[1] << 24);
      v.b = G[1] | (B[1] << 8) | (R[2] << 16) | ((uint32_t)G[2] << 24);
      v.c = B[2] | (R[3] << 8) | (G[3] << 16) | ((uint32_t)B[3] << 24);
      *(HIPDEC_GLOBAL colordev::U3*)o = v;
    } else {
      for (int i = 0; i < npx; i++) { o[3 * i] = (uint8_t)R[i]; o[3 * i + 1] = (uint8_t)G[i]; o[3 * i + 2] = (uint8_t)B[i]; }
    }
  }
  lds_barrier();   // the next tile overwrites the staged tiles
  }
}

// ---- the lean form of k_sao_rgb --------------------------------------------------------------------------------------------------------------------
// k_sao_rgb spends 1.7 vector wave-instructions per pixel (109 lane operations: byte-wise LDS reads with run-time neighbour offsets, compare /
// select chains, per-sample colour arithmetic) and is bound by vector issue at 43 ms per 2048 4K stills (30 % of the HBM roofline; its fetched bytes
// are at the algorithmic figure since the XCD-aware tile order).  Tiles in the INTERIOR of a plain 8-bit 4:2:0 picture (the tile inside the output
// and one sample away from the picture's borders, no lossless / unfiltered PCM units, neighbours free, integer RGB24 arithmetic, aligned crop) take this
// kernel instead: four samples per operation.
//   * a quad is ONE LDS dword; its two edge neighbours are dword reads at byte addresses (gfx950 reads LDS dwords at any alignment,
//     tools/ubench/lds_unaligned.hip) - 3 reads instead of 12;
//   * the four bytes are split into two registers of 2 x 16 bits (samples 0, 2 / samples 1, 3: v_perm_b32) and everything runs on v_pk_*_i16: the
//     two signs are clamps of differences, edgeIdx - or the band index - of the four samples becomes the SELECTOR of one v_perm_b32 that looks the four
//     offsets up in an 8-byte table built once per thread from the CTB's parameters (edge: o0 o1 0 o2 o3, band: o0 o1 o2 o3 0), and the offsets are added
//     and clipped as pairs;
//   * the chroma threads leave the three colour terms of their samples ((i_r_cr (Cr - 128) + 128) >> 8 ...: yuv2rgb.cc:377-421, 32-bit arithmetic as
//     there) in LDS as 16-bit pairs, so that R, G, B of a luma quad are three packed additions with clamps per register, and three v_perm_b32
//     interleave them into the 12 bytes of RGB24.
// About 30 lane operations per pixel.  Everything else of the tile - staging, the planes it writes, the arithmetic results - is k_sao_rgb's; tiles that do
// not qualify are left to it (launch_sao_rgb runs both; each returns at once from the other's tiles).

// does picture P (with colour stage cp) take the lean kernel?  The same answer on the host (launch_sao_rgb) and in both kernels.
__host__ __device__ inline bool sao_rgb_pic_is_lean(const PicParams& P, const colordev::ColorParams& cp)
{
  return P.chroma_format_idc == 1 && P.bit_depth_luma == 8 && P.bit_depth_chroma == 8 &&
         P.transquant_bypass_enabled == 0 && !(P.pcm_enabled && P.pcm_loop_filter_disabled) && P.sao_free_neighbours != 0 &&
         (cp.arith == colordev::AR_INT88 || cp.arith == colordev::AR_FLOAT) && cp.bpp == 8 && ((cp.os | (uintptr_t)cp.o0) & 3) == 0 &&
         (P.crop_x & 7) == 0 && (P.crop_y & 31) == 0 && P.log2_ctb >= 5 &&                  // quads are LDS dwords, a tile's 32 rows share one CTB row
         (P.out_width & 7) == 0 && (P.out_height & 1) == 0 && (P.width & 7) == 0 && (P.height & 1) == 0;   // a quad is inside the output or outside it, luma and chroma
}
// a CTB's SAO parameters as the lean kernel wants them
struct SaoLeanTab { uint32_t lo, hi, cls2; int type, cls, a_off; };
__device__ __forceinline__ SaoLeanTab sao_lean_tab(const uint32_t w[3], int rowb)
{
  SaoLeanTab t;
  t.type = (int)(w[0] & 255u);
  const int cls = (int)((w[0] >> 8) & 255u);
  t.cls = cls;
  // the low bytes of the four offsets (SaoOffsetVal fits a signed byte at 8 bits): o0 = w0 >> 16, o1 = w1 & 0xffff, o2 = w1 >> 16, o3 = w2 & 0xffff
  const uint32_t p3 = swar::perm(w[1], w[0], 0x0c060402u);            // o0 o1 o2 -
  const uint32_t p4 = swar::perm(w[2], p3, 0x04020100u);              // o0 o1 o2 o3
  const bool edge = t.type == 2, band = t.type == 1;
  t.lo = edge ? swar::perm(0u, p4, 0x020c0100u) : (band ? p4 : 0u);   // edge: edgeIdx' 0 1 2 3 4 -> o0 o1 0 o2 o3 (8.7.3.2); band: k 0 .. 3 -> o0 .. o3, 4 -> 0
  t.hi = edge ? (p4 >> 24) : 0u;
  t.cls2 = (uint32_t)cls * 0x00010001u;                               // band: sao_band_position for both halves
  const int hx = cls == 1 ? 0 : (cls == 3 ? 1 : -1), hy = cls == 0 ? 0 : -1;   // edge: the first neighbour (the second one is its mirror image)
  t.a_off = hy * rowb + hx;
  return t;
}
// SAO of the four samples at byte `base` of the staged tile `tb`: returns them as four bytes and as 16-bit pairs (samples 0, 2 / 1, 3).
// keep: bytes 0xff where an edge-offset sample stays as it is because a neighbour lies outside the picture (8.7.3: picture-border samples; 0 inside)
__device__ __forceinline__ uint32_t sao_quad_lean(const uint8_t* tb, int base, const SaoLeanTab& t, uint32_t keep, uint32_t& rlo, uint32_t& rhi)
{
  using namespace swar;
  const uint32_t v = lds32u(tb + base);
  const uint32_t vlo = perm(0u, v, kEven), vhi = perm(0u, v, kOdd);
  uint32_t sel = 0;
  if (t.type == 2) {
    const uint32_t a = lds32u(tb + base + t.a_off), b = lds32u(tb + base - t.a_off);
    const uint32_t alo = perm(0u, a, kEven), ahi = perm(0u, a, kOdd), blo = perm(0u, b, kEven), bhi = perm(0u, b, kOdd);
    // Sign(v - a) + Sign(v - b) + 2 per sample: 0 .. 4
    const uint32_t slo = pk_add(pk_min_c(pk_max_c(pk_sub(vlo, alo), 0xffffffffu), 0x00010001u), pk_min_c(pk_max_c(pk_sub(vlo, blo), 0xffffffffu), 0x00010001u));
    const uint32_t shi = pk_add(pk_min_c(pk_max_c(pk_sub(vhi, ahi), 0xffffffffu), 0x00010001u), pk_min_c(pk_max_c(pk_sub(vhi, bhi), 0xffffffffu), 0x00010001u));
    sel = pk_add_c(slo, 0x00020002u) | (pk_add_c(shi, 0x00020002u) << 8);
    sel = (sel & ~keep) | (0x02020202u & keep);        // edgeIdx' 2: offset 0
  } else if (t.type == 1) {
    // band position relative to sao_band_position, bands 4 .. 31 share the table's zero
    const uint32_t klo = pk_min_c(pk_sub((vlo >> 3) & 0x001f001fu, t.cls2) & 0x001f001fu, 0x00040004u);
    const uint32_t khi = pk_min_c(pk_sub((vhi >> 3) & 0x001f001fu, t.cls2) & 0x001f001fu, 0x00040004u);
    sel = klo | (khi << 8);
  }
  const uint32_t off4 = perm(t.hi, t.lo, sel);            // the four offsets, signed bytes
  const uint32_t off4s = off4 << 8;
  const uint32_t olo = perm(off4s, off4s, kSextOdd), ohi = perm(off4, off4, kSextOdd);
  rlo = pk_min_c(pk_max_c(pk_add(vlo, olo), 0u), 0x00ff00ffu);
  rhi = pk_min_c(pk_max_c(pk_add(vhi, ohi), 0u), 0x00ff00ffu);
  return rlo | (rhi << 8);
}
// the keep mask of a quad whose first sample is (x0, y) in a component plane of W x H samples (tiles at the picture's border only)
__device__ __forceinline__ uint32_t sao_lean_keep(const SaoLeanTab& t, int x0, int y, int W, int H)
{
  const bool hx = t.cls != 1, hy = t.cls != 0;         // the class looks left / right, up / down
  uint32_t keep = 0;
  if (hx && x0 == 0) keep |= 0x000000ffu;
  if (hx && x0 + 4 == W) keep |= 0xff000000u;
  if (hy && (y == 0 || y == H - 1)) keep = 0xffffffffu;
  return keep;
}

__global__ __launch_bounds__(256) void k_sao_rgb_lean(FilterArgs A, const colordev::ColorParams* __restrict__ cps)
{
  typedef uint8_t Pix;
  constexpr int ROW_WORDS = ((SAO_TW + 2) + 3) / 4 + 2, ROWB = ROW_WORDS * 4;
  constexpr int CROW_WORDS = ((SAO_CW + 2) + 3) / 4 + 2, CROWB = CROW_WORDS * 4;
  // (one word in front of each staged tile: at the picture's left / upper border a neighbour read reaches one byte in front of it - the sample is masked)
  __shared__ uint32_t tile_buf[1 + (SAO_TH + 2) * ROW_WORDS];
  __shared__ uint32_t tile_c_buf[2][1 + (SAO_CH + 2) * CROW_WORDS];
  // the colour terms of the tile's chroma positions: integer arithmetic R, G, B as 16-bit values; float arithmetic f_r_cr cr, f_g_cb cb, f_g_cr cr, f_b_cb cb
  __shared__ union { int16_t i[3][SAO_CH][SAO_CW]; float4 f[SAO_CH][SAO_CW]; } term;
  uint32_t* const tile = tile_buf + 1;
  if (*A.status != 0) return;
  const PicParams& P = A.pics[blockIdx.y];
  const colordev::ColorParams cp = cps[blockIdx.y];
  if (!sao_rgb_pic_is_lean(P, cp)) return;
  // A workgroup takes SAO_TPW tiles one BELOW the other (group g: tile column g % tiles_x, tile rows SAO_TPW (g / tiles_x) ..): its horizontal
  // neighbours - the groups g - 1 and g + 1, on the same XCD (sao_tile_of_block) and started with it - walk down beside it, so that the lines the
  // tiles share at their left and right ends are fetched by both at about the same time and once from HBM.  (Four tiles side by side, one after
  // the other: 3.2 B/px fetched instead of 1.55 - a tile's end lines had left the L2 when its neighbour came to them.)
  const int tiles_x = (P.out_width + SAO_TW - 1) / SAO_TW, tiles_y = (P.out_height + SAO_TH - 1) / SAO_TH;
  const int group = sao_tile_of_block(tiles_x * ((tiles_y + SAO_TPW - 1) / SAO_TPW));
  if (group < 0) return;
  const int tile_col = group % tiles_x, row0 = (group / tiles_x) * SAO_TPW;
  const SaoComp<Pix> SY = sao_comp<Pix>(A, P, 0, false, false), SB = sao_comp<Pix>(A, P, 1, false, false), SR = sao_comp<Pix>(A, P, 2, false, false);
  const int tid = threadIdx.x;
  // thread -> samples as in k_sao_rgb: a wave stays inside one CTB column
  const int wv = tid >> 6, ln = tid & 63;
  const int tx = (wv & 1) * 64 + (ln & 15) * 4, ty = (wv >> 1) * 16 + (ln >> 4);
  const int ctx = (wv & 1) * 32 + (ln & 7) * 4, cty = (wv >> 1) * 8 + (ln >> 3);
  constexpr int RSTEP = 4;
  const bool int88 = cp.arith == colordev::AR_INT88;
  const int first = row0, last = row0 + SAO_TPW < tiles_y ? row0 + SAO_TPW : tiles_y;   // tile rows
  // what a tile needs from global memory - the SAO parameters of the thread's CTBs and its words of the three source tiles - is requested one tile AHEAD:
  // while a tile is worked on, the next one's loads are in flight (with 5 workgroups per CU and a load - barrier - arithmetic - barrier sequence per
  // tile the kernel waited for memory most of the time)
  constexpr int NL_Y = ((SAO_TH + 2) * ROW_WORDS + 255) / 256, NL_C = ((SAO_CH + 2) * CROW_WORDS + 255) / 256;
  uint32_t vy[NL_Y], vb[NL_C], vr[NL_C], spw_y[3], spw_b[3], spw_r[3];
  auto request = [&](int t) {
    const int ox = tile_col * SAO_TW, oy = t * SAO_TH;
    sao_params_at(SY, ox + tx, oy + ty, spw_y);
    sao_params_at(SB, ox / 2 + ctx, oy / 2 + cty, spw_b);
    sao_params_at(SR, ox / 2 + ctx, oy / 2 + cty, spw_r);
    (void)sao_stage_load<Pix, SAO_TH, ROW_WORDS, 256, NL_Y>(SY, ox, oy, tid, vy);
    (void)sao_stage_load<Pix, SAO_CH, CROW_WORDS, 256, NL_C>(SB, ox / 2, oy / 2, tid, vb);
    (void)sao_stage_load<Pix, SAO_CH, CROW_WORDS, 256, NL_C>(SR, ox / 2, oy / 2, tid, vr);
  };
  request(first);
  for (int tile_idx = first; tile_idx < last; tile_idx++) {
    const int ox_t = tile_col * SAO_TW, oy_t = tile_idx * SAO_TH;
    const int xs0 = ox_t + SY.crop_xc, ys0 = oy_t + SY.crop_yc, cxs0 = xs0 >> 1, cys0 = ys0 >> 1;
    sao_stage_store<SAO_TH, ROW_WORDS, 256, NL_Y>(tile, tid, vy);
    sao_stage_store<SAO_CH, CROW_WORDS, 256, NL_C>(tile_c_buf[0] + 1, tid, vb);
    sao_stage_store<SAO_CH, CROW_WORDS, 256, NL_C>(tile_c_buf[1] + 1, tid, vr);
    const SaoLeanTab ty_tab = sao_lean_tab(spw_y, ROWB), tb_tab = sao_lean_tab(spw_b, CROWB), tr_tab = sao_lean_tab(spw_r, CROWB);
    lds_barrier();
    if (tile_idx + 1 < last) request(tile_idx + 1);
    // the staged rows start at byte ab of the plane's rows (4 bytes left of the tile's first sample; 0 at the picture's left border), one row above the tile
    const int xoff = xs0 - (xs0 > 0 ? xs0 - 4 : 0), cxoff = cxs0 - (cxs0 > 0 ? cxs0 - 4 : 0);
    // a tile at a border of the picture or of the output: quads outside the output are not stored, edge-offset samples at the picture's border are kept
    const bool border = xs0 == 0 || xs0 + SAO_TW >= SY.W || ys0 == 0 || ys0 + SAO_TH >= SY.H || ox_t + SAO_TW > SY.ow || oy_t + SAO_TH > SY.oh;
    // ---- luma: four quads, rows ty + 4 rr
    uint32_t ylo[SAO_RPT], yhi[SAO_RPT];
    const bool col_ok = !border || ox_t + tx + 4 <= SY.ow;
#pragma unroll
    for (int rr = 0; rr < SAO_RPT; rr++) {
      const int ly = ty + rr * RSTEP;
      const uint32_t keep = border ? sao_lean_keep(ty_tab, xs0 + tx, ys0 + ly, SY.W, SY.H) : 0u;
      const uint32_t r4 = sao_quad_lean((const uint8_t*)tile, (ly + 1) * ROWB + tx + xoff, ty_tab, keep, ylo[rr], yhi[rr]);
      if (col_ok && (!border || oy_t + ly < SY.oh)) *(uint32_t*)(SY.out + (size_t)(oy_t + ly) * SY.os + ox_t + tx) = r4;
    }
    // ---- Cb, Cr: one quad each, and the colour terms of its four positions
    uint32_t c4[2];
    const bool c_ok = !border || (ox_t / 2 + ctx + 4 <= SB.ow && oy_t / 2 + cty < SB.oh);
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const SaoComp<Pix>& SC = c == 0 ? SB : SR;
      const SaoLeanTab& tab = c == 0 ? tb_tab : tr_tab;
      const uint32_t keep = border ? sao_lean_keep(tab, cxs0 + ctx, cys0 + cty, SC.W, SC.H) : 0u;
      uint32_t lo, hi;
      c4[c] = sao_quad_lean((const uint8_t*)(tile_c_buf[c] + 1), (cty + 1) * CROWB + ctx + cxoff, tab, keep, lo, hi);
      if (c_ok) *(uint32_t*)(SC.out + (size_t)(oy_t / 2 + cty) * SC.os + ox_t / 2 + ctx) = c4[c];
    }
    if (int88) {
      int tr[4], tg[4], tb[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {   // yuv2rgb.cc:377-421
        const int cb = (int)((c4[0] >> (8 * k)) & 255u) - 128, cr = (int)((c4[1] >> (8 * k)) & 255u) - 128;
        tr[k] = (cp.i_r_cr * cr + 128) >> 8;
        tg[k] = (cp.i_g_cb * cb + cp.i_g_cr * cr + 128) >> 8;
        tb[k] = (cp.i_b_cb * cb + 128) >> 8;
      }
      *(uint2*)&term.i[0][cty][ctx] = make_uint2(((uint32_t)tr[0] & 0xffffu) | ((uint32_t)tr[1] << 16), ((uint32_t)tr[2] & 0xffffu) | ((uint32_t)tr[3] << 16));
      *(uint2*)&term.i[1][cty][ctx] = make_uint2(((uint32_t)tg[0] & 0xffffu) | ((uint32_t)tg[1] << 16), ((uint32_t)tg[2] & 0xffffu) | ((uint32_t)tg[3] << 16));
      *(uint2*)&term.i[2][cty][ctx] = make_uint2(((uint32_t)tb[0] & 0xffffu) | ((uint32_t)tb[1] << 16), ((uint32_t)tb[2] & 0xffffu) | ((uint32_t)tb[3] << 16));
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {   // yuv2rgb.cc:267-282, the products of colordev::convert_px's float arm (8 bits: halfRange 128)
        float cb = (float)((int)((c4[0] >> (8 * k)) & 255u) - 128), cr = (float)((int)((c4[1] >> (8 * k)) & 255u) - 128);
        if (!cp.full_range) { cb = cb * 1.1429f; cr = cr * 1.1429f; }
        term.f[cty][ctx + k] = make_float4(cp.f_r_cr * cr, cp.f_g_cb * cb, cp.f_g_cr * cr, cp.f_b_cb * cb);
      }
    }
    lds_barrier();
    // ---- RGB24 of the luma quads (nearest-neighbour chroma: samples 0, 1 take the terms of chroma position tx / 2, samples 2, 3 those of the next one)
#pragma unroll
    for (int rr = 0; rr < SAO_RPT; rr++) {
      using namespace swar;
      const int ly = ty + rr * RSTEP;
      colordev::U3 v;
      if (int88) {
        // exactly the pairing of the 16-bit halves: (Y0, Y2) and (Y1, Y3) both add the pair (term(c0), term(c1))
        const uint32_t t_r = *(const uint32_t*)&term.i[0][ly >> 1][tx >> 1], t_g = *(const uint32_t*)&term.i[1][ly >> 1][tx >> 1], t_b = *(const uint32_t*)&term.i[2][ly >> 1][tx >> 1];
        const uint32_t r_lo = pk_min_c(pk_max_c(pk_add(ylo[rr], t_r), 0u), 0x00ff00ffu), r_hi = pk_min_c(pk_max_c(pk_add(yhi[rr], t_r), 0u), 0x00ff00ffu);
        const uint32_t g_lo = pk_min_c(pk_max_c(pk_add(ylo[rr], t_g), 0u), 0x00ff00ffu), g_hi = pk_min_c(pk_max_c(pk_add(yhi[rr], t_g), 0u), 0x00ff00ffu);
        const uint32_t b_lo = pk_min_c(pk_max_c(pk_add(ylo[rr], t_b), 0u), 0x00ff00ffu), b_hi = pk_min_c(pk_max_c(pk_add(yhi[rr], t_b), 0u), 0x00ff00ffu);
        const uint32_t rg_lo = r_lo | (g_lo << 8);   // R0 G0 R2 G2
        const uint32_t br_x = b_lo | (r_hi << 8);    // B0 R1 B2 R3
        const uint32_t gb_hi = g_hi | (b_hi << 8);   // G1 B1 G3 B3
        v.a = perm(br_x, rg_lo, 0x05040100u);        // R0 G0 B0 R1
        v.b = perm(gb_hi, rg_lo, 0x03020504u);       // G1 B1 R2 G2
        v.c = perm(gb_hi, br_x, 0x07060302u);        // B2 R3 G3 B3
      } else {
        const float4 t0 = term.f[ly >> 1][tx >> 1], t1 = term.f[ly >> 1][(tx >> 1) + 1];
        int R[4], G[4], B[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int Y = (int)(((i & 1) ? yhi[rr] : ylo[rr]) >> ((i >> 1) * 16)) & 255;
          const float4 t = (i >> 1) ? t1 : t0;
          float yv = (float)Y;
          if (!cp.full_range) yv = (yv - 16.0f) * 1.1689f;
          R[i] = colordev::clip_f(yv + t.x, 255);
          G[i] = colordev::clip_f(yv + t.y + t.z, 255);
          B[i] = colordev::clip_f(yv + t.w, 255);
        }
        v.a = R[0] | (G[0] << 8) | (B[0] << 16) | ((uint32_t)R[1] << 24);
        v.b = G[1] | (B[1] << 8) | (R[2] << 16) | ((uint32_t)G[2] << 24);
        v.c = B[2] | (R[3] << 8) | (G[3] << 16) | ((uint32_t)B[3] << 24);
      }
      if (col_ok && (!border || oy_t + ly < SY.oh))
        *(HIPDEC_GLOBAL colordev::U3*)((HIPDEC_GLOBAL uint8_t*)cp.o0 + (size_t)(oy_t + ly) * cp.os + (size_t)(ox_t + tx) * 3) = v;
    }
    lds_barrier();   // the next tile overwrites the staged tiles and the terms
  }
}

void launch_deblock(const FilterArgs& a, int n_pics, int max_w, int max_h, bool wide, hipStream_t s, bool one_pass)
{
  static const bool two_pass_forced = getenv("HIPDEC_DEBLOCK_TWO_PASS") != nullptr;   // A/B knob
  if (one_pass && !two_pass_forced) {
    // intra pictures with 4:0:0 / 4:2:0 sampling only: both edge directions in one pass over each plane (k_deblock_fused); workgroups: a multiple of 8
    // (sao_tile_of_block)
    const int blocks_y = ((max_w >> 3) + 1) * ((max_h >> 3) + 1), blocks_c = ((max_w >> 4) + 1) * ((max_h >> 4) + 1);
    if (wide) {
      hipLaunchKernelGGL((k_deblock_fused<uint16_t, 0>), dim3(((blocks_y + 255) / 256 + 63) & ~63, n_pics), dim3(256), 0, s, a);
      hipLaunchKernelGGL((k_deblock_fused<uint16_t, 1>), dim3(((blocks_c + 255) / 256 + 63) & ~63, n_pics), dim3(256), 0, s, a);
    } else {
      hipLaunchKernelGGL((k_deblock_fused<uint8_t, 0>), dim3(((blocks_y + 255) / 256 + 63) & ~63, n_pics), dim3(256), 0, s, a);
      hipLaunchKernelGGL((k_deblock_fused<uint8_t, 1>), dim3(((blocks_c + 255) / 256 + 63) & ~63, n_pics), dim3(256), 0, s, a);
    }
    return;
  }
  const int uw = (max_w + 3) / 4, uh = (max_h + 3) / 4;
  const int work_v = ((max_w + 7) / 8) * uh, work_h = ((max_h + 7) / 8) * uw;
  if (wide) {
    hipLaunchKernelGGL((k_deblock<uint16_t, 0>), dim3((work_v + 255) / 256, n_pics), dim3(256), 0, s, a);
    hipLaunchKernelGGL((k_deblock<uint16_t, 1>), dim3((work_h + 255) / 256, n_pics), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((k_deblock<uint8_t, 0>), dim3((work_v + 255) / 256, n_pics), dim3(256), 0, s, a);
    hipLaunchKernelGGL((k_deblock<uint8_t, 1>), dim3((work_h + 255) / 256, n_pics), dim3(256), 0, s, a);
  }
}

// kernel variants: without the "samples stay as they are" paths (no lossless CUs / unfiltered PCM in the batch) and / or without the per-neighbour
// slice / tile checks (every picture has sao_free_neighbours) — the common batch takes the leanest one
#define HIPDEC_SAO_DISPATCH(LAUNCH)                          \
  do {                                                       \
    if (may_keep && restricted) { LAUNCH(true, true); }      \
    else if (may_keep) { LAUNCH(true, false); }              \
    else if (restricted) { LAUNCH(false, true); }            \
    else { LAUNCH(false, false); }                           \
  } while (0)

void launch_sao_rgb(const FilterArgs& a, const void* color_params_dev, int n_pics, int max_out_w, int max_out_h, hipStream_t s, bool may_keep, bool restricted,
                    const PicParams* host_pics, const void* host_color_params)
{
  const int tiles_x = (max_out_w + SAO_TW - 1) / SAO_TW, tiles_y = (max_out_h + SAO_TH - 1) / SAO_TH;
  const int tiles = ((tiles_x * tiles_y + SAO_TPW - 1) / SAO_TPW + 7) & ~7;          // workgroups: SAO_TPW tiles each, a multiple of 8 (sao_tile_of_block)
  const int tiles_lean = (tiles_x * ((tiles_y + SAO_TPW - 1) / SAO_TPW) + 7) & ~7;   // the lean kernel's groups are columns of SAO_TPW tiles
  // the lean kernel takes plain 8-bit 4:2:0 pictures (sao_rgb_pic_is_lean), the general one the rest; each returns at once from the other's pictures, and a
  // kernel without a picture is not launched when the host copies of the parameter blocks say so
  static const bool no_lean = getenv("HIPDEC_SAO_NO_LEAN") != nullptr;   // A/B knob
  const int skip_lean = no_lean ? 0 : 1;
  bool any_lean = skip_lean != 0, any_general = true;
  if (skip_lean && host_pics && host_color_params) {
    any_lean = any_general = false;
    for (int i = 0; i < n_pics; i++) {
      if (sao_rgb_pic_is_lean(host_pics[i], ((const colordev::ColorParams*)host_color_params)[i])) any_lean = true; else any_general = true;
    }
  }
  if (any_lean) hipLaunchKernelGGL(k_sao_rgb_lean, dim3(tiles_lean, n_pics), dim3(256), 0, s, a, (const colordev::ColorParams*)color_params_dev);
#define L_RGB(K, R) hipLaunchKernelGGL((k_sao_rgb<K, R>), dim3(tiles, n_pics), dim3(256), 0, s, a, (const colordev::ColorParams*)color_params_dev, skip_lean)
  if (any_general) HIPDEC_SAO_DISPATCH(L_RGB);
#undef L_RGB
}

void launch_sao(const FilterArgs& a, int n_pics, int max_out_w, int max_out_h, bool wide, hipStream_t s, bool may_keep, bool restricted)
{
  const int tiles = (((max_out_w + SAO_TW - 1) / SAO_TW) * ((max_out_h + SAO_TH - 1) / SAO_TH) + 7) & ~7;   // (sao_tile_of_block)
#define L_16(K, R) hipLaunchKernelGGL((k_sao<uint16_t, K, R>), dim3(tiles, n_pics, 3), dim3(256), 0, s, a)
#define L_8(K, R) hipLaunchKernelGGL((k_sao<uint8_t, K, R>), dim3(tiles, n_pics, 3), dim3(256), 0, s, a)
  if (wide) HIPDEC_SAO_DISPATCH(L_16); else HIPDEC_SAO_DISPATCH(L_8);
#undef L_16
#undef L_8
}

}  // namespace hipdec
