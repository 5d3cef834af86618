// residual_kernel.hip — dequantisation + inverse DCT/DST of every coded transform block, in place.
//
// Stands in for libde265's scaling / transform stages behind de265_decode() (reference call site
// libheif/plugins/decoder_libde265.cc:402).  ITU-T H.265 8.6.2-8.6.4: scaling with flat lists,
// transform skip, cu_transquant_bypass, DST-VII 4x4 for intra luma, DCT 4..32, 16-bit intermediate clip.
//
// MI355X mapping: residuals do not depend on prediction, so they are taken off the intra-prediction
// dependency chain entirely: this kernel runs over ALL transform blocks of the batch in parallel (one
// 256-thread workgroup per CTB) and overwrites each TU-contiguous int16 coefficient block with its int16
// residual block; the reconstruction wavefront (recon_kernel.hip) then only adds.
//
// Round 6 form.  The CTB's coded blocks are sorted by size into LDS lists and a wave pass takes 64 / n blocks of
// size n x n at once (16 4x4, 8 8x8 or 4 16x16 blocks), ONE THREAD PER ROW / COLUMN: a thread loads its row of
// levels with one or two 16-byte loads (the lanes of a block cover its n * n * 2 contiguous bytes), scales it,
// hands it through LDS to the thread that owns the column, which runs the column's 1-D inverse transform as
// an even-odd butterfly in registers (constants are instruction operands: no transform matrix in LDS, no table
// prologue), hands the clipped intermediate back through LDS by rows and runs the row transform the same way.
// Per 16x16 block that is ~120 wave instructions (0.47 per sample) where the wave-per-block form of rounds
// 1 - 5 (n MACs per output sample as v_dot2 chains over LDS operands: two LDS reads per MAC pair, 16 of 64
// lanes busy in the second pass of an 8x8 block, ~100 instructions of bookkeeping per block) spent 360 (1.4)
// and 3.2 per sample of an 8x8 block.  32x32 blocks (rare, and 32 + 32 live values per thread would cost the
// common sizes their occupancy) keep the wave-per-block form below.  Integer butterflies, no MFMA: 16-bit
// clipping sits between the two passes.  Traffic: reads and writes 2 B per coded sample.
#include <hip/hip_runtime.h>
#include <cstddef>
#include "hevc_device.h"
#include "kernels.h"

namespace hipdec {
namespace {

// small per-block lookups as packed immediates: a __constant__ array indexed at run time is a global load plus a wait in front of every block
__device__ __forceinline__ int chroma_qp_table(int qpi)   // table 8-10 for ChromaArrayType 1, qPi in [30, 43]
{
  constexpr uint64_t kT = 0ull | (1ull << 4) | (2ull << 8) | (3ull << 12) | (4ull << 16) | (4ull << 20) | (5ull << 24) | (5ull << 28) |
                          (6ull << 32) | (6ull << 36) | (7ull << 40) | (7ull << 44) | (8ull << 48) | (8ull << 52);
  return 29 + (int)((kT >> ((qpi - 30) * 4)) & 15u);
}
__device__ __forceinline__ int chroma_qp(int qpi, bool not420)   // 8.6.1: QpC from qPi: table 8-10 for ChromaArrayType 1, Min(qPi, 51) otherwise
{
  if (not420) return qpi < 51 ? qpi : 51;
  return qpi < 30 ? qpi : (qpi >= 44 ? qpi - 6 : chroma_qp_table(qpi));
}
// Scaling (8.6.3, flat m = 16) in 32 bits: level * 16 * levelScale < 2^27, and with q = qP / 6, b = bdShift
//   ((p << q) + (1 << (b - 1))) >> b  ==  q < b ? (p + (1 << (b - q - 1))) >> (b - q) : p << (q - b)      (q - b <= 3)
// `rs` / `ls` are the right / left shift of the block (one of them is 0), `rnd` = rs ? 1 << (rs - 1) : 0.
__device__ __forceinline__ int scale_level(int level, int f, int rs, int ls, int rnd)
{
  const int v = ((__mul24(level, f) + rnd) >> rs) << ls;   // |level| < 2^15, f <= 16 * 72
  return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}
// ... with a scaling list: f = m[x][y] * levelScale <= 255 * 72, level * f < 2^30; the left-shift case (q - b <= 3) is clamped
// first so that it cannot wrap before the 16-bit clip
__device__ __forceinline__ int scale_level_sl(int level, int f, int rs, int ls, int rnd)
{
  int v = (__mul24(level, f) + rnd) >> rs;   // f <= 255 * 72 < 2^15: the product is below 2^30
  v = v < -(1 << 27) ? -(1 << 27) : (v > (1 << 27) ? (1 << 27) : v);
  v <<= ls;
  return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}
__device__ __forceinline__ int level_scale(int r)         // levelScale[qP % 6] (8.6.3)
{
  constexpr uint64_t kS = 40ull | (45ull << 8) | (51ull << 16) | (57ull << 24) | (64ull << 32) | (72ull << 40);
  return (int)((kS >> (r * 8)) & 255u);
}

// ---- LDS of one workgroup (one CTB) ------------------------------------------------------------------------------------------------------
// 32x32 blocks (wave-per-block form): transposed, j-contiguous operands so that both 1-D passes are chains of v_dot2_i32_i16 (two MACs per
// instruction, one 32-bit LDS read per operand pair).  Rows are padded by 2 samples: consecutive lanes then hit distinct banks.
constexpr int RPAD = 2;
// 4x4 / 8x8 / 16x16 blocks (thread-per-row form): a wave's 64 / n blocks of one pass, rows of n int16, the blocks BS bytes apart so that
// the column accesses of a pass (lane = block * n + column reads / writes ONE int16 of row j, the same j in every lane) fall into distinct
// banks: a block's n lanes cover n * 2 contiguous bytes = n / 2 banks, and 64 / n blocks at a bank distance of n / 2 fill the 32 banks once
//   n = 16: BS = 512 + 32 (8 banks per block, 4 blocks), n = 8: 128 + 16 (4 banks, 8 blocks), n = 4: 32 + 8 (2 banks at multiples of 10: all 16 distinct)
constexpr int kSmallBufBytes = 4 * (512 + 32);
struct ResLds {
  union {
    struct { alignas(4) int16_t et32[32 * 32];            // E^T[i][j] of the 32-point matrix (filled only by a CTB that has a 32x32 block)
             alignas(4) int16_t blk[4][32 * (32 + RPAD)];   // per wave: scaled levels, TRANSPOSED: blk[x][j] = d[j][x]
             alignas(4) int16_t tmp[4][32 * (32 + RPAD)];   // per wave: first-stage output tmp[y][j]
    } big;
    struct { alignas(16) uint8_t buf[4][kSmallBufBytes]; } small;   // per wave
  } u;
  uint8_t m_size[256], m_flags[256], m_ipm[256];
  int8_t m_qp[256];
  // block lists by size, entry = z | component << 8 | lower 4:2:2 block << 10 | inter coded unit << 11 (z = the unit that carries the TU's flags).
  // Capacities: a 64x64 CTB in 4:4:4 (three components of 256 / 64 / 16 / 4 blocks); 4:2:2 stays below them
  uint16_t l4[768], l8[192], l16[48], l32[12];
  uint32_t count[4];     // entries in l4, l8, l16, l32
  uint32_t colmask[4];   // 32x32 form, per wave: columns of the current block that hold a nonzero level (DS atomic OR of the lanes' bits; zero between blocks)
};

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
// A wave's blk / tmp areas are its own: its lanes only need their LDS traffic drained before they read each other's values.  (A workgroup-scope
// fence here also waited for the wave's outstanding GLOBAL loads and stores - vmcnt(0) - i.e. for the block's result stores and for the next
// block's prefetched levels, three times per block.)
__device__ __forceinline__ void lds_sync()
{
#ifndef HIPDEC_HOST_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t compact1by1(uint32_t v)
{
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu;
  return v;
}

#ifndef HIPDEC_HOST_EMU
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc)
{
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), acc, false);
}
#else   // CPU-test build (tests/emu): v_dot2_i32_i16 spelled out
inline int dot2(uint32_t a, uint32_t b, int acc)
{
  return acc + (int)(int16_t)(a & 0xffff) * (int)(int16_t)(b & 0xffff) + (int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16);
}
#endif

// one transform block, by one wave: coef (global, n*n int16, raster) -> residual in place
// SL: the block's ScalingFactor table m[y * n + x] (read from the picture's 2 KB table in HBM: cache-resident, and only
// streams with scaling lists pay for it - no LDS is set aside) replaces the flat factor 16 (8.6.4.2)
// first_raw: the lane's first four levels (coef[lane * 4 ..], zero where lane * 4 >= n * n), loaded by the caller one block AHEAD: a wave works on one
// block at a time and used to start each with a global load it needed at once - the HBM latency of that load, block after block, was what the kernel
// waited for (PMC: 51 % of its wave cycles in s_waitcnt; removing 40 % of its instructions changed its time by 2 %)
template <bool SL>
__device__ __forceinline__ void residual_block(ResLds& L, int wave, int lane, int16_t* coef, int log2n, int bit_depth, int qp, int dst,
                                               int transform_skip, int bypass, const uint8_t* m, uint2 first_raw)
{
  if (bypass) return;  // cu_transquant_bypass: the coefficient levels are the residual (8.6.2)
  const int n = 1 << log2n, nn = n * n, rs = n + RPAD;
  int16_t* blk = L.u.big.blk[wave];
  int16_t* tmp = L.u.big.tmp[wave];
  const int16_t* et = L.u.big.et32;   // (this form only serves 32x32 blocks since round 6: E_32 itself)
  // ---- scaling (8.6.3, flat m = 16) + nonzero extent ----
  const int bd_shift = bit_depth + log2n - 5;
  const int q6 = qp / 6, ls6 = level_scale(qp - 6 * q6), f = 16 * ls6;
  const int sh_r = q6 < bd_shift ? bd_shift - q6 : 0, sh_l = q6 < bd_shift ? 0 : q6 - bd_shift, rnd = sh_r ? 1 << (sh_r - 1) : 0;
  const int bd_shift2 = 20 - bit_depth;
  // nonzero extent (max_row, max_col) without cross-lane shuffles: rows grow with the lane index, so the last
  // nonzero row falls out of one ballot per pass; the nonzero columns are OR-ed into one LDS word per wave by a DS atomic in front of the
  // fence the first stage needs anyway (bit by bit with five dependent ballots this was a sixth of the kernel's instructions)
  int max_row = -1;
  uint32_t my_cols = 0;   // nonzero columns held by this lane
  for (int base = 0; base < nn; base += 256) {   // wave-uniform trip count: every lane takes part in the ballot
    const int idx = base + lane * 4;
    const bool active = idx < nn;
    uint2 raw = make_uint2(0, 0);
    if (base == 0) raw = first_raw;
    else if (active) raw = *(const uint2*)&coef[idx];
    int16_t c[4] = {(int16_t)(raw.x & 0xffff), (int16_t)(raw.x >> 16), (int16_t)(raw.y & 0xffff), (int16_t)(raw.y >> 16)};
    int16_t d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      d[k] = (int16_t)(SL ? scale_level_sl(c[k], (active ? (int)m[idx + k] : 16) * ls6, sh_r, sh_l, rnd) : scale_level(c[k], f, sh_r, sh_l, rnd));
    }
    my_cols |= (uint32_t)((c[0] != 0) | ((c[1] != 0) << 1) | ((c[2] != 0) << 2) | ((c[3] != 0) << 3)) << (idx & (n - 1));   // (an inactive lane holds zeros)
    if (active) {
      if (transform_skip) {  // 8.6.4.2: r = d << 7, then the second-stage shift; no transform, no LDS
        int16_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = (int16_t)(((int)d[k] * 128 + (1 << (bd_shift2 - 1))) >> bd_shift2);
        *(uint2*)&coef[idx] = make_uint2((uint16_t)r[0] | ((uint32_t)(uint16_t)r[1] << 16), (uint16_t)r[2] | ((uint32_t)(uint16_t)r[3] << 16));
      } else {
        const int y = idx >> log2n, x0 = idx & (n - 1);   // the lane's 4 levels sit in row y, columns x0..x0+3
#pragma unroll
        for (int k = 0; k < 4; k++) blk[(x0 + k) * rs + y] = d[k];
      }
    }
    const unsigned long long nz = __ballot((raw.x | raw.y) != 0);
    if (nz) { const int last_lane = 63 - __clzll((long long)nz); const int r = (base + last_lane * 4) >> log2n; max_row = r > max_row ? r : max_row; }
  }
  if (transform_skip) return;
  if (my_cols) __hip_atomic_fetch_or(&L.colmask[wave], my_cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  lds_sync();
  const uint32_t cols = L.colmask[wave];
  const int max_col = cols ? 31 - __clz((int)cols) : -1;
  __builtin_amdgcn_wave_barrier();      // every lane has read the word
  if (lane == 0) L.colmask[wave] = 0;   // (a wave's LDS operations execute in order: the next block's atomics come after this store)
  const int rows_nz2 = (max_row + 2) >> 1, cols_nz = max_col + 1, cols_nz2 = (max_col + 2) >> 1;   // nonzero extents (pairs)
  // first stage (columns): tmp[i][x] = clip16((sum_j E[j][i] * d[j][x] + 64) >> 7); only x < cols_nz can be nonzero.
  // (pairs beyond max_row read zeros: the levels there are zero and were written)
  for (int idx = lane; idx < nn; idx += 64) {
    const int x = idx & (n - 1), i = idx >> log2n;
    int sum = 0;
    if (x < cols_nz) {
      const uint32_t* e = (const uint32_t*)(et + i * n);
      const uint32_t* v = (const uint32_t*)(blk + x * rs);
      for (int j = 0; j < rows_nz2; j++) sum = dot2(e[j], v[j], sum);
    }
    tmp[i * rs + x] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
  }
  lds_sync();
  // second stage (rows): res[y][i] = (sum_j E[j][i] * tmp[y][j] + rnd) >> bd_shift2, j < cols_nz
  for (int idx = lane * 4; idx < nn; idx += 256) {
    const int y = idx >> log2n, i0 = idx & (n - 1);
    const uint32_t* v = (const uint32_t*)(tmp + y * rs);
    int sum[4] = {0, 0, 0, 0};
    for (int j = 0; j < cols_nz2; j++) {
      const uint32_t t = v[j];
#pragma unroll
      for (int k = 0; k < 4; k++) sum[k] = dot2(((const uint32_t*)(et + (i0 + k) * n))[j], t, sum[k]);
    }
    int16_t r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = (int16_t)((sum[k] + (1 << (bd_shift2 - 1))) >> bd_shift2);
    *(uint2*)&coef[idx] = make_uint2((uint16_t)r[0] | ((uint32_t)(uint16_t)r[1] << 16), (uint16_t)r[2] | ((uint32_t)(uint16_t)r[3] << 16));
  }
  lds_sync();
}

// ---- thread-per-row form (4x4, 8x8, 16x16) -------------------------------------------------------------------------------------------------
// 8.6.4.2: the n-point matrix is E_n[j][i] = M32[j * 32 / n][i], M32[j][i] = +-c[(2 i + 1) j mod 128, folded into 0 .. 32], c = the 33 magnitudes kMag
constexpr int kMag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
constexpr int m32(int j, int i)
{
  int k = ((2 * i + 1) * j) & 127;
  if (k > 64) k = 128 - k;
  return k <= 32 ? kMag[k] : -kMag[64 - k];
}
#ifdef HIPDEC_HOST_EMU
#define RES_UNROLL
#else
#define RES_UNROLL _Pragma("unroll")
#endif
// y[i] = rnd + sum_j E_N[j][i] x[j] as the even-odd butterfly: E_N[j][N - 1 - i] = (-1)^j E_N[j][i], and the even rows of E_N are E_(N/2), so
//   y[k] = even[k] + odd[k], y[N - 1 - k] = even[k] - odd[k]   with even = the N/2-point transform of x[0], x[2], ... and odd[k] = sum over odd j.
// All sums are exact 32-bit integers (|x| < 2^15, sum of |E| over a column < 2^12): the same numbers as the n-MACs-per-sample form.
// The inputs arrive PACKED in pairs that one v_dot2_i32_i16 consumes (two MACs per instruction against a constant pair), in the order the recursion
// wants them - "butterfly order": the even-index inputs first (in the butterfly order of the half-size transform), then the odd ones as
// (x1, x3), (x5, x7), ...  For N = 16: (x0, x8) (x4, x12) (x2, x6) (x10, x14) (x1, x3) (x5, x7) (x9, x11) (x13, x15).  The transposes through LDS
// deliver that order for free: the writer of element j puts it at position bfly_pos(j) of its row.  16 points: 44 dot2 + 30 additions.
constexpr uint32_t pk16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int bfly_pos(int j, int n)      // position of input j in the butterfly order of an n-point transform
{
  int base = 0;
  for (; n > 2; n >>= 1) {
    if (j & 1) return base + (n >> 1) + (j >> 1);
    j >>= 1;
  }
  return base + j;
}
template <int N>
struct Idct {
  static __device__ __forceinline__ void run(const uint32_t* p, int rnd, int* y)
  {
    int ev[N / 2];
    Idct<N / 2>::run(p, rnd, ev);
    RES_UNROLL
    for (int k = 0; k < N / 2; k++) {
      int od = 0;
      RES_UNROLL
      for (int m = 0; m < N / 4; m++) od = dot2(p[N / 4 + m], pk16(m32((4 * m + 1) * (32 / N), k), m32((4 * m + 3) * (32 / N), k)), od);
      y[k] = ev[k] + od;
      y[N - 1 - k] = ev[k] - od;
    }
  }
};
template <>
struct Idct<2> {
  static __device__ __forceinline__ void run(const uint32_t* p, int rnd, int* y) { y[0] = dot2(p[0], pk16(64, 64), rnd); y[1] = dot2(p[0], pk16(64, -64), rnd); }
};
// 4x4 DST-VII of intra luma blocks (8.6.4.2: transMatrix rows {29 55 74 84} {74 74 0 -74} {84 -29 -74 55} {55 -84 74 -29}); p = (x0, x2) (x1, x3)
__device__ __forceinline__ void idst4(const uint32_t* p, int rnd, int* y)
{
  y[0] = dot2(p[1], pk16(74, 55), dot2(p[0], pk16(29, 84), rnd));
  y[1] = dot2(p[1], pk16(74, -84), dot2(p[0], pk16(55, -29), rnd));
  y[2] = dot2(p[1], pk16(0, 74), dot2(p[0], pk16(74, -74), rnd));
  y[3] = dot2(p[1], pk16(-74, -29), dot2(p[0], pk16(84, 55), rnd));
}

// what the kernel knows about its CTB / picture
struct ResCtx {
  int16_t *coef_y, *coef_cb, *coef_cr;
  const uint8_t* sl_tab;
  int cfi, bd_luma, bd_chroma, cb_off, cr_off;
};
// a list entry -> the block's levels and everything its scaling needs (per lane: the lanes of a pass work on different blocks)
struct BlkMeta {
  int16_t* p;          // the block's n * n levels (TU-contiguous, raster)
  const uint8_t* m;    // its ScalingFactor table (scaling lists only)
  int ls6, sh_r, sh_l, rnd, bd2, ts, dst;
};
template <bool GEN>
__device__ __forceinline__ BlkMeta block_meta(const ResLds& L, const ResCtx& cx, int entry, int log2n)
{
  const bool c444 = GEN && cx.cfi == 3, c422 = GEN && cx.cfi == 2;
  const int z = entry & 255, c = (entry >> 8) & 3, low = c422 ? (entry >> 10) & 1 : 0, inter = (entry >> 11) & 1;
  const int t = L.m_size[z] & 15, fl = L.m_flags[z], ipm = L.m_ipm[low ? (z ^ 1) : z], qp_y = L.m_qp[z];
  BlkMeta b;
  int bit_depth, qp;
  if (c == 0) {
    b.p = cx.coef_y + z * 16; bit_depth = cx.bd_luma; qp = qp_y + 6 * (cx.bd_luma - 8); b.ts = (fl & UF_TS_LUMA) != 0;
    b.dst = log2n == 2 && !inter;   // DST-VII for the 4x4 luma blocks of intra coded units only
  } else {
    // the chroma blocks of four 4x4 luma blocks hang off the quad's first unit; 4:2:2: the lower block follows the upper one
    const int zc = (t > 2 || c444) ? z : (z & ~3);
    const int off_c = 6 * (cx.bd_chroma - 8);
    const int qpi = clip3(-off_c, 57, qp_y + (c == 1 ? cx.cb_off : cx.cr_off));
    b.p = (c == 1 ? cx.coef_cb : cx.coef_cr) + zc * (c444 ? 16 : (c422 ? 8 : 4)) + (low << (2 * log2n));
    bit_depth = cx.bd_chroma; qp = chroma_qp(qpi, GEN && cx.cfi != 1) + off_c; b.ts = (ipm & (c == 1 ? 64 : 128)) != 0;
    b.dst = 0;
  }
  const int q6 = (qp * 43) >> 8;           // qp / 6 for 0 <= qp < 128
  const int bd_shift = bit_depth + log2n - 5;
  b.ls6 = level_scale(qp - 6 * q6);
  b.sh_r = q6 < bd_shift ? bd_shift - q6 : 0; b.sh_l = q6 < bd_shift ? 0 : q6 - bd_shift; b.rnd = b.sh_r ? 1 << (b.sh_r - 1) : 0;
  if (!cx.sl_tab) b.ls6 = (16 * b.ls6) << b.sh_l;   // flat lists: the whole factor (m = 16, levelScale, the left shift; q - b <= 3: < 2^14)
  b.bd2 = 20 - bit_depth;
  // ScalingFactor tables (hevc_device.h, PicParams::off_scaling): component c at c * 336 (4x4, 8x8 at + 16, 16x16 at + 80), inter coded units 2048 bytes on
  b.m = cx.sl_tab ? cx.sl_tab + (inter << 11) + c * 336 + (log2n == 2 ? 0 : (log2n == 3 ? 16 : 80)) : nullptr;
  return b;
}
template <int N> struct RowRaw { uint32_t w[N / 2]; };     // a row of N int16
template <int N>
__device__ __forceinline__ RowRaw<N> row_load(const int16_t* p)
{
  RowRaw<N> r;
  if constexpr (N == 4) { const uint2 v = *(const uint2*)p; r.w[0] = v.x; r.w[1] = v.y; }
  else {
    RES_UNROLL
    for (int q = 0; q < N / 8; q++) { const uint4 v = ((const uint4*)p)[q]; r.w[4 * q] = v.x; r.w[4 * q + 1] = v.y; r.w[4 * q + 2] = v.z; r.w[4 * q + 3] = v.w; }
  }
  return r;
}
template <int N>
__device__ __forceinline__ void row_store(int16_t* p, const RowRaw<N>& r)
{
  if constexpr (N == 4) *(uint2*)p = make_uint2(r.w[0], r.w[1]);
  else {
    RES_UNROLL
    for (int q = 0; q < N / 8; q++) { uint4 v; v.x = r.w[4 * q]; v.y = r.w[4 * q + 1]; v.z = r.w[4 * q + 2]; v.w = r.w[4 * q + 3]; ((uint4*)p)[q] = v; }
  }
}
template <int N> __device__ __forceinline__ int row_get(const RowRaw<N>& r, int k) { return (int)(int16_t)(r.w[k >> 1] >> ((k & 1) * 16)); }
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

// All blocks of one size (N = 1 << LG) of the CTB: pass q takes the blocks q * BPP .. of `list`, lane = block * N + row; a wave takes the passes
// first, first + 4, ...  The NEXT pass's rows are requested before the current pass is worked on.
template <int LG, bool SL, bool GEN>
__device__ __forceinline__ void residual_class(ResLds& L, const ResCtx& cx, int wave, int lane, int first, int count, const uint16_t* list)
{
  constexpr int N = 1 << LG, BPP = 64 >> LG;
  constexpr int BS = LG == 4 ? 512 + 32 : (LG == 3 ? 128 + 16 : 32 + 8);
  static_assert(BPP * BS <= kSmallBufBytes, "a pass fits the wave's buffer");
  const int g = lane >> LG, r = lane & (N - 1);
  uint8_t* const blk = L.u.small.buf[wave] + g * BS;           // this lane's block
  int16_t* const my_row = (int16_t*)(blk + r * (2 * N));      // row r of the block's N x N int16
  int16_t* const my_pos = (int16_t*)blk + bfly_pos(r, N);      // this lane's slot in row j: my_pos[j * N]
  const int passes = (count + BPP - 1) / BPP;
  bool valid = false;
  BlkMeta bm{};
  bm.bd2 = 12;                                                  // (lanes without a block compute on zeros with legal shift counts)
  RowRaw<N> raw{};
  RowRaw<N / 2> mrow{};                                        // the row's ScalingFactors (N bytes; scaling lists only)
  auto fetch = [&](int q, BlkMeta& b, RowRaw<N>& rw, decltype(mrow)& mr) -> bool {
    const int idx = q * BPP + g;
    const bool ok = idx < count;
    rw = RowRaw<N>{};
    if (ok) {
      b = block_meta<GEN>(L, cx, (int)list[idx], LG);
      rw = row_load<N>(b.p + r * N);
      if (SL) {
        if constexpr (N == 4) mr.w[0] = *(const uint32_t*)(b.m + r * 4);
        else if constexpr (N == 8) { const uint2 v = *(const uint2*)(b.m + r * 8); mr.w[0] = v.x; mr.w[1] = v.y; }
        else { const uint4 v = *(const uint4*)(b.m + r * 16); mr.w[0] = v.x; mr.w[1] = v.y; mr.w[2] = v.z; mr.w[3] = v.w; }
      }
    }
    return ok;
  };
  if (first < passes) valid = fetch(first, bm, raw, mrow);
  for (int q = first; q < passes; q += 4) {
    const bool cur_valid = valid;
    const BlkMeta b = bm;
    const RowRaw<N> cur = raw;
    const auto cur_m = mrow;
    if (q + 4 < passes) valid = fetch(q + 4, bm, raw, mrow);
    // ---- scaling (8.6.3) of this lane's row d[r][.]; element k goes to the thread of column k: row k of the buffer, slot bfly_pos(r) ----
    int d[N];
    RES_UNROLL
    for (int k = 0; k < N; k++) {
      const int lev = row_get<N>(cur, k);      // (a lane without a block holds zeros)
      // flat lists: one of the two shifts is 0, and a left shift of the product is a left shift of the factor: (level * (f << ls) + rnd) >> rs
      d[k] = SL ? scale_level_sl(lev, (int)((cur_m.w[k >> 2] >> ((k & 3) * 8)) & 255u) * b.ls6, b.sh_r, b.sh_l, b.rnd)
                : clip3(-32768, 32767, (__mul24(lev, b.ls6) + b.rnd) >> b.sh_r);
      my_pos[k * N] = (int16_t)d[k];
    }
    lds_sync();
    // ---- first stage (columns): tmp[i][x] = clip16((sum_j E[j][i] d[j][x] + 64) >> 7), this lane owns column x = r: row r of the buffer holds
    //      d[.][r] in butterfly order.  Its outputs go to the threads of the rows: row i of the buffer, slot bfly_pos(r).  (Reads of a wave complete before
    //      its writes: the buffer is reused in place.)
    int y[N];
    {
      const RowRaw<N> t = row_load<N>(my_row);
      if (N == 4 && b.dst) idst4(t.w, 64, y); else Idct<N>::run(t.w, 64, y);
    }
    lds_sync();
    RES_UNROLL
    for (int i = 0; i < N; i++) my_pos[i * N] = (int16_t)clip3(-32768, 32767, y[i] >> 7);
    lds_sync();
    // ---- second stage (rows): res[y][i] = (sum_j E[j][i] tmp[y][j] + rnd) >> bdShift, this lane owns row y = r ----
    const int rnd2 = 1 << (b.bd2 - 1);
    {
      const RowRaw<N> t = row_load<N>(my_row);
      if (N == 4 && b.dst) idst4(t.w, rnd2, y); else Idct<N>::run(t.w, rnd2, y);
    }
    RowRaw<N> out;
    RES_UNROLL
    for (int k = 0; k < N / 2; k++) {
      int r0 = y[2 * k] >> b.bd2, r1 = y[2 * k + 1] >> b.bd2;
      if (N == 4 && b.ts) { r0 = (d[2 * k] * 128 + rnd2) >> b.bd2; r1 = (d[2 * k + 1] * 128 + rnd2) >> b.bd2; }   // 8.6.4.2 with transform_skip_flag: r = d << 7
      out.w[k] = pack16(r0, r1);
    }
    if (cur_valid) row_store<N>(b.p + r * N, out);
    lds_sync();      // the next pass's rows overwrite the buffer
  }
}

}  // namespace

// blockIdx.x = CTB (raster) of picture blockIdx.y.  GEN = false: a build for batches of 4:0:0 / 4:2:0 pictures only (the 4:2:2 / 4:4:4 block
// addressing costs the common batch a few per cent of this kernel)
template <bool GEN>
__global__ __launch_bounds__(256) void k_residual(FilterArgs A)
{
  __shared__ ResLds L;
  if (*A.status != 0) return;
  const PicParams& P = A.pics[blockIdx.y];
  const int cfi_p = GEN ? P.chroma_format_idc : (P.chroma_format_idc ? 1 : 0);
  const int n_ctb = P.ctb_w * P.ctb_h;
  const int ctb_rs = blockIdx.x;
  if (ctb_rs >= n_ctb) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int units = 1 << P.units_per_ctb_log2;
  const int ctb = 1 << P.log2_ctb;
  const size_t base = (size_t)ctb_rs * units;
  // Everything the prologue reads from global memory is requested FIRST and used afterwards: the CTB's unit maps (four bytes per thread, five in
  // P / B pictures), its slice's chroma QP offsets
  uint8_t u_size = 0, u_flags = 0, u_ipm = 0, u_qp = 0, u_ipmc = 0;
  if (tid < units) {
    u_size = A.arena[P.off_u_size + base + tid]; u_flags = A.arena[P.off_u_flags + base + tid];
    u_ipm = A.arena[P.off_u_ipm + base + tid]; u_qp = A.arena[P.off_u_qp + base + tid];
    if (P.is_inter) u_ipmc = A.arena[P.off_u_ipmc + base + tid];
  }
  const CtbInfo ci = ((const CtbInfo*)(A.arena + P.off_ctb_info))[ctb_rs];
  const bool c444 = cfi_p == 3;   // chroma blocks have the luma blocks' size and position
  const bool c422 = cfi_p == 2;   // two chroma blocks of half the luma block's size, one above the other
  if (tid < 4) { L.count[tid] = 0; L.colmask[tid] = 0; }
  if (tid < units) { L.m_size[tid] = u_size; L.m_flags[tid] = u_flags; L.m_ipm[tid] = u_ipm; L.m_qp[tid] = (int8_t)u_qp; }
  __syncthreads();
  // ---- lists of coded blocks by size, one entry per block and component: z | c << 8 with z the unit that carries the flags (a TU's
  //      first unit; for the chroma blocks of four 4x4 luma TUs the 4th unit, where the parser leaves their flags) ----
  auto push = [&](int log2n, uint16_t entry) {
    if (log2n == 2) L.l4[atomicAdd(&L.count[0], 1u)] = entry;
    else if (log2n == 3) L.l8[atomicAdd(&L.count[1], 1u)] = entry;
    else if (log2n == 4) L.l16[atomicAdd(&L.count[2], 1u)] = entry;
    else L.l32[atomicAdd(&L.count[3], 1u)] = entry;
  };
  if (tid < units) {
    const int z = tid;
    const int ux = (int)compact1by1((uint32_t)z), uy = (int)compact1by1((uint32_t)z >> 1);
    const int x_ctb = (ctb_rs % P.ctb_w) << P.log2_ctb, y_ctb = (ctb_rs / P.ctb_w) << P.log2_ctb;
    if (x_ctb + ux * 4 < P.width && y_ctb + uy * 4 < P.height) {
      const int t = u_size & 15, fl = u_flags;
      const int first = t >= 2 && t <= 5 && (z & ((1 << (2 * (t - 2))) - 1)) == 0;
      if (first && !(fl & UF_BYPASS)) {
        // (P pictures: bit 11 = the block belongs to an inter coded unit: its 4x4 luma transform is the DCT, not the DST of intra blocks, 8.6.4.2;
        //  with scaling lists the bit also selects the matrices of inter coded units for every block size and component, Table 7-4)
        const int inter_bit = (P.is_inter && (u_ipmc & UM_INTER)) ? 0x800 : 0;
        if (fl & UF_CBF_LUMA) push(t, (uint16_t)(z | inter_bit));
        // chroma blocks hang off the unit that carries their flags: a block's first unit, or the 4th unit of a quad of 4x4 luma blocks.  4:2:2 has
        // two chroma blocks per unit; the lower one's flags sit in unit z ^ 1 (so in a 4:2:2 quad only the 4th unit's flags are block flags)
        if (cfi_p && !(c422 && t == 2 && (z & 3) != 3))
          for (int c = 1; c < 3; c++)
            for (int low = 0; low < (c422 ? 2 : 1); low++)
              if ((low ? L.m_flags[z ^ 1] : fl) & (c == 1 ? UF_CBF_CB : UF_CBF_CR))
                push(c444 ? t : (t > 3 ? t - 1 : 2), (uint16_t)(z | (c << 8) | (low << 10) | inter_bit));
      }
    }
  }
  __syncthreads();
  const int n4 = (int)L.count[0], n8 = (int)L.count[1], n16 = (int)L.count[2], n32 = (int)L.count[3];
  const SliceParams sl = ((const SliceParams*)(A.arena + P.off_slices))[ci.slice_idx];
  const int cc_shift = c444 ? 0 : (c422 ? 1 : 2);
  ResCtx cx;
  cx.coef_y = (int16_t*)(A.arena + P.off_coeff[0]) + (size_t)ctb_rs * ctb * ctb;
  // (two named pointers, selected by comparison: an array indexed by the component goes through memory and comes back as a generic pointer -
  //  FLAT loads and stores, which also count in lgkmcnt, so that every LDS wait waited for the global traffic too)
  cx.coef_cb = (int16_t*)(A.arena + P.off_coeff[1]) + (size_t)ctb_rs * ((ctb * ctb) >> cc_shift);
  cx.coef_cr = (int16_t*)(A.arena + P.off_coeff[2]) + (size_t)ctb_rs * ((ctb * ctb) >> cc_shift);
  cx.sl_tab = P.scaling_lists ? A.arena + P.off_scaling : nullptr;   // ScalingFactor tables of the picture
  cx.cfi = cfi_p; cx.bd_luma = P.bit_depth_luma; cx.bd_chroma = P.bit_depth_chroma; cx.cb_off = sl.cb_qp_offset; cx.cr_off = sl.cr_qp_offset;
  const bool use_sl = cx.sl_tab != nullptr;
  // the passes of the three sizes behind one another, dealt to the four waves round-robin (largest blocks first)
  const int p16 = (n16 + 3) >> 2, p8 = (n8 + 7) >> 3;
  const int f16 = wave, f8 = (wave - p16) & 3, f4 = (wave - p16 - p8) & 3;
  if (use_sl) {
    residual_class<4, true, GEN>(L, cx, wave, lane, f16, n16, L.l16);
    residual_class<3, true, GEN>(L, cx, wave, lane, f8, n8, L.l8);
    residual_class<2, true, GEN>(L, cx, wave, lane, f4, n4, L.l4);
  } else {
    residual_class<4, false, GEN>(L, cx, wave, lane, f16, n16, L.l16);
    residual_class<3, false, GEN>(L, cx, wave, lane, f8, n8, L.l8);
    residual_class<2, false, GEN>(L, cx, wave, lane, f4, n4, L.l4);
  }
  if (n32 == 0) return;
  // ---- 32x32 blocks, one per wave pass (wave-per-block form); the buffers above are dead behind the barrier ----
  __syncthreads();
  for (int idx = tid; idx < 1024; idx += 256) { const int i = idx >> 5, j = idx & 31; L.u.big.et32[idx] = (int16_t)m32(j, i); }   // E^T[i][j]
  __syncthreads();
  for (int e = wave; e < n32; e += 4) {
    const int entry = L.l32[e];
    const int z = entry & 255, c = (entry >> 8) & 3, sl_off = ((entry >> 11) & 1) << 11;
    const int fl = L.m_flags[z], ipm = L.m_ipm[z], qp_y = L.m_qp[z];
    int16_t* const cc = c == 0 ? cx.coef_y + z * 16 : (c == 1 ? cx.coef_cb : cx.coef_cr) + z * 16;   // (a 32x32 chroma block: 4:4:4 only)
    const uint2 first_raw = *(const uint2*)&cc[lane * 4];
    if (c == 0) {
      if (use_sl) residual_block<true>(L, wave, lane, cc, 5, cx.bd_luma, qp_y + 6 * (cx.bd_luma - 8), 0, (fl & UF_TS_LUMA) != 0, 0, cx.sl_tab + sl_off + 1008, first_raw);
      else residual_block<false>(L, wave, lane, cc, 5, cx.bd_luma, qp_y + 6 * (cx.bd_luma - 8), 0, (fl & UF_TS_LUMA) != 0, 0, nullptr, first_raw);
    } else {
      const int off_c = 6 * (cx.bd_chroma - 8);
      const int qpi = clip3(-off_c, 57, qp_y + (c == 1 ? cx.cb_off : cx.cr_off));
      const int qpc = chroma_qp(qpi, cfi_p != 1);
      if (use_sl) residual_block<true>(L, wave, lane, cc, 5, cx.bd_chroma, qpc + off_c, 0, (ipm & (c == 1 ? 64 : 128)) != 0, 0, cx.sl_tab + 2048 + (c - 1) * 1024, first_raw);
      else residual_block<false>(L, wave, lane, cc, 5, cx.bd_chroma, qpc + off_c, 0, (ipm & (c == 1 ? 64 : 128)) != 0, 0, nullptr, first_raw);
    }
  }
}

void launch_residual(const FilterArgs& a, int n_pics, int max_ctbs, bool general_chroma, hipStream_t s)
{
  if (n_pics <= 0 || max_ctbs <= 0) return;
  if (general_chroma) hipLaunchKernelGGL(k_residual<true>, dim3(max_ctbs, n_pics), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_residual<false>, dim3(max_ctbs, n_pics), dim3(256), 0, s, a);
}

}  // namespace hipdec
