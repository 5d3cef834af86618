// residual_kernel.hip — dequantisation + inverse DCT/DST of every coded transform block, in place.
//
// Stands in for libde265's scaling / transform stages behind de265_decode() (reference call site
// libheif/plugins/decoder_libde265.cc:402).  ITU-T H.265 8.6.2-8.6.4: scaling with flat lists,
// transform skip, cu_transquant_bypass, DST-VII 4x4 for intra luma, DCT 4..32, 16-bit intermediate clip.
//
// MI355X mapping: residuals do not depend on prediction, so they are taken off the intra-prediction
// dependency chain entirely: this kernel runs over ALL transform blocks of the batch in parallel (one
// 256-thread workgroup per CTB, its four waves take the CTB's coded blocks round-robin) and overwrites each
// TU-contiguous int16 coefficient block with its int16 residual block; the reconstruction wavefront
// (recon_kernel.hip) then only adds.  Per block the wave stages the dequantised coefficients in LDS and runs
// the two 1-D passes as n MACs per output sample against the LDS-resident 32-point matrix, skipping the
// all-zero high-frequency rows / columns that dominate real content.  Integer butterflies, no MFMA: the
// largest block is 32x32x32 int16 MACs with 16-bit clipping between the passes (not a dense contraction
// worth matrix cores).  Traffic: reads and writes 2 B per coded sample (<= 3 B per luma pixel each way).
#include <hip/hip_runtime.h>
#include <cstddef>
#include "hevc_device.h"
#include "kernels.h"

namespace hipdec {
namespace {

// the 33 magnitudes of the 32-point DCT matrix (8.6.4.2: M32[j][i] = +-c[(2 i + 1) j mod 128 folded]) and the 4x4 DST-VII matrix (row-major),
// behind one another: the table fill of the kernel's prologue reads entry (is DST ? 33 + i : k) with ONE unconditional load per pass, so
// that the passes' loads are all in flight together (a load in each arm of a branch is waited for where the arms meet)
__constant__ int8_t r_tab[33 + 16] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0,
                                      29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
// small per-block lookups as packed immediates: a __constant__ array indexed at run time is a global load plus a wait
// in front of every block
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
  const int v = ((level * f + rnd) >> rs) << ls;
  return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}
// ... with a scaling list: f = m[x][y] * levelScale <= 255 * 72, level * f < 2^30; the left-shift case (q - b <= 3) is clamped
// first so that it cannot wrap before the 16-bit clip
__device__ __forceinline__ int scale_level_sl(int level, int f, int rs, int ls, int rnd)
{
  int v = (level * f + rnd) >> rs;
  v = v < -(1 << 27) ? -(1 << 27) : (v > (1 << 27) ? (1 << 27) : v);
  v <<= ls;
  return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}
__device__ __forceinline__ int level_scale(int r)         // levelScale[qP % 6] (8.6.3)
{
  constexpr uint64_t kS = 40ull | (45ull << 8) | (51ull << 16) | (57ull << 24) | (64ull << 32) | (72ull << 40);
  return (int)((kS >> (r * 8)) & 255u);
}

// Transposed, j-contiguous operands so that both 1-D passes are chains of v_dot2_i32_i16 (two MACs per instruction,
// one 32-bit LDS read per operand pair).  Rows are padded by 2 samples: consecutive lanes then hit distinct banks.
constexpr int RPAD = 2;
constexpr int LIST_N = 896;
struct ResLds {
  alignas(4) int16_t et32[32 * 32], et16[16 * 16], et8[8 * 8], et4[4 * 4], est4[4 * 4];   // E^T[i][j] per size, DST last (filled as ONE array of 1376 entries)
  alignas(4) int16_t blk[4][32 * (32 + RPAD)];   // per wave: scaled levels, TRANSPOSED: blk[x][j] = d[j][x]
  alignas(4) int16_t tmp[4][32 * (32 + RPAD)];   // per wave: first-stage output tmp[y][j]
  uint8_t m_size[256], m_flags[256], m_ipm[256];
  int8_t m_qp[256];
  // block lists, entry = z | component << 8 (z = the unit that carries the TU's flags), in ONE array: blocks larger than 4x4 fill it from the
  // front (list[e]), 4x4 blocks — four of them per wave pass — from the back (list4(i) = list[LIST_N - 1 - i]).  An 8x8 luma area brings at
  // most 3 entries of the first kind or 12 of the second (4:4:4; 4:2:0: 1 + 0 or 4 + 2), so the two never meet: <= 768 entries in all.
  // (LDS is sized to the byte for 7 workgroups per CU: 160 KB / 7 in 512 B granules)
  uint16_t list[LIST_N];
  uint32_t count, count4;
  uint32_t colmask[4];   // per wave: columns of the current block that hold a nonzero level (DS atomic OR of the lanes' bits; zero between blocks)
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
  int16_t* blk = L.blk[wave];
  int16_t* tmp = L.tmp[wave];
  const int16_t* et = dst ? L.est4 : (log2n == 2 ? L.et4 : (log2n == 3 ? L.et8 : (log2n == 4 ? L.et16 : L.et32)));
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

// Four independent 4x4 blocks per wave pass: lane = 16 g + 4 y + x works on sample (y, x) of block g.  entry = z | c << 8.
// (8.6.2 - 8.6.4 as in residual_block; both 1-D stages are two v_dot2 per output, no nonzero-extent bookkeeping)
// the levels of the 4x4 block `entry` names (z | component << 8 | lower 4:2:2 block << 10)
__device__ __forceinline__ int16_t* quad_levels(const ResLds& L, int entry, int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr, int cfi)
{
  const bool c444 = cfi == 3, c422 = cfi == 2;
  const int z = entry & 255, c = (entry >> 8) & 3, low = c422 ? (entry >> 10) & 1 : 0;
  if (c == 0) return coef_y + z * 16;
  const int t = L.m_size[z] & 15;
  const int zc = (t > 2 || c444) ? z : (z & ~3);
  return (c == 1 ? coef_cb : coef_cr) + zc * (c444 ? 16 : (c422 ? 8 : 4)) + low * 16;
}
// first_lev: this lane's level, loaded by the caller one pass ahead (0 for an invalid / bypass block: such a block is not written either)
__device__ __forceinline__ void residual_quad(ResLds& L, int bd_luma, int bd_chroma, int cb_qp_offset, int cr_qp_offset, const uint8_t* sl_tab, int wave, int lane, int entry, bool valid,
                                              int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr, int cfi, int first_lev)
{
  const bool c444 = cfi == 3, c422 = cfi == 2;
  const int g = lane >> 4, l = lane & 15;
  const int z = entry & 255, c = (entry >> 8) & 3, low = c422 ? (entry >> 10) & 1 : 0;   // low: the lower chroma block of a 4:2:2 unit (its flags sit in unit z ^ 1)
  const int t = L.m_size[z] & 15, fl = L.m_flags[z], ipm = L.m_ipm[low ? (z ^ 1) : z], qp_y = L.m_qp[z];
  const bool bypass = (fl & UF_BYPASS) != 0;
  int16_t* coef;
  int bit_depth, qp, ts;
  if (c == 0) {
    coef = coef_y + z * 16; bit_depth = bd_luma; qp = qp_y + 6 * (bd_luma - 8); ts = (fl & UF_TS_LUMA) != 0;
  } else {
    const int zc = (t > 2 || c444) ? z : (z & ~3);
    const int off_c = 6 * (bd_chroma - 8);
    const int qpi = clip3(-off_c, 57, qp_y + (c == 1 ? cb_qp_offset : cr_qp_offset));
    const int qpc = chroma_qp(qpi, cfi != 1);
    coef = (c == 1 ? coef_cb : coef_cr) + zc * (c444 ? 16 : (c422 ? 8 : 4)) + low * 16; bit_depth = bd_chroma; qp = qpc + off_c; ts = (ipm & (c == 1 ? 64 : 128)) != 0;
  }
  const bool act = valid && !bypass;      // cu_transquant_bypass: the coefficient levels are the residual
  const int q6 = (qp * 43) >> 8;          // qp / 6 for 0 <= qp < 128
  const int bd_shift = bit_depth - 3;     // bitDepth + log2(4) - 5
  const int bd_shift2 = 20 - bit_depth;
  const bool use_sl = sl_tab != nullptr;
  const int mfac = use_sl ? (int)sl_tab[c * 336 + l + (((entry >> 11) & 1) << 11)] : 16;    // 4x4 ScalingFactor of this lane's coefficient (8.6.4.2); inter coded units: the second block of tables
  const int f = mfac * level_scale(qp - 6 * q6);
  const int sh_r = q6 < bd_shift ? bd_shift - q6 : 0, sh_l = q6 < bd_shift ? 0 : q6 - bd_shift, rnd = sh_r ? 1 << (sh_r - 1) : 0;
  const int lev = act ? first_lev : 0;
  const int d = use_sl ? scale_level_sl(lev, f, sh_r, sh_l, rnd) : scale_level(lev, f, sh_r, sh_l, rnd);
  int16_t* blk = L.blk[wave] + g * 16;    // blk[x][j] = d[j][x]
  int16_t* tmp = L.tmp[wave] + g * 16;    // tmp[i][x]
  const int y = l >> 2, x = l & 3;
  blk[x * 4 + y] = (int16_t)d;
  lds_sync();
  const bool use_dst = c == 0 && !((entry >> 11) & 1);   // DST-VII for the 4x4 luma blocks of intra coded units only
  const uint32_t* e = (const uint32_t*)((use_dst ? L.est4 : L.et4) + y * 4);   // E^T row of this lane's output index
  {
    // first stage, lane (i = y, x): tmp[i][x] = clip16((sum_j E[j][i] d[j][x] + 64) >> 7)
    const uint32_t* v = (const uint32_t*)(blk + x * 4);
    const int sum = dot2(e[1], v[1], dot2(e[0], v[0], 0));
    tmp[y * 4 + x] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
  }
  lds_sync();
  int r;
  {
    // second stage, lane (y, i = x): res[y][i] = (sum_j E[j][i] tmp[y][j] + rnd) >> bdShift
    const uint32_t* ei = (const uint32_t*)((use_dst ? L.est4 : L.et4) + x * 4);
    const uint32_t* v = (const uint32_t*)(tmp + y * 4);
    const int sum = dot2(ei[1], v[1], dot2(ei[0], v[0], 0));
    r = (sum + (1 << (bd_shift2 - 1))) >> bd_shift2;
  }
  if (ts) r = (d * 128 + (1 << (bd_shift2 - 1))) >> bd_shift2;   // 8.6.4.2 with transform_skip_flag: r = d << 7
  if (act) coef[l] = (int16_t)r;
  lds_sync();
}

}  // namespace

// blockIdx.x = CTB (raster) of picture blockIdx.y.  GEN = false: a build for batches of 4:0:0 / 4:2:0 pictures only (the 4:2:2 / 4:4:4 block
// addressing costs the common batch 3 % of this kernel: 84 -> 87 ms per 2048 4K stills)
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
  // Everything the prologue reads from global memory is requested FIRST and used afterwards - the CTB's unit maps (four bytes per thread) and the
  // table entries below (one byte each, six per thread): as a loop of load - wait - store the table alone was six memory latencies in a row at the
  // start of every workgroup (one workgroup per CTB, ~18 blocks per wave: the prologue is a visible part of its life).
  uint8_t u_size = 0, u_flags = 0, u_ipm = 0, u_qp = 0;
  if (tid < units) {
    u_size = A.arena[P.off_u_size + base + tid]; u_flags = A.arena[P.off_u_flags + base + tid];
    u_ipm = A.arena[P.off_u_ipm + base + tid]; u_qp = A.arena[P.off_u_qp + base + tid];
  }
  // E^T[i][j] for every transform size (8.6.4.2): E_n[j][i] = M32[j * 32/n][i], the 32-point matrix from its 33 magnitudes
  constexpr int kTabEntries = 1024 + 256 + 64 + 16 + 16, kTabPasses = (kTabEntries + 255) / 256;
  int tab_v[kTabPasses], tab_neg[kTabPasses];
#pragma unroll
  for (int it = 0; it < kTabPasses; it++) {
    const int idx = tid + 256 * it;
    const int lg = idx < 1024 ? 5 : (idx < 1280 ? 4 : (idx < 1344 ? 3 : 2));
    const int off = idx - (idx < 1024 ? 0 : (idx < 1280 ? 1024 : (idx < 1344 ? 1280 : (idx < 1360 ? 1344 : 1360))));
    const int nsz = 1 << lg, i = off >> lg, j = off & (nsz - 1);
    const int mm = j * (32 >> lg);          // row of the 32-point matrix
    int k = ((2 * i + 1) * mm) & 127;
    if (k > 64) k = 128 - k;
    const bool dst = idx >= 1360;           // est4[off]: r_dst[(off & 3) * 4 + (off >> 2)]
    const int e = dst ? 33 + ((off & 3) * 4 + ((off >> 2) & 3)) : (k <= 32 ? k : 64 - k);
    tab_neg[it] = !dst && k > 32;
    tab_v[it] = r_tab[e];                   // (idx past the tables: some valid entry, not stored)
  }
#pragma unroll
  for (int it = 0; it < kTabPasses; it++) tab_v[it] = tab_neg[it] ? -tab_v[it] : tab_v[it];
  static_assert(offsetof(ResLds, et32) == 0 && offsetof(ResLds, et16) == 2 * 1024 && offsetof(ResLds, et8) == 2 * 1280 && offsetof(ResLds, et4) == 2 * 1344 &&
                offsetof(ResLds, est4) == 2 * 1360, "the transform tables are filled as one array");
  int16_t* const tab = reinterpret_cast<int16_t*>(&L);
#pragma unroll
  for (int it = 0; it < kTabPasses; it++) {
    const int idx = tid + 256 * it;
    if (idx < kTabEntries) tab[idx] = (int16_t)tab_v[it];
  }
  const uint8_t* sl_tab = P.scaling_lists ? A.arena + P.off_scaling : nullptr;   // ScalingFactor tables of the picture
  const bool use_sl = sl_tab != nullptr;
  const bool c444 = cfi_p == 3;   // chroma blocks have the luma blocks' size and position
  const bool c422 = cfi_p == 2;   // two chroma blocks of half the luma block's size, one above the other
  if (tid == 0) { L.count = 0; L.count4 = 0; }
  if (tid < 4) L.colmask[tid] = 0;
  if (tid < units) { L.m_size[tid] = u_size; L.m_flags[tid] = u_flags; L.m_ipm[tid] = u_ipm; L.m_qp[tid] = (int8_t)u_qp; }
  __syncthreads();
  // ---- lists of coded blocks, one entry per block and component: z | c << 8 with z the unit that carries the flags (a TU's
  //      first unit; for the chroma blocks of four 4x4 luma TUs the 4th unit, where the parser leaves their flags) ----
  if (tid < units) {
    const int z = tid;
    const int ux = (int)compact1by1((uint32_t)z), uy = (int)compact1by1((uint32_t)z >> 1);
    const int x_ctb = (ctb_rs % P.ctb_w) << P.log2_ctb, y_ctb = (ctb_rs / P.ctb_w) << P.log2_ctb;
    if (x_ctb + ux * 4 < P.width && y_ctb + uy * 4 < P.height) {
      const int t = L.m_size[z] & 15, fl = L.m_flags[z];
      const int first = t >= 2 && t <= 5 && (z & ((1 << (2 * (t - 2))) - 1)) == 0;
      if (first && !(fl & UF_BYPASS)) {
        const int inter_bit = (P.is_inter && (A.arena[P.off_u_ipmc + base + z] & UM_INTER)) ? 0x800 : 0;
        if (fl & UF_CBF_LUMA) {
          // (P pictures: bit 11 = the block belongs to an inter coded unit: its 4x4 luma transform is the DCT, not the DST of intra blocks, 8.6.4.2)
          // (with scaling lists the bit also selects the matrices of inter coded units for every block size and component, Table 7-4)
          if (t == 2) L.list[LIST_N - 1 - atomicAdd(&L.count4, 1u)] = (uint16_t)(z | inter_bit);
          else L.list[atomicAdd(&L.count, 1u)] = (uint16_t)(z | inter_bit);
        }
        // chroma blocks hang off the unit that carries their flags: a block's first unit, or the 4th unit of a quad of 4x4 luma blocks.  4:2:2 has
        // two chroma blocks per unit; the lower one's flags sit in unit z ^ 1 (so in a 4:2:2 quad only the 4th unit's flags are block flags)
        if (cfi_p && !(c422 && t == 2 && (z & 3) != 3))
          for (int c = 1; c < 3; c++)
            for (int low = 0; low < (c422 ? 2 : 1); low++)
              if ((low ? L.m_flags[z ^ 1] : fl) & (c == 1 ? UF_CBF_CB : UF_CBF_CR)) {
                const uint16_t entry = (uint16_t)(z | (c << 8) | (low << 10) | inter_bit);
                if (t <= (c444 ? 2 : 3)) L.list[LIST_N - 1 - atomicAdd(&L.count4, 1u)] = entry;
                else L.list[atomicAdd(&L.count, 1u)] = entry;
              }
      }
    }
  }
  __syncthreads();
  const int count = (int)L.count, count4 = (int)L.count4;
  const CtbInfo ci = ((const CtbInfo*)(A.arena + P.off_ctb_info))[ctb_rs];
  const SliceParams sl = ((const SliceParams*)(A.arena + P.off_slices))[ci.slice_idx];
  int16_t* coef_y = (int16_t*)(A.arena + P.off_coeff[0]) + (size_t)ctb_rs * ctb * ctb;
  const int cc_shift = c444 ? 0 : (c422 ? 1 : 2);
  // (two named pointers, selected by comparison: an array indexed by the component goes through memory and comes back as a generic pointer -
  //  FLAT loads and stores, which also count in lgkmcnt, so that every LDS wait waited for the global traffic too)
  int16_t* const coef_cb = (int16_t*)(A.arena + P.off_coeff[1]) + (size_t)ctb_rs * ((ctb * ctb) >> cc_shift);
  int16_t* const coef_cr = (int16_t*)(A.arena + P.off_coeff[2]) + (size_t)ctb_rs * ((ctb * ctb) >> cc_shift);
  const int bd_luma = P.bit_depth_luma, bd_chroma = P.bit_depth_chroma, cb_off = sl.cb_qp_offset, cr_off = sl.cr_qp_offset;
  // 4x4 blocks, four per wave pass
  {
    auto quad_entry = [&](int q, bool& valid) -> int { const int idx = q * 4 + (lane >> 4); valid = idx < (int)count4; return valid ? (int)L.list[LIST_N - 1 - idx] : 0; };
    bool valid = false, valid_next = false;
    int entry = 0, ahead = 0;
    if (wave * 4 < (int)count4) { entry = quad_entry(wave, valid); if (valid) ahead = quad_levels(L, entry, coef_y, coef_cb, coef_cr, cfi_p)[lane & 15]; }
    for (int q = wave; q * 4 < (int)count4; q += 4) {
      const int lev = ahead, cur = entry;
      const bool cur_valid = valid;
      if ((q + 4) * 4 < (int)count4) {   // the next pass's levels are requested before this pass is worked on
        entry = quad_entry(q + 4, valid_next); valid = valid_next; ahead = 0;
        if (valid) ahead = quad_levels(L, entry, coef_y, coef_cb, coef_cr, cfi_p)[lane & 15];
      }
      residual_quad(L, bd_luma, bd_chroma, cb_off, cr_off, sl_tab, wave, lane, cur, cur_valid, coef_y, coef_cb, coef_cr, cfi_p, lev);
    }
  }
  // larger blocks, one per wave pass; the first levels of the NEXT block are requested before the current one is worked on
  auto block_of = [&](int e, int& c, int& t, int& tc, int& fl, int& ipm, int& qp_y, int& sl_off) -> int16_t* {   // list entry -> component, sizes, flags, levels
    const int z = L.list[e] & 255, low = c422 ? (L.list[e] >> 10) & 1 : 0;
    sl_off = ((L.list[e] >> 11) & 1) << 11;   // scaling lists: the tables of inter coded units follow the intra ones (P / B pictures: 2048-byte blocks)
    c = (L.list[e] >> 8) & 3;
    t = L.m_size[z] & 15; fl = L.m_flags[z]; ipm = L.m_ipm[low ? (z ^ 1) : z]; qp_y = L.m_qp[z];
    tc = c == 0 ? t : (c444 ? t : t - 1);    // log2 size of the block
    return c == 0 ? coef_y + z * 16 : (c == 1 ? coef_cb : coef_cr) + z * (c444 ? 16 : (c422 ? 8 : 4)) + (low << (2 * tc));
  };
  auto first_levels = [&](const int16_t* cc, int tc) -> uint2 {
    return lane * 4 < (1 << (2 * tc)) ? *(const uint2*)&cc[lane * 4] : make_uint2(0, 0);
  };
  int c = 0, t = 0, tc = 0, fl = 0, ipm = 0, qp_y = 0, sl_off = 0;
  int16_t* cc = nullptr;
  uint2 ahead = make_uint2(0, 0);
  if (wave < count) { cc = block_of(wave, c, t, tc, fl, ipm, qp_y, sl_off); ahead = first_levels(cc, tc); }
  for (int e = wave; e < count; e += 4) {
    const uint2 first_raw = ahead;
    int16_t* const cur = cc;
    const int cur_c = c, cur_t = t, cur_tc = tc, cur_fl = fl, cur_ipm = ipm, cur_qp = qp_y, cur_sl = sl_off;
    if (e + 4 < count) { cc = block_of(e + 4, c, t, tc, fl, ipm, qp_y, sl_off); ahead = first_levels(cc, tc); }
    if (cur_c == 0) {
      if (use_sl) residual_block<true>(L, wave, lane, cur, cur_t, bd_luma, cur_qp + 6 * (bd_luma - 8), 0, (cur_fl & UF_TS_LUMA) != 0, 0,
                                       sl_tab + cur_sl + (cur_t == 5 ? 1008 : (cur_t == 3 ? 16 : 80)), first_raw);
      else residual_block<false>(L, wave, lane, cur, cur_t, bd_luma, cur_qp + 6 * (bd_luma - 8), 0, (cur_fl & UF_TS_LUMA) != 0, 0, nullptr, first_raw);
    }
    else {
      const int off_c = 6 * (bd_chroma - 8);
      const int qpi = clip3(-off_c, 57, cur_qp + (cur_c == 1 ? cb_off : cr_off));
      const int qpc = chroma_qp(qpi, cfi_p != 1);
      if (use_sl) residual_block<true>(L, wave, lane, cur, cur_tc, bd_chroma, qpc + off_c, 0, (cur_ipm & (cur_c == 1 ? 64 : 128)) != 0, 0,
                                       cur_tc == 5 ? sl_tab + 2048 + (cur_c - 1) * 1024 : sl_tab + cur_sl + cur_c * 336 + (cur_tc == 3 ? 16 : 80), first_raw);   // 32x32 chroma: 4:4:4 only
      else residual_block<false>(L, wave, lane, cur, cur_tc, bd_chroma, qpc + off_c, 0, (cur_ipm & (cur_c == 1 ? 64 : 128)) != 0, 0, nullptr, first_raw);
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
