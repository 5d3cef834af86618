// recon_kernel.hip — intra sample prediction + residual add: the reconstruction wavefront.
//
// Stands in for libde265's intra-prediction / reconstruction stage behind de265_decode()
// (reference call site libheif/plugins/decoder_libde265.cc:402).  ITU-T H.265 8.4.4.2 (reference sample
// availability + substitution, [1 2 1] / strong smoothing, planar / DC / angular prediction with their edge
// filters).  The residuals were already produced in place of the coefficients by residual_kernel.hip.
//
// MI355X mapping
//   * intra prediction makes every block depend on its left / above / above-right neighbours, so the
//     parallelism is the 2-CTB-lag wavefront over CTB rows, times luma / chroma (they predict independently):
//     one 64-lane wavefront per CTB-row chain of the luma plane and one per chain of the chroma planes, for every
//     picture of the batch, all running concurrently.  The chroma wave reconstructs Cb on lanes 0..31 and Cr on
//     lanes 32..63 with ONE instruction stream (block geometry, neighbour availability, substitution pattern and
//     prediction mode are shared; only samples, residuals and coded-block flags differ).  Tickets are handed out
//     row by row so a wave's predecessor (the row above, same planes) always holds an earlier ticket, i.e. is
//     resident or finished.
//   * the CTB being reconstructed lives in LDS (tile(s) + top / left borders + reference-sample line(s), one packed
//     word per 4x4 unit, a bitmap of the units decoded so far), so gathering, substitution, smoothing and
//     prediction never touch HBM; the finished CTB leaves LDS once with row-contiguous stores.
//   * the kernel is VALU-issue bound (a 4x4 block uses 16 of the lanes but costs whole wave instructions), so the
//     block path avoids quarter-rate multiplies (v_mul_i32_i24), constant-memory tables (packed immediates) and
//     anything that waits on global memory besides the prefetched residual.
//   * row-to-row hand-off without fences (cdna guide, Guideline 16 form R1): the bottom sample row of every
//     CTB goes to a per-picture line buffer with write-through (sc1) stores, is drained with s_waitcnt and
//     announced with one relaxed agent-scope progress store; the row below polls that word and reads its
//     top border from the line buffer with sc1 loads.  The reconstruction planes themselves are plain
//     stores (only the next kernel reads them).
//   * HBM traffic per luma pixel: 1.5*s written + <= 3 B residual and 0.3 B unit maps read.
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "kernels.h"

namespace hipdec {

// 7 waves per SIMD (<= 72 VGPRs): the reconstruction wavefront hides its LDS / HBM latencies with resident waves
#ifndef HIPDEC_HOST_EMU
#ifndef HIPDEC_RECON_OCC
#define HIPDEC_RECON_OCC 7   // (measurement builds: tools/ab_variant.sh rocc6 -DHIPDEC_RECON_OCC=6 ...)
#endif
#define RECON_OCCUPANCY __attribute__((amdgpu_waves_per_eu(HIPDEC_RECON_OCC, 8)))
#else
#define RECON_OCCUPANCY
#endif

namespace {

// intraPredAngle / invAngle (8.4.4.2.6, tables 8-4 and 8-5) are functions of the distance k of the mode from the pure horizontal (10) or vertical (26)
// direction.  The mode of a block is wave-uniform (a scalar register), so a table indexed by it is ONE scalar load from the constant cache whose
// latency hides behind the reference-sample gather (the same table indexed per lane would be a vector load and a vmcnt(0) wait - which also waits
// for the residual prefetch - in every angular block).  Entry: intraPredAngle in the low byte (signed), invAngle in the high half (signed; 0 where
// the angle is not negative); it replaces ~ 17 scalar instructions of mode arithmetic per angular block.
struct AngleTable {
  int32_t v[64];
  constexpr AngleTable() : v{}
  {
    constexpr int mag[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
    constexpr int inv[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
    for (int mode = 2; mode <= 34; mode++) {
      const bool vertical = mode >= 18;
      const int dm = mode - (vertical ? 26 : 10), k = dm < 0 ? -dm : dm;
      const int angle = ((dm < 0) == vertical) ? -mag[k] : mag[k];
      const int inv_angle = angle < 0 ? -inv[k] : 0;
      v[mode] = (int32_t)(((uint32_t)angle & 255u) | ((uint32_t)inv_angle << 16));
    }
  }
};
__constant__ AngleTable k_angles;
// 8.4.4.2.3: the reference samples of a block are filtered when minDistVerHor = Min(|mode - 26|, |mode - 10|) exceeds intraHorVerDistThres[nTbS] (7 / 1 / 0
// for 8 / 16 / 32; planar counts with its mode number 0, DC never, 4x4 never): one bit per mode (bits 35..63 as the formula gives for them)
constexpr uint64_t smooth_mode_mask(int n)
{
  uint64_t m = 0;
  for (int mode = 0; mode < 64; mode++) {
    if (mode == 1 || n == 4) continue;
    int d1 = mode - 26, d2 = mode - 10;
    d1 = d1 < 0 ? -d1 : d1; d2 = d2 < 0 ? -d2 : d2;
    const int thres = n == 8 ? 7 : (n == 16 ? 1 : 0);
    if ((d1 < d2 ? d1 : d2) > thres) m |= 1ull << mode;
  }
  return m;
}

template <typename Pix>
struct ReconLds {
  Pix tile[64 * 64];        // the CTB of this wave's component
  uint32_t top_raw[72];     // words of the line buffer covering x_ctb - 1 .. x_ctb + 2 * ctb - 1 (+ one pad word in front)
  Pix left[128];            // right column of the previous CTB (the chroma pair: Cb at [0, 64), Cr at [64, 128))
  uint16_t refbuf0[134];    // reference samples in scan order, one pad element in front (an unused weight-0 tap may read index -1); filtered in place (8.4.4.2.3)
  // availability of the neighbourhood in 4x4-luma units, one row of bits per unit row: row uy + 1, bit ux + 1 for the
  // units ux, uy in [-1, 2 * units_per_side): row 0 / bit 0 are the borders owned by the neighbouring CTBs, rows and bits
  // past the CTB stay 0 (not decoded yet), a unit of the CTB is set when its block has been reconstructed
  uint64_t avrow[33];
  // per 4x4-luma unit (z order): log2 TU size | UM_OUTSIDE / UM_INVALID | UF_* flags << 8 | intra mode of this component << 16 |
  // unit x << 24 | unit y << 28
  uint32_t m_unit[256];
};

constexpr uint32_t UM_OUTSIDE = 16u, UM_INVALID = 32u;   // unit-word bits 4 / 5 (ReconLds::m_unit): the unit lies outside the picture / carries an impossible transform size

__device__ __forceinline__ uint32_t compact1by1(uint32_t v)
{
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu;
  return v;
}
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
// full-rate 24-bit multiply (v_mul_i32_i24); a plain `*` becomes the quarter-rate v_mul_lo_u32 unless the compiler can
// prove the operand ranges.  All operands here are sample values, block coordinates or table entries (< 2^17).
__device__ __forceinline__ int mul24(int a, int b)
{
#ifndef HIPDEC_HOST_EMU
  int r;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a * b;
#endif
}
// The workgroup is ONE wave: lanes only need their LDS traffic drained before they read each other's values.  (A
// __syncthreads() would also wait for the outstanding global loads, i.e. serialise the residual prefetch.)
__device__ __forceinline__ void lds_sync()
{
#ifndef HIPDEC_HOST_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();
}
// bits into a word of the availability map: a DS atomic without return value (a plain |= is a read, a wait and a write)
__device__ __forceinline__ void lds_or(uint64_t* p, uint64_t bits)
{
  __hip_atomic_fetch_or(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// all of this wave's global stores have left it (the CPU-test build of tests/emu orders them with a fence instead)
__device__ __forceinline__ void drain_stores()
{
#ifndef HIPDEC_HOST_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}

struct Ctx {
  int lane;
  int ctbc, lg_ctbc;  // CTB size in component samples (and its log2)
  int ush;            // component samples -> 4x4-luma units: >> ush (2 for luma, 1 for 4:2:0 chroma)
  int ushy;           // the chroma pair: rows -> 4x4-luma units (1 for 4:2:0, 2 for 4:2:2 whose chroma is not subsampled vertically); columns: >> 1
  int lg_ctbh;        // the chroma pair: log2 of the CTB height in chroma rows (the tiles are 1 << lg_ctbc wide and 1 << lg_ctbh tall)
  int bit_depth, maxv;
  int luma;           // c_idx == 0: DC / horizontal / vertical boundary filters (8.4.4.2.6) apply
  int smooth;         // reference-sample filtering (8.4.4.2.3) applies: c_idx == 0 or ChromaArrayType == 3
  int strong;         // sps strong_intra_smoothing_enabled_flag (c_idx == 0 only)
  int cip;            // constrained_intra_pred_flag of a picture with P / B slices: inter coded units never become "available" for intra prediction
};

// A transform block of a coding unit that is NOT intra coded (P pictures): the prediction samples are in the reconstruction plane already
// (k_mc, inter_kernels.hip); the block enters the LDS tile with its residual added, so that intra blocks next to it predict from it and the
// CTB leaves LDS as a whole.  LW lanes (64, or 32 per half of the chroma pair) cover the block; `pred` points at the block in the plane.
// the units of an inter coded block become available to the intra blocks behind them (not with constrained_intra_pred_flag: their samples stay "not
// available for intra prediction", 8.4.4.2.2)
template <typename Pix>
__device__ __forceinline__ void mark_inter_available(ReconLds<Pix>& L, const Ctx& C, int xb, int yb, int log2n, int ushx, int ushy)
{
  if (!C.cip) {
    const int n = 1 << log2n;
    const int kx = n >> ushx, rows = n >> ushy;    // units per row of the block (>= 1), unit rows
    if (C.lane < (rows > 0 ? rows : 1)) lds_or(&L.avrow[(yb >> ushy) + 1 + C.lane], ((1ull << (kx > 0 ? kx : 1)) - 1ull) << ((xb >> ushx) + 1));
  }
  lds_sync();
}

template <typename Pix>
__device__ __forceinline__ void reconstruct_inter_block(ReconLds<Pix>& L, const Ctx& C, Pix* tile, const Pix* pred, uint32_t pstride, int xb, int yb, int log2n, int cbf,
                                                        const int16_t* res, int l, int LW, int ushx, int ushy)
{
  const int n = 1 << log2n, nn = n * n, lg_ctbc = C.lg_ctbc, maxv = C.maxv;
  for (int idx = l; idx < nn; idx += LW) {
    const int x = idx & (n - 1), y = idx >> log2n;
    int v = (int)pred[(size_t)y * pstride + x];
    if (cbf) v = clip3(0, maxv, v + (int)res[idx]);
    tile[((yb + y) << lg_ctbc) + xb + x] = (Pix)v;
  }
  mark_inter_available<Pix>(L, C, xb, yb, log2n, ushx, ushy);
}

// One transform block: prediction (+ residual) into the LDS tile.
//   (xb, yb): block origin inside the CTB in component samples; log2n: block size
template <typename Pix>
__device__ __forceinline__ void reconstruct_block(ReconLds<Pix>& L, const Ctx& C, const Pix* top, int xb, int yb, int log2n, int mode, int cbf, const int16_t* res)
{
  const int lane = C.lane;
  const int n = 1 << log2n, n2 = 2 * n, N = 4 * n + 1, nn = n * n;
  const int lg_ctbc = C.lg_ctbc, maxv = C.maxv;
  Pix* tile = L.tile;
  uint16_t* ref0 = L.refbuf0 + 1;

  // the block's residual is requested from HBM first, so that its latency hides behind the prediction
  // (lane l owns samples l, l + 64, ...; the first 4 cover blocks up to 16x16, a 32x32 block reads the rest late)
  int rp0 = 0, rp1 = 0, rp2 = 0, rp3 = 0;
  if (cbf) {
    if (lane < nn) rp0 = res[lane];
    if (nn > 64) { rp1 = res[lane + 64]; rp2 = res[lane + 128]; rp3 = res[lane + 192]; }
  }
  // the residual of the current iteration is always rp0: the four prefetched values rotate (a register array indexed by
  // the iteration would live in scratch memory), a 32x32 block refills the free slot four iterations ahead
#define NEXT_RES(it) do { rp0 = rp1; rp1 = rp2; rp2 = rp3; rp3 = (cbf && (it) + 4 < iters) ? (int)res[lane + 64 * ((it) + 4)] : 0; } while (0)

  // ---- 8.4.4.2.2 reference samples: gather + availability ----
  // scan order e: left column bottom-up (e < 2n), corner (e = 2n), top row left to right.  Lanes take e = lane + 64 j for
  // the J = 1 (2 for 32x32) full passes; blocks of 16x16 and up have one more sample (e = 64 J = 4n, the last one of the
  // top row), which all lanes handle uniformly.
  const int J = n == 32 ? 2 : 1;
  uint64_t m[2] = {0, 0};
  int av[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 2; j++) {
    if (j >= J) continue;               // wave-uniform
    const int e = lane + 64 * j;
    const int is_left = e < n2;
    const int px = is_left ? -1 : e - n2 - 1, py = is_left ? n2 - 1 - e : -1;
    const int X = xb + px, Y = yb + py;
    int a = 0;
    if (e < N) a = (int)((L.avrow[(Y >> C.ush) + 1] >> ((X >> C.ush) + 1)) & 1u);
    if (a) {
      const Pix* src = Y < 0 ? &top[X + 1] : (X < 0 ? &L.left[Y] : &tile[(Y << lg_ctbc) + X]);
      ref0[e] = (uint16_t)*src;
    }
    av[j] = a;
    m[j] = __ballot(a);
  }
  int ax = 1;          // availability of the extra sample (blocks below 16x16 have none: counts as present)
  const int has_x = n >= 16;
  if (has_x) {
    const int X = xb + n2 - 1, Y = yb - 1;
    ax = (int)((L.avrow[(Y >> C.ush) + 1] >> ((X >> C.ush) + 1)) & 1u);
    if (ax && lane == 0) ref0[N - 1] = (uint16_t)(Y < 0 ? top[X + 1] : tile[(Y << lg_ctbc) + X]);
  }
  lds_sync();
  // ---- substitution process, only where something is missing (wave-uniform) ----
  const int n_av = __popcll(m[0]) + __popcll(m[1]) + (has_x ? ax : 0);
  if (n_av != N) {
    if (n_av == 0) {
      const uint16_t half = (uint16_t)(1 << (C.bit_depth - 1));
      for (int e = lane; e < N; e += 64) ref0[e] = half;
    } else {
      // an unavailable sample takes the nearest available one below it in scan order, those below the first available
      // one take that one
      const int hi0 = m[0] ? 63 - __clzll((long long)m[0]) : -1;
      const int hi1 = m[1] ? 127 - __clzll((long long)m[1]) : -1;
      const int first = m[0] ? __ffsll((long long)m[0]) - 1 : (m[1] ? 63 + __ffsll((long long)m[1]) : N - 1);
      int val[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (j >= J) continue;
        const int e = lane + 64 * j;
        if (e < N && !av[j]) {
          const uint64_t mm = m[j] & ((1ull << lane) - 1ull);   // available samples below this lane's
          const int src = mm ? 64 * j + 63 - __clzll((long long)mm) : (j == 1 && hi0 >= 0 ? hi0 : first);
          val[j] = ref0[src];
        }
      }
      int valx = 0;
      if (has_x && !ax) valx = ref0[hi1 >= 0 ? hi1 : hi0];   // something below is available (n_av > 0)
      // (reads hit available positions, writes unavailable ones: no barrier needed in between)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (j >= J) continue;
        const int e = lane + 64 * j;
        if (e < N && !av[j]) ref0[e] = (uint16_t)val[j];
      }
      if (has_x && !ax && lane == 0) ref0[N - 1] = (uint16_t)valx;
    }
    lds_sync();
  }
  // accessors in scan order: left column p[-1][k-1] = ref[2n - k], top row p[k-1][-1] = ref[2n + k]
  const uint16_t* ref = ref0;
  // ---- 8.4.4.2.3 smoothing of the reference samples (luma; chroma too with ChromaArrayType 3) ----
  if (C.smooth && n != 4) {
    constexpr uint64_t k8 = smooth_mode_mask(8), k16 = smooth_mode_mask(16), k32 = smooth_mode_mask(32);
    const uint64_t filtered_modes = n == 8 ? k8 : (n == 16 ? k16 : k32);
    if ((filtered_modes >> (mode & 63)) & 1ull) {
      int strong = 0;
      if (C.strong && n == 32) {
        const int c0 = ref[n2], tl = ref[0], tm = ref[n], rt = ref[N - 1], rm = ref[n2 + n];  // corner, p[-1][63], p[-1][31], p[63][-1], p[31][-1]
        int a1 = c0 + rt - 2 * rm, a2 = c0 + tl - 2 * tm;
        a1 = a1 < 0 ? -a1 : a1; a2 = a2 < 0 ? -a2 : a2;
        strong = a1 < (1 << (C.bit_depth - 5)) && a2 < (1 << (C.bit_depth - 5));
      }
      // in place: every lane filters its (at most three) samples into registers, then all write back
      int fv[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int e = lane + 64 * j;
        if (64 * j >= N) continue;      // wave-uniform
        if (e < N) {
          int v;
          if (e == 0 || e == N - 1) v = ref0[e];
          else if (strong) {
            if (e == n2) v = ref0[n2];
            else if (e < n2) { const int y = 63 - e; v = ((63 - y) * ref0[n2] + (y + 1) * ref0[0] + 32) >> 6; }
            else { const int x = e - n2 - 1; v = ((63 - x) * ref0[n2] + (x + 1) * ref0[N - 1] + 32) >> 6; }
          } else v = (ref0[e - 1] + 2 * ref0[e] + ref0[e + 1] + 2) >> 2;
          fv[j] = v;
        }
      }
      lds_sync();
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int e = lane + 64 * j;
        if (64 * j >= N) continue;
        if (e < N) ref0[e] = (uint16_t)fv[j];
      }
      lds_sync();
    }
  }
#define RL(k) ((int)ref[n2 - (k)])
#define RT(k) ((int)ref[n2 + (k)])
  // ---- prediction 8.4.4.2.4 - 8.4.4.2.6, residual add, store into the tile ----
  const int edge = C.luma && n < 32;  // DC / horizontal / vertical boundary smoothing
  const int iters = nn > 64 ? nn >> 6 : 1;
  Pix* dst0 = &tile[(yb << lg_ctbc) + xb];
  if (mode == 0) {
    const int tr = RT(n + 1), bl = RL(n + 1);
    for (int it = 0; it < iters; it++) {
      const int idx = lane + 64 * it, x = idx & (n - 1), y = idx >> log2n;
      if (idx < nn) {
        int v = (mul24(n - 1 - x, RL(y + 1)) + mul24(x + 1, tr) + mul24(n - 1 - y, RT(x + 1)) + mul24(y + 1, bl) + n) >> (log2n + 1);
        if (cbf) v = (cbf & UF_PCM) ? rp0 : clip3(0, maxv, v + rp0);   // a PCM unit's "residual" is the sample itself
        dst0[(y << lg_ctbc) + x] = (Pix)v;
      }
      NEXT_RES(it);
    }
  } else if (mode == 1) {
    int part = lane < n ? RT(lane + 1) + RL(lane + 1) : 0;
    for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o);      // n <= 32 lanes carry a term: lanes 0..31 hold the sum
    const int dc_val = (__builtin_amdgcn_readfirstlane(part) + n) >> (log2n + 1);
    for (int it = 0; it < iters; it++) {
      const int idx = lane + 64 * it, x = idx & (n - 1), y = idx >> log2n;
      if (idx < nn) {
        int v = dc_val;
        if (edge) {
          if (x == 0 && y == 0) v = (RL(1) + 2 * dc_val + RT(1) + 2) >> 2;
          else if (y == 0) v = (RT(x + 1) + 3 * dc_val + 2) >> 2;
          else if (x == 0) v = (RL(y + 1) + 3 * dc_val + 2) >> 2;
        }
        if (cbf) v = (cbf & UF_PCM) ? rp0 : clip3(0, maxv, v + rp0);   // a PCM unit's "residual" is the sample itself
        dst0[(y << lg_ctbc) + x] = (Pix)v;
      }
      NEXT_RES(it);
    }
  } else {
    // angular: with the roles of the two reference arms swapped for the horizontal modes (2..17) both families read
    // main(k) = ref[2n + s k], side(k) = ref[2n - s k]  (s = +1 vertical, -1 horizontal)
    const int vertical = mode >= 18;
    const int s = vertical ? 1 : -1;
    const int32_t ae = k_angles.v[mode & 63];                // modes 11..25 point up-left: negative angle, taps projected with invAngle
    const int angle = (int)(int8_t)(ae & 255), inv_angle = ae >> 16;
    const int pure = edge && angle == 0;                     // pure vertical / horizontal (modes 26 / 10) with boundary smoothing
    for (int it = 0; it < iters; it++) {
      const int idx = lane + 64 * it, x = idx & (n - 1), y = idx >> log2n;
      if (idx < nn) {
        const int a = vertical ? x : y, b = vertical ? y : x;   // along / across the main arm
        const int t = mul24(b + 1, angle), i_idx = t >> 5, i_fact = t & 31;
        const int k0 = a + i_idx + 1, k1 = k0 + 1;
        // taps with a negative index come from the other arm, projected with invAngle (24-bit multiplies: full rate)
        const int p0 = -((mul24(k0, inv_angle) + 128) >> 8), p1 = -((mul24(k1, inv_angle) + 128) >> 8);
        const int r0 = ref[n2 + mul24(s, k0 >= 0 ? k0 : p0)];
        const int r1 = ref[n2 + mul24(s, k1 >= 0 ? k1 : p1)];
        int v = (mul24(32 - i_fact, r0) + mul24(i_fact, r1) + 16) >> 5;      // i_fact == 0 gives r0
        if (pure && a == 0) v = clip3(0, maxv, (int)ref[n2 + s] + (((int)ref[n2 - mul24(s, b + 1)] - (int)ref[n2]) >> 1));
        if (cbf) v = (cbf & UF_PCM) ? rp0 : clip3(0, maxv, v + rp0);   // a PCM unit's "residual" is the sample itself
        dst0[(y << lg_ctbc) + x] = (Pix)v;
      }
      NEXT_RES(it);
    }
  }
#undef RL
#undef RT
#undef NEXT_RES
  // ---- the block's units are decoded now ----
  {
    const int k = n >> C.ush;    // units per side (>= 1)
    if (lane < k) lds_or(&L.avrow[(yb >> C.ush) + 1 + lane], ((1ull << k) - 1ull) << ((xb >> C.ush) + 1));
  }
  lds_sync();
}

// The same for 4x4 and 8x8 blocks (LG = 2, 3: three quarters of a picture's block calls at the benchmark's QP), REGISTER-RESIDENT: the 4n + 1 <= 33
// reference samples stay one per lane - gathered, substituted ([8.4.4.2.2]: the nearest available sample below in scan order) and smoothed with
// wave shuffles - and the prediction reads them with ds_bpermute instead of through a reference line in LDS: no line to write, no passes over 64 j + lane,
// no LDS round trips between the stages (the general form's four wave-level LDS fences per block were most of a small block's latency, its run-time
// sizes a third of its instructions).  Every sample of the block is one lane (16 or 64 of them).  Same arithmetic as reconstruct_block.
template <typename Pix, int LG>
__device__ __forceinline__ void reconstruct_block_reg(ReconLds<Pix>& L, const Ctx& C, const Pix* top, int xb, int yb, int mode, int cbf, const int16_t* res)
{
  constexpr int n = 1 << LG, n2 = 2 * n, N = 4 * n + 1, nn = n * n;
  static_assert(N <= 64 && nn <= 64, "one lane per reference sample and per block sample");
  const int lane = C.lane, lg_ctbc = C.lg_ctbc, maxv = C.maxv;
  Pix* tile = L.tile;
  int rp0 = 0;
  if (cbf && lane < nn) rp0 = res[lane];   // requested first: its latency hides behind the prediction
  // ---- reference samples, scan order e = lane: left column bottom-up (e < 2n), corner (e = 2n), top row left to right ----
  const int e = lane;
  const int is_left = e < n2;
  const int X = xb + (is_left ? -1 : e - n2 - 1), Y = yb + (is_left ? n2 - 1 - e : -1);
  int a = 0;
  if (e < N) a = (int)((L.avrow[(Y >> C.ush) + 1] >> ((X >> C.ush) + 1)) & 1u);
  int r = 0;
  if (a) r = (int)(Y < 0 ? top[X + 1] : (X < 0 ? L.left[Y] : tile[(Y << lg_ctbc) + X]));
  const uint64_t m = __ballot(a);
  if (m != (1ull << N) - 1ull) {   // (wave-uniform) substitution
    if (m == 0) r = 1 << (C.bit_depth - 1);
    else {
      const int first = __ffsll((long long)m) - 1;
      const uint64_t below = m & ((1ull << lane) - 1ull);   // available samples below this lane's
      const int src = a ? lane : (below ? 63 - __clzll((long long)below) : first);
      r = __shfl(r, src);
    }
  }
  // ---- 8.4.4.2.3 smoothing (8x8; 4x4 blocks are never filtered): [1 2 1] on everything but the two ends ----
  if (LG == 3 && C.smooth) {
    constexpr uint64_t k8 = smooth_mode_mask(8);
    if ((k8 >> (mode & 63)) & 1ull) {
      const int lo = __shfl(r, lane - 1), hi = __shfl(r, lane + 1);
      const int f = (lo + 2 * r + hi + 2) >> 2;
      r = (e == 0 || e >= N - 1) ? r : f;
    }
  }
  // ref[k] for a lane-varying k: __shfl(r, k); left column p[-1][k-1] = ref[2n - k], top row p[k-1][-1] = ref[2n + k]
  const int idx = lane & (nn - 1), x = idx & (n - 1), y = idx >> LG;   // (lanes >= nn compute on a copy of a block sample and store nothing)
  const int edge = C.luma;   // DC / horizontal / vertical boundary smoothing (n < 32)
  int v;
  if (mode == 0) {
    const int tr = __shfl(r, n2 + n + 1), bl = __shfl(r, n2 - n - 1), rl = __shfl(r, n2 - (y + 1)), rt = __shfl(r, n2 + (x + 1));
    v = (mul24(n - 1 - x, rl) + mul24(x + 1, tr) + mul24(n - 1 - y, rt) + mul24(y + 1, bl) + n) >> (LG + 1);
  } else if (mode == 1) {
    // sum of p[-1][0 .. n-1] and p[0 .. n-1][-1]: the lanes e in [n, 2n) and (2n, 3n]
    int part = ((e >= n && e < n2) || (e > n2 && e <= n2 + n)) ? r : 0;
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    const int dc_val = (part + n) >> (LG + 1);
    const int rl = __shfl(r, n2 - (y + 1)), rt = __shfl(r, n2 + (x + 1));
    v = dc_val;
    if (edge) {
      if (x == 0 && y == 0) v = (rl + 2 * dc_val + rt + 2) >> 2;
      else if (y == 0) v = (rt + 3 * dc_val + 2) >> 2;
      else if (x == 0) v = (rl + 3 * dc_val + 2) >> 2;
    }
  } else {
    const int vertical = mode >= 18;
    const int s = vertical ? 1 : -1;
    const int32_t ae = k_angles.v[mode & 63];
    const int angle = (int)(int8_t)(ae & 255), inv_angle = ae >> 16;
    const int a_ = vertical ? x : y, b_ = vertical ? y : x;   // along / across the main arm
    const int t = mul24(b_ + 1, angle), i_idx = t >> 5, i_fact = t & 31;
    const int k0 = a_ + i_idx + 1, k1 = k0 + 1;
    const int p0 = -((mul24(k0, inv_angle) + 128) >> 8), p1 = -((mul24(k1, inv_angle) + 128) >> 8);
    const int r0 = __shfl(r, n2 + mul24(s, k0 >= 0 ? k0 : p0));
    const int r1 = __shfl(r, n2 + mul24(s, k1 >= 0 ? k1 : p1));
    v = (mul24(32 - i_fact, r0) + mul24(i_fact, r1) + 16) >> 5;
    if (edge && angle == 0) {   // pure vertical / horizontal (modes 26 / 10) with boundary smoothing (wave-uniform)
      const int c0 = __shfl(r, n2), m1 = __shfl(r, n2 + s), o1 = __shfl(r, n2 - mul24(s, b_ + 1));
      if (a_ == 0) v = clip3(0, maxv, m1 + ((o1 - c0) >> 1));
    }
  }
  if (cbf) v = (cbf & UF_PCM) ? rp0 : clip3(0, maxv, v + rp0);   // a PCM unit's "residual" is the sample itself
  if (lane < nn) tile[((yb + y) << lg_ctbc) + xb + x] = (Pix)v;
  // ---- the block's units are decoded now ----
  {
    const int k = n >> C.ush;    // units per side (>= 1)
    if (lane < k) lds_or(&L.avrow[(yb >> C.ush) + 1 + lane], ((1ull << k) - 1ull) << ((xb >> C.ush) + 1));
  }
  lds_sync();
}

// Cb and Cr blocks of one position TOGETHER: lanes 0..31 work on Cb, lanes 32..63 on Cr.  Geometry, availability, substitution
// pattern and prediction mode are the same for both (4:2:0 chroma has neither reference smoothing nor boundary filters), only
// the samples, the residuals and the coded-block flags differ — so one instruction stream reconstructs both blocks.
//   LDS: the Cr tile sits behind the Cb tile (each 1 << lg_ctbc wide, 1 << lg_ctbh tall), left borders at left[0..63] / left[64..127], reference lines
//   at refbuf0[1 + 67 h ...];  `top`, `cbf` and `res` are this lane's half's.
template <typename Pix>
__device__ __forceinline__ void reconstruct_chroma_pair(ReconLds<Pix>& L, const Ctx& C, const Pix* top, int xb, int yb, int log2n, int mode, int cbf,
                                                        const int16_t* res)
{
  const int lane = C.lane, l = lane & 31, h = lane >> 5;
  const int n = 1 << log2n, n2 = 2 * n, N = 4 * n + 1, nn = n * n;
  const int lg_ctbc = C.lg_ctbc, maxv = C.maxv;
  Pix* tile = L.tile + (h << (lg_ctbc + C.lg_ctbh));
  const Pix* left = L.left + h * 64;
  uint16_t* ref0 = L.refbuf0 + 1 + h * 67;
  const int ushy = C.ushy;
  const int iters = nn > 32 ? nn >> 5 : 1;   // 32 samples per pass and half: 1 / 2 / 8 passes for 4x4 / 8x8 / 16x16

  int rp0 = 0, rp1 = 0, rp2 = 0, rp3 = 0;
  if (cbf) {
    if (l < nn) rp0 = res[l];
    if (nn > 32) rp1 = res[l + 32];
    if (nn > 64) { rp2 = res[l + 64]; rp3 = res[l + 96]; }
  }
#define NEXT_RES(it) do { rp0 = rp1; rp1 = rp2; rp2 = rp3; rp3 = (cbf && (it) + 4 < iters) ? (int)res[l + 32 * ((it) + 4)] : 0; } while (0)

  // ---- reference samples: e = l + 32 j for J = 1 (2 for 16x16) passes; blocks from 8x8 up have the extra sample e = 4n ----
  const int J = n == 16 ? 2 : 1;
  uint64_t m0 = 0;        // availability over e < 64 (identical in both halves: taken from the Cb lanes)
  int av[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 2; j++) {
    if (j >= J) continue;
    const int e = l + 32 * j;
    const int is_left = e < n2;
    const int px = is_left ? -1 : e - n2 - 1, py = is_left ? n2 - 1 - e : -1;
    const int X = xb + px, Y = yb + py;
    int a = 0;
    if (e < N) a = (int)((L.avrow[(Y >> ushy) + 1] >> ((X >> 1) + 1)) & 1u);
    if (a) {
      const Pix* src = Y < 0 ? &top[X + 1] : (X < 0 ? &left[Y] : &tile[(Y << lg_ctbc) + X]);
      ref0[e] = (uint16_t)*src;
    }
    av[j] = a;
    m0 |= (__ballot(a) & 0xffffffffull) << (32 * j);
  }
  int ax = 1;
  const int has_x = n >= 8;
  if (has_x) {
    const int X = xb + n2 - 1, Y = yb - 1;
    ax = (int)((L.avrow[(Y >> ushy) + 1] >> ((X >> 1) + 1)) & 1u);
    if (ax && l == 0) ref0[N - 1] = (uint16_t)(Y < 0 ? top[X + 1] : tile[(Y << lg_ctbc) + X]);
  }
  lds_sync();
  const int n_av = __popcll(m0) + (has_x ? ax : 0);
  if (n_av != N) {
    if (n_av == 0) {
      const uint16_t half = (uint16_t)(1 << (C.bit_depth - 1));
      for (int e = l; e < N; e += 32) ref0[e] = half;
    } else {
      const int hi0 = m0 ? 63 - __clzll((long long)m0) : -1;
      const int first = m0 ? __ffsll((long long)m0) - 1 : N - 1;
      int val[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (j >= J) continue;
        const int e = l + 32 * j;
        if (e < N && !av[j]) {
          const uint64_t mm = m0 & ((1ull << e) - 1ull);
          val[j] = ref0[mm ? 63 - __clzll((long long)mm) : first];
        }
      }
      int valx = 0;
      if (has_x && !ax) valx = ref0[hi0];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (j >= J) continue;
        const int e = l + 32 * j;
        if (e < N && !av[j]) ref0[e] = (uint16_t)val[j];
      }
      if (has_x && !ax && l == 0) ref0[N - 1] = (uint16_t)valx;
    }
    lds_sync();
  }
  const uint16_t* ref = ref0;
#define RL(k) ((int)ref[n2 - (k)])
#define RT(k) ((int)ref[n2 + (k)])
  Pix* dst0 = &tile[(yb << lg_ctbc) + xb];
  if (mode == 0) {
    const int tr = RT(n + 1), bl = RL(n + 1);
    for (int it = 0; it < iters; it++) {
      const int idx = l + 32 * it, x = idx & (n - 1), y = idx >> log2n;
      if (idx < nn) {
        int v = (mul24(n - 1 - x, RL(y + 1)) + mul24(x + 1, tr) + mul24(n - 1 - y, RT(x + 1)) + mul24(y + 1, bl) + n) >> (log2n + 1);
        if (cbf) v = (cbf & UF_PCM) ? rp0 : clip3(0, maxv, v + rp0);   // a PCM unit's "residual" is the sample itself
        dst0[(y << lg_ctbc) + x] = (Pix)v;
      }
      NEXT_RES(it);
    }
  } else if (mode == 1) {
    int part = l < n ? RT(l + 1) + RL(l + 1) : 0;
    for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o);      // every lane of a half ends up with its half's sum
    const int dc_val = (part + n) >> (log2n + 1);
    for (int it = 0; it < iters; it++) {
      const int idx = l + 32 * it, x = idx & (n - 1), y = idx >> log2n;
      if (idx < nn) {
        int v = dc_val;
        if (cbf) v = (cbf & UF_PCM) ? rp0 : clip3(0, maxv, v + rp0);   // a PCM unit's "residual" is the sample itself
        dst0[(y << lg_ctbc) + x] = (Pix)v;
      }
      NEXT_RES(it);
    }
  } else {
    const int vertical = mode >= 18;
    const int s = vertical ? 1 : -1;
    const int32_t ae = k_angles.v[mode & 63];
    const int angle = (int)(int8_t)(ae & 255), inv_angle = ae >> 16;
    for (int it = 0; it < iters; it++) {
      const int idx = l + 32 * it, x = idx & (n - 1), y = idx >> log2n;
      if (idx < nn) {
        const int a = vertical ? x : y, b = vertical ? y : x;
        const int t = mul24(b + 1, angle), i_idx = t >> 5, i_fact = t & 31;
        const int k0 = a + i_idx + 1, k1 = k0 + 1;
        const int p0 = -((mul24(k0, inv_angle) + 128) >> 8), p1 = -((mul24(k1, inv_angle) + 128) >> 8);
        const int r0 = ref[n2 + mul24(s, k0 >= 0 ? k0 : p0)];
        const int r1 = ref[n2 + mul24(s, k1 >= 0 ? k1 : p1)];
        int v = (mul24(32 - i_fact, r0) + mul24(i_fact, r1) + 16) >> 5;
        if (cbf) v = (cbf & UF_PCM) ? rp0 : clip3(0, maxv, v + rp0);   // a PCM unit's "residual" is the sample itself
        dst0[(y << lg_ctbc) + x] = (Pix)v;
      }
      NEXT_RES(it);
    }
  }
#undef RL
#undef RT
#undef NEXT_RES
  {
    const int k = n >> 1, rows = n >> ushy;    // units per row of the block, unit rows (4:2:2: a 4x4 block is two units wide and one tall)
    if (lane < rows) lds_or(&L.avrow[(yb >> ushy) + 1 + lane], ((1ull << k) - 1ull) << ((xb >> 1) + 1));
  }
  lds_sync();
}

// The 4x4 blocks of the chroma pair register-resident (reconstruct_block_reg's scheme in each half of the wave: 17 reference samples and 16 block
// samples per half; 4:2:0 / 4:2:2 chroma has neither reference smoothing nor boundary filters)
template <typename Pix>
__device__ __forceinline__ void reconstruct_chroma_pair_reg4(ReconLds<Pix>& L, const Ctx& C, const Pix* top, int xb, int yb, int mode, int cbf, const int16_t* res)
{
  constexpr int LG = 2, n = 4, n2 = 8, N = 17, nn = 16;
  const int lane = C.lane, l = lane & 31, hb = lane & 32;   // hb: the first lane of this lane's half
  const int lg_ctbc = C.lg_ctbc, maxv = C.maxv, ushy = C.ushy;
  Pix* tile = L.tile + ((lane >> 5) << (lg_ctbc + C.lg_ctbh));
  const Pix* left = L.left + (lane >> 5) * 64;
  int rp0 = 0;
  if (cbf && l < nn) rp0 = res[l];
  const int e = l;
  const int is_left = e < n2;
  const int X = xb + (is_left ? -1 : e - n2 - 1), Y = yb + (is_left ? n2 - 1 - e : -1);
  int a = 0;
  if (e < N) a = (int)((L.avrow[(Y >> ushy) + 1] >> ((X >> 1) + 1)) & 1u);
  int r = 0;
  if (a) r = (int)(Y < 0 ? top[X + 1] : (X < 0 ? left[Y] : tile[(Y << lg_ctbc) + X]));
  const uint32_t m = (uint32_t)__ballot(a);   // availability is the same in both halves: the Cb lanes' bits
  if (m != (1u << N) - 1u) {
    if (m == 0) r = 1 << (C.bit_depth - 1);
    else {
      const int first = __ffsll((long long)m) - 1;
      const uint32_t below = m & ((1u << l) - 1u);
      const int src = a ? l : (below ? 63 - __clzll((long long)below) : first);
      r = __shfl(r, hb + src);
    }
  }
  const int idx = l & (nn - 1), x = idx & (n - 1), y = idx >> LG;
  int v;
  if (mode == 0) {
    const int tr = __shfl(r, hb + n2 + n + 1), bl = __shfl(r, hb + n2 - n - 1), rl = __shfl(r, hb + n2 - (y + 1)), rt = __shfl(r, hb + n2 + (x + 1));
    v = (mul24(n - 1 - x, rl) + mul24(x + 1, tr) + mul24(n - 1 - y, rt) + mul24(y + 1, bl) + n) >> (LG + 1);
  } else if (mode == 1) {
    int part = ((e >= n && e < n2) || (e > n2 && e <= n2 + n)) ? r : 0;
    for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o);      // every lane of a half ends up with its half's sum
    v = (part + n) >> (LG + 1);
  } else {
    const int vertical = mode >= 18;
    const int s = vertical ? 1 : -1;
    const int32_t ae = k_angles.v[mode & 63];
    const int angle = (int)(int8_t)(ae & 255), inv_angle = ae >> 16;
    const int a_ = vertical ? x : y, b_ = vertical ? y : x;
    const int t = mul24(b_ + 1, angle), i_idx = t >> 5, i_fact = t & 31;
    const int k0 = a_ + i_idx + 1, k1 = k0 + 1;
    const int p0 = -((mul24(k0, inv_angle) + 128) >> 8), p1 = -((mul24(k1, inv_angle) + 128) >> 8);
    const int r0 = __shfl(r, hb + n2 + mul24(s, k0 >= 0 ? k0 : p0));
    const int r1 = __shfl(r, hb + n2 + mul24(s, k1 >= 0 ? k1 : p1));
    v = (mul24(32 - i_fact, r0) + mul24(i_fact, r1) + 16) >> 5;
  }
  if (cbf) v = (cbf & UF_PCM) ? rp0 : clip3(0, maxv, v + rp0);   // a PCM unit's "residual" is the sample itself
  if (l < nn) tile[((yb + y) << lg_ctbc) + xb + x] = (Pix)v;
  {
    const int k = n >> 1, rows = n >> ushy;    // units per row of the block, unit rows (4:2:2: a 4x4 block is two units wide and one tall)
    if (lane < (rows > 0 ? rows : 1)) lds_or(&L.avrow[(yb >> ushy) + 1 + lane], ((1ull << k) - 1ull) << ((xb >> 1) + 1));
  }
  lds_sync();
}

}  // namespace

// The CTB rows of one wave.  DUAL = false: ONE plane at luma resolution, 64 lanes per block — the luma plane (plane 0) or, for a 4:4:4
// picture, the Cb / Cr plane (plane 1 / 2: same geometry as luma, the chroma modes / flags / bit depth, no boundary filters).
// DUAL = true: the 4:2:0 Cb (lanes 0..31) and Cr (lanes 32..63) side by side — h / l below are a lane's half and its index inside
// the half, LW the lanes one component has.
template <typename Pix, bool DUAL, bool INTER>
__device__ __forceinline__ void recon_rows(const ReconArgs& A, const ReconWave& wd, ReconLds<Pix>& L, int lane, int plane)
{
  constexpr int ES = (int)sizeof(Pix);
  constexpr int LW = DUAL ? 32 : 64;
  constexpr int PPW = 4 / ES;            // pixels per 32-bit word
  const int h = DUAL ? lane >> 5 : 0, l = DUAL ? lane & 31 : lane;
  const int comp = DUAL ? 1 + h : plane; // colour component this lane works on
  const int c_idx = DUAL ? 1 : plane;    // progress words / wave table: 0 = luma, 1 = the chroma pair (4:4:4: 1 = Cb, 2 = Cr)
  const bool chroma = DUAL || plane != 0;
  const PicParams& P = A.pics[wd.pic];
  const int sub = DUAL ? 2 : 1;                                   // horizontal subsampling of this wave's plane(s)
  const int suby = (DUAL && P.chroma_format_idc != 2) ? 2 : 1;    // vertical: the 4:2:2 pair has the luma rows
  const int ctb = 1 << P.log2_ctb, ctbc = ctb / sub, ctbch = ctb / suby;   // CTB width / height in component samples
  const int units = 1 << P.units_per_ctb_log2;
  const CtbInfo* ctb_info = (const CtbInfo*)(A.arena + P.off_ctb_info);
  Pix* rec = (Pix*)(A.arena + P.off_rec[comp]);
  const uint32_t stride = P.rec_stride[c_idx] / sizeof(Pix);           // Cb and Cr planes share their geometry
  // line buffer: bottom sample row of every CTB row of this component, row stride = rec_stride
  uint32_t* line = (uint32_t*)(A.arena + P.off_line[comp]);
  const uint32_t line_words = P.rec_stride[c_idx] / 4;
  const int16_t* coeff = (const int16_t*)(A.arena + P.off_coeff[comp]);
  const size_t off_mode = chroma ? P.off_u_ipmc : P.off_u_ipm, off_size = P.off_u_size, off_flags = P.off_u_flags;
  const bool is_inter = INTER && P.is_inter != 0;   // (INTER = false: the build for batches without P pictures, the benchmarked path)
  const uint32_t mode_mask = is_inter ? 127u : (chroma ? 255u : 63u);   // u_ipm carries the chroma transform-skip flags in bits 6 and 7; P pictures: bit 6 = inter
  int err = 0;
  uint32_t my_row = 0;
  Ctx C;
  C.lane = lane; C.ctbc = ctbc; C.lg_ctbc = P.log2_ctb - (DUAL ? 1 : 0); C.ush = DUAL ? 1 : 2;
  C.ushy = suby == 2 ? 1 : 2; C.lg_ctbh = P.log2_ctb - (suby == 2 ? 1 : 0);
  C.bit_depth = chroma ? P.bit_depth_chroma : P.bit_depth_luma; C.maxv = (1 << C.bit_depth) - 1;
  C.luma = !chroma; C.smooth = !DUAL; C.strong = !chroma && P.strong_intra_smoothing;
  C.cip = (INTER && P.is_inter && P.constrained_intra_pred) ? 1 : 0;
  const bool from_plane = INTER && is_inter && A.inter_from_plane != 0;
  const int Wc = DUAL ? P.cwidth : P.width, Hc = DUAL ? P.cheight : P.height;   // component plane size in samples
  const int pic_w = P.width, pic_h = P.height, ctb_w = P.ctb_w, ctb_h = P.ctb_h, log2_ctb = P.log2_ctb;
  const int side = 1 << (log2_ctb - 2);                                         // 4x4-luma units per CTB side
  const int cbf_bit = DUAL ? (h ? UF_CBF_CR : UF_CBF_CB) : (plane == 0 ? UF_CBF_LUMA : (plane == 1 ? UF_CBF_CB : UF_CBF_CR));
  // this lane's slices of the shared LDS arrays
  Pix* tile = L.tile + (DUAL ? h << (C.lg_ctbc + C.lg_ctbh) : 0);
  Pix* left = L.left + (DUAL ? h * 64 : 0);
  uint32_t* top_raw = L.top_raw + (DUAL ? h * 36 : 0);
  const int lg_wpr = C.lg_ctbc - (ES == 1 ? 2 : 1);      // log2 of the words per tile row
  const int wpr = 1 << lg_wpr;

  for (int cy = (int)wd.first_row; cy < ctb_h && !err; cy += (int)wd.stride) {
  my_row = wd.base_row + (uint32_t)cy;     // batch row index
  uint32_t* my_progress = A.row_progress + (size_t)my_row * 3 + c_idx;
  const uint32_t* up_progress = my_progress - 3;
  for (int cx = 0; cx < ctb_w && !err; cx++) {
    const int ctb_rs = cy * ctb_w + cx;
    const CtbInfo ci = ctb_info[ctb_rs];
    const int x_ctb = cx << log2_ctb, y_ctb = cy << log2_ctb;   // luma origin of the CTB
    const int xc0 = x_ctb / sub;  // component x of the CTB
    // ---- a CTB of a P / B picture without a single intra coded unit (the common one): k_mc has left it complete in the plane, nothing is predicted here,
    //      nothing above it is read - its bottom row goes to the line buffer, its right column becomes the next CTB's left border, done.  (Round 5 walked
    //      its 256 units marking each available: 25 us per CTB, and the pixel step of a 720p picture is a 42-CTB wavefront of those.)
    if (INTER && from_plane && x_ctb + ctb <= pic_w && y_ctb + ctb <= pic_h) {
      const size_t mbase = (size_t)ctb_rs * units;
      bool all_inter = true;
      for (int i = lane * 4; i < units; i += 256) all_inter = all_inter && (*(const uint32_t*)(A.arena + P.off_u_ipmc + mbase + i) & 0x40404040u) == 0x40404040u;
      if (__ballot(!all_inter) == 0) {
        const int yc0 = y_ctb / suby;
        uint32_t* dst = line + (size_t)cy * line_words + (size_t)xc0 * ES / 4;
        const Pix* bottom = rec + (size_t)(yc0 + ctbch - 1) * stride + xc0;
        for (int i = l; i < wpr; i += LW) __hip_atomic_store(dst + i, *(const uint32_t*)&bottom[i * PPW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = l; i < ctbch; i += LW) left[i] = rec[(size_t)(yc0 + i) * stride + xc0 + ctbc - 1];
        lds_sync();
        drain_stores();
        if (lane == 0) __hip_atomic_store(my_progress, (uint32_t)(cx + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
    }
    // ---- wait for the row above: above-right CTB done (or the row end) ----
    const Pix* top = (const Pix*)(top_raw + 1);
    if (cy > 0) {
      uint32_t need = cx == 0 ? wd.start_lag : (uint32_t)(cx + 2);   // 2 = above-right CTB; a larger start distance decouples the rows
      if (need > (uint32_t)ctb_w) need = (uint32_t)ctb_w;
      uint32_t spins = 0;
      while (__hip_atomic_load(up_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1u << 22) || __hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { err = DEV_ERR_TIMEOUT; break; }
      }
      if (err) break;
      // top border samples xc0 - 1 .. xc0 + 2 * ctbc - 1 from the line buffer (write-through data: sc1 loads)
      const int b0 = (xc0 - 1) * ES;                              // byte offset of the above-left sample
      const int start = b0 < 0 ? 0 : (b0 & ~3);
      int endb = (xc0 + 2 * ctbc) * ES;
      if (endb > (int)(line_words * 4)) endb = (int)(line_words * 4);
      const int nwords = (endb - start + 3) >> 2;
      const uint32_t* src = line + (size_t)(cy - 1) * line_words + (start >> 2);
      for (int i = l; i < nwords; i += LW) top_raw[1 + i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      top = (const Pix*)((const uint8_t*)(top_raw + 1) + (b0 - start));   // top[0] = above-left sample
    }
    // ---- stage the CTB's maps, one packed word per unit ----
    {
      const size_t base = (size_t)ctb_rs * units;
      int inter_units = 0;
      for (int i = lane * 4; i < units; i += 256) {
        const uint32_t sz = *(const uint32_t*)(A.arena + off_size + base + i);
        const uint32_t fl = *(const uint32_t*)(A.arena + off_flags + base + i);
        uint32_t md = *(const uint32_t*)(A.arena + off_mode + base + i);
        if (is_inter) {   // P picture: bit 6 of a unit's mode byte = the unit is not intra coded (u_ipmc carries it; the luma wave reads u_ipm)
          const uint32_t pc = *(const uint32_t*)(A.arena + P.off_u_ipmc + base + i);
          md = (md & 0x3f3f3f3fu) | (pc & 0x40404040u);
          inter_units |= (int)(pc & 0x40404040u);
        }
        const uint32_t ux = compact1by1((uint32_t)i), uy = compact1by1((uint32_t)i >> 1);   // i is a multiple of 4: units i..i+3 are a 2x2 quad
#pragma nounroll
        for (int k = 0; k < 4; k++) {
          // units right of / below the picture are skipped, a transform size outside 4..32 is a broken map: both are decided here by the lanes
          const uint32_t tbk = (sz >> (8 * k)) & 15u, uxk = ux + (k & 1), uyk = uy + (k >> 1);
          const uint32_t special = (x_ctb + (int)uxk * 4 >= pic_w || y_ctb + (int)uyk * 4 >= pic_h) ? UM_OUTSIDE : ((tbk - 2u) > 3u ? UM_INVALID : 0u);
          L.m_unit[i + k] = tbk | special | (((fl >> (8 * k)) & 255u) << 8) | (((md >> (8 * k)) & mode_mask) << 16) | (uxk << 24) | (uyk << 28);
        }
      }
      // availability rows: the borders come from the neighbouring CTBs (slice / tile / picture limits are in ci.avail,
      // units right of or below the picture never become available), the CTB's own units are set block by block
      // constrained_intra_pred_flag (P / B pictures): units of the neighbouring CTBs that are not intra coded stay unavailable too.  Lane j looks at
      // the unit above the CTB at column j - 1 (above-left, above, above-right) and, lanes 1 .. side, at the unit left of unit row lane - 1
      uint64_t up_inter = 0;
      bool left_inter = false;
      if (INTER && C.cip) {
        const uint8_t* pc = A.arena + P.off_u_ipmc;
        auto unit_is_inter = [&](int gx, int gy) -> bool {   // (gx, gy): 4x4-luma unit in the picture
          if (gx < 0 || gy < 0 || gx * 4 >= pic_w || gy * 4 >= pic_h) return false;
          const int cxx = gx >> (log2_ctb - 2), cyy = gy >> (log2_ctb - 2);
          uint32_t x = (uint32_t)(gx & (side - 1)), y = (uint32_t)(gy & (side - 1));
          x = (x | (x << 2)) & 0x33u; x = (x | (x << 1)) & 0x55u; y = (y | (y << 2)) & 0x33u; y = (y | (y << 1)) & 0x55u;
          return (pc[((size_t)(cyy * ctb_w + cxx) << (2 * (log2_ctb - 2))) + (x | (y << 1))] & UM_INTER) != 0;
        };
        const bool up = lane <= 2 * side && unit_is_inter((x_ctb >> 2) - 1 + lane, (y_ctb >> 2) - 1);
        up_inter = __ballot(up);
        left_inter = lane >= 1 && lane <= side && unit_is_inter((x_ctb >> 2) - 1, (y_ctb >> 2) + lane - 1);
      }
      if (lane < 33) {
        const int usz = 4 / sub, uszy = 4 / suby;                       // component samples per unit, horizontally / vertically
        uint64_t row = 0;
        if (lane == 0) {
          int nu = (Wc - xc0 + usz - 1) / usz;                          // units up to the right picture edge
          nu = nu > 2 * side ? 2 * side : nu;
          const int n_up = nu < side ? nu : side;
          if (ci.avail & AV_UPLEFT) row |= 1ull;
          if (ci.avail & AV_UP) row |= ((1ull << n_up) - 1ull) << 1;
          if ((ci.avail & AV_UPRIGHT) && nu > side) row |= ((1ull << (nu - side)) - 1ull) << (side + 1);
          row &= ~up_inter;
        } else if (lane <= side && (ci.avail & AV_LEFT) && y_ctb / suby + (lane - 1) * uszy < Hc && !left_inter) row = 1ull;
        L.avrow[lane] = row;
      }
      // inter_from_plane: k_mc left the inter coded units complete in the plane (prediction + residual) - the tile starts as a copy of the CTB (the
      // positions of the intra blocks hold whatever was there: they are written before anything reads them) and the walk below only marks those units
      if (INTER && from_plane && __ballot(inter_units != 0) != 0) {
        const int yc0 = y_ctb / suby;
        const int y0 = l >> lg_wpr, xw = l & (wpr - 1);
        uint32_t off = (uint32_t)(yc0 + y0) * stride + (uint32_t)(xc0 + xw * PPW);
        const uint32_t step = ((uint32_t)LW >> lg_wpr) * stride;
        for (int i = l; i < wpr * ctbch; i += LW, off += step) *(uint32_t*)&tile[i * PPW] = *(const uint32_t*)&rec[off];
      }
    }
    lds_sync();

    // ---- the blocks of the CTB in z-scan order ----
    const int16_t* res_base = coeff + (size_t)ctb_rs * (ctb * ctb / (sub * suby));
    int z = 0;
    while (z < units) {
      const uint32_t w = L.m_unit[z];
      if (w & (UM_OUTSIDE | UM_INVALID)) {
        if (w & UM_OUTSIDE) { z++; continue; }
        err = DEV_ERR_SYNTAX; break;
      }
      const int ux = (int)((w >> 24) & 15u), uy = (int)(w >> 28);
      const int tb = (int)(w & 15u), fl = (int)((w >> 8) & 255u), mode = (int)((w >> 16) & 255u);
      // the chroma pair: the 4x4 chroma blocks of four 4x4 luma TUs hang off the quad's 4th unit (their flags are there) - straight to it
      if (DUAL && tb == 2 && (z & 3) != 3) { z |= 3; continue; }
      if (INTER && (mode & 64) && from_plane) {   // a unit of an inter coded CU, complete in the plane and in the tile already
        if (!DUAL) mark_inter_available<Pix>(L, C, ux * 4, uy * 4, tb, 2, 2);
        else if (tb > 2 || (z & 3) == 3) {
          const int quad = tb == 2;
          mark_inter_available<Pix>(L, C, (quad ? (ux & ~1) : ux) * 2, (quad ? (uy & ~1) : uy) * 2, quad ? 2 : tb - 1, 1, 1);
        }
      } else if (INTER && (mode & 64)) {   // the same with the residual added here: prediction from the plane + residual
        if (!DUAL) {
          const Pix* pred = rec + (size_t)(y_ctb + uy * 4) * stride + (size_t)(x_ctb + ux * 4);
          reconstruct_inter_block<Pix>(L, C, tile, pred, stride, ux * 4, uy * 4, tb, fl & cbf_bit, res_base + z * 16, lane, 64, 2, 2);
        } else if (tb > 2 || (z & 3) == 3) {
          const int quad = tb == 2;
          const int zc = quad ? (z & ~3) : z, cux = quad ? (ux & ~1) : ux, cuy = quad ? (uy & ~1) : uy;
          const int lgc = quad ? 2 : tb - 1;
          const Pix* pred = rec + (size_t)(y_ctb / 2 + cuy * 2) * stride + (size_t)(xc0 + cux * 2);
          reconstruct_inter_block<Pix>(L, C, tile, pred, stride, cux * 2, cuy * 2, lgc, fl & cbf_bit, res_base + zc * 4, l, 32, 1, 1);
        }
      } else if (!DUAL) {
        if (tb == 2) reconstruct_block_reg<Pix, 2>(L, C, top, ux * 4, uy * 4, mode, fl & (cbf_bit | UF_PCM), res_base + z * 16);
        else if (tb == 3) reconstruct_block_reg<Pix, 3>(L, C, top, ux * 4, uy * 4, mode, fl & (cbf_bit | UF_PCM), res_base + z * 16);
        else reconstruct_block<Pix>(L, C, top, ux * 4, uy * 4, tb, mode, fl & (cbf_bit | UF_PCM), res_base + z * 16);
      } else if (tb > 2 || (z & 3) == 3) {
        // the 4x4 chroma blocks of four 4x4 luma TUs hang off the 4th unit (their flags are there); they sit at the quad's origin
        const int quad = tb == 2;
        const int zc = quad ? (z & ~3) : z, cux = quad ? (ux & ~1) : ux, cuy = quad ? (uy & ~1) : uy;
        const int lgc = quad ? 2 : tb - 1;
        if (suby == 2 && lgc == 2) reconstruct_chroma_pair_reg4<Pix>(L, C, top, cux * 2, cuy * 2, mode, fl & (cbf_bit | UF_PCM), res_base + zc * 4);
        else if (suby == 2) reconstruct_chroma_pair<Pix>(L, C, top, cux * 2, cuy * 2, lgc, mode, fl & (cbf_bit | UF_PCM), res_base + zc * 4);
        else {
          // 4:2:2: two blocks one above the other, the upper one first (the lower one predicts from it); the lower one's flags sit in unit z ^ 1
          const int fl2 = (int)((L.m_unit[z ^ 1] >> 8) & 255u);
#pragma nounroll
          for (int lower = 0; lower < 2; lower++)
            reconstruct_chroma_pair<Pix>(L, C, top, cux * 2, cuy * 4 + (lower << lgc), lgc, mode, (lower ? fl2 : fl) & (cbf_bit | UF_PCM),
                                         res_base + zc * 8 + (lower << (2 * lgc)));
        }
      }
      z += 1 << (2 * (tb - 2));
    }
    lds_sync();

    // ---- write the CTB out (planes are allocated CTB-aligned, so no edge guards), hand its bottom row to the
    //      row below and keep its right column as the next CTB's left border ----
    {
      const int yc0 = y_ctb / suby;
      // LW lanes cover LW / wpr whole tile rows per pass (wpr <= 32 is a power of two): the plane offset advances by a
      // wave-uniform step, no per-pass multiplies or divisions
      const int y0 = l >> lg_wpr, xw = l & (wpr - 1);
      uint32_t off = (uint32_t)(yc0 + y0) * stride + (uint32_t)(xc0 + xw * PPW);
      const uint32_t step = ((uint32_t)LW >> lg_wpr) * stride;
      for (int i = l; i < wpr * ctbch; i += LW, off += step)
        *(uint32_t*)&rec[off] = *(const uint32_t*)&tile[i * PPW];     // tile rows are wpr words: word i of the tile
      uint32_t* dst = line + (size_t)cy * line_words + (size_t)xc0 * ES / 4;
      for (int i = l; i < wpr; i += LW)
        __hip_atomic_store(dst + i, *(const uint32_t*)&tile[(ctbch - 1) * ctbc + i * PPW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lds_sync();
      for (int i = l; i < ctbch; i += LW) left[i] = tile[i * ctbc + ctbc - 1];
    }
    lds_sync();
    drain_stores();   // the line-buffer stores have left this wave
    if (lane == 0) __hip_atomic_store(my_progress, (uint32_t)(cx + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }
  if (err && lane == 0) atomicCAS((int*)A.status, 0, err | (int)(0x40000000u) | (int)(my_row << 8));
}

template <typename Pix, bool INTER>
__device__ __forceinline__ void recon_wave(const ReconArgs& A)
{
  __shared__ ReconLds<Pix> L;
  const int lane = threadIdx.x;
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(A.ticket, 1u);
  const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  if (ticket >= A.num_waves) return;
  if (__hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // failed parse: maps are garbage
  const ReconWave wd = A.waves[ticket];
  const int cfi = A.pics[wd.pic].chroma_format_idc;
  if (wd.comp == 0 || cfi == 3) { if (wd.comp == 0 || cfi) recon_rows<Pix, false, INTER>(A, wd, L, lane, (int)wd.comp); }   // luma; 4:4:4: one wave per plane
  else if (cfi) recon_rows<Pix, true, INTER>(A, wd, L, lane, 1);                            // 4:2:0: comp 1 = Cb and Cr together
}

// 8-bit pictures: 7 waves per SIMD by registers (<= 72 VGPRs), 26 per CU by LDS (6072 B per wave, handed out in 512 B granules).  What was measured
// on the way here (1024 4K stills, profiles/r04_recon_variants.txt): the kernel is bound by instruction issue, not by latency hiding - a compact unit
// map that allowed 28 waves per CU bought nothing, 6 / 5 waves per SIMD with fewer spills cost 1 % / 10 %; fewer scalar instructions per block did
// pay (78.0 -> 72.8 ms: mode tables, DS atomics on the availability map, units classified at staging, the chroma pair skipping to a quad's 4th
// unit); size-specialised copies of the block functions gained 5 % before those changes and nothing after them (74.8 ms; twice the code), so the
// block functions keep their run-time sizes.  The 16-bit variant is limited by its 10 KB of LDS per wave either way.
__global__ __launch_bounds__(64) RECON_OCCUPANCY void k_recon8(ReconArgs A) { recon_wave<uint8_t, false>(A); }
// (16-bit samples: 10.3 KB of LDS per wave allow 15 waves per CU; at 128 VGPRs - 4 waves per SIMD - the kernel runs 512 Main10 4K stills in 47.3 ms, unconstrained
//  (132 VGPRs, 3 waves) in 50.0, at 5 waves per SIMD (spills) in 49.9: GPU call 64)
#ifndef HIPDEC_HOST_EMU
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_recon16(ReconArgs A) { recon_wave<uint16_t, false>(A); }
#else
__global__ __launch_bounds__(64) void k_recon16(ReconArgs A) { recon_wave<uint16_t, false>(A); }
#endif
// batches with P pictures (sequence tracks): inter coded blocks take their prediction from the plane (k_mc) instead of the intra predictor
__global__ __launch_bounds__(64) void k_recon8_inter(ReconArgs A) { recon_wave<uint8_t, true>(A); }
__global__ __launch_bounds__(64) void k_recon16_inter(ReconArgs A) { recon_wave<uint16_t, true>(A); }

void launch_recon(const ReconArgs& a, bool wide, hipStream_t s, bool inter)
{
  if (!a.num_waves) return;
  if (inter) {
    if (wide) hipLaunchKernelGGL(k_recon16_inter, dim3(a.num_waves), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_recon8_inter, dim3(a.num_waves), dim3(64), 0, s, a);
  } else if (wide) hipLaunchKernelGGL(k_recon16, dim3(a.num_waves), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(k_recon8, dim3(a.num_waves), dim3(64), 0, s, a);
}

}  // namespace hipdec
