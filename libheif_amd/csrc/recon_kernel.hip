// recon_kernel.hip — intra sample prediction + residual add: the reconstruction wavefront.
//
// Stands in for libde265's intra-prediction / reconstruction stage behind de265_decode()
// (reference call site libheif/plugins/decoder_libde265.cc:402).  ITU-T H.265 8.4.4.2 (reference sample
// availability + substitution, [1 2 1] / strong smoothing, planar / DC / angular prediction with their edge
// filters).  The residuals were already produced in place of the coefficients by residual_kernel.hip.
//
// MI355X mapping
//   * intra prediction makes every block depend on its left / above / above-right neighbours, so the
//     parallelism is the 2-CTB-lag wavefront over CTB rows, times the colour components (Y, Cb, Cr predict
//     independently): one 64-lane wavefront per (CTB row, component) of every picture of the batch, all
//     running concurrently; tickets are handed out row by row so a wave's predecessor (the row above, same
//     component) always holds an earlier ticket, i.e. is resident or finished.
//   * the CTB being reconstructed lives in LDS (tile + top / left borders + reference-sample line), so
//     gathering, substitution, smoothing and prediction never touch HBM; the finished CTB leaves LDS once
//     with row-contiguous stores.
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


namespace {

__constant__ int8_t c_angle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                   -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
__constant__ int16_t c_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};

template <typename Pix>
struct ReconLds {
  Pix tile[64 * 64];        // the CTB of this wave's component
  uint32_t top_raw[72];     // words of the line buffer covering x_ctb - 1 .. x_ctb + 2 * ctb - 1 (+ one pad word in front)
  Pix left[64];
  uint16_t ref0[132], ref1[132];
  uint8_t m_size[256], m_flags[256], m_ipm[256], m_ipmc[256];
};

__device__ __forceinline__ uint32_t interleave4(uint32_t x, uint32_t y)
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
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
// The workgroup is ONE wave: lanes only need their LDS traffic drained before they read each other's values.  (A
// __syncthreads() would also wait for the outstanding global loads, i.e. serialise the residual prefetch.)
__device__ __forceinline__ void lds_sync()
{
#ifndef HIPDEC_HOST_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();
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

__device__ __forceinline__ int find_src(int e, uint64_t m0, uint64_t m1, uint64_t m2)
{
  const int j = e >> 6, b = e & 63;
  uint64_t mm = (j == 0 ? m0 : (j == 1 ? m1 : m2)) & ((1ull << b) - 1ull);
  if (mm) return j * 64 + 63 - __clzll((long long)mm);
  if (j >= 2 && m1) return 64 + 63 - __clzll((long long)m1);
  if (j >= 1 && m0) return 63 - __clzll((long long)m0);
  if (m0) return __ffsll((long long)m0) - 1;
  if (m1) return 64 + __ffsll((long long)m1) - 1;
  if (m2) return 128 + __ffsll((long long)m2) - 1;
  return -1;
}

struct Ctx {
  const PicParams* pp;
  int lane;
  int x_ctb, y_ctb;   // luma origin of the CTB
  int avail;          // CtbInfo.avail
  int ctb;            // CTB size in luma samples
};

// One transform block: prediction (+ residual) into the LDS tile.
//   c_idx: component; (xb, yb): block origin inside the CTB in component samples; log2n: block size
//   z_cur: z-index (4x4 luma units) of the block that defines "already decoded" for availability
template <typename Pix>
__device__ __forceinline__ void reconstruct_block(ReconLds<Pix>& L, const Ctx& C, const Pix* top, int c_idx, int xb, int yb, int log2n, int z_cur, int mode,
                                                  int cbf, const int16_t* res)
{
  const PicParams& P = *C.pp;
  const int lane = C.lane;
  const int n = 1 << log2n, n2 = 2 * n, N = 4 * n + 1;
  const int sub = c_idx ? 2 : 1;                      // 4:2:0
  const int ctbc = C.ctb / sub;                       // CTB size in component samples
  const int Wc = c_idx ? P.cwidth : P.width, Hc = c_idx ? P.cheight : P.height;
  const int x_abs0 = C.x_ctb / sub, y_abs0 = C.y_ctb / sub;
  const int bit_depth = c_idx ? P.bit_depth_chroma : P.bit_depth_luma;
  const int maxv = (1 << bit_depth) - 1;
  Pix* tile = L.tile;
  const Pix* left = L.left;

  // the block's residual is requested from HBM first, so that its latency hides behind the prediction
  // (lane l owns samples l, l + 64, ...; the first 4 cover blocks up to 16x16, a 32x32 block reads the rest late)
  int16_t rpre[4];
  const int nn = n * n;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int idx = lane + 64 * k; rpre[k] = (cbf && idx < nn) ? res[idx] : (int16_t)0; }

  // ---- 8.4.4.2.2 reference samples: gather + availability + substitution ----
  uint64_t m[3] = {0, 0, 0};
  int val[3] = {0, 0, 0}, av[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    if (64 * j >= N) continue;          // wave-uniform: small blocks have 17 / 33 / 65 reference samples
    const int e = lane + 64 * j;
    int a = 0, v = 0;
    if (e < N) {
      int px, py;
      if (e < n2) { px = -1; py = n2 - 1 - e; }
      else if (e == n2) { px = -1; py = -1; }
      else { px = e - n2 - 1; py = -1; }
      const int X = xb + px, Y = yb + py;
      const int inside = (x_abs0 + X) < Wc && (y_abs0 + Y) < Hc && (x_abs0 + X) >= 0 && (y_abs0 + Y) >= 0;
      if (Y < 0) {           // row above the CTB
        const int bit = X < 0 ? AV_UPLEFT : (X < ctbc ? AV_UP : AV_UPRIGHT);
        a = inside && (C.avail & bit);
        if (a) v = top[X + 1];
      } else if (X < 0) {    // column left of the CTB
        a = inside && Y < ctbc && (C.avail & AV_LEFT);
        if (a) v = left[Y];
      } else if (X < ctbc && Y < ctbc) {
        const int z = (int)interleave4((uint32_t)(X * sub) >> 2, (uint32_t)(Y * sub) >> 2);
        a = inside && z < z_cur;
        if (a) v = tile[Y * ctbc + X];
      }
    }
    av[j] = a; val[j] = v;
    m[j] = __ballot(a);
    if (e < N && a) L.ref0[e] = (uint16_t)v;
  }
  lds_sync();
  const int n_av = __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]);
  if (n_av != N) {          // substitution process only where something is missing (wave-uniform)
    const int any = n_av != 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (64 * j >= N) continue;
      const int e = lane + 64 * j;
      if (e < N && !av[j]) {
        int v;
        if (!any) v = 1 << (bit_depth - 1);
        else v = L.ref0[find_src(e, m[0], m[1], m[2])];
        val[j] = v;
      }
    }
    lds_sync();
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (64 * j >= N) continue;
      const int e = lane + 64 * j;
      if (e < N && !av[j]) L.ref0[e] = (uint16_t)val[j];
    }
    lds_sync();
  }
  // accessors in scan order: left column p[-1][k-1] = ref[2n - k], top row p[k-1][-1] = ref[2n + k]
  uint16_t* ref = L.ref0;
  // ---- 8.4.4.2.3 smoothing of the reference samples (luma only in 4:2:0) ----
  if (c_idx == 0 && mode != 1 && n != 4) {
    int d1 = mode - 26, d2 = mode - 10;
    d1 = d1 < 0 ? -d1 : d1; d2 = d2 < 0 ? -d2 : d2;
    const int min_dist = d1 < d2 ? d1 : d2;
    const int thres = n == 8 ? 7 : (n == 16 ? 1 : 0);
    if (min_dist > thres) {
      int strong = 0;
      if (P.strong_intra_smoothing && n == 32) {
        const int c0 = ref[n2], tl = ref[0], tm = ref[n], rt = ref[N - 1], rm = ref[n2 + n];  // corner, p[-1][63], p[-1][31], p[63][-1], p[31][-1]
        int a1 = c0 + rt - 2 * rm, a2 = c0 + tl - 2 * tm;
        a1 = a1 < 0 ? -a1 : a1; a2 = a2 < 0 ? -a2 : a2;
        strong = a1 < (1 << (bit_depth - 5)) && a2 < (1 << (bit_depth - 5));
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        if (64 * j >= N) continue;
        const int e = lane + 64 * j;
        if (e < N) {
          int v;
          if (e == 0 || e == N - 1) v = ref[e];
          else if (strong) {
            if (e == n2) v = ref[n2];
            else if (e < n2) { const int y = 63 - e; v = ((63 - y) * ref[n2] + (y + 1) * ref[0] + 32) >> 6; }
            else { const int x = e - n2 - 1; v = ((63 - x) * ref[n2] + (x + 1) * ref[N - 1] + 32) >> 6; }
          } else v = (ref[e - 1] + 2 * ref[e] + ref[e + 1] + 2) >> 2;
          L.ref1[e] = (uint16_t)v;
        }
      }
      lds_sync();
      ref = L.ref1;
    }
  }
#define RL(k) ((int)ref[n2 - (k)])
#define RT(k) ((int)ref[n2 + (k)])
  // ---- prediction 8.4.4.2.4 - 8.4.4.2.6 ----
  int dc_val = 0;
  if (mode == 1) {
    int part = lane < n ? RT(lane + 1) + RL(lane + 1) : 0;
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    dc_val = (part + n) >> (log2n + 1);
  }
  const int angle = c_angle[mode];
  const int inv_angle = (mode >= 11 && mode <= 25) ? c_inv_angle[mode - 11] : 0;
  const int edge = c_idx == 0 && n < 32;  // DC / horizontal / vertical boundary smoothing
  for (int idx = lane; idx < n * n; idx += 64) {
    const int x = idx & (n - 1), y = idx >> log2n;
    int v;
    if (mode == 0) {
      v = ((n - 1 - x) * RL(y + 1) + (x + 1) * RT(n + 1) + (n - 1 - y) * RT(x + 1) + (y + 1) * RL(n + 1) + n) >> (log2n + 1);
    } else if (mode == 1) {
      v = dc_val;
      if (edge) {
        if (x == 0 && y == 0) v = (RL(1) + 2 * dc_val + RT(1) + 2) >> 2;
        else if (y == 0) v = (RT(x + 1) + 3 * dc_val + 2) >> 2;
        else if (x == 0) v = (RL(y + 1) + 3 * dc_val + 2) >> 2;
      }
    } else if (mode >= 18) {
      const int i_idx = ((y + 1) * angle) >> 5, i_fact = ((y + 1) * angle) & 31;
      const int k0 = x + i_idx + 1;
      // ref[k] = p[-1+k][-1] for k >= 0, projected left column for k < 0
      const int r0 = k0 >= 0 ? RT(k0) : RL((k0 * inv_angle + 128) >> 8);
      if (i_fact) {
        const int k1 = k0 + 1;
        const int r1 = k1 >= 0 ? RT(k1) : RL((k1 * inv_angle + 128) >> 8);
        v = ((32 - i_fact) * r0 + i_fact * r1 + 16) >> 5;
      } else v = r0;
      if (mode == 26 && edge && x == 0) v = clip3(0, maxv, RT(1) + ((RL(y + 1) - RL(0)) >> 1));
    } else {
      const int i_idx = ((x + 1) * angle) >> 5, i_fact = ((x + 1) * angle) & 31;
      const int k0 = y + i_idx + 1;
      const int r0 = k0 >= 0 ? RL(k0) : RT((k0 * inv_angle + 128) >> 8);
      if (i_fact) {
        const int k1 = k0 + 1;
        const int r1 = k1 >= 0 ? RL(k1) : RT((k1 * inv_angle + 128) >> 8);
        v = ((32 - i_fact) * r0 + i_fact * r1 + 16) >> 5;
      } else v = r0;
      if (mode == 10 && edge && y == 0) v = clip3(0, maxv, RL(1) + ((RT(x + 1) - RT(0)) >> 1));
    }
    tile[(yb + y) * ctbc + xb + x] = (Pix)v;
  }
#undef RL
#undef RT
  lds_sync();
  if (!cbf) return;

  // ---- residual (already scaled + inverse transformed in place of the coefficients) ----
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int idx = lane + 64 * k;
    if (idx < nn) {
      const int x = idx & (n - 1), y = idx >> log2n;
      Pix* p = &tile[(yb + y) * ctbc + xb + x];
      *p = (Pix)clip3(0, maxv, (int)*p + (int)rpre[k]);
    }
  }
  for (int idx = lane + 256; idx < nn; idx += 64) {   // 32x32 only
    const int x = idx & (n - 1), y = idx >> log2n;
    Pix* p = &tile[(yb + y) * ctbc + xb + x];
    *p = (Pix)clip3(0, maxv, (int)*p + (int)res[idx]);
  }
  lds_sync();
}

}  // namespace

template <typename Pix>
__global__ __launch_bounds__(64) void k_recon(ReconArgs A)
{
  __shared__ ReconLds<Pix> L;
  const int lane = threadIdx.x;
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(A.ticket, 1u);
  const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  if (ticket >= A.num_waves) return;
  if (__hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // failed parse: maps are garbage
  const ReconWave wd = A.waves[ticket];
  const int c_idx = (int)wd.comp;
  const PicParams& P = A.pics[wd.pic];
  if (c_idx > 0 && !P.chroma_format_idc) return;
  const int sub = c_idx ? 2 : 1;
  const int ctb = 1 << P.log2_ctb, ctbc = ctb / sub;
  const int units = 1 << P.units_per_ctb_log2;
  const CtbInfo* ctb_info = (const CtbInfo*)(A.arena + P.off_ctb_info);
  Pix* rec = (Pix*)(A.arena + P.off_rec[c_idx]);
  const uint32_t stride = P.rec_stride[c_idx] / sizeof(Pix);
  // line buffer: bottom sample row of every CTB row of this component, row stride = rec_stride
  uint32_t* line = (uint32_t*)(A.arena + P.off_line[c_idx]);
  const uint32_t line_words = P.rec_stride[c_idx] / 4;
  constexpr int ES = (int)sizeof(Pix);
  int err = 0;
  uint32_t my_row = 0;

  for (int cy = (int)wd.first_row; cy < P.ctb_h && !err; cy += (int)wd.stride) {
  my_row = wd.base_row + (uint32_t)cy;     // batch row index
  uint32_t* my_progress = A.row_progress + (size_t)my_row * 3 + c_idx;
  const uint32_t* up_progress = my_progress - 3;
  for (int cx = 0; cx < P.ctb_w && !err; cx++) {
    const int ctb_rs = cy * P.ctb_w + cx;
    const CtbInfo ci = ctb_info[ctb_rs];
    Ctx C;
    C.pp = &P; C.lane = lane; C.x_ctb = cx << P.log2_ctb; C.y_ctb = cy << P.log2_ctb; C.avail = ci.avail; C.ctb = ctb;
    const int xc0 = C.x_ctb / sub;  // component x of the CTB
    // ---- wait for the row above: above-right CTB done (or the row end) ----
    const Pix* top = (const Pix*)(L.top_raw + 1);
    if (cy > 0) {
      uint32_t need = cx == 0 ? wd.start_lag : (uint32_t)(cx + 2);   // 2 = above-right CTB; a larger start distance decouples the rows
      if (need > (uint32_t)P.ctb_w) need = (uint32_t)P.ctb_w;
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
      for (int i = lane; i < nwords; i += 64) L.top_raw[1 + i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      top = (const Pix*)((const uint8_t*)(L.top_raw + 1) + (b0 - start));   // top[0] = above-left sample
    }
    // ---- stage the CTB's maps ----
    {
      const size_t base = (size_t)ctb_rs * units;
      for (int i = lane * 4; i < units; i += 256) {
        *(uint32_t*)&L.m_size[i] = *(const uint32_t*)(A.arena + P.off_u_size + base + i);
        *(uint32_t*)&L.m_flags[i] = *(const uint32_t*)(A.arena + P.off_u_flags + base + i);
        *(uint32_t*)&L.m_ipm[i] = *(const uint32_t*)(A.arena + P.off_u_ipm + base + i);
        *(uint32_t*)&L.m_ipmc[i] = *(const uint32_t*)(A.arena + P.off_u_ipmc + base + i);
      }
    }
    lds_sync();

    // ---- this component's blocks of the CTB in z-scan order ----
    const int16_t* res_base = (const int16_t*)(A.arena + P.off_coeff[c_idx]) + (size_t)ctb_rs * (ctb * ctb / (sub * sub));
    int z = 0;
    while (z < units) {
      const int ux = (int)compact1by1((uint32_t)z), uy = (int)compact1by1((uint32_t)z >> 1);
      if (C.x_ctb + ux * 4 >= P.width || C.y_ctb + uy * 4 >= P.height) { z++; continue; }
      const int tb = L.m_size[z] & 15;
      if (tb < 2 || tb > 5) { err = DEV_ERR_SYNTAX; break; }
      const int fl = L.m_flags[z];
      if (c_idx == 0) {
        reconstruct_block<Pix>(L, C, top, 0, ux * 4, uy * 4, tb, z, L.m_ipm[z] & 63, fl & UF_CBF_LUMA, res_base + z * 16);
      } else {
        int do_c = 0, zc = z, tc = tb - 1;
        if (tb > 2) do_c = 1;
        else if ((z & 3) == 3) { do_c = 1; zc = z & ~3; tc = 2; }
        if (do_c) {
          const int cux = (int)compact1by1((uint32_t)zc), cuy = (int)compact1by1((uint32_t)zc >> 1);
          reconstruct_block<Pix>(L, C, top, c_idx, cux * 2, cuy * 2, tc, zc, L.m_ipmc[z], fl & (c_idx == 1 ? UF_CBF_CB : UF_CBF_CR), res_base + zc * 4);
        }
      }
      z += 1 << (2 * (tb - 2));
    }
    lds_sync();

    // ---- write the CTB out (planes are allocated CTB-aligned, so no edge guards), hand its bottom row to the
    //      row below and keep its right column as the next CTB's left border ----
    {
      constexpr int PPW = 4 / ES;            // pixels per 32-bit word
      const int wpr = ctbc / PPW;            // words per tile row
      const int yc0 = C.y_ctb / sub;
      for (int i = lane; i < wpr * ctbc; i += 64) {
        const int y = i / wpr, xw = i - y * wpr;
        *(uint32_t*)&rec[(size_t)(yc0 + y) * stride + xc0 + xw * PPW] = *(const uint32_t*)&L.tile[y * ctbc + xw * PPW];
      }
      uint32_t* dst = line + (size_t)cy * line_words + (size_t)xc0 * ES / 4;
      for (int i = lane; i < wpr; i += 64)
        __hip_atomic_store(dst + i, *(const uint32_t*)&L.tile[(ctbc - 1) * ctbc + i * PPW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lds_sync();
      for (int i = lane; i < ctbc; i += 64) L.left[i] = L.tile[i * ctbc + ctbc - 1];
    }
    lds_sync();
    drain_stores();   // the line-buffer stores have left this wave
    if (lane == 0) __hip_atomic_store(my_progress, (uint32_t)(cx + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }
  if (err && lane == 0) atomicCAS((int*)A.status, 0, err | (int)(0x40000000u) | (int)(my_row << 8));
}

void launch_recon(const ReconArgs& a, bool wide, hipStream_t s)
{
  if (!a.num_waves) return;
  if (wide) hipLaunchKernelGGL((k_recon<uint16_t>), dim3(a.num_waves), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((k_recon<uint8_t>), dim3(a.num_waves), dim3(64), 0, s, a);
}

}  // namespace hipdec
