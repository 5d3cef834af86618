// recon_kernel.hip — intra prediction + dequantisation + inverse DCT/DST + reconstruction.
//
// Stands in for libde265's decode_TU / intra-prediction / transform stages behind de265_decode()
// (reference call site libheif/plugins/decoder_libde265.cc:402).  ITU-T H.265 8.4.4.2 (intra sample
// prediction incl. reference substitution and smoothing), 8.6.2-8.6.4 (scaling, transforms).
//
// MI355X mapping
//   * intra prediction makes every block depend on its left / above / above-right neighbours, so the
//     parallelism is the classic 2-CTB-lag wavefront over CTB rows: one 64-lane wavefront per CTB row
//     of each picture; rows of ALL pictures of a batch run concurrently (ticket-ordered so a row's
//     predecessor is always resident; progress words use agent-scope release/acquire).
//   * the CTB being reconstructed lives in LDS (luma + chroma tiles with their top / left borders), so
//     reference-sample gathering, smoothing, prediction and the two transform passes never touch
//     HBM; the finished CTB leaves LDS once with row-contiguous stores.
//   * inside a block the 64 lanes split the samples (prediction: n*n/64 samples per lane; transform:
//     n*n/64 outputs per lane per pass), coefficients arrive as one contiguous int16 block per TU.
//   * HBM traffic per luma pixel: 3 B coefficients (only where cbf) + 0.3 B maps in, 1.5*s out — the
//     kernel is bound by the dependency wavefront, not by bandwidth.
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "kernels.h"

namespace hipdec {


namespace {

__constant__ int8_t c_dct_c[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                   61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
__constant__ int8_t c_dst[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
__constant__ int8_t c_angle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                   -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
__constant__ int16_t c_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};
__constant__ uint8_t c_chroma_qp[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
__constant__ int c_level_scale[6] = {40, 45, 51, 57, 64, 72};

template <typename Pix>
struct ReconLds {
  Pix tile_y[64 * 64];
  Pix tile_c[2][32 * 32];
  Pix top_y[132], top_c[2][68];
  Pix left_y[64], left_c[2][32];
  uint16_t ref0[132], ref1[132];
  int16_t blk[32 * 32], tmp[32 * 32];
  int8_t dct[32 * 32];
  uint8_t m_size[256], m_flags[256], m_ipm[256], m_ipmc[256];
  int8_t m_qp[256];
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
  SliceParams sl;
};

// One transform block: prediction (+ residual) into the LDS tile.
//   c_idx: component; (xb, yb): block origin inside the CTB in component samples; log2n: block size
//   z_cur: z-index (4x4 luma units) of the block that defines "already decoded" for availability
template <typename Pix>
__device__ __forceinline__ void reconstruct_block(ReconLds<Pix>& L, const Ctx& C, int c_idx, int xb, int yb, int log2n, int z_cur, int mode, int cbf,
                                  int transform_skip, int bypass, int qp_y, const int16_t* coef)
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
  Pix* tile = c_idx == 0 ? L.tile_y : L.tile_c[c_idx - 1];
  const Pix* top = c_idx == 0 ? L.top_y : L.top_c[c_idx - 1];
  const Pix* left = c_idx == 0 ? L.left_y : L.left_c[c_idx - 1];

  // ---- 8.4.4.2.2 reference samples: gather + availability + substitution ----
  uint64_t m[3] = {0, 0, 0};
  int val[3] = {0, 0, 0}, av[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < 3; j++) {
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
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int e = lane + 64 * j;
    if (e < N && av[j]) L.ref0[e] = (uint16_t)val[j];
  }
  __syncthreads();
  const int any = (m[0] | m[1] | m[2]) != 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int e = lane + 64 * j;
    if (e < N && !av[j]) {
      int v;
      if (!any) v = 1 << (bit_depth - 1);
      else v = L.ref0[find_src(e, m[0], m[1], m[2])];
      val[j] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int e = lane + 64 * j;
    if (e < N && !av[j]) L.ref0[e] = (uint16_t)val[j];
  }
  __syncthreads();
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
      __syncthreads();
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
  __syncthreads();
  if (!cbf) return;

  // ---- residual: 8.6.2 scaling, 8.6.4 transformation ----
  const int nn = n * n;
  if (bypass) {
    for (int idx = lane; idx < nn; idx += 64) {
      const int x = idx & (n - 1), y = idx >> log2n;
      Pix* p = &tile[(yb + y) * ctbc + xb + x];
      *p = (Pix)clip3(0, maxv, (int)*p + (int)coef[idx]);
    }
    __syncthreads();
    return;
  }
  int qp;
  if (c_idx == 0) qp = qp_y + 6 * (P.bit_depth_luma - 8);
  else {
    const int off_c = 6 * (P.bit_depth_chroma - 8);
    int qpi = clip3(-off_c, 57, qp_y + (c_idx == 1 ? C.sl.cb_qp_offset : C.sl.cr_qp_offset));
    int qpc = qpi < 30 ? qpi : (qpi >= 44 ? qpi - 6 : c_chroma_qp[qpi - 30]);
    qp = qpc + off_c;
  }
  {
    const int bd_shift = bit_depth + log2n - 5;
    const long long scale = (long long)(16 * c_level_scale[qp % 6]) << (qp / 6);
    const long long rnd = 1ll << (bd_shift - 1);
    for (int idx = lane; idx < nn; idx += 64) {
      long long v = ((long long)coef[idx] * scale + rnd) >> bd_shift;
      L.blk[idx] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
    }
  }
  __syncthreads();
  const int bd_shift2 = 20 - bit_depth;
  if (transform_skip) {
    for (int idx = lane; idx < nn; idx += 64) {
      const int x = idx & (n - 1), y = idx >> log2n;
      const int r = ((int)L.blk[idx] * 128 + (1 << (bd_shift2 - 1))) >> bd_shift2;
      Pix* p = &tile[(yb + y) * ctbc + xb + x];
      *p = (Pix)clip3(0, maxv, (int)*p + r);
    }
    __syncthreads();
    return;
  }
  const int dst = c_idx == 0 && n == 4;
  const int step = 32 >> log2n;  // row stride into the 32-point matrix
  // first stage: columns.  tmp[i][x] = clip16((sum_j M[j][i] * blk[j][x] + 64) >> 7)
  for (int idx = lane; idx < nn; idx += 64) {
    const int x = idx & (n - 1), i = idx >> log2n;
    int sum = 0;
    if (dst) { for (int j = 0; j < 4; j++) sum += (int)c_dst[j * 4 + i] * (int)L.blk[j * 4 + x]; }
    else { for (int j = 0; j < n; j++) sum += (int)L.dct[(j * step) * 32 + i] * (int)L.blk[j * n + x]; }
    L.tmp[idx] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
  }
  __syncthreads();
  // second stage: rows.  res[y][i] = (sum_j M[j][i] * tmp[y][j] + rnd) >> bd_shift2
  for (int idx = lane; idx < nn; idx += 64) {
    const int i = idx & (n - 1), y = idx >> log2n;
    int sum = 0;
    if (dst) { for (int j = 0; j < 4; j++) sum += (int)c_dst[j * 4 + i] * (int)L.tmp[y * 4 + j]; }
    else { for (int j = 0; j < n; j++) sum += (int)L.dct[(j * step) * 32 + i] * (int)L.tmp[y * n + j]; }
    const int r = (sum + (1 << (bd_shift2 - 1))) >> bd_shift2;
    Pix* p = &tile[(yb + y) * ctbc + xb + i];
    *p = (Pix)clip3(0, maxv, (int)*p + r);
  }
  __syncthreads();
}

}  // namespace

template <typename Pix>
__global__ __launch_bounds__(64) void k_recon(ReconArgs A)
{
  __shared__ ReconLds<Pix> L;
  __shared__ uint32_t s_ticket;
  const int lane = threadIdx.x;
  if (lane == 0) s_ticket = atomicAdd(A.ticket, 1u);
  // 32-point DCT matrix (8.6.4.2): M[m][n] from the 33 distinct magnitudes
  for (int idx = lane; idx < 1024; idx += 64) {
    const int mm = idx >> 5, nx = idx & 31;
    int k = ((2 * nx + 1) * mm) & 127;
    if (k > 64) k = 128 - k;
    L.dct[idx] = (int8_t)(k <= 32 ? c_dct_c[k] : -c_dct_c[64 - k]);
  }
  __syncthreads();
  if (s_ticket >= A.num_rows) return;
  const RowDesc rd = A.rows[s_ticket];
  const PicParams& P = A.pics[rd.pic];
  const uint32_t my_row = s_ticket;  // batch row index == ticket (rows are listed picture by picture)
  const int cy = (int)rd.row;
  const int ctb = 1 << P.log2_ctb, ctbc = ctb >> 1;
  const int units = 1 << P.units_per_ctb_log2;
  const int has_chroma = P.chroma_format_idc != 0;
  const CtbInfo* ctb_info = (const CtbInfo*)(A.arena + P.off_ctb_info);
  const SliceParams* slices = (const SliceParams*)(A.arena + P.off_slices);
  Pix* rec[3];
  uint32_t stride[3];
  for (int c = 0; c < 3; c++) { rec[c] = (Pix*)(A.arena + P.off_rec[c]); stride[c] = P.rec_stride[c] / sizeof(Pix); }
  int err = 0;

  for (int cx = 0; cx < P.ctb_w && !err; cx++) {
    const int ctb_rs = cy * P.ctb_w + cx;
    const CtbInfo ci = ctb_info[ctb_rs];
    Ctx C;
    C.pp = &P; C.lane = lane; C.x_ctb = cx << P.log2_ctb; C.y_ctb = cy << P.log2_ctb; C.avail = ci.avail; C.ctb = ctb;
    C.sl = slices[ci.slice_idx];
    // ---- wait for the row above: above-right CTB done (or the row end) ----
    if (cy > 0) {
      const uint32_t need = (uint32_t)(cx + 2 < P.ctb_w ? cx + 2 : P.ctb_w);
      uint32_t spins = 0;
      while (__hip_atomic_load(&A.row_progress[my_row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 22) || __hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { err = DEV_ERR_TIMEOUT; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
      if (err) break;
    }
    // ---- stage the CTB's maps and its top border ----
    {
      const size_t base = (size_t)ctb_rs * units;
      for (int i = lane * 4; i < units; i += 256) {
        *(uint32_t*)&L.m_size[i] = *(const uint32_t*)(A.arena + P.off_u_size + base + i);
        *(uint32_t*)&L.m_flags[i] = *(const uint32_t*)(A.arena + P.off_u_flags + base + i);
        *(uint32_t*)&L.m_ipm[i] = *(const uint32_t*)(A.arena + P.off_u_ipm + base + i);
        *(uint32_t*)&L.m_ipmc[i] = *(const uint32_t*)(A.arena + P.off_u_ipmc + base + i);
        *(uint32_t*)&L.m_qp[i] = *(const uint32_t*)(A.arena + P.off_u_qp + base + i);
      }
      if (cy > 0) {
        // luma: x_ctb - 1 .. x_ctb + 2*ctb - 1  (index 0 = above-left corner)
        for (int i = lane; i <= 2 * ctb; i += 64) {
          const int x = C.x_ctb - 1 + i;
          if (x >= 0 && x < P.width) L.top_y[i] = rec[0][(size_t)(C.y_ctb - 1) * stride[0] + x];
        }
        if (has_chroma)
          for (int c = 0; c < 2; c++)
            for (int i = lane; i <= 2 * ctbc; i += 64) {
              const int x = C.x_ctb / 2 - 1 + i;
              if (x >= 0 && x < P.cwidth) L.top_c[c][i] = rec[c + 1][(size_t)(C.y_ctb / 2 - 1) * stride[c + 1] + x];
            }
      }
    }
    __syncthreads();

    // ---- transform blocks of the CTB in z-scan order ----
    const int16_t* coef_y = (const int16_t*)(A.arena + P.off_coeff[0]) + (size_t)ctb_rs * ctb * ctb;
    const int16_t* coef_cb = (const int16_t*)(A.arena + P.off_coeff[1]) + (size_t)ctb_rs * (ctb * ctb / 4);
    const int16_t* coef_cr = (const int16_t*)(A.arena + P.off_coeff[2]) + (size_t)ctb_rs * (ctb * ctb / 4);
    int z = 0;
    while (z < units) {
      const int ux = (int)compact1by1((uint32_t)z), uy = (int)compact1by1((uint32_t)z >> 1);
      if (C.x_ctb + ux * 4 >= P.width || C.y_ctb + uy * 4 >= P.height) { z++; continue; }
      const int sz = L.m_size[z];
      const int t = sz & 15;
      if (t < 2 || t > 5) { err = DEV_ERR_SYNTAX; break; }
      const int fl = L.m_flags[z], ipm = L.m_ipm[z];
      const int qp_y = L.m_qp[z];
      const int bypass = (fl & UF_BYPASS) != 0;
      reconstruct_block<Pix>(L, C, 0, ux * 4, uy * 4, t, z, ipm & 63, fl & UF_CBF_LUMA, (fl & UF_TS_LUMA) != 0, bypass, qp_y, coef_y + z * 16);
      if (has_chroma) {
        int do_c = 0, zc = z, tc = t - 1;
        if (t > 2) do_c = 1;
        else if ((z & 3) == 3) { do_c = 1; zc = z & ~3; tc = 2; }
        if (do_c) {
          const int cux = (int)compact1by1((uint32_t)zc), cuy = (int)compact1by1((uint32_t)zc >> 1);
          const int cmode = L.m_ipmc[z];
          reconstruct_block<Pix>(L, C, 1, cux * 2, cuy * 2, tc, zc, cmode, fl & UF_CBF_CB, (ipm & 64) != 0, bypass, qp_y, coef_cb + zc * 4);
          reconstruct_block<Pix>(L, C, 2, cux * 2, cuy * 2, tc, zc, cmode, fl & UF_CBF_CR, (ipm & 128) != 0, bypass, qp_y, coef_cr + zc * 4);
        }
      }
      z += 1 << (2 * (t - 2));
    }
    __syncthreads();

    // ---- write the CTB out (planes are allocated CTB-aligned, so no edge guards) and keep its
    //      right column as the next CTB's left border ----
    {
      constexpr int PPW = 4 / sizeof(Pix);  // pixels per 32-bit word
      const int wpr = ctb / PPW;            // words per luma row
      for (int i = lane; i < wpr * ctb; i += 64) {
        const int y = i / wpr, xw = i - y * wpr;
        *(uint32_t*)&rec[0][(size_t)(C.y_ctb + y) * stride[0] + C.x_ctb + xw * PPW] = *(const uint32_t*)&L.tile_y[y * ctb + xw * PPW];
      }
      if (has_chroma) {
        const int wprc = ctbc / PPW;
        for (int c = 0; c < 2; c++)
          for (int i = lane; i < wprc * ctbc; i += 64) {
            const int y = i / wprc, xw = i - y * wprc;
            *(uint32_t*)&rec[c + 1][(size_t)(C.y_ctb / 2 + y) * stride[c + 1] + C.x_ctb / 2 + xw * PPW] = *(const uint32_t*)&L.tile_c[c][y * ctbc + xw * PPW];
          }
      }
      __syncthreads();
      for (int i = lane; i < ctb; i += 64) L.left_y[i] = L.tile_y[i * ctb + ctb - 1];
      if (has_chroma)
        for (int c = 0; c < 2; c++)
          for (int i = lane; i < ctbc; i += 64) L.left_c[c][i] = L.tile_c[c][i * ctbc + ctbc - 1];
    }
    __syncthreads();
    if (lane == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&A.row_progress[my_row], (uint32_t)(cx + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (err && lane == 0) atomicCAS((int*)A.status, 0, err | (int)(0x40000000u) | (int)(my_row << 8));
}

void launch_recon(const ReconArgs& a, bool wide, hipStream_t s)
{
  if (!a.num_rows) return;
  if (wide) hipLaunchKernelGGL((k_recon<uint16_t>), dim3(a.num_rows), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((k_recon<uint8_t>), dim3(a.num_rows), dim3(64), 0, s, a);
}

}  // namespace hipdec
