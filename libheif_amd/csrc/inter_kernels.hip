// inter_kernels.hip — P and B pictures of sequence tracks: motion vector derivation and motion-compensated prediction.
//
// Stands in for libde265's inter decoding behind de265_decode() for the samples libheif pushes one by one for a track
// (libheif/sequences/track_visual.cc:200-280; libheif/plugins/decoder_libde265.cc:360, :417-419).  ITU-T H.265 6.4.2 (prediction block
// availability), 8.5.3.2.2 - 8.5.3.2.5 (merge mode: spatial, temporal, combined bi-predictive and zero candidates), 8.5.3.2.6 - 8.5.3.2.9 (motion
// vector prediction: spatial candidates with scaling, the collocated candidate), 8.5.3.3.3 (fractional sample interpolation), 8.5.3.3.4.2 /
// 8.5.3.3.4.3 (default and explicit weighted sample prediction, one or two lists).
// Scope as the host front end enforces it: P and B slices, short-term and long-term reference pictures, 4:0:0 / 4:2:0.
//
// MI355X mapping
//   * the entropy decoder (parse_core.h, HIPDEC_PARSE_INTER build) only PARSES prediction units into MotionSyntax records: HEVC keeps parsing free
//     of the neighbours' motion.  k_motion turns them into the picture's motion field in decoding order.  A candidate list reads the left,
//     above, above-left and above-RIGHT neighbours, so the parallelism is the 2-CTB-lag wavefront over CTB rows - one wave per CTB row, rows
//     handed out by ticket so that a row's predecessor is resident or finished, progress published per CTB (release) and awaited (acquire).
//     Inside a CTB the derivation is a serial chain over prediction units (lane 0); the 64 lanes write each unit's motion to its 4x4 units.
//     Everything the chain reads is staged in LDS by all lanes first (14 KB): the CTB being derived and the one to its left, the bottom unit rows of
//     the three CTBs above, the CTB's syntax records and unit maps; only the collocated picture's units (temporal candidates) come from HBM.
//   * k_mc is embarrassingly parallel once the motion field exists: one thread per sample of each plane, taps gathered from the reference
//     picture (coordinates clamped to the picture: the padding of 8.5.3.3.3.1), separable 8-tap (luma) / 4-tap (chroma) filters with the
//     intermediate precision of the specification, written to the reconstruction plane where k_recon adds the residual.  HBM-bound in
//     principle (1.5 s in + 1.5 s out per pixel with cache-resident taps); sequences are not the benchmarked path.
#include <hip/hip_runtime.h>
#include "hevc_device.h"
#include "kernels.h"
#include "color_device.h"

namespace hipdec {

namespace {

__device__ __forceinline__ uint32_t mk_compact(uint32_t v)
{
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu;
  return v;
}
__device__ __forceinline__ uint32_t mk_interleave(uint32_t x, uint32_t y)   // z-index of unit (x, y), x, y < 16
{
  x = (x | (x << 2)) & 0x33; x = (x | (x << 1)) & 0x55;
  y = (y | (y << 2)) & 0x33; y = (y | (y << 1)) & 0x55;
  return x | (y << 1);
}
__device__ __forceinline__ int mk_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int mk_abs(int v) { return v < 0 ? -v : v; }

// all of this wave's global stores have left it (the CPU-test build orders them with a fence instead)
__device__ __forceinline__ void mk_drain_stores()
{
#ifndef HIPDEC_HOST_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
__device__ __forceinline__ void mk_lds_sync()
{
#ifndef HIPDEC_HOST_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();
}

// Motion of a prediction block while it is being derived.  Plain scalars and select-based accessors on purpose: the derivation runs on ONE lane, and
// any array indexed with a run-time value (candidate lists, [list] members) is placed in scratch memory by the compiler - a global-memory round trip
// per access for a lone lane (the first version spent 11.5 ms on a 720p picture that way).  This version uses no scratch.
struct Mo { int x0, y0, x1, y1, r0, r1; };   // list 0 / list 1 vector (quarter samples) and reference index (< 0: the list is not used)
__device__ __forceinline__ Mo mo_none() { return Mo{0, 0, 0, 0, -1, -1}; }
__device__ __forceinline__ void mo_set(Mo& m, int L, int x, int y, int r) { if (L) { m.x1 = x; m.y1 = y; m.r1 = r; } else { m.x0 = x; m.y0 = y; m.r0 = r; } }
// the i-th candidate that is present, in list order (the lists are never stored: a run-time index into one would put it in scratch)
__device__ __forceinline__ Mo nth_present(int i, int f0, const Mo& m0, int f1, const Mo& m1, int f2, const Mo& m2, int f3, const Mo& m3, int f4, const Mo& m4,
                                          int f5, const Mo& m5)
{
  Mo r = mo_none();
  int c = 0;
  if (f0) { if (c == i) r = m0; c++; }
  if (f1) { if (c == i) r = m1; c++; }
  if (f2) { if (c == i) r = m2; c++; }
  if (f3) { if (c == i) r = m3; c++; }
  if (f4) { if (c == i) r = m4; c++; }
  if (f5) { if (c == i) r = m5; c++; }
  return r;
}
// [list] members of a stored unit without run-time indexing
__device__ __forceinline__ int unit_ref(const MotionUnit& u, int L) { return L ? u.ref_idx[1] : u.ref_idx[0]; }
__device__ __forceinline__ int unit_slot(const MotionUnit& u, int L) { return (int)((L ? u.slot_pred[1] : u.slot_pred[0]) & 63u); }
__device__ __forceinline__ int unit_mvx(const MotionUnit& u, int L) { return L ? u.mv[1][0] : u.mv[0][0]; }
__device__ __forceinline__ int unit_mvy(const MotionUnit& u, int L) { return L ? u.mv[1][1] : u.mv[0][1]; }
__device__ __forceinline__ int unit_poc_delta(const MotionUnit& u, int L) { return L ? u.poc_delta[1] : u.poc_delta[0]; }

// everything the derivation of one CTB needs (lane 0 only)
struct MotionCtx {
  const PicParams* P;
  const RefFrame* reftab;
  const SliceParams* slice;
  const MotionUnit* field;      // the picture's motion field in HBM
  const MotionUnit* cur;        // the current CTB's units (LDS)
  const MotionUnit* left;       // the CTB to the left, as it was derived a moment ago (LDS)
  const MotionUnit* up;         // bottom unit rows of the CTBs above-left, above and above-right (LDS): [3][units per CTB side]
  HIPDEC_GLOBAL const MotionUnit* col;   // the collocated picture's motion field in HBM (nullptr: no temporal candidates / an intra picture)
  const MotionUnit* col_lds;    // its units on the 16x16 grid the CTB's candidates can name, staged in LDS: [(y - y_ctb) / 16][(x - x_ctb) / 16], x up to one
                                // grid column to the right of the CTB (bottom-right candidates; never below the CTB row, 8.5.3.2.8)
  const int* ref_poc;           // PicOrderCnt of the reference picture table's slots (LDS)
  int col_w;                    // grid columns staged
  int lt_mask;                  // PicParams::lt_mask: the reference picture table's long-term slots
  int cx, cy, avail;            // CTB position, AV_* bits
  int log2_ctb, units_log2, lmt, ctb_w, width, height;
  int poc, par_mrg;
};

__device__ __forceinline__ int slot_of(const MotionCtx& C, int X, int ref_idx) { return X ? C.slice->ref_slot_l1[ref_idx] : C.slice->ref_slot[ref_idx]; }
__device__ __forceinline__ int num_ref_of(const MotionCtx& C, int X) { return X ? C.slice->num_ref_idx_l1 : C.slice->num_ref_idx; }

__device__ __forceinline__ size_t unit_index(const MotionCtx& C, int x, int y)
{
  const uint32_t z = mk_interleave((uint32_t)((x >> 2) & ((1 << (C.log2_ctb - 2)) - 1)), (uint32_t)((y >> 2) & ((1 << (C.log2_ctb - 2)) - 1)));
  return ((size_t)((y >> C.log2_ctb) * C.ctb_w + (x >> C.log2_ctb)) << C.units_log2) + z;
}

// the motion of the 4x4 unit that covers luma sample (x, y); the caller has checked availability.  Every spatial neighbour a candidate list can
// name lies in the current CTB, in the CTB to the left, or in the bottom unit row of the three CTBs above: all of them are in LDS
__device__ __forceinline__ MotionUnit unit_at(const MotionCtx& C, int x, int y)
{
  const int ncx = x >> C.log2_ctb, ncy = y >> C.log2_ctb, side = 1 << (C.log2_ctb - 2);
  const uint32_t ux = (uint32_t)((x >> 2) & (side - 1)), uy = (uint32_t)((y >> 2) & (side - 1));
  // Every source below is LDS, and it has to stay that way: a fall-back to the field in HBM here (it was never taken: the three regions cover every
  // neighbour a candidate list can name) made the compiler merge the LDS and the global pointer into a GENERIC one - every neighbour read, field
  // by field, became a FLAT load with global-memory latency on the one lane that derives (64 of them per unit walk: 12 000 cycles per prediction unit)
  if (ncy == C.cy) {
    if (ncx == C.cx) return C.cur[mk_interleave(ux, uy)];
    if (ncx == C.cx - 1) return C.left[mk_interleave(ux, uy)];
  } else if (ncy == C.cy - 1 && (int)uy == side - 1 && ncx >= C.cx - 1 && ncx <= C.cx + 1) return C.up[(ncx - C.cx + 1) * side + (int)ux];
  MotionUnit none{};   // (not reachable for an available neighbour: reads as an intra coded unit)
  none.ref_idx[0] = none.ref_idx[1] = -1;
  return none;
}

// 6.4.1 z-scan order availability of (xN, yN) seen from (xC, yC) of the current CTB: inside the picture, decoded earlier, same slice, same tile
__device__ __forceinline__ int avail_z(const MotionCtx& C, int xC, int yC, int xN, int yN)
{
  if (xN < 0 || yN < 0 || xN >= C.width || yN >= C.height) return 0;
  const int dx = (xN >> C.log2_ctb) - C.cx, dy = (yN >> C.log2_ctb) - C.cy;
  if (dx == 0 && dy == 0) {   // same CTB: order of the minimum transform blocks in z-scan
    const int m = (1 << (C.log2_ctb - C.lmt)) - 1;
    const uint32_t zn = mk_interleave((uint32_t)((xN >> C.lmt) & m), (uint32_t)((yN >> C.lmt) & m));
    const uint32_t zc = mk_interleave((uint32_t)((xC >> C.lmt) & m), (uint32_t)((yC >> C.lmt) & m));
    return zn <= zc;
  }
  if (dx == -1 && dy == 0) return (C.avail & AV_LEFT) != 0;
  if (dx == 0 && dy == -1) return (C.avail & AV_UP) != 0;
  if (dx == 1 && dy == -1) return (C.avail & AV_UPRIGHT) != 0;
  if (dx == -1 && dy == -1) return (C.avail & AV_UPLEFT) != 0;
  return 0;   // to the right of / below the current CTB: decoded later
}

struct PbGeom { int xCb, yCb, nCbS, xPb, yPb, nPbW, nPbH, partIdx; };

// 6.4.2 prediction block availability; *out = the neighbour's motion when it is available (and not intra coded)
__device__ __forceinline__ int pb_available(const MotionCtx& C, const PbGeom& g, int xN, int yN, MotionUnit* out)
{
  const int same_cb = g.xCb <= xN && g.yCb <= yN && g.xCb + g.nCbS > xN && g.yCb + g.nCbS > yN;
  int av;
  if (!same_cb) av = avail_z(C, g.xPb, g.yPb, xN, yN);
  else if ((g.nPbW << 1) == g.nCbS && (g.nPbH << 1) == g.nCbS && g.partIdx == 1 && g.yCb + g.nPbH <= yN && g.xCb + g.nPbW > xN) av = 0;
  else av = 1;
  if (!av) return 0;
  const MotionUnit m = unit_at(C, xN, yN);
  if (m.ref_idx[0] < 0 && m.ref_idx[1] < 0) return 0;   // MODE_INTRA
  *out = m;
  return 1;
}

// (the fields of a list that is not used are stored as 0 / -1, so whole-record comparison is the comparison of 8.5.3.2.3)
__device__ __forceinline__ int same_motion(const MotionUnit& a, const MotionUnit& b)
{
  return a.mv[0][0] == b.mv[0][0] && a.mv[0][1] == b.mv[0][1] && a.mv[1][0] == b.mv[1][0] && a.mv[1][1] == b.mv[1][1] &&
         a.ref_idx[0] == b.ref_idx[0] && a.ref_idx[1] == b.ref_idx[1];
}
__device__ __forceinline__ Mo to_mo(const MotionUnit& u) { return Mo{u.mv[0][0], u.mv[0][1], u.mv[1][0], u.mv[1][1], u.ref_idx[0], u.ref_idx[1]}; }

__device__ __forceinline__ void scale_mv(int& mvx, int& mvy, int td, int tb)
{
  td = mk_clip3(-128, 127, td); tb = mk_clip3(-128, 127, tb);
  const int tx = (16384 + (mk_abs(td) >> 1)) / td;
  const int dsf = mk_clip3(-4096, 4095, (tb * tx + 32) >> 6);
  const int vx = dsf * mvx, vy = dsf * mvy;
  mvx = mk_clip3(-32768, 32767, (vx < 0 ? -1 : 1) * ((mk_abs(vx) + 127) >> 8));
  mvy = mk_clip3(-32768, 32767, (vy < 0 ? -1 : 1) * ((mk_abs(vy) + 127) >> 8));
}

// 8.5.3.2.9: the motion vector the collocated picture stored for the 16x16 block that covers (x, y), scaled to the distance of the target picture
__device__ __forceinline__ int collocated_mv(const MotionCtx& C, int x, int y, int ref_idx, int X, int& mvx, int& mvy)
{
  // (a lone lane fetching this from HBM - one or two dependent global loads per candidate, up to four per prediction unit - was most of
  //  k_motion's 8 ms per 720p picture: the grid positions a CTB can name are staged by all lanes before the chain starts)
  const int gx = (x >> 4) - ((C.cx << C.log2_ctb) >> 4), gy = (y >> 4) - ((C.cy << C.log2_ctb) >> 4);
  if (gx < 0 || gx >= C.col_w || gy < 0 || gy >= (1 << (C.log2_ctb - 4))) return 0;   // (not reachable: the staged grid covers every position 8.5.3.2.8 can name)
  const MotionUnit cu = C.col_lds[gy * C.col_w + gx];   // LDS only: mixing in a global fall-back turns the read into FLAT loads (see unit_at)
  const int f0 = cu.ref_idx[0] >= 0, f1 = cu.ref_idx[1] >= 0;
  if (!f0 && !f1) return 0;   // intra coded
  int L;
  if (!f0) L = 1;
  else if (!f1) L = 0;
  else L = C.slice->no_backward ? X : C.slice->col_from_l0;
  int vx = unit_mvx(cu, L), vy = unit_mvy(cu, L);
  const int tgt_slot = slot_of(C, X, ref_idx);
  const int col_diff = unit_poc_delta(cu, L), cur_diff = C.poc - C.ref_poc[tgt_slot];
  // LongTermRefPic(current) != LongTermRefPic(collocated block, as marked when ITS picture was decoded): no candidate; a long-term target takes the
  // collocated vector as it is
  const int col_lt = (cu.slot_pred[1] >> (6 + L)) & 1, cur_lt = (C.lt_mask >> tgt_slot) & 1;
  if (col_lt != cur_lt) return 0;
  if (!cur_lt && col_diff != cur_diff) {
    if (col_diff == 0) return 0;
    scale_mv(vx, vy, col_diff, cur_diff);
  }
  mvx = vx; mvy = vy;
  return 1;
}

// 8.5.3.2.8: the bottom-right candidate (inside the CTB row and the picture), then the centre
__device__ __forceinline__ int temporal_mv(const MotionCtx& C, int xPb, int yPb, int nPbW, int nPbH, int ref_idx, int X, int& mvx, int& mvy)
{
  if (!C.col) return 0;
  const int xBr = xPb + nPbW, yBr = yPb + nPbH;
  if ((yPb >> C.log2_ctb) == (yBr >> C.log2_ctb) && yBr < C.height && xBr < C.width && collocated_mv(C, xBr, yBr, ref_idx, X, mvx, mvy)) return 1;
  return collocated_mv(C, xPb + (nPbW >> 1), yPb + (nPbH >> 1), ref_idx, X, mvx, mvy);
}

// 8.5.3.2.2 - 8.5.3.2.5: merge candidate merge_idx (spatial candidates A1, B1, B0, A0, B2, the temporal candidate, combined bi-predictive candidates
// of a B slice, zero candidates)
__device__ __forceinline__ Mo derive_merge(const MotionCtx& C, PbGeom g, int part_mode, int merge_idx)
{
  const int pl = C.par_mrg;
  const int orig_w = g.nPbW, orig_h = g.nPbH;
  if (pl > 2 && g.nCbS == 8) { g.xPb = g.xCb; g.yPb = g.yCb; g.nPbW = g.nPbH = g.nCbS; g.partIdx = 0; part_mode = 0; }   // singleMCLFlag
  const int xPb = g.xPb, yPb = g.yPb, nPbW = g.nPbW, nPbH = g.nPbH;
  const int max_cand = C.slice->max_merge_cand, is_b = C.slice->is_b;
#define MK_SAME_MER(xn, yn) ((xPb >> pl) == ((xn) >> pl) && (yPb >> pl) == ((yn) >> pl))
  MotionUnit A1{}, B1{}, B0{}, A0{}, B2{};
  // availableN: 6.4.2 minus the merge-estimation-region / second-partition exclusions; flagN: after pruning.  Comparisons read availableN of the other
  // candidate, not its flag; only B2's "all four present" rule counts flags (8.5.3.2.3).
  // The list is only needed up to merge_idx, and a candidate's flag depends on EARLIER candidates only: the walk stops at the merge_idx-th
  // present one (round 5: every prediction unit evaluated all five neighbours - availability, an LDS read and the pruning compares each - on
  // the one lane that derives, whatever merge_idx was; small indices are the common case)
  Mo out = mo_none();
  bool found = false;
  int avA1 = 0, avB1 = 0, avB0 = 0, avA0 = 0, avB2 = 0, fA1 = 0, fB1 = 0, fB0 = 0, fA0 = 0, fB2 = 0, n = 0, fCol = 0;
  Mo mA1 = mo_none(), mB1 = mo_none(), mB0 = mo_none(), mA0 = mo_none(), mB2 = mo_none(), mCol = mo_none();
  do {
    avA1 = pb_available(C, g, xPb - 1, yPb + nPbH - 1, &A1);
    if (MK_SAME_MER(xPb - 1, yPb + nPbH - 1) || (g.partIdx == 1 && (part_mode == 2 || part_mode == 6 || part_mode == 7))) avA1 = 0;   // Nx2N, nLx2N, nRx2N
    fA1 = avA1; mA1 = to_mo(A1);
    if (fA1 && n == merge_idx) { out = mA1; found = true; break; }
    n += fA1;
    avB1 = pb_available(C, g, xPb + nPbW - 1, yPb - 1, &B1);
    if (MK_SAME_MER(xPb + nPbW - 1, yPb - 1) || (g.partIdx == 1 && (part_mode == 1 || part_mode == 4 || part_mode == 5))) avB1 = 0;   // 2NxN, 2NxnU, 2NxnD
    fB1 = avB1 && !(avA1 && same_motion(A1, B1)); mB1 = to_mo(B1);
    if (fB1 && n == merge_idx) { out = mB1; found = true; break; }
    n += fB1;
    avB0 = pb_available(C, g, xPb + nPbW, yPb - 1, &B0);
    if (MK_SAME_MER(xPb + nPbW, yPb - 1)) avB0 = 0;
    fB0 = avB0 && !(avB1 && same_motion(B1, B0)); mB0 = to_mo(B0);
    if (fB0 && n == merge_idx) { out = mB0; found = true; break; }
    n += fB0;
    avA0 = pb_available(C, g, xPb - 1, yPb + nPbH, &A0);
    if (MK_SAME_MER(xPb - 1, yPb + nPbH)) avA0 = 0;
    fA0 = avA0 && !(avA1 && same_motion(A1, A0)); mA0 = to_mo(A0);
    if (fA0 && n == merge_idx) { out = mA0; found = true; break; }
    n += fA0;
    avB2 = pb_available(C, g, xPb - 1, yPb - 1, &B2);
    if (MK_SAME_MER(xPb - 1, yPb - 1)) avB2 = 0;
    fB2 = avB2 && !(avA1 && same_motion(A1, B2)) && !(avB1 && same_motion(B1, B2)) && fA0 + fA1 + fB0 + fB1 != 4; mB2 = to_mo(B2);
    if (fB2 && n == merge_idx) { out = mB2; found = true; break; }
    n += fB2;
  } while (0);
#undef MK_SAME_MER
  if (found) {
    if (out.r0 >= 0 && out.r1 >= 0 && orig_w + orig_h == 12) { out.r1 = -1; out.x1 = out.y1 = 0; }   // 8x4 / 4x8: uni-prediction
    return out;
  }
  if (C.slice->tmvp && n <= merge_idx) {   // the temporal candidate: reference index 0 in each list of the slice (only needed when merge_idx reaches it)
    int vx = 0, vy = 0;
    if (temporal_mv(C, xPb, yPb, nPbW, nPbH, 0, 0, vx, vy)) mo_set(mCol, 0, vx, vy, 0);
    if (is_b && temporal_mv(C, xPb, yPb, nPbW, nPbH, 0, 1, vx, vy)) mo_set(mCol, 1, vx, vy, 0);
    fCol = mCol.r0 >= 0 || mCol.r1 >= 0;
    n += fCol;
  }
  if (n > max_cand) n = max_cand;
#define MK_ORIG(i) nth_present((i), fA1, mA1, fB1, mB1, fB0, mB0, fA0, mA0, fB2, mB2, fCol, mCol)
  if (merge_idx < n) out = MK_ORIG(merge_idx);
  else {
    int found_comb = 0;
    if (is_b && n > 1 && n < max_cand) {   // 8.5.3.2.4 combined bi-predictive candidates: walked in list order until merge_idx is reached
      const int num_orig = n;
      for (int comb = 0; comb < num_orig * (num_orig - 1) && n < max_cand && !found_comb; comb++) {
        // l0CandIdx / l1CandIdx of Table 8-7: (0,1) (1,0) (0,2) (2,0) (1,2) (2,1) (0,3) (3,0) (1,3) (3,1) (2,3) (3,2), one nibble per combIdx
        const int l0 = (int)((0x323130212010ull >> (4 * comb)) & 3u), l1 = (int)((0x231303120201ull >> (4 * comb)) & 3u);
        const Mo a = MK_ORIG(l0), b = MK_ORIG(l1);
        if (a.r0 >= 0 && b.r1 >= 0 && (slot_of(C, 0, a.r0) != slot_of(C, 1, b.r1) || a.x0 != b.x1 || a.y0 != b.y1)) {
          if (n == merge_idx) { out = Mo{a.x0, a.y0, b.x1, b.y1, a.r0, b.r1}; found_comb = 1; }
          n++;
        }
      }
    }
    if (!found_comb) {   // zero candidates: the reference index counts up while below the number of reference pictures (B: of the shorter list), then 0
      const int zero_idx = merge_idx - n;
      const int num_ref = is_b ? (C.slice->num_ref_idx < C.slice->num_ref_idx_l1 ? C.slice->num_ref_idx : C.slice->num_ref_idx_l1) : C.slice->num_ref_idx;
      const int r = zero_idx < num_ref ? zero_idx : 0;
      out = Mo{0, 0, 0, 0, r, is_b ? r : -1};
    }
  }
#undef MK_ORIG
  if (out.r0 >= 0 && out.r1 >= 0 && orig_w + orig_h == 12) { out.r1 = -1; out.x1 = out.y1 = 0; }   // 8x4 / 4x8: uni-prediction
  return out;
}

// the motion vector of neighbour m that points at the target picture (its own list X first, then the other list); 8.5.3.2.7
__device__ __forceinline__ int nb_same_pic(const MotionUnit& m, int X, int tgt_slot, int& mvx, int& mvy)
{
  if (unit_ref(m, X) >= 0 && unit_slot(m, X) == tgt_slot) { mvx = unit_mvx(m, X); mvy = unit_mvy(m, X); return 1; }
  if (unit_ref(m, 1 - X) >= 0 && unit_slot(m, 1 - X) == tgt_slot) { mvx = unit_mvx(m, 1 - X); mvy = unit_mvy(m, 1 - X); return 1; }
  return 0;
}
__device__ __forceinline__ int nb_scaled(const MotionCtx& C, const MotionUnit& m, int X, int tgt_slot, int& mvx, int& mvy)
{
  // 8.5.3.2.7 (7): the neighbour's vector of list X, else of the other list, counts when its reference picture and the target are both long-term
  // or both short-term pictures; it is scaled only between short-term pictures
  const int tgt_lt = (C.lt_mask >> tgt_slot) & 1;
  int L = -1;
  if (unit_ref(m, X) >= 0 && ((C.lt_mask >> unit_slot(m, X)) & 1) == tgt_lt) L = X;
  else if (unit_ref(m, 1 - X) >= 0 && ((C.lt_mask >> unit_slot(m, 1 - X)) & 1) == tgt_lt) L = 1 - X;
  if (L < 0) return 0;
  mvx = unit_mvx(m, L); mvy = unit_mvy(m, L);
  const int nb_slot = unit_slot(m, L);
  if (!tgt_lt && nb_slot != tgt_slot) scale_mv(mvx, mvy, C.poc - C.ref_poc[nb_slot], C.poc - C.ref_poc[tgt_slot]);
  return 1;
}

// 8.5.3.2.6 - 8.5.3.2.8: motion vector predictor mvp_flag of list X for reference index ref_idx
__device__ __forceinline__ void derive_mvp(const MotionCtx& C, const PbGeom& g, int X, int ref_idx, int mvp_flag, int& out_x, int& out_y)
{
  const int xPb = g.xPb, yPb = g.yPb, nPbW = g.nPbW, nPbH = g.nPbH;
  const int tgt_slot = slot_of(C, X, ref_idx);
  MotionUnit a0{}, a1{};
  const int av_a0 = pb_available(C, g, xPb - 1, yPb + nPbH, &a0), av_a1 = pb_available(C, g, xPb - 1, yPb + nPbH - 1, &a1);
  const int is_scaled = av_a0 || av_a1;
  int flagA = 0, ax = 0, ay = 0;
  if (av_a0) flagA = nb_same_pic(a0, X, tgt_slot, ax, ay);
  if (!flagA && av_a1) flagA = nb_same_pic(a1, X, tgt_slot, ax, ay);
  if (!flagA && av_a0) flagA = nb_scaled(C, a0, X, tgt_slot, ax, ay);
  if (!flagA && av_a1) flagA = nb_scaled(C, a1, X, tgt_slot, ax, ay);
  if (flagA && !mvp_flag) { out_x = ax; out_y = ay; return; }   // mvpListLX[0] is the A candidate whenever there is one: B and the temporal candidate are not needed
  MotionUnit b0{}, b1{}, b2{};
  const int av_b0 = pb_available(C, g, xPb + nPbW, yPb - 1, &b0), av_b1 = pb_available(C, g, xPb + nPbW - 1, yPb - 1, &b1), av_b2 = pb_available(C, g, xPb - 1, yPb - 1, &b2);
  int flagB = 0, bx = 0, by = 0;
  if (av_b0) flagB = nb_same_pic(b0, X, tgt_slot, bx, by);
  if (!flagB && av_b1) flagB = nb_same_pic(b1, X, tgt_slot, bx, by);
  if (!flagB && av_b2) flagB = nb_same_pic(b2, X, tgt_slot, bx, by);
  if (!is_scaled && flagB) { flagA = 1; ax = bx; ay = by; }
  if (!is_scaled) {
    flagB = 0;
    if (av_b0) flagB = nb_scaled(C, b0, X, tgt_slot, bx, by);
    if (!flagB && av_b1) flagB = nb_scaled(C, b1, X, tgt_slot, bx, by);
    if (!flagB && av_b2) flagB = nb_scaled(C, b2, X, tgt_slot, bx, by);
  }
  // the list: A, B unless it repeats A, the temporal candidate only when the spatial ones left a place, zero vectors behind
  int l0x = 0, l0y = 0, l1x = 0, l1y = 0, n = 0;
  if (flagA) { l0x = ax; l0y = ay; n = 1; }
  if (flagB && !(flagA && ax == bx && ay == by)) { if (n == 0) { l0x = bx; l0y = by; } else { l1x = bx; l1y = by; } n++; }
  if (n < 2 && C.slice->tmvp) {
    int cx = 0, cy = 0;
    if (temporal_mv(C, xPb, yPb, nPbW, nPbH, ref_idx, X, cx, cy)) { if (n == 0) { l0x = cx; l0y = cy; } else { l1x = cx; l1y = cy; } n++; }
  }
  out_x = mvp_flag ? l1x : l0x; out_y = mvp_flag ? l1y : l0y;
}

struct MotionLds {
  MotionUnit cur[2][256];  // the CTB being derived and the one before it (its left neighbour), z-order; they alternate
  MotionUnit up[3 * 16];   // bottom unit rows of the CTBs above-left, above, above-right
  MotionSyntax msyn[256];  // what the parser read for the CTB's prediction units
  uint8_t usize[256], uipmc[256];   // the CTB's coding block sizes / prediction modes (unit maps)
  MotionUnit col[5 * 4];   // the collocated picture's units on the 16x16 grid of the CTB (+ one column to its right)
  SliceParams slice;       // the CTB's slice (the chain reads its fields at every prediction unit)
  int ref_poc[16];         // PicOrderCnt of the reference picture table's slots
  MotionUnit pu;           // the motion of the prediction unit lane 0 just derived (broadcast to the lanes that fill its units)
  int pu_geom[4];          // its rectangle in units relative to the CTB: x, y, w, h
  int pu_bad;
};

}  // namespace

__global__ __launch_bounds__(64) void k_motion(MotionArgs A)
{
  __shared__ MotionLds L;
  const int lane = (int)threadIdx.x;
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(A.ticket, 1u);
  const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  if (ticket >= A.num_rows) return;
  if (__hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // failed parse: the maps are garbage
  const RowDesc rd = A.rows[ticket];
  const PicParams& P = A.pics[rd.pic];
  if (!P.is_inter) return;
  const int cy = (int)rd.row, ctb_w = P.ctb_w, log2_ctb = P.log2_ctb, units_log2 = P.units_per_ctb_log2, units = 1 << units_log2;
  uint32_t* my_progress = A.row_progress + (size_t)(P.first_row + rd.row) * 3 + 2;   // slot 2 of the row: the reconstruction waves of a 4:2:0 picture use 0 and 1
  const uint32_t* up_progress = my_progress - 3;
  MotionUnit* field = (MotionUnit*)(A.arena + P.off_mf);
  const MotionSyntax* msyn_base = (const MotionSyntax*)(A.arena + P.off_msyn);
  const uint8_t* u_size = A.arena + P.off_u_size;
  const uint8_t* u_ipmc = A.arena + P.off_u_ipmc;
  const CtbInfo* ctb_info = (const CtbInfo*)(A.arena + P.off_ctb_info);
  const SliceParams* slices = (const SliceParams*)(A.arena + P.off_slices);
  int err = 0;
  for (int cx = 0; cx < ctb_w && !err; cx++) {
    if (cy > 0) {   // the row above: its CTB above-right is done (or the row is)
      uint32_t need = (uint32_t)(cx + 2);
      if (need > (uint32_t)ctb_w) need = (uint32_t)ctb_w;
      uint32_t spins = 0;
      // (relaxed poll + write-through data, the reconstruction kernel's protocol: an ACQUIRE here and a RELEASE fence per CTB on the producer side cost an
      //  L2 invalidate / write-back per CTB - on a chip with one L2 per XCD that was most of this kernel's 7.7 ms per 720p picture, not the derivation)
      while (__hip_atomic_load(up_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22) || __hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { err = DEV_ERR_TIMEOUT; break; }
      }
      if (err) break;
    }
    const int ctb_rs = cy * ctb_w + cx;
    const size_t base = (size_t)ctb_rs << units_log2;
    const CtbInfo ci = ctb_info[ctb_rs];
    MotionUnit* const cur = L.cur[cx & 1];
    MotionCtx C;
    C.P = &P; C.reftab = (const RefFrame*)(A.arena + P.off_reftab); C.field = field; C.cur = cur;
    C.left = L.cur[(cx & 1) ^ 1]; C.up = L.up;
    // the slice's parameters and the reference pictures' POCs: read at every prediction unit by the chain, so they live in LDS (16 + 16 dwords)
    if (lane < 16) ((uint32_t*)&L.slice)[lane] = ((const uint32_t*)(slices + ci.slice_idx))[lane];
    else if (lane < 32) L.ref_poc[lane - 16] = C.reftab[lane - 16].poc;
    mk_lds_sync();
    C.slice = &L.slice; C.ref_poc = L.ref_poc; C.col_lds = L.col;
    C.col = C.slice->tmvp ? (HIPDEC_GLOBAL const MotionUnit*)(uintptr_t)C.reftab[C.slice->col_slot].mf : nullptr;
    C.cx = cx; C.cy = cy; C.avail = ci.avail; C.log2_ctb = log2_ctb; C.units_log2 = units_log2; C.lmt = P.log2_min_tb; C.ctb_w = ctb_w;
    C.width = P.width; C.height = P.height; C.poc = P.poc; C.par_mrg = P.log2_par_mrg_level; C.lt_mask = P.lt_mask;
    const int x_ctb = cx << log2_ctb, y_ctb = cy << log2_ctb;
    // what the serial derivation reads goes to LDS first, with all lanes: the CTB's syntax records and unit maps, and the bottom unit rows of the
    // CTBs above (a lone lane chasing them through HBM one by one was 12 ms of a 720p picture's 60)
    const int side = 1 << (log2_ctb - 2);
    for (int i = lane; i < units; i += 64) { MotionUnit z{}; z.ref_idx[0] = z.ref_idx[1] = -1; cur[i] = z; L.msyn[i] = msyn_base[base + i]; }
    for (int i = lane; i < units / 4; i += 64) {
      ((uint32_t*)L.usize)[i] = ((const uint32_t*)(u_size + base))[i];
      ((uint32_t*)L.uipmc)[i] = ((const uint32_t*)(u_ipmc + base))[i];
    }
    if (cy > 0)
      for (int i = lane; i < 3 * side; i += 64) {
        const int ncx = cx - 1 + i / side;
        if (ncx >= 0 && ncx < ctb_w) {   // written by another wave of this launch: read with sc1 loads (four dwords per unit)
          const uint32_t* src = (const uint32_t*)&field[((size_t)((cy - 1) * ctb_w + ncx) << units_log2) + mk_interleave((uint32_t)(i % side), (uint32_t)(side - 1))];
          uint32_t* dst = (uint32_t*)&L.up[i];
          for (int k = 0; k < 4; k++) dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    C.col_w = (1 << (log2_ctb - 4)) + 1;
    if (C.col) {   // the collocated units the CTB's temporal candidates can name
      // The collocated picture may be an earlier picture of the same chain whose motion this very launch derives (its rows hold earlier tickets): the
      // candidates of this CTB name units of its CTB (cx, cy) and of the grid column right of it - never below the CTB row, 8.5.3.2.8 - so its row
      // cy must be past CTB cx + 1.  Pictures of a chain thus follow each other at a 2-CTB distance instead of one launch per picture.
      const uint32_t col_row = C.reftab[C.slice->col_slot].progress_row;
      if (col_row) {
        const uint32_t* col_progress = A.row_progress + (size_t)(col_row - 1 + (uint32_t)cy) * 3 + 2;
        uint32_t need = (uint32_t)(cx + 2);
        if (need > (uint32_t)ctb_w) need = (uint32_t)ctb_w;
        uint32_t spins = 0;
        while (__hip_atomic_load(col_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 22) || __hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { err = DEV_ERR_TIMEOUT; break; }
        }
        if (err) break;
      }
      const int rows16 = 1 << (log2_ctb - 4);
      for (int i = lane; i < C.col_w * rows16; i += 64) {
        const int x = x_ctb + 16 * (i % C.col_w), y = y_ctb + 16 * (i / C.col_w);
        if (x < P.width && y < P.height) {   // (possibly written by another wave of this launch: write-through data, sc1 loads - four dwords per unit)
          HIPDEC_GLOBAL const uint32_t* src = (HIPDEC_GLOBAL const uint32_t*)&C.col[unit_index(C, x, y)];
          uint32_t* dst = (uint32_t*)&L.col[i];
          for (int k = 0; k < 4; k++) dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    mk_lds_sync();
    int z = 0;
    while (z < units && !err) {
      const int ux = (int)mk_compact((uint32_t)z), uy = (int)mk_compact((uint32_t)z >> 1);
      if (x_ctb + ux * 4 >= P.width || y_ctb + uy * 4 >= P.height) { z++; continue; }
      const int log2cb = L.usize[z] >> 4;
      if (log2cb < 3 || log2cb > log2_ctb) { err = DEV_ERR_SYNTAX; break; }
      const int n_units = 1 << (2 * (log2cb - 2));
      const uint32_t pm = L.uipmc[z];
      if (pm & UM_INTER) {
        const int nCbS = 1 << log2cb, xCb = x_ctb + ux * 4, yCb = y_ctb + uy * 4;
        const MotionSyntax s0 = L.msyn[z];
        const int part_mode = (int)((s0.w0 >> 9) & 7u);
        const int q = nCbS >> 2, hf = nCbS >> 1;
        const int n_parts = part_mode == 0 ? 1 : (part_mode == 3 ? 4 : 2);
        for (int k = 0; k < n_parts && !err; k++) {
          // the partition's rectangle inside the coding block (Table 7-10)
          int rx = 0, ry = 0, rw = nCbS, rh = nCbS;
          switch (part_mode) {
            case 1: rh = hf; ry = k * hf; break;
            case 2: rw = hf; rx = k * hf; break;
            case 3: rw = rh = hf; rx = (k & 1) * hf; ry = (k >> 1) * hf; break;
            case 4: rh = k ? nCbS - q : q; ry = k ? q : 0; break;
            case 5: rh = k ? q : nCbS - q; ry = k ? nCbS - q : 0; break;
            case 6: rw = k ? nCbS - q : q; rx = k ? q : 0; break;
            case 7: rw = k ? q : nCbS - q; rx = k ? nCbS - q : 0; break;
            default: break;
          }
          if (lane == 0) {
            PbGeom g{xCb, yCb, nCbS, xCb + rx, yCb + ry, rw, rh, k};
            const uint32_t zk = mk_interleave((uint32_t)(ux + (rx >> 2)), (uint32_t)(uy + (ry >> 2)));
            const MotionSyntax sy = L.msyn[zk];
            Mo m = mo_none();
            int bad = !(sy.w0 & 0x8000u) || (int)((sy.w0 >> 12) & 3u) != k;
            if (!bad) {
              if (sy.w0 & 1u) m = derive_merge(C, g, part_mode, (int)((sy.w0 >> 1) & 7u));
              else {
                const int idc = C.slice->is_b ? (int)((sy.w0 >> 16) & 3u) : 0;
                if (idc > 2 || (idc == 2 && rw + rh == 12)) bad = 1;
                if (!bad && idc != 1) {   // list 0
                  const int ref_idx = (int)((sy.w0 >> 4) & 15u);
                  if (ref_idx >= num_ref_of(C, 0)) bad = 1;
                  else {
                    int px_ = 0, py_ = 0;
                    derive_mvp(C, g, 0, ref_idx, (int)((sy.w0 >> 8) & 1u), px_, py_);
                    const int ux_ = (px_ + (int16_t)(sy.mvd[0] & 0xffffu) + 65536) & 65535, uy_ = (py_ + (int16_t)(sy.mvd[0] >> 16) + 65536) & 65535;   // 8.5.3.2.1: wrapped into 16 bits
                    mo_set(m, 0, ux_ >= 32768 ? ux_ - 65536 : ux_, uy_ >= 32768 ? uy_ - 65536 : uy_, ref_idx);
                  }
                }
                if (!bad && idc != 0) {   // list 1
                  const int ref_idx = (int)((sy.w0 >> 18) & 15u);
                  if (ref_idx >= num_ref_of(C, 1)) bad = 1;
                  else {
                    int px_ = 0, py_ = 0;
                    derive_mvp(C, g, 1, ref_idx, (int)((sy.w0 >> 22) & 1u), px_, py_);
                    const int ux_ = (px_ + (int16_t)(sy.mvd[1] & 0xffffu) + 65536) & 65535, uy_ = (py_ + (int16_t)(sy.mvd[1] >> 16) + 65536) & 65535;
                    mo_set(m, 1, ux_ >= 32768 ? ux_ - 65536 : ux_, uy_ >= 32768 ? uy_ - 65536 : uy_, ref_idx);
                  }
                }
              }
              if (!bad) {
                if (m.r0 < 0 && m.r1 < 0) bad = 1;
                if (m.r0 >= num_ref_of(C, 0) || m.r1 >= num_ref_of(C, 1)) bad = 1;
              }
            }
            MotionUnit o{};
            o.ref_idx[0] = o.ref_idx[1] = -1;
            if (!bad) {
              if (m.r0 >= 0) {
                const int slot = slot_of(C, 0, m.r0);
                o.mv[0][0] = (int16_t)m.x0; o.mv[0][1] = (int16_t)m.y0; o.ref_idx[0] = (int8_t)m.r0;
                o.poc_delta[0] = (int16_t)mk_clip3(-32768, 32767, C.poc - C.ref_poc[slot]); o.slot_pred[0] = (uint8_t)slot;
              }
              if (m.r1 >= 0) {
                const int slot = slot_of(C, 1, m.r1);
                o.mv[1][0] = (int16_t)m.x1; o.mv[1][1] = (int16_t)m.y1; o.ref_idx[1] = (int8_t)m.r1;
                o.poc_delta[1] = (int16_t)mk_clip3(-32768, 32767, C.poc - C.ref_poc[slot]); o.slot_pred[1] = (uint8_t)slot;
              }
              // what later pictures' temporal candidates ask about this unit: were its reference pictures long-term ones NOW
              if (m.r0 >= 0) o.slot_pred[1] = (uint8_t)(o.slot_pred[1] | (((C.lt_mask >> (o.slot_pred[0] & 63)) & 1) << 6));
              if (m.r1 >= 0) o.slot_pred[1] = (uint8_t)(o.slot_pred[1] | (((C.lt_mask >> (o.slot_pred[1] & 63)) & 1) << 7));
            }
            o.slot_pred[0] = (uint8_t)(o.slot_pred[0] | (((pm & UM_SKIP) ? 2u : 1u) << 6));
            L.pu = o;
            L.pu_bad = bad;
            L.pu_geom[0] = ux + (rx >> 2); L.pu_geom[1] = uy + (ry >> 2); L.pu_geom[2] = rw >> 2; L.pu_geom[3] = rh >> 2;
          }
          mk_lds_sync();
          const MotionUnit o = L.pu;
          if (L.pu_bad) { err = DEV_ERR_SYNTAX; break; }
          const int gx = L.pu_geom[0], gy = L.pu_geom[1], gw = L.pu_geom[2], gh = L.pu_geom[3];
          for (int i = lane; i < gw * gh; i += 64) cur[mk_interleave((uint32_t)(gx + i % gw), (uint32_t)(gy + i / gw))] = o;
          mk_lds_sync();
        }
      }
      z += n_units;
    }
    if (err) break;
    // the CTB's units leave LDS (coalesced 16-byte stores; later KERNELS read them), the bottom unit row - what the row below reads while this
    // launch runs - additionally as write-through dword stores; the stores are drained and ONE relaxed progress store announces them (no fence)
    for (int i = lane; i < units; i += 64) field[base + i] = cur[i];
    for (int i = lane; i < 4 * side; i += 64) {
      const uint32_t zi = mk_interleave((uint32_t)(i >> 2), (uint32_t)(side - 1));
      __hip_atomic_store((uint32_t*)&field[base + zi] + (i & 3), ((const uint32_t*)&cur[zi])[i & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ... and the units on the 16x16 grid: what the temporal candidates of a LATER picture of the chain read while this launch runs
    for (int i = lane; i < units / 4; i += 64) {
      const uint32_t zi = (uint32_t)(i >> 2) << 4;
      __hip_atomic_store((uint32_t*)&field[base + zi] + (i & 3), ((const uint32_t*)&cur[zi])[i & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    mk_lds_sync();
    mk_drain_stores();
    if (lane == 0) __hip_atomic_store(my_progress, (uint32_t)(cx + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (err && lane == 0) atomicCAS((int*)A.status, 0, err | (int)(0x50000000u) | (int)(ticket << 8));
}

// ---- 8.5.3.3: motion-compensated prediction of every sample that belongs to an inter coded unit -------------------------------------------
namespace {
__device__ __forceinline__ int luma_tap(int frac, int i)
{
  // fL[frac][i] (Table 8-11), rows 1..3; row 0 is the integer position
  constexpr int8_t t[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
  return t[frac][i];
}
__device__ __forceinline__ int chroma_tap(int frac, int i)
{
  constexpr int8_t t[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
  return t[frac][i];
}
}  // namespace

// the 14-bit prediction sample of list X at (x, y) of the plane (8.5.3.3.3): integer copy, one separable pass, or both
template <typename Pix>
__device__ __forceinline__ int mc_sample(HIPDEC_GLOBAL const Pix* ref, size_t rstride, int W, int H, int plane, int bit_depth, int x, int y, int mvx, int mvy)
{
  const int shift1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4, shift3 = 14 - bit_depth > 2 ? 14 - bit_depth : 2;
  const int fbits = plane ? 3 : 2, taps = plane ? 4 : 8, before = plane ? 1 : 3;
  const int xf = mvx & ((1 << fbits) - 1), yf = mvy & ((1 << fbits) - 1);
  const int xi = x + (mvx >> fbits), yi = y + (mvy >> fbits);
#define MK_REF(xx, yy) ((int)ref[(size_t)mk_clip3(0, H - 1, (yy)) * rstride + (size_t)mk_clip3(0, W - 1, (xx))])
#define MK_TAP(f, i) (plane ? chroma_tap((f), (i)) : luma_tap((f), (i)))
  int v;
  if (!xf && !yf) v = MK_REF(xi, yi) << shift3;
  else if (!yf) { int a = 0; for (int i = 0; i < taps; i++) a += MK_TAP(xf, i) * MK_REF(xi + i - before, yi); v = a >> shift1; }
  else if (!xf) { int a = 0; for (int i = 0; i < taps; i++) a += MK_TAP(yf, i) * MK_REF(xi, yi + i - before); v = a >> shift1; }
  else {
    int a = 0;
    for (int j = 0; j < taps; j++) {
      int t = 0;
      for (int i = 0; i < taps; i++) t += MK_TAP(xf, i) * MK_REF(xi + i - before, yi + j - before);
      a += MK_TAP(yf, j) * (t >> shift1);
    }
    v = a >> 6;
  }
#undef MK_REF
#undef MK_TAP
  return v;
}

// add_residual: the sample leaves with its residual added (8.6.7 for the units of inter coded CUs) - then the reconstruction wavefront, a serial walk over
// the blocks of a CTB, has nothing left to do for them (ReconArgs::inter_from_plane)
template <typename Pix>
__global__ __launch_bounds__(256) void k_mc(FilterArgs A, int add_residual)
{
  const int pic = (int)blockIdx.z / 3, plane = (int)blockIdx.z % 3;
  if (__hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  const PicParams& P = A.pics[pic];
  if (!P.is_inter || (plane && !P.chroma_format_idc)) return;
  const int W = plane ? P.cwidth : P.width, H = plane ? P.cheight : P.height;
  const int x = (int)(blockIdx.x * 16 + (threadIdx.x & 15)), y = (int)(blockIdx.y * 16 + (threadIdx.x >> 4));
  if (x >= W || y >= H) return;
  const int sub = plane ? 1 : 0;                       // log2 subsampling (4:2:0 only)
  const int xl = x << sub, yl = y << sub;              // the luma sample that decides which prediction unit the sample belongs to
  const int log2_ctb = P.log2_ctb;
  const int cx = xl >> log2_ctb, cy = yl >> log2_ctb, m = (1 << (log2_ctb - 2)) - 1;
  const uint32_t z = mk_interleave((uint32_t)((xl >> 2) & m), (uint32_t)((yl >> 2) & m));
  const MotionUnit mu = ((const MotionUnit*)(A.arena + P.off_mf))[((size_t)(cy * P.ctb_w + cx) << P.units_per_ctb_log2) + z];
  if (mu.ref_idx[0] < 0 && mu.ref_idx[1] < 0) return;   // intra coded: k_recon predicts it
  const RefFrame* reftab = (const RefFrame*)(A.arena + P.off_reftab);
  const int bit_depth = plane ? P.bit_depth_chroma : P.bit_depth_luma;
  int pred[2] = {0, 0};
  for (int X = 0; X < 2; X++) {
    if (mu.ref_idx[X] < 0) continue;
    const RefFrame rf = reftab[mu.slot_pred[X] & 63u];
    pred[X] = mc_sample<Pix>((HIPDEC_GLOBAL const Pix*)(uintptr_t)rf.plane[plane],   // (an integer turned pointer is generic - FLAT loads - unless typed as global memory)
                             rf.stride[plane] / sizeof(Pix), W, H, plane, bit_depth, x, y, mu.mv[X][0], mu.mv[X][1]);
  }
  const int bi = mu.ref_idx[0] >= 0 && mu.ref_idx[1] >= 0, one = mu.ref_idx[0] >= 0 ? 0 : 1;
  const int shift1 = 14 - bit_depth, maxv = (1 << bit_depth) - 1;   // (bit depth <= 12)
  const SliceParams& sl = ((const SliceParams*)(A.arena + P.off_slices))[((const CtbInfo*)(A.arena + P.off_ctb_info))[cy * P.ctb_w + cx].slice_idx];
  int v;
  if (!sl.weighted) {   // 8.5.3.3.4.2
    v = bi ? (pred[0] + pred[1] + (1 << shift1)) >> (shift1 + 1) : (pred[one] + (1 << (shift1 - 1))) >> shift1;
  } else {              // 8.5.3.3.4.3
    const WeightTable& wt = ((const WeightTable*)(A.arena + P.off_wp))[sl.wp_index];
    const int log2wd = (plane ? sl.chroma_log2_wd : sl.luma_log2_wd) + shift1;
    if (bi) {
      const int w0 = wt.w[0][mu.ref_idx[0]][plane], w1 = wt.w[1][mu.ref_idx[1]][plane];
      const int o0 = wt.o[0][mu.ref_idx[0]][plane] << (bit_depth - 8), o1 = wt.o[1][mu.ref_idx[1]][plane] << (bit_depth - 8);
      v = (pred[0] * w0 + pred[1] * w1 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
    } else {
      const int w = wt.w[one][mu.ref_idx[one]][plane], o = wt.o[one][mu.ref_idx[one]][plane] << (bit_depth - 8);
      v = ((pred[one] * w + (1 << (log2wd - 1))) >> log2wd) + o;   // (log2WD >= 2: shift1 >= 2)
    }
  }
  v = mk_clip3(0, maxv, v);
  if (add_residual) {
    // the transform block that covers the sample: its size sits in the low nibble of the unit's size byte, its flags and its residual at its first unit
    // (z order; the residual arrays hold a CTB's blocks one behind the other, each block in raster order: residual_kernel.hip).  4:2:0 chroma: a block of
    // half the luma size - the 4x4 block of four 4x4 luma blocks hangs off the quad's 4th unit and sits at the quad's origin
    const size_t ctb_rs = (size_t)(cy * P.ctb_w + cx), base = ctb_rs << P.units_per_ctb_log2;
    const uint8_t* u_size = A.arena + P.off_u_size + base;
    const uint8_t* u_flags = A.arena + P.off_u_flags + base;
    const int tb = u_size[z] & 15;
    if (tb >= 2 && tb <= 5) {   // (anything else is a broken map: k_recon reports it)
      const uint32_t ux = (uint32_t)((xl >> 2) & m), uy = (uint32_t)((yl >> 2) & m), tmask = ~(uint32_t)((1 << (tb - 2)) - 1);
      const uint32_t z0 = mk_interleave(ux & tmask, uy & tmask);
      const int ctb2 = 1 << (2 * log2_ctb);
      int res = 0;
      if (!plane) {
        if (u_flags[z0] & UF_CBF_LUMA) res = ((const int16_t*)(A.arena + P.off_coeff[0]))[ctb_rs * (size_t)ctb2 + z0 * 16u + (uint32_t)(((y & ((1 << tb) - 1)) << tb) + (x & ((1 << tb) - 1)))];
      } else {
        const int16_t* coeff = (const int16_t*)(A.arena + P.off_coeff[plane]) + ctb_rs * (size_t)(ctb2 >> 2);
        const int bit = plane == 1 ? UF_CBF_CB : UF_CBF_CR;
        if (tb > 2) {
          const int lgc = tb - 1;
          if (u_flags[z0] & bit) res = coeff[z0 * 4u + (uint32_t)(((y & ((1 << lgc) - 1)) << lgc) + (x & ((1 << lgc) - 1)))];
        } else {
          const uint32_t zc = z & ~3u;
          if (u_flags[zc | 3u] & bit) res = coeff[zc * 4u + (uint32_t)(((y & 3) << 2) + (x & 3))];
        }
      }
      v = mk_clip3(0, maxv, v + res);
    }
  }
  Pix* rec = (Pix*)(A.arena + P.off_rec[plane]);
  rec[(size_t)y * (P.rec_stride[plane ? 1 : 0] / sizeof(Pix)) + x] = (Pix)v;
}

void launch_motion(const MotionArgs& a, hipStream_t s)
{
  if (!a.num_rows) return;
  hipLaunchKernelGGL(k_motion, dim3(a.num_rows), dim3(64), 0, s, a);
}

void launch_mc(const FilterArgs& a, int n_pics, int max_w, int max_h, bool wide, hipStream_t s, bool add_residual)
{
  if (n_pics <= 0) return;
  const dim3 grid((unsigned)((max_w + 15) / 16), (unsigned)((max_h + 15) / 16), (unsigned)(n_pics * 3));
  if (wide) hipLaunchKernelGGL(k_mc<uint16_t>, grid, dim3(256), 0, s, a, add_residual ? 1 : 0);
  else hipLaunchKernelGGL(k_mc<uint8_t>, grid, dim3(256), 0, s, a, add_residual ? 1 : 0);
}

}  // namespace hipdec
