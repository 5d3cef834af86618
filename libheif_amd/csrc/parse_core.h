// parse_core.h — the CABAC entropy decoder + slice-data syntax parser, written as ONE wave-uniform
// instruction stream over a handful of lane-indexed registers.
//
// Stands in for libde265's slice-data parser behind de265_decode() (reference call site
// libheif/plugins/decoder_libde265.cc:402).  Syntax and context selection per ITU-T H.265 7.3.8 /
// 9.3 (intra slices).
//
// MI355X mapping (why the code looks the way it does)
//   * CABAC is a serial dependency chain, so one 64-lane wavefront decodes one substream as ONE wave-uniform instruction stream
//     (every branch is a scalar branch).  All tables and all mutable state that a CPU decoder would keep in memory live in VGPRs
//     used as 64-entry register files addressed with v_readlane / v_writelane (a few cycles) instead of LDS (~50 cycles per
//     dependent access, microarch guide):
//        - context variables: three VGPRs, one context per lane (groups A / B / C below), pStateIdx | valMps << 6
//        - rangeTabLps (4 bytes per pStateIdx lane), transIdxLps, the 8x8 diagonal scan and its inverse
//        - a 256-byte bitstream window + the prefetched next window (one coalesced 256-B load per 256 bytes of bitstream;
//          emulation-prevention bytes are looked for once per window by all lanes)
//        - the CTB's per-4x4-unit maps (size, flags, intra modes, QP): four units per lane in z-scan order, so every CU / TU
//          is a contiguous lane range that is filled with one masked vector move and published with one coalesced store per map
//        - the previous CTB's maps (left neighbour), the row above's sizes, SAO parameters
//   * WHERE a wave-uniform value lives decides which pipe its arithmetic issues on.  The scalar ALU is shared by the CU's four
//     SIMDs (one instruction per cycle per CU), each SIMD has its own vector ALU; measured, the parser was scalar-bound (32 SALU +
//     8 branch against 23 VALU instructions per pixel, profiles/r02g_pmc_parse_b512.txt).  So the arithmetic decoder's state
//     (range / value / bit count: UReg) and the context-state update are deliberately kept in VECTOR registers although they are
//     uniform; syntax control flow, lane selects and the bin values the syntax branches on stay scalar.  After the rebalancing:
//     26.4 SALU + 26.1 VALU + 8 branch per pixel (profiles/r02l_pmc_parse_b512_after_valu_state.txt); round 3's hand-scheduled
//     statements for the bins: 23.4 + 21.9 + 7.3 (profiles/r03b_pmc_parse_b512.txt).
//   * the 64 lanes do the data-parallel side jobs (context init, window loads and emulation-prevention scan, map fills, per-sub-
//     block context selection and level finalisation, coefficient block flush, WPP context save / restore).
//
// The same source compiles for the device (parse_kernel.hip) and, with HIPDEC_HOST_EMU, as a
// lane-emulating host build used ONLY by the CPU tests (tests/emu) to check the parser's logic
// against the oracle without a GPU.  The emulation is never linked into libheifhip.so.
#pragma once
#include <stdint.h>
#include "hevc_device.h"

#if defined(HIPDEC_HOST_EMU)
#include <string.h>
#define PC_DEV static inline
struct VReg { uint32_t v[64]; };
#define PC_VEC_BEGIN for (int lane = 0; lane < 64; lane++) { (void)lane;
#define PC_VEC_END }
#define PC_L(r) ((r).v[lane])
PC_DEV uint32_t pc_rdlane(const VReg& r, int l) { return r.v[l & 63]; }
PC_DEV void pc_wrlane(VReg& r, int l, uint32_t x) { r.v[l & 63] = x; }
PC_DEV void pc_wrlane_v(VReg& r, int l, uint32_t x) { r.v[l & 63] = x; }   // value given in a vector register on the device
PC_DEV uint32_t pc_uni(uint32_t x) { return x; }
PC_DEV int pc_clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }
PC_DEV int pc_ffs(uint32_t x) { return __builtin_ffs((int)x); }
PC_DEV int pc_popc(uint32_t x) { return __builtin_popcount(x); }
PC_DEV uint64_t pc_ballot(const VReg& r) { uint64_t m = 0; for (int l = 0; l < 64; l++) if (r.v[l]) m |= 1ull << l; return m; }
// a wave-uniform value deliberately kept in a VECTOR register (see "VALU-resident arithmetic decoder" below)
typedef uint32_t UReg;
PC_DEV UReg pc_vec(uint32_t x) { return x; }
PC_DEV bool pc_any(bool b) { return b; }
PC_DEV float pc_rcp(float x) { return 1.0f / x; }
PC_DEV uint32_t pc_mul24(uint32_t a, uint32_t b) { return a * b; }
#define PC_LDS_SYNC() do { } while (0)
#define PC_CONST static const
// value of the lane below (lane 0: its own); only valid inside PC_VEC_BEGIN .. PC_VEC_END
#define PC_FROM_LANE_BELOW(r) ((r).v[lane ? lane - 1 : 0])
#else
#include <hip/hip_runtime.h>
#define PC_DEV __device__ __forceinline__
typedef uint32_t VReg;
#define PC_VEC_BEGIN { const int lane = (int)threadIdx.x; (void)lane;
#define PC_VEC_END }
#define PC_L(r) (r)
PC_DEV uint32_t pc_rdlane(const VReg& r, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)r, l); }
// v_writelane_b32 (hipcc 7.2 exposes no builtin for it): two instructions instead of move + compare + wait state + select.  gfx9
// encodings may read only ONE SGPR besides M0, so the lane select travels in M0 (what the compiler's own legalisation of this
// instruction does).  The value must be provably uniform for the "s" constraint.
PC_DEV void pc_wrlane(VReg& r, int l, uint32_t x)
{
  // (M0 in the clobber list draws a "reserved register" warning: the compiler does not allocate it; nothing else in these kernels
  //  uses it — gfx9 LDS instructions no longer need it — and every use here sets it immediately before reading it)
  // (readfirstlane is free for a value the compiler already holds in an SGPR and moves a uniform value out of a VGPR otherwise: the
  //  "s" constraint does not legalise by itself)
  const uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
  const int ls = __builtin_amdgcn_readfirstlane(l);
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(r) : "s"(xs), "s"(ls) : "m0");
}
// the same with the (wave-uniform) value in a VECTOR register: compare + select, no scalar instruction
PC_DEV void pc_wrlane_v(VReg& r, int l, uint32_t x) { r = ((int)threadIdx.x == l) ? x : r; }
PC_DEV uint32_t pc_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
PC_DEV int pc_clz(uint32_t x) { return __clz((int)x); }
PC_DEV int pc_ffs(uint32_t x) { return __ffs((int)x); }
PC_DEV int pc_popc(uint32_t x) { return __popc(x); }
PC_DEV uint64_t pc_ballot(const VReg& r) { return __ballot(r != 0); }
// a wave-uniform value deliberately kept in a VECTOR register: the asm move hides its uniformity from the compiler,
// so arithmetic on it is issued to the SIMD's VALU instead of the CU-shared scalar pipe
typedef uint32_t UReg;
PC_DEV UReg pc_vec(uint32_t x) { UReg r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
PC_DEV bool pc_any(bool b) { return __ballot(b) != 0; }   // uniform branch condition from a (uniform-valued) vector compare
PC_DEV float pc_rcp(float x) { return __builtin_amdgcn_rcpf(x); }                       // v_rcp_f32, 1 ulp
PC_DEV uint32_t pc_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }              // operands < 2^24
#define PC_LDS_SYNC() __syncthreads()
#define PC_CONST __constant__
#define PC_FROM_LANE_BELOW(r) ((uint32_t)__shfl_up((int)(r), 1))
#endif

#include "parse_tables.h"

#ifndef HIPDEC_PARSE_CHROMA_GENERAL
#define HIPDEC_PARSE_CHROMA_GENERAL 1   // 0: a build for 4:0:0 / 4:2:0 pictures only (see pc_is444)
#endif
#ifndef HIPDEC_PARSE_LDS_CTX
#define HIPDEC_PARSE_LDS_CTX 0          // 1: context variables and the rangeTabLps / transIdxLps tables live in LDS (the throughput kernel, see "LDS-resident contexts" below)
#endif
#ifndef HIPDEC_PARSE_INTER
#define HIPDEC_PARSE_INTER 0            // 1: the build that also parses P slices (sequence tracks; parse_kernel_inter.hip, the CPU emulation)
#endif

namespace hipdec {
namespace pcore {

struct Lds {
  alignas(16) int16_t coef[32 * 32];  // coefficient block being parsed (zero outside the parse of a block)
  alignas(16) uint32_t park[SAVE_DWORDS];   // staging image of a parked row's record (save_row_state / load_row_state)
#if HIPDEC_PARSE_LDS_CTX
  // LDS-resident contexts (round 6).  Measured on MI355X (profiles/r06_issue_model_*.txt): a VALU instruction that touches the scalar register
  // file - an SGPR operand, VCC, v_readlane / v_writelane, every v_cmp and v_cndmask - issues at HALF the rate of one that only reads and writes
  // VGPRs / inline constants (0.9 against 1.76 instructions per cycle and CU at 8 waves per SIMD), and k_parse sat at exactly that 0.9.  The
  // register-file form of a context-coded bin is 9 such instructions (two v_readlane, the write-back mask + select, operands that came out of a
  // v_readlane ...).  Here the context variable and its rangeTabLps row come from LDS with wave-uniform addresses (a broadcast read, the third
  // issue pipe of the CU, idle in this kernel), everything between them is pure-VGPR arithmetic on wave-uniform values, and only the two decisions
  // (MPS / LPS, renormalise or not) cross to the scalar side.
  alignas(16) uint32_t ctx[3 * 64];   // context variables, groups A | B | C: (p' << 2) | valMps << 16 with p' = 62 - pStateIdx (the low half is the byte offset of its row in tlps)
  uint32_t tlps[64];                  // entry p': rangeTabLps[62 - p'][0..3], one byte per qRangeIdx
  uint32_t tnext[64];                 // entry p': the variable after an LPS - (p'_next << 2), bit 16 set where the LPS flips valMps (pStateIdx 0)
  uint32_t vctx[16];                  // sig_coeff_flag run: byte address (inside this struct) of the context variable of each scan position
#endif
};
#if HIPDEC_PARSE_LDS_CTX
struct CtxGroup { int base; };         // a context group is an index range of Lds::ctx
typedef CtxGroup CtxRef;
#else
typedef VReg& CtxRef;                  // ... or a 64-lane register
#endif

// everything here is wave-uniform unless it is a VReg
struct PS {
#ifdef HIPDEC_PARSE_CYCLES   // measurement build (tools/ab_variant.sh cyc -DHIPDEC_PARSE_CYCLES): where does a row's time go?  printed per row
  unsigned long long c_resid = 0, c_flush = 0, c_wait = 0, c_cu = 0, c_pub = 0; uint32_t n_resid = 0;
#endif
  // ---- arithmetic decoder: range / value / bits_needed are wave-uniform values kept in VECTOR registers (UReg), so the
  //      decoder's arithmetic issues on the SIMD's VALU while the CU-shared scalar pipe keeps the syntax control flow
  UReg range, value, bits_needed;
  uint32_t pos, end, win_base;
  uint32_t fast_limit;   // bytes [pos, fast_limit) of the current window hold no emulation-prevention candidate: read without the 00 00 03 tracking
  int32_t zeros;
  int32_t err;
  const uint8_t* bs;
  // ---- lane-indexed register files
#if HIPDEC_PARSE_LDS_CTX
  static constexpr CtxGroup ctxA{0}, ctxB{64}, ctxC{128};
  VReg t_next;                                  // (only the scan tables in bits 13:8 / 29:24)
#else
  VReg ctxA, ctxB, ctxC;
  VReg t_lps, t_next;
#endif
  VReg win, win_next;
  VReg m_size, m_flags, m_ipm, m_ipmc, m_qp;  // 4 units per lane, z-scan order
  VReg p_left;                                  // left neighbour CTB: lane y = size byte | intra mode byte << 8 of its rightmost unit in unit row y
  VReg up;                                      // hand-off record of the CTB above: lanes 0..8 SaoParams dwords, lanes 9..12 the
                                                //   size bytes of its bottom unit row (4 units per lane)
  VReg sao, sao_left;                           // lanes 0..8: 3 dwords per component (SaoParams)
  // ---- picture constants (copied out of PicParams once)
  int32_t width, height, log2_ctb, log2_min_cb, log2_min_tb, log2_max_tb, max_th_depth_intra;
  int32_t chroma_format_idc, bit_depth_luma, bit_depth_chroma, log2_min_cu_qp_delta_size;
  uint32_t tools;  // bit0 sign_data_hiding 1 transform_skip 2 cu_qp_delta 3 transquant_bypass
  uint32_t pcm;    // PicParams bytes pcm_enabled, pcm_bd_luma, pcm_bd_chroma, pcm_cb_range
  // ---- slice
  int32_t slice_qp_y, deblock, sao_luma, sao_chroma;
  // ---- CTB
  int32_t x_ctb, y_ctb, ctb_avail;
  // ---- QP
  int32_t is_cu_qp_delta_coded, cu_qp_delta_val, qpy_pred, last_qp_y, cur_qp_y;
  int32_t cu_tq_bypass;
  Lds* L;
#if HIPDEC_PARSE_INTER
  // ---- P slices: slice_type P, num_ref_idx_l0_active, MaxNumMergeCand, initType; amp_enabled_flag, max_transform_hierarchy_depth_inter;
  //      the motion syntax records of the CTB being parsed (one per prediction unit, at the unit index of its first 4x4 unit)
  int32_t is_p, num_ref_idx, max_merge_cand, init_type, amp, max_th_depth_inter;
  int32_t is_b, num_ref_idx_l1, mvd_l1_zero;   // B slices: slice_type B, num_ref_idx_l1_active, mvd_l1_zero_flag
  MotionSyntax* msyn;
#endif
};
enum : uint32_t { TOOL_SDH = 1, TOOL_TS = 2, TOOL_CUQPD = 4, TOOL_TQBYPASS = 8 };

PC_DEV uint32_t interleave4(uint32_t x, uint32_t y)  // z-index of unit (x,y), x,y < 16
{
  x = (x | (x << 2)) & 0x33; x = (x | (x << 1)) & 0x55;
  y = (y | (y << 2)) & 0x33; y = (y | (y << 1)) & 0x55;
  return x | (y << 1);
}
PC_DEV uint32_t compact1by1(uint32_t v)
{
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu;
  return v;
}

// ---- lane-indexed byte maps ---------------------------------------------------------------------
// ChromaArrayType 3 / 2.  The throughput kernels of parse_kernel.hip are compiled with HIPDEC_PARSE_CHROMA_GENERAL = 0: the 4:4:4 / 4:2:2 block loops
// cost the scalar-bound 4:2:0 parser 1.5 % (measured), so batches that hold such pictures run the general build (parse_kernel_general.hip) instead
PC_DEV bool pc_is444(const PS& s) { return HIPDEC_PARSE_CHROMA_GENERAL && s.chroma_format_idc == 3; }
PC_DEV bool pc_is422(const PS& s) { return HIPDEC_PARSE_CHROMA_GENERAL && s.chroma_format_idc == 2; }
PC_DEV uint32_t map_get(const VReg& m, int z) { return (pc_rdlane(m, z >> 2) >> ((z & 3) * 8)) & 255u; }
// fills units [zb, zb + n) with byte b; n is 1 or a multiple of 4 with zb aligned to it
PC_DEV void map_fill(VReg& m, int zb, int n, uint32_t b)
{
  if (n >= 4) {
    const int l0 = zb >> 2, nl = n >> 2;
    const uint32_t w = b * 0x01010101u;
    PC_VEC_BEGIN
      if ((uint32_t)(lane - l0) < (uint32_t)nl) PC_L(m) = w;
    PC_VEC_END
  } else {
    const int sh = (zb & 3) * 8;
    uint32_t w = pc_rdlane(m, zb >> 2);
    w = (w & ~(255u << sh)) | (b << sh);
    pc_wrlane(m, zb >> 2, w);
  }
}

// ---- bitstream window + CABAC engine (9.3.4.3, scaled-window formulation) ---------------------
// Emulation-prevention bytes (00 00 03, one per ~4 MB of random payload) are looked for ONCE per 256-byte window, by all 64
// lanes: a window without a candidate ("03" behind a zero byte, or an "03" in front whose predecessors are unknown) is read
// through the short path of read_byte(), which neither tracks the zero run nor tests the byte; only windows with a candidate
// take the per-byte path.  Entering such a window at its first byte, the zero run is recovered from the two bytes in front of
// it (the short path does not maintain it); entered in the middle (a resumed row), the saved run is the per-byte path's own.
PC_DEV void load_window(PS& s, uint32_t base, uint32_t first_pos)
{
  const bool contiguous = base == s.win_base + 256u;
  uint32_t tail;             // the two bytes in front of the window
  if (contiguous) {
    tail = pc_rdlane(s.win, 63) >> 16;
    PC_VEC_BEGIN PC_L(s.win) = PC_L(s.win_next); PC_VEC_END
  } else {
    tail = base >= 4u ? pc_uni(*(const uint32_t*)(s.bs + base - 4u)) >> 16 : 0x0101u;
    PC_VEC_BEGIN PC_L(s.win) = *(const uint32_t*)(s.bs + base + 4u * (uint32_t)lane); PC_VEC_END
  }
  PC_VEC_BEGIN PC_L(s.win_next) = *(const uint32_t*)(s.bs + base + 256u + 4u * (uint32_t)lane); PC_VEC_END
  s.win_base = base;
  VReg cand;
  PC_VEC_BEGIN
    const uint32_t w = PC_L(s.win);
    const uint32_t below = PC_FROM_LANE_BELOW(s.win);
    const uint32_t b0 = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, b3 = w >> 24;
    const uint32_t p3 = lane ? below >> 24 : 0u;
    PC_L(cand) = ((b0 == 3u && p3 == 0u) || (b1 == 3u && b0 == 0u) || (b2 == 3u && b1 == 0u) || (b3 == 3u && b2 == 0u)) ? 1u : 0u;
  PC_VEC_END
  if (pc_ballot(cand) == 0) {
    const uint32_t lim = base + 256u;
    s.fast_limit = lim < s.end ? lim : s.end;
  } else {
    s.fast_limit = 0;
    if (first_pos == base) s.zeros = (tail & 0xffffu) == 0 ? 2 : ((tail >> 8) == 0 ? 1 : 0);
  }
}
PC_DEV uint32_t fetch_byte(PS& s, uint32_t pos)
{
  if ((pos & ~255u) != s.win_base) load_window(s, pos & ~255u, pos);
  return (pc_rdlane(s.win, (int)((pos >> 2) & 63u)) >> ((pos & 3u) * 8u)) & 255u;
}
PC_DEV uint32_t read_byte(PS& s)
{
  if (__builtin_expect(s.pos < s.fast_limit, 1)) {   // inside a window without emulation-prevention candidates, before the end
    const uint32_t p = s.pos++;
    return (pc_rdlane(s.win, (int)((p >> 2) & 63u)) >> ((p & 3u) * 8u)) & 255u;
  }
  uint32_t b;
  for (;;) {   // (a loop so that the window fetch is emitted once per call site)
    if (s.pos >= s.end) { s.pos++; if (s.pos > s.end + 8) s.err = DEV_ERR_BITSTREAM_END; return 0; }
    b = fetch_byte(s, s.pos++);
    if (s.fast_limit) break;                                                   // a fresh window without candidates
    if (s.zeros >= 2 && b == 3 && s.pos < s.end) { s.zeros = 0; continue; }  // emulation_prevention_three_byte
    s.zeros = b == 0 ? s.zeros + 1 : 0;
    break;
  }
  return b;
}
// the same byte as a wave-uniform VECTOR value: the arithmetic decoder adds it to its (vector-resident) window, so position masking, shift and
// extraction issue on the SIMD instead of the CU-shared scalar pipe (5 of the 8 scalar instructions of a byte read)
PC_DEV UReg read_byte_v(PS& s)
{
  if (__builtin_expect(s.pos < s.fast_limit, 1)) {
    const uint32_t p = s.pos++;
    const uint32_t dw = pc_rdlane(s.win, (int)(p >> 2));     // (v_readlane takes the lane from bits 5:0 of the select)
    const UReg sh = (pc_vec(p) & 3u) << 3;
    return ((UReg)dw >> sh) & 255u;
  }
  return pc_vec(read_byte(s));
}
PC_DEV void cabac_start(PS& s, uint32_t start, uint32_t end)
{
  s.pos = start; s.end = end; s.zeros = 0; s.win_base = 0xfffff000u; s.fast_limit = 0;
  s.range = pc_vec(510u << 7); s.bits_needed = pc_vec((uint32_t)-8);
  const uint32_t b0 = read_byte(s), b1 = read_byte(s);
  s.value = pc_vec((b0 << 8) | b1);
}
// Representation of the arithmetic decoder (9.3.4.3, scaled-window formulation):
//   s.range        ivlCurrRange << 7 (the scaled range the window is compared with: no shift per bin)
//   s.value        the scaled window (ivlOffset with 7 look-ahead bits), s.bits_needed in -8 .. -1 counts the shifts until the next byte
//   context var.   p' | valMps << 16 with p' = 62 - pStateIdx.  The MPS transition (pStateIdx + 1, saturating at 62: table 9-47) is then ONE
//                  packed 16-bit saturating subtraction (v_pk_sub_u16 ... clamp: low half p' - 1 >= 0, high half valMps - 0), and the variable
//                  itself is the lane select of both table reads (v_readlane takes the lane from bits 5:0).
//   t_lps          lane p': rangeTabLps[62 - p'][0..3], one byte per qRangeIdx
//   t_next         lane p': bits 5:0 the state after an LPS (62 - transIdxLps[62 - p']), bit 16 set where the LPS flips valMps (pStateIdx 0);
//                  bits 13:8 / 29:24 carry the 8x8 diagonal scan and its inverse (lane = scan position / raster index)
// Both pipes matter: the scalar ALU is shared by the CU's four SIMDs, so the decoder's arithmetic runs on the VALU (UReg) and only lane
// selects, the bin value and the syntax control flow are scalar (profiles/r02g_pmc_parse_b512.txt).
PC_DEV void refill_byte(PS& s) { s.value += read_byte_v(s) << s.bits_needed; s.bits_needed -= 8u; }

#if HIPDEC_PARSE_LDS_CTX
// the C++ form over the LDS-resident contexts (host emulation of this build, -DHIPDEC_PARSE_CXX_BINS on the device): same arithmetic as below with
// the variable's low half scaled by four
PC_DEV int decode_bin_cxx(PS& s, CtxRef grp, int ctx_lane)
{
  uint32_t* const cv = &s.L->ctx[grp.base + ctx_lane];
  const uint32_t st = pc_uni(*cv);
  const uint32_t row = pc_uni(s.L->tlps[(st & 0xfcu) >> 2]);
  const UReg lps = (row >> ((s.range >> 10) & 24u)) & 255u;
  UReg range = s.range - (lps << 7);
  uint32_t nst;
  int bin;
  UReg nb;
  if (__builtin_expect(pc_any(s.value < range), 1)) {
    bin = (int)(st >> 16);
    nst = st - (((st & 0xfcu) != 0u) ? 4u : 0u);
    nb = 1u - (range >> 15);
    range <<= nb;
  } else {
    bin = (int)((st >> 16) ^ 1u);
    nb = (UReg)pc_clz(lps) - 23u;
    s.value -= range;
    range = lps << (nb + 7u);
    nst = pc_uni(s.L->tnext[(st & 0xfcu) >> 2]) ^ (st & 0x10000u);
  }
  PC_VEC_BEGIN if (lane == 0) *cv = nst; PC_VEC_END
  s.range = range;
  s.value <<= nb;
  s.bits_needed += nb;
  if (__builtin_expect(pc_any((int32_t)s.bits_needed >= 0), 0)) refill_byte(s);
  return bin;
}
#else
PC_DEV int decode_bin_cxx(PS& s, VReg& grp, int ctx_lane)
{
  const uint32_t st = pc_rdlane(grp, ctx_lane);
  const uint32_t row = pc_rdlane(s.t_lps, (int)st);
  const UReg lps = (row >> ((s.range >> 10) & 24u)) & 255u;
  UReg range = s.range - (lps << 7);
  const UReg vst = pc_vec(st);
  UReg nst;
  int bin;
  UReg nb;
  if (__builtin_expect(pc_any(s.value < range), 1)) {   // MPS: at most one renormalisation shift
    bin = (int)(st >> 16);
    nst = vst - (((vst & 63u) != 0u) ? 1u : 0u);
    nb = 1u - (range >> 15);                                    // ivlCurrRange < 256 <=> range < 2^15 (range < 2^16 always)
    range <<= nb;
  } else {                                                      // LPS
    bin = (int)((st >> 16) ^ 1u);
    nb = (UReg)pc_clz(lps) - 23u;
    s.value -= range;
    range = lps << (nb + 7u);
    const UReg tr = pc_vec(pc_rdlane(s.t_next, (int)st));
    nst = (tr & 0x1003fu) ^ (vst & 0x10000u);
  }
  pc_wrlane_v(grp, ctx_lane, nst);
  s.range = range;
  s.value <<= nb;
  s.bits_needed += nb;
  if (__builtin_expect(pc_any((int32_t)s.bits_needed >= 0), 0)) refill_byte(s);
  return bin;
}
#endif

#if !defined(HIPDEC_HOST_EMU) && !defined(HIPDEC_PARSE_CXX_BINS)
#define PC_ASM_BINS 1
#if HIPDEC_PARSE_LDS_CTX
#include "parse_bins_lds_gfx950.h"
#else
#include "parse_bins_gfx950.h"
#endif
#else
PC_DEV int decode_bin(PS& s, CtxRef grp, int ctx_lane) { return decode_bin_cxx(s, grp, ctx_lane); }
PC_DEV int decode_unary_ctx_run(PS& s, CtxRef grp, int base_lane, int shift, int max)
{
  int i = 0;
  while (i < max && decode_bin(s, grp, base_lane + (i >> shift))) i++;
  return i;
}
PC_DEV uint32_t decode_g1_run(PS& s, int base_lane, int n, int& g)
{
  uint32_t gb = 0;
  for (int i = 0; i < n; i++) {
    const int b = decode_bin(s, s.ctxC, base_lane + (g > 3 ? 3 : g));
    gb = (gb << 1) | (uint32_t)b;
    if (b) g = 0; else if (g > 0) g++;
  }
  return gb;
}
PC_DEV uint32_t decode_sig_run(PS& s, const VReg& vctx, int n_start)
{
  uint32_t sig = 0;
  int k = n_start;
  do sig |= (uint32_t)decode_bin(s, s.ctxB, (int)pc_rdlane(vctx, k)) << k; while (--k > 0);
  return sig;
}
#endif

PC_DEV int decode_bypass(PS& s)
{
  s.value <<= 1;
  s.bits_needed += 1u;
  if (pc_any((int32_t)s.bits_needed >= 0)) { s.bits_needed = pc_vec((uint32_t)-8); s.value += read_byte_v(s); }
  if (pc_any(s.value >= s.range)) { s.value -= s.range; return 1; }
  return 0;
}
// n <= 8 bypass bins at once: n steps of 9.3.4.3.4 are one long division of the scaled window by the scaled
// range (quotient = the bins, MSB first; remainder = the new offset); at most one byte is needed
PC_DEV uint32_t decode_bypass_multi(PS& s, int n)
{
  s.value <<= n;
  s.bits_needed += (uint32_t)n;
  if (pc_any((int32_t)s.bits_needed >= 0)) refill_byte(s);
  const UReg scaled = s.range;
  // value < scaled * 2^n <= 2^24 and scaled < 2^16 are exact in fp32: the quotient estimate from one reciprocal is off by at
  // most one, which the remainder check repairs (an integer division expands to ~40 instructions)
  UReg q = (UReg)((float)s.value * pc_rcp((float)scaled));
  UReg r = s.value - pc_mul24(q, scaled);
  if (pc_any((int32_t)r < 0)) { q -= 1u; r += scaled; }
  else if (pc_any(r >= scaled)) { q += 1u; r -= scaled; }
  const uint32_t qmax = (1u << n) - 1u;
  if (pc_any(q > qmax)) { r += pc_mul24(q - qmax, scaled); q = pc_vec(qmax); }   // only reachable on a corrupt stream
  s.value = r;
  return pc_uni(q);
}
PC_DEV int decode_bypass_bits(PS& s, int n)   // n <= 32, MSB first
{
  if (n <= 8) {   // the common case (rice suffixes, most sign groups, last-position suffixes): one division, or one bin
    if (n >= 2) return (int)decode_bypass_multi(s, n);
    return n ? decode_bypass(s) : 0;
  }
  uint32_t v = 0;
  while (n > 0) {
    const int c = n > 8 ? 8 : n;
    if (c >= 3) v = (v << c) | decode_bypass_multi(s, c);
    else { for (int i = 0; i < c; i++) v = (v << 1) | (uint32_t)decode_bypass(s); }
    n -= c;
  }
  return (int)v;
}
PC_DEV int decode_terminate(PS& s)
{
  s.range -= 2u << 7;
  if (pc_any(s.value >= s.range)) return 1;
  if (pc_any(s.range < (256u << 7))) {
    s.range <<= 1;
    s.value <<= 1;
    s.bits_needed += 1u;
    if (pc_any((int32_t)s.bits_needed == 0)) { s.bits_needed = pc_vec((uint32_t)-8); s.value += read_byte(s); }
  }
  return 0;
}

// ---- context initialisation 9.3.2.2 (lane-parallel: one context per lane and group) ------------
PC_DEV void init_contexts(PS& s)
{
  const int qp = s.slice_qp_y < 0 ? 0 : (s.slice_qp_y > 51 ? 51 : s.slice_qp_y);
  PC_VEC_BEGIN
    for (int g = 0; g < 3; g++) {
#if HIPDEC_PARSE_INTER
      const int init = s.init_type ? c_init_p[s.init_type - 1][g][lane] : c_init[g][lane];
#else
      const int init = c_init[g][lane];
#endif
      const int m = (init >> 4) * 5 - 45, n = ((init & 15) << 3) - 16;
      int pre = ((m * qp) >> 4) + n;
      pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
      const int mps = pre <= 63 ? 0 : 1;
      const int p_state = mps ? pre - 64 : 63 - pre;
#if HIPDEC_PARSE_LDS_CTX
      s.L->ctx[g * 64 + lane] = (uint32_t)(((62 - p_state) << 2) | (mps << 16));
#else
      const uint32_t v = (uint32_t)((62 - p_state) | (mps << 16));   // p' | valMps << 16
      if (g == 0) PC_L(s.ctxA) = v; else if (g == 1) PC_L(s.ctxB) = v; else PC_L(s.ctxC) = v;
#endif
    }
  PC_VEC_END
  PC_LDS_SYNC();
}
PC_DEV void load_tables(PS& s)
{
  PC_VEC_BEGIN
    const int ps = lane < 63 ? 62 - lane : 63;   // lane p' serves pStateIdx 62 - p' (lane 63: the terminate state's row, never addressed)
    const uint32_t lps_row = (uint32_t)c_range_lps[ps * 4] | ((uint32_t)c_range_lps[ps * 4 + 1] << 8) | ((uint32_t)c_range_lps[ps * 4 + 2] << 16) |
                             ((uint32_t)c_range_lps[ps * 4 + 3] << 24);
#if HIPDEC_PARSE_LDS_CTX
    s.L->tlps[lane] = lps_row;
    s.L->tnext[lane] = (lane < 63 ? (uint32_t)(62 - c_next_lps[ps]) << 2 : 0u) | (lane == 62 ? 0x10000u : 0u);
    PC_L(s.t_next) = (uint32_t)c_diag8[lane] << 8;
#else
    PC_L(s.t_lps) = lps_row;
    PC_L(s.t_next) = (lane < 63 ? (uint32_t)(62 - c_next_lps[ps]) : 0u) | (lane == 62 ? 0x10000u : 0u) | ((uint32_t)c_diag8[lane] << 8);
#endif
  PC_VEC_END
  {   // inverse of the 8x8 diagonal scan, scattered with one masked move per position (once per substream)
    VReg inv;
    PC_VEC_BEGIN PC_L(inv) = 0u; PC_VEC_END
    for (int k = 0; k < 64; k++) pc_wrlane(inv, (int)((pc_rdlane(s.t_next, k) >> 8) & 63u), (uint32_t)k);
    PC_VEC_BEGIN PC_L(s.t_next) |= PC_L(inv) << 24; PC_VEC_END
  }
}

// ---- neighbour helpers over the z-ordered maps ---------------------------------------------------
// log2 CB size of the unit left of / above unit (ux, uy) of the current CTB, or 0 if unavailable
PC_DEV int left_cb_log2(PS& s, int ux, int uy)
{
  if (ux > 0) return (int)(map_get(s.m_size, (int)interleave4((uint32_t)ux - 1, (uint32_t)uy)) >> 4);
  if (s.ctb_avail & AV_LEFT) return (int)((pc_rdlane(s.p_left, uy) & 255u) >> 4);
  return 0;
}
PC_DEV int up_cb_log2(PS& s, int ux, int uy)
{
  if (uy > 0) return (int)(map_get(s.m_size, (int)interleave4((uint32_t)ux, (uint32_t)uy - 1)) >> 4);
  if (s.ctb_avail & AV_UP) return (int)(((pc_rdlane(s.up, 9 + (ux >> 2)) >> ((ux & 3) * 8)) & 255u) >> 4);
  return 0;
}
// 8.6.1 (qPY_A / qPY_B only count inside the current CTB)
PC_DEV void derive_qp_pred(PS& s, int ux, int uy)
{
  const int prev = s.last_qp_y;
  int a = prev, b = prev;
  if (ux > 0) a = (int8_t)map_get(s.m_qp, (int)interleave4((uint32_t)ux - 1, (uint32_t)uy));
  if (uy > 0) b = (int8_t)map_get(s.m_qp, (int)interleave4((uint32_t)ux, (uint32_t)uy - 1));
  s.qpy_pred = (a + b + 1) >> 1;
}
PC_DEV void set_qp_y(PS& s)
{
  const int off = 6 * (s.bit_depth_luma - 8);
  s.cur_qp_y = ((s.qpy_pred + s.cu_qp_delta_val + 52 + 2 * off) % (52 + off)) - off;
}

// ---- coefficient block staging (LDS block -> TU-contiguous int16 in HBM) -------------------------
struct alignas(16) Coef8 { int16_t v[8]; };
struct alignas(8) Coef4 { int16_t v[4]; };
PC_DEV void flush_coef(PS& s, int16_t* dst, int n2)
{
  PC_LDS_SYNC();
  PC_VEC_BEGIN
    if (n2 >= 64) {
      for (int i = lane * 8; i < n2; i += 512) {
        *(Coef8*)&dst[i] = *(const Coef8*)&s.L->coef[i];
        *(Coef8*)&s.L->coef[i] = Coef8{{0, 0, 0, 0, 0, 0, 0, 0}};
      }
    } else if (lane < 4) {  // 4x4: 32 bytes
      *(Coef4*)&dst[lane * 4] = *(const Coef4*)&s.L->coef[lane * 4];
      *(Coef4*)&s.L->coef[lane * 4] = Coef4{{0, 0, 0, 0}};
    }
  PC_VEC_END
  PC_LDS_SYNC();
}

// ---- 7.3.8.11 residual_coding ----------------------------------------------------------------
PC_DEV int decode_remaining(PS& s, int rice)
{
  int prefix = 0;
  while (prefix < 32 && decode_bypass(s)) prefix++;
  if (prefix >= 32) { s.err = DEV_ERR_SYNTAX; return 0; }
  if (prefix <= 3) return (prefix << rice) + decode_bypass_bits(s, rice);
  return (((1 << (prefix - 3)) + 3 - 1) << rice) + decode_bypass_bits(s, prefix - 3 + rice);
}
// coeff_abs_level_remaining (9.3.3.11: unary prefix, then rice / escape suffix - all bypass bins) in ONE step when the whole code is at most 8 bins
// long and the window needs no special handling: the next 8 bypass bins are the quotient of the window extended by the next byte (looked at,
// not consumed) by the range - sequential bypass decoding is long division, so the first L bins of that quotient are the L bins sequential
// decoding would produce -; prefix length, code length L and value come from the quotient's bits, and exactly L bins are committed (window,
// position) with the quotient already known.  Everything is vector arithmetic on wave-uniform values: the bin-by-bin form cost ~25 scalar
// instructions per coefficient.  Longer codes and windows with emulation-prevention candidates take decode_remaining().
PC_DEV UReg decode_remaining_v(PS& s, UReg rice)
{
  if (__builtin_expect(s.pos < s.fast_limit, 1)) {
    const uint32_t p = s.pos;
    const uint32_t dw = pc_rdlane(s.win, (int)(p >> 2));
    const UReg byte = ((UReg)dw >> ((pc_vec(p) & 3u) << 3)) & 255u;
    const UReg bn = s.bits_needed;                                  // -8 .. -1
    const UReg v8 = (s.value << 8) + (byte << (bn + 8u));           // what 8 steps of 9.3.4.3.4 would have shifted in (the refill always happens within 8)
    const UReg scaled = s.range;
    UReg q = (UReg)((float)v8 * pc_rcp((float)scaled));             // v8 < 2^24, scaled < 2^16: exact in fp32 up to an error of one, repaired below
    UReg r = v8 - pc_mul24(q, scaled);
    if (pc_any((int32_t)r < 0)) q -= 1u;
    else if (pc_any(r >= scaled)) q += 1u;
    if (__builtin_expect(!pc_any(q > 255u), 1)) {
      const UReg inv = (~q) & 255u;
      const UReg prefix = (UReg)pc_clz(inv | 1u) - 24u + ((inv == 0u) ? 1u : 0u);   // leading ones of the 8 bins; 8 if all are ones
      const UReg suffix_len = prefix <= 3u ? rice : prefix - 3u + rice;
      const UReg len = prefix + 1u + suffix_len;
      if (__builtin_expect(pc_any(len <= 8u), 1)) {
        const UReg suffix = (q >> (8u - len)) & ((1u << suffix_len) - 1u);
        const UReg val = (prefix <= 3u ? (prefix << rice) : ((((1u << (prefix - 3u)) + 2u) << rice))) + suffix;
        // commit len bins: the window after len shifts (with the byte if the shifts reached it) minus the bins' multiples of the range
        const UReg bn2 = bn + len;
        const bool took = pc_any((int32_t)bn2 >= 0);
        UReg vl = s.value << len;
        if (took) { vl += byte << bn2; s.pos = p + 1u; }
        s.value = vl - pc_mul24(q >> (8u - len), scaled);
        s.bits_needed = took ? bn2 - 8u : bn2;
        return val;
      }
      // codes of 9 .. 16 bins (escape codes of large levels: the first coefficients of a busy sub-block, before cRiceParam has grown): the next 8
      // bins are a second division of the first one's remainder extended by the byte after (both bytes only looked at)
      if (__builtin_expect(s.pos + 1u < s.fast_limit, 1)) {
        const uint32_t dw2 = pc_rdlane(s.win, (int)((p + 1u) >> 2));   // (the window register holds 256 bytes: p + 1 < fast_limit stays inside it)
        const UReg byte2 = ((UReg)dw2 >> ((pc_vec(p + 1u) & 3u) << 3)) & 255u;
        const UReg r8 = v8 - pc_mul24(q, scaled);                      // remainder after the first 8 bins (bits_needed is back at bn)
        const UReg v16 = (r8 << 8) + (byte2 << (bn + 8u));
        UReg q2 = (UReg)((float)v16 * pc_rcp((float)scaled));
        UReg r2 = v16 - pc_mul24(q2, scaled);
        if (pc_any((int32_t)r2 < 0)) q2 -= 1u;
        else if (pc_any(r2 >= scaled)) q2 += 1u;
        if (__builtin_expect(!pc_any(q2 > 255u), 1)) {
          const UReg bins = (q << 8) | q2;                              // 16 bins, first bin in bit 15
          const UReg inv16 = (~bins) & 0xffffu;
          const UReg prefix16 = (UReg)pc_clz(inv16 | 1u) - 16u + ((inv16 == 0u) ? 1u : 0u);
          const UReg suffix_len16 = prefix16 <= 3u ? rice : prefix16 - 3u + rice;
          const UReg len16 = prefix16 + 1u + suffix_len16;
          if (__builtin_expect(pc_any(len16 <= 16u), 1)) {
            const UReg suffix = (bins >> (16u - len16)) & ((1u << suffix_len16) - 1u);
            const UReg val = (prefix16 <= 3u ? (prefix16 << rice) : ((((1u << (prefix16 - 3u)) + 2u) << rice))) + suffix;
            const UReg m = len16 - 8u;                                  // bins taken from the second group (1 .. 8)
            const UReg bn3 = bn + m;
            const bool took2 = pc_any((int32_t)bn3 >= 0);
            UReg vl = r8 << m;
            if (took2) vl += byte2 << bn3;
            s.pos = p + (took2 ? 2u : 1u);
            s.value = vl - pc_mul24(q2 >> (8u - m), scaled);
            s.bits_needed = took2 ? bn3 - 8u : bn3;
            return val;
          }
        }
      }
    }
  }
  return pc_vec((uint32_t)decode_remaining(s, (int)pc_uni(rice)));
}
// scan of sub-blocks: lg = log2 of the sub-block grid width (0..3)
PC_DEV void scan_sb(PS& s, int lg, int scan_idx, int i, int& xs, int& ys)
{
  if (lg == 0) { xs = 0; ys = 0; }
  else if (lg == 1) {
    if (scan_idx == 1) { xs = i & 1; ys = i >> 1; }  // horizontal
    else { xs = i >> 1; ys = i & 1; }                // diagonal and vertical coincide for 2x2
  } else if (lg == 2) { const uint32_t v = (uint32_t)(PC_DIAG4 >> (i * 4)) & 15u; xs = (int)(v & 3u); ys = (int)(v >> 2); }
  else { const uint32_t v = (pc_rdlane(s.t_next, i) >> 8) & 63u; xs = (int)(v & 7u); ys = (int)(v >> 3); }
}

// returns transform_skip_flag; coefficients go to L->coef (raster, n x n)
PC_DEV int residual_coding(PS& s, int log2n, int c_idx, int pred_mode)
{
  const int n = 1 << log2n;
  int ts = 0;
  if ((s.tools & TOOL_TS) && !s.cu_tq_bypass && log2n <= 2) ts = decode_bin(s, s.ctxA, A_TRANSFORM_SKIP + (c_idx ? 1 : 0));
  int ctx_offset, ctx_shift;
  if (c_idx == 0) { ctx_offset = 3 * (log2n - 2) + ((log2n - 1) >> 2); ctx_shift = (log2n + 1) >> 2; }
  else { ctx_offset = 15; ctx_shift = log2n - 2; }
  const int c_max = (log2n << 1) - 1;
  const int px = decode_unary_ctx_run(s, s.ctxA, A_LAST_X + ctx_offset, ctx_shift, c_max);
  const int py = decode_unary_ctx_run(s, s.ctxA, A_LAST_Y + ctx_offset, ctx_shift, c_max);
  int last_x = px, last_y = py;
  if (px > 3) last_x = (1 << ((px >> 1) - 1)) * (2 + (px & 1)) + decode_bypass_bits(s, (px >> 1) - 1);
  if (py > 3) last_y = (1 << ((py >> 1) - 1)) * (2 + (py & 1)) + decode_bypass_bits(s, (py >> 1) - 1);
  int scan_idx = 0;
  if (log2n == 2 || (log2n == 3 && (c_idx == 0 || pc_is444(s)))) {   // 7.4.9.11: 8x8 chroma blocks too with ChromaArrayType 3
    if (pred_mode >= 6 && pred_mode <= 14) scan_idx = 2;
    else if (pred_mode >= 22 && pred_mode <= 30) scan_idx = 1;
  }
  if (scan_idx == 2) { const int t = last_x; last_x = last_y; last_y = t; }
  if (last_x >= n || last_y >= n) { s.err = DEV_ERR_SYNTAX; return ts; }
  const uint64_t scan4 = scan_idx == 0 ? PC_DIAG4 : (scan_idx == 1 ? PC_HORZ4 : PC_VERT4);

  // locate the last position in scan order — sub-block (last_x >> 2, last_y >> 2), position inside it — through the inverse scans
  const int lg = log2n - 2;  // log2 of the sub-block grid width
  int last_sb, last_pos;
  {
    const int xs_t = last_x >> 2, ys_t = last_y >> 2;
    const uint32_t r_t = (uint32_t)((last_x & 3) | ((last_y & 3) << 2));
    const uint64_t inv4 = scan_idx == 0 ? PC_INV_DIAG4 : (scan_idx == 1 ? PC_INV_HORZ4 : PC_INV_VERT4);
    last_pos = (int)((uint32_t)(inv4 >> (r_t * 4u)) & 15u);
    if (lg == 0) last_sb = 0;
    else if (lg == 1) last_sb = scan_idx == 1 ? (xs_t | (ys_t << 1)) : ((xs_t << 1) | ys_t);   // see scan_sb
    else if (lg == 2) last_sb = (int)((uint32_t)(PC_INV_DIAG4 >> ((uint32_t)(xs_t | (ys_t << 2)) * 4u)) & 15u);
    else last_sb = (int)((pc_rdlane(s.t_next, xs_t | (ys_t << 3)) >> 24) & 63u);
  }
  uint64_t csbf = 0;  // coded_sub_block_flag bitmap, bit (ys*8 + xs)
  const int sbw = 1 << lg;
  int g1_carry = 1, first_sb_with_g1 = 1;
  const int sdh = (s.tools & TOOL_SDH) != 0;
  VReg vovf;   // per lane: the largest |level| - (level < 0) it has stored for this block
  PC_VEC_BEGIN PC_L(vovf) = 0u; PC_VEC_END
  for (int i = last_sb; i >= 0; i--) {
    int xs, ys;
    scan_sb(s, lg, scan_idx, i, xs, ys);
    int infer_dc = 0, coded;
    const int right = (xs < sbw - 1) ? (int)((csbf >> (ys * 8 + xs + 1)) & 1) : 0;
    const int below = (ys < sbw - 1) ? (int)((csbf >> ((ys + 1) * 8 + xs)) & 1) : 0;
    if (i < last_sb && i > 0) {
      coded = decode_bin(s, s.ctxA, A_CODED_SUB_BLOCK + ((right | below) ? 1 : 0) + (c_idx ? 2 : 0));
      infer_dc = 1;
    } else coded = 1;
    if (!coded) continue;
    csbf |= 1ull << (ys * 8 + xs);
    // per-sub-block sig_coeff_flag context selection: pattern word + constant offset
    uint32_t pat;
    int sig_off;
    if (log2n == 2) { pat = 0; sig_off = c_idx ? 27 : 0; }
    else {
      const int prev_csbf = right | (below << 1);
      pat = prev_csbf == 0 ? PC_SIGPAT0 : (prev_csbf == 1 ? PC_SIGPAT1 : (prev_csbf == 2 ? PC_SIGPAT2 : PC_SIGPAT3));
      if (c_idx == 0) sig_off = ((xs | ys) ? 3 : 0) + ((log2n == 3) ? (scan_idx == 0 ? 9 : 15) : 21);
      else sig_off = 27 + ((log2n == 3) ? 9 : 12);
    }
    // sig_coeff_flag contexts of the 16 scan positions, one per lane (vector), then the serial bin loop
    VReg vctx;
    {
      const int dc_sb = (xs | ys) == 0;
      PC_VEC_BEGIN
        const uint32_t r = (uint32_t)(scan4 >> ((lane & 15) * 4)) & 15u;
        uint32_t c;
        if (log2n == 2) c = (uint32_t)sig_off + (uint32_t)((PC_CTXIDXMAP4 >> (r * 4)) & 15u);
        else if (dc_sb && r == 0) c = c_idx ? 27u : 0u;
        else c = (uint32_t)sig_off + ((pat >> (r * 2)) & 3u);
        PC_L(vctx) = c;
#if HIPDEC_PARSE_LDS_CTX
        if (lane < 16) s.L->vctx[lane] = (uint32_t)__builtin_offsetof(Lds, ctx) + 4u * (uint32_t)(PS::ctxB.base + B_SIG_COEFF) + 4u * c;
#endif
      PC_VEC_END
    }
    uint32_t sig = 0;  // bit k = sig_coeff_flag at scan position k
    int n_start = 15;
    if (i == last_sb) { sig = 1u << last_pos; n_start = last_pos - 1; }
    if (n_start > 0) {
      sig |= decode_sig_run(s, vctx, n_start);
    }
    if (n_start >= 0) {   // position 0: inferred significant when the sub-block was signalled coded and nothing else is
      if (infer_dc && !sig) sig = 1u;
      else sig |= (uint32_t)decode_bin(s, s.ctxB, B_SIG_COEFF + (int)pc_rdlane(vctx, 0));
    }
    if (!sig) continue;
    // greater1 / greater2 flags
    int ctx_set = (i == 0 || c_idx > 0) ? 0 : 2;
    if (!first_sb_with_g1 && g1_carry == 0) ctx_set++;
    first_sb_with_g1 = 0;
    const int last_sig_pos = 31 - pc_clz(sig), first_sig_pos = pc_ffs(sig) - 1;
    const int n_sig = pc_popc(sig), n_g1 = n_sig < 8 ? n_sig : 8;
    int g1_ctx = 1;
    const uint32_t gbits = decode_g1_run(s, C_GREATER1 + ctx_set * 4 + (c_idx ? 16 : 0), n_g1, g1_ctx);
    g1_carry = g1_ctx;
    // the run's flags back at their scan positions: the r-th significant position from the top carries flag r (r < 8)
    uint32_t g1, g1_coded;
    {
      VReg vcoded, vg1;
      PC_VEC_BEGIN
        const int k = lane & 15;
        const int rank = pc_popc(sig >> (k + 1));
        const uint32_t coded = (lane < 16 && ((sig >> k) & 1u) && rank < 8) ? 1u : 0u;
        PC_L(vcoded) = coded;
        PC_L(vg1) = coded & (gbits >> ((n_g1 - 1 - rank) & 31));
      PC_VEC_END
      g1_coded = (uint32_t)pc_ballot(vcoded);
      g1 = (uint32_t)pc_ballot(vg1);
    }
    const int sign_hidden = s.cu_tq_bypass ? 0 : (sdh && (last_sig_pos - first_sig_pos > 3));
    const uint32_t first_g1_bit = g1 ? 1u << (31 - pc_clz(g1)) : 0u;   // the first flag that was 1 (descending scan order)
    uint32_t g2 = 0;
    if (first_g1_bit && decode_bin(s, s.ctxB, B_GREATER2 + ctx_set + (c_idx ? 4 : 0))) g2 = first_g1_bit;
    // coeff_sign_flag: all of the sub-block's sign bins in one multi-bit bypass read (MSB = highest scan position)
    const uint32_t sig_signed = sign_hidden ? sig & ~(1u << first_sig_pos) : sig;
    const int n_signs = pc_popc(sig_signed);
    const uint32_t sign_bits = (uint32_t)decode_bypass_bits(s, n_signs);
    // coeff_abs_level_remaining for the positions whose base level hit its cap (9.3.3.11 order: descending k)
    const uint32_t need_rem = (g1_coded & g1 & ~(first_g1_bit & ~g2)) | (sig & ~g1_coded);
    VReg vrem, vbase;   // vbase: baseLevel of scan position k (1 + greater1 + greater2) on lane k, computed once for the sub-block
    PC_VEC_BEGIN PC_L(vrem) = 0u; PC_L(vbase) = 1u + ((g1 >> (lane & 15)) & 1u) + ((g2 >> (lane & 15)) & 1u); PC_VEC_END
    {
      uint32_t rem = need_rem;
      UReg rice = pc_vec(0u);   // cRiceParam and the level arithmetic stay on the vector side
      while (rem) {
        const int k = 31 - pc_clz(rem);
        rem &= ~(1u << k);
        const UReg r = decode_remaining_v(s, rice);
        const UReg abs_level = (UReg)pc_rdlane(vbase, k) + r;
        rice = (abs_level > (3u << rice) && rice < 4u) ? rice + 1u : rice;
        pc_wrlane_v(vrem, k, r);
      }
    }
    // levels, signs (incl. the hidden one) and positions of the 16 scan positions in parallel, one lane each
    VReg vabs, vneg, vodd;
    PC_VEC_BEGIN
      const int k = lane & 15;
      const uint32_t on = lane < 16 ? (sig >> k) & 1u : 0u;
      const uint32_t a = on ? 1u + ((g1 >> k) & 1u) + ((g2 >> k) & 1u) + PC_L(vrem) : 0u;
      const int rank = pc_popc(sig_signed >> (k + 1));       // sign bins decoded before this position's
      const uint32_t sgn = (lane < 16 && ((sig_signed >> k) & 1u)) ? (sign_bits >> (n_signs - 1 - rank)) & 1u : 0u;
      PC_L(vabs) = a; PC_L(vneg) = sgn; PC_L(vodd) = a & 1u;
    PC_VEC_END
    const uint32_t flip_first = sign_hidden ? (uint32_t)(pc_popc((uint32_t)pc_ballot(vodd)) & 1) : 0u;   // 9.3.4.? sign data hiding: parity of sumAbsLevel
    const int sb_base = (ys << 2) * n + (xs << 2);
    PC_VEC_BEGIN
      const int k = lane & 15;
      const uint32_t a = PC_L(vabs);
      uint32_t neg = PC_L(vneg);
      if (k == first_sig_pos) neg ^= flip_first;
      // a level outside -32768 .. 32767 (a > 32768, or 32768 without the sign): the largest a - neg of the block is looked at ONCE, behind the last
      // sub-block (a ballot per sub-block was 0.4 wave-instructions per pixel)
      { const uint32_t m = a ? a - neg : 0u; PC_L(vovf) = m > PC_L(vovf) ? m : PC_L(vovf); }   // (lanes 16 .. 63 hold a = 0)
      if (lane < 16 && a) {
        const uint32_t r = (uint32_t)(scan4 >> (k * 4)) & 15u;
        s.L->coef[sb_base + (int)(r >> 2) * n + (int)(r & 3u)] = (int16_t)(neg ? -(int32_t)a : (int32_t)a);
      }
    PC_VEC_END
  }
  {
    VReg vbad;
    PC_VEC_BEGIN PC_L(vbad) = PC_L(vovf) > 32767u ? 1u : 0u; PC_VEC_END
    if (pc_ballot(vbad)) s.err = DEV_ERR_SYNTAX;
  }
  return ts;
}

// ---- 7.3.8.14 cu_qp_delta -----------------------------------------------------------------------
PC_DEV void parse_cu_qp_delta(PS& s)
{
  int v = 0;
  if (decode_bin(s, s.ctxA, A_CU_QP_DELTA)) {
    v = 1;
    while (v < 5 && decode_bin(s, s.ctxA, A_CU_QP_DELTA + 1)) v++;
    if (v == 5) {
      int k = 0;
      while (decode_bypass(s)) { v += 1 << k; k++; if (k > 16) { s.err = DEV_ERR_SYNTAX; break; } }
      while (k-- > 0) v += decode_bypass(s) << k;
    }
  }
  const int sign = v ? decode_bypass(s) : 0;
  s.is_cu_qp_delta_coded = 1;
  s.cu_qp_delta_val = sign ? -v : v;
  const int off = 6 * (s.bit_depth_luma - 8);
  if (s.cu_qp_delta_val < -(26 + off / 2) || s.cu_qp_delta_val > 25 + off / 2) s.err = DEV_ERR_SYNTAX;
  set_qp_y(s);
}

// maps of one transform block (units zu .. zu + tu_units - 1): flags with the deblocking edges of its left column / top row (8.7.2.2 /
// 8.7.2.3), size byte, intra mode byte
PC_DEV void fill_tu_maps(PS& s, int zu, int tu_units, uint32_t fl, uint32_t ipm, uint32_t szb)
{
  const int tux0 = (int)compact1by1((uint32_t)zu), tuy0 = (int)compact1by1((uint32_t)zu >> 1);
  const int edge_l = s.deblock && (tux0 > 0 || (s.ctb_avail & AV_EDGE_LEFT));
  const int edge_t = s.deblock && (tuy0 > 0 || (s.ctb_avail & AV_EDGE_UP));
  const uint32_t ve = edge_l ? UF_VEDGE : 0u, he = edge_t ? UF_HEDGE : 0u;
  if (tu_units >= 4) {
    const int l0 = zu >> 2, nl = tu_units >> 2;
    PC_VEC_BEGIN
      const uint32_t rel = (uint32_t)(lane - l0);
      if (rel < (uint32_t)nl) {
        uint32_t wf = 0;
        for (int k = 0; k < 4; k++) {
          const uint32_t i = rel * 4u + (uint32_t)k;   // unit index inside the TU, z-order
          uint32_t f = fl;
          if ((i & 0x55555555u) == 0) f |= ve;         // x == 0
          if ((i & 0xAAAAAAAAu) == 0) f |= he;         // y == 0
          wf |= f << (8 * k);
        }
        PC_L(s.m_flags) = wf;
        PC_L(s.m_size) = szb * 0x01010101u;
        PC_L(s.m_ipm) = ipm * 0x01010101u;
      }
    PC_VEC_END
  } else {
    map_fill(s.m_flags, zu, 1, fl | ve | he);
    map_fill(s.m_size, zu, 1, szb);
    map_fill(s.m_ipm, zu, 1, ipm);
  }
}

// 7.3.8.7 pcm_sample of a coding unit whose pcm_flag (a terminate bin) was 1.  The arithmetic decoder stops on a byte boundary of the
// payload: with the scaled window of this engine the next unread byte (s.pos) is the first sample byte (the decoder has consumed
// 8 * bytes - (-bits_needed - 1) bits; the bit it read last lies in the byte before s.pos, pcm_alignment_zero_bits fill the rest of it).
// The samples go where a transform block of CU size would put its levels (one block per component), so that the reconstruction kernel
// finds them as the block's "residual"; the decoder is initialised again behind them (9.3.2.5), the context variables are kept.
PC_DEV void pcm_coding_unit(PS& s, int zb, int log2cb, int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr)
{
  set_qp_y(s);
  const int n_units = 1 << (2 * (log2cb - 2));
  uint32_t acc = 0;
  int nacc = 0;
  // chroma: one block of half the CU size (4:2:0), of CU size (4:4:4), or - 4:2:2: half as wide, as tall - two square blocks one above the other,
  // which in raster order are simply the first and the second half of the samples
  const int cfi = pc_is444(s) ? 3 : (pc_is422(s) ? 2 : (s.chroma_format_idc ? 1 : 0));
  for (int k = 0; k < (cfi ? (cfi == 2 ? 5 : 3) : 1); k++) {
    const int c = cfi == 2 ? (k + 1) >> 1 : k, t = cfi == 2 ? ((k + 1) & 1) : 0;   // 4:2:2: k = 0 luma, 1 / 2 Cb upper / lower, 3 / 4 Cr
    const int lg = (c && cfi != 3) ? log2cb - 1 : log2cb, n2 = 1 << (2 * lg);
    const int depth = (int)((s.pcm >> (c ? 16 : 8)) & 255u), shift = (c ? s.bit_depth_chroma : s.bit_depth_luma) - depth;
    for (int i = 0; i < n2; i++) {
      while (nacc < depth) { acc = (acc << 8) | read_byte(s); nacc += 8; }
      const uint32_t v = (acc >> (nacc - depth)) & ((1u << depth) - 1u);
      nacc -= depth;
      PC_VEC_BEGIN if (lane == 0) s.L->coef[i] = (int16_t)(v << shift); PC_VEC_END
    }
    flush_coef(s, c == 0 ? coef_y + zb * 16 : (c == 1 ? coef_cb : coef_cr) + zb * (cfi == 3 ? 16 : (cfi == 2 ? 8 : 4)) + t * n2, n2);
  }
  s.range = pc_vec(510u << 7); s.bits_needed = pc_vec((uint32_t)-8);
  {
    const uint32_t b0 = read_byte(s), b1 = read_byte(s);
    s.value = pc_vec((b0 << 8) | b1);
  }
  fill_tu_maps(s, zb, n_units, (uint32_t)(UF_PCM | (s.cu_tq_bypass ? UF_BYPASS : 0)), 1u, (uint32_t)((log2cb << 4) | log2cb));
  map_fill(s.m_ipmc, zb, n_units, 1u);
  map_fill(s.m_qp, zb, n_units, (uint32_t)(uint8_t)(int8_t)s.cur_qp_y);
  s.last_qp_y = s.cur_qp_y;
}

#if HIPDEC_PARSE_INTER
// ---- P slices: 7.3.8.5 (cu_skip_flag, pred_mode_flag, inter part_mode), 7.3.8.6 prediction_unit, 7.3.8.9 mvd_coding, rqt_root_cbf ---------------
// The parser only PARSES: what a prediction unit codes (merge_flag / merge_idx, or ref_idx / mvd / mvp flag) goes to its MotionSyntax record;
// the candidate lists need the neighbours' final motion, i.e. the 2-CTB wavefront of k_motion (inter_kernels.hip), not the entropy decoder's order.
enum : int { PM_2Nx2N = 0, PM_2NxN, PM_Nx2N, PM_NxN, PM_2NxnU, PM_2NxnD, PM_nLx2N, PM_nRx2N };   // Table 7-10

PC_DEV int decode_egk(PS& s, int k)   // k-th order Exp-Golomb, bypass bins (9.3.3.5)
{
  int v = 0;
  while (decode_bypass(s)) { v += 1 << k; k++; if (k > 20) { s.err = DEV_ERR_SYNTAX; return 0; } }
  if (k) v += decode_bypass_bits(s, k);
  return v;
}

PC_DEV int parse_part_mode_inter(PS& s, int log2cb)
{
  if (decode_bin(s, s.ctxA, A_PART_MODE)) return PM_2Nx2N;
  if (log2cb == s.log2_min_cb) {
    if (decode_bin(s, s.ctxC, C_PART_MODE_INTER + 0)) return PM_2NxN;
    if (log2cb == 3) return PM_Nx2N;
    return decode_bin(s, s.ctxC, C_PART_MODE_INTER + 1) ? PM_Nx2N : PM_NxN;
  }
  if (!s.amp) return decode_bin(s, s.ctxC, C_PART_MODE_INTER + 0) ? PM_2NxN : PM_Nx2N;
  if (decode_bin(s, s.ctxC, C_PART_MODE_INTER + 0)) {
    if (decode_bin(s, s.ctxC, C_PART_MODE_INTER + 2)) return PM_2NxN;
    return decode_bypass(s) ? PM_2NxnD : PM_2NxnU;
  }
  if (decode_bin(s, s.ctxC, C_PART_MODE_INTER + 2)) return PM_Nx2N;
  return decode_bypass(s) ? PM_nRx2N : PM_nLx2N;
}

// mvd_coding (7.3.8.9): mvd_x | mvd_y << 16 as int16 halves
PC_DEV uint32_t parse_mvd(PS& s)
{
  int mvd_x = 0, mvd_y = 0;
  const int gx = decode_bin(s, s.ctxC, C_MVD_GT0), gy = decode_bin(s, s.ctxC, C_MVD_GT0);
  int g1x = 0, g1y = 0;
  if (gx) g1x = decode_bin(s, s.ctxC, C_MVD_GT1);
  if (gy) g1y = decode_bin(s, s.ctxC, C_MVD_GT1);
  if (gx) { int a = 1; if (g1x) a = decode_egk(s, 1) + 2; if (a > 32768) s.err = DEV_ERR_SYNTAX; mvd_x = decode_bypass(s) ? -a : a; }
  if (gy) { int a = 1; if (g1y) a = decode_egk(s, 1) + 2; if (a > 32768) s.err = DEV_ERR_SYNTAX; mvd_y = decode_bypass(s) ? -a : a; }
  return ((uint32_t)mvd_x & 0xffffu) | ((uint32_t)mvd_y << 16);
}

PC_DEV int parse_ref_idx(PS& s, int num_ref)   // TR, cMax = num_ref_idx_lX_active_minus1: bins 0 and 1 context coded, the rest bypass
{
  int ref_idx = 0;
  const int cmax = num_ref - 1;
  while (ref_idx < cmax) {
    const int b = ref_idx < 2 ? decode_bin(s, s.ctxC, C_REF_IDX + ref_idx) : decode_bypass(s);
    if (!b) break;
    ref_idx++;
  }
  return ref_idx;
}

// one prediction unit: its syntax into the record at unit index z (CTB-local z-order); returns merge_flag.  ct_depth: the coding quadtree depth
// of the unit (context of inter_pred_idc); small: an 8x4 / 4x8 block (no bi-prediction: inter_pred_idc has one bin)
PC_DEV int prediction_unit(PS& s, int z, int part_mode, int part_idx, int cu_skip, int ct_depth, int small)
{
  int merge_flag = 1, merge_idx = 0, idc = 0, ref_idx[2] = {0, 0}, mvp_flag[2] = {0, 0};
  uint32_t mvd[2] = {0, 0};
  if (!cu_skip) merge_flag = decode_bin(s, s.ctxC, C_MERGE_FLAG);
  if (merge_flag) {
    if (s.max_merge_cand > 1 && decode_bin(s, s.ctxC, C_MERGE_IDX)) {
      merge_idx = 1;
      while (merge_idx < s.max_merge_cand - 1 && decode_bypass(s)) merge_idx++;
    }
  } else {
    if (s.is_b) {   // inter_pred_idc (9.3.3.8): PRED_L0 0, PRED_L1 1, PRED_BI 2
      if (!small && decode_bin(s, s.ctxC, C_INTER_PRED_IDC + ct_depth)) idc = 2;
      else idc = decode_bin(s, s.ctxC, C_INTER_PRED_IDC + 4);
    }
    if (idc != 1) {
      ref_idx[0] = parse_ref_idx(s, s.num_ref_idx);
      mvd[0] = parse_mvd(s);
      mvp_flag[0] = decode_bin(s, s.ctxC, C_MVP_FLAG);
    }
    if (idc != 0) {
      ref_idx[1] = parse_ref_idx(s, s.num_ref_idx_l1);
      if (!(s.mvd_l1_zero && idc == 2)) mvd[1] = parse_mvd(s);
      mvp_flag[1] = decode_bin(s, s.ctxC, C_MVP_FLAG);
    }
  }
  const uint32_t w0 = (uint32_t)merge_flag | ((uint32_t)merge_idx << 1) | ((uint32_t)ref_idx[0] << 4) | ((uint32_t)mvp_flag[0] << 8) | ((uint32_t)part_mode << 9) |
                      ((uint32_t)part_idx << 12) | 0x8000u | ((uint32_t)idc << 16) | ((uint32_t)ref_idx[1] << 18) | ((uint32_t)mvp_flag[1] << 22);
  MotionSyntax* dst = s.msyn + z;
  const uint32_t m0 = mvd[0], m1 = mvd[1];
  PC_VEC_BEGIN if (lane == 0) { dst->w0 = w0; dst->mvd[0] = m0; dst->mvd[1] = m1; } PC_VEC_END
  return merge_flag;
}

// a coding unit of a P slice that is not intra coded (cu_skip_flag = 1, or pred_mode_flag = 0)
PC_DEV void inter_coding_unit(PS& s, int zb, int log2cb, int cu_skip, int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr)
{
  const int ux0 = (int)compact1by1((uint32_t)zb), uy0 = (int)compact1by1((uint32_t)zb >> 1);
  const int n_units = 1 << (2 * (log2cb - 2)), nw = 1 << (log2cb - 2);   // units in the CU, units per CU side
  set_qp_y(s);
  const uint32_t bypass = s.cu_tq_bypass ? UF_BYPASS : 0u;
  map_fill(s.m_size, zb, n_units, (uint32_t)(log2cb << 4));
  map_fill(s.m_flags, zb, n_units, bypass);
  map_fill(s.m_ipm, zb, n_units, 1u);                                     // a neighbour that is not intra coded counts as INTRA_DC (8.4.2)
  map_fill(s.m_ipmc, zb, n_units, 1u | UM_INTER | (cu_skip ? UM_SKIP : 0u));
  int part_mode = PM_2Nx2N;
  if (!cu_skip) {
    part_mode = parse_part_mode_inter(s, log2cb);
    if (part_mode == PM_NxN && log2cb == 3) { s.err = DEV_ERR_SYNTAX; part_mode = PM_2Nx2N; }
  }
  // partitions (Table 7-10): offsets in 4x4 units; AMP quarters are whole units because AMP needs a coding block of 16 and up
  const int q = nw >> 2, hf = nw >> 1;
  int n_parts = 1, px[4] = {0, 0, 0, 0}, py[4] = {0, 0, 0, 0};
  int vx = -1, hy = -1;   // prediction block edges inside the unit (deblocking edges, 8.7.2.3), in units from the CU origin
  switch (part_mode) {
    case PM_2NxN: n_parts = 2; py[1] = hf; hy = hf; break;
    case PM_Nx2N: n_parts = 2; px[1] = hf; vx = hf; break;
    case PM_NxN: n_parts = 4; px[1] = hf; py[2] = hf; px[3] = hf; py[3] = hf; vx = hy = hf; break;
    case PM_2NxnU: n_parts = 2; py[1] = q; hy = q; break;
    case PM_2NxnD: n_parts = 2; py[1] = nw - q; hy = nw - q; break;
    case PM_nLx2N: n_parts = 2; px[1] = q; vx = q; break;
    case PM_nRx2N: n_parts = 2; px[1] = nw - q; vx = nw - q; break;
    default: break;
  }
  int merge0 = 0;
  for (int k = 0; k < n_parts && !s.err; k++) {
    const int z = (int)interleave4((uint32_t)(ux0 + px[k]), (uint32_t)(uy0 + py[k]));
    const int mf = prediction_unit(s, z, part_mode, k, cu_skip, s.log2_ctb - log2cb, log2cb == 3 && part_mode != PM_2Nx2N);
    if (k == 0) merge0 = mf;
  }
  int rqt_root_cbf = 0;
  if (!cu_skip) {
    rqt_root_cbf = 1;
    if (!(part_mode == PM_2Nx2N && merge0)) rqt_root_cbf = decode_bin(s, s.ctxC, C_RQT_ROOT_CBF);
  }
  if (!rqt_root_cbf) {
    // no transform tree: for the maps the unit is covered by transform blocks of min(CB size, 32) without coefficients whose only deblocking
    // edges are the coding unit's own left / top edge (fill_tu_maps over the whole unit marks exactly those)
    fill_tu_maps(s, zb, n_units, bypass, 1u, (uint32_t)((log2cb << 4) | (log2cb > 5 ? 5 : log2cb)));
  } else {
    // ---- transform tree of an inter coding unit (7.3.8.8): no IntraSplitFlag, interSplitFlag (7.4.9.8), cbf_luma inferred 1 at a root leaf
    //      without chroma coefficients; chroma 4:2:0 / 4:0:0 only (the host refuses P slices of other formats)
    const int max_trafo_depth = s.max_th_depth_inter;
    const int inter_split = s.max_th_depth_inter == 0 && part_mode != PM_2Nx2N;
    uint32_t cbf_cb_bits = 0, cbf_cr_bits = 0;
    int qn = 0;
    while (qn < n_units && !s.err) {
      int t;
      if (qn == 0) t = log2cb; else { t = 2 + ((pc_ffs((uint32_t)qn) - 1) >> 1); if (t > log2cb) t = log2cb; }
      for (;;) {
        const int depth = log2cb - t;
        int split;
        if (t <= s.log2_max_tb && t > s.log2_min_tb && depth < max_trafo_depth) split = decode_bin(s, s.ctxA, A_SPLIT_TRANSFORM + 5 - t);
        else split = (t > s.log2_max_tb || (inter_split && depth == 0)) ? 1 : 0;
        if (s.chroma_format_idc) {
          const uint32_t bit = 1u << depth, pbit = depth ? (1u << (depth - 1)) : 0;
          if (t > 2) {
            int cb = 0, cr = 0;
            if (depth == 0 || (cbf_cb_bits & pbit)) cb = decode_bin(s, s.ctxA, A_CBF_CHROMA + depth);
            if (depth == 0 || (cbf_cr_bits & pbit)) cr = decode_bin(s, s.ctxA, A_CBF_CHROMA + depth);
            cbf_cb_bits = (cbf_cb_bits & ~bit) | (cb ? bit : 0);
            cbf_cr_bits = (cbf_cr_bits & ~bit) | (cr ? bit : 0);
          } else {
            cbf_cb_bits = (cbf_cb_bits & ~bit) | ((cbf_cb_bits & pbit) << 1);
            cbf_cr_bits = (cbf_cr_bits & ~bit) | ((cbf_cr_bits & pbit) << 1);
          }
        }
        if (!split) break;
        t--;
      }
      const int depth = log2cb - t;
      const int zu = zb + qn;
      const int tu_units = 1 << (2 * (t - 2));
      const int cbf_cb = (int)((cbf_cb_bits >> depth) & 1u), cbf_cr = (int)((cbf_cr_bits >> depth) & 1u);
      int cbf_luma = 1;
      if (depth != 0 || cbf_cb || cbf_cr) cbf_luma = decode_bin(s, s.ctxA, A_CBF_LUMA + (depth == 0 ? 1 : 0));
      if ((cbf_luma | cbf_cb | cbf_cr) && (s.tools & TOOL_CUQPD) && !s.is_cu_qp_delta_coded) parse_cu_qp_delta(s);
      int do_chroma = 0, zc = zu, tc = t - 1;
      if (s.chroma_format_idc) {
        if (t > 2) do_chroma = 1;
        else if ((qn & 3) == 3) { do_chroma = 1; zc = zb + (qn & ~3); tc = 2; }
      }
      uint32_t ts_bits = 0;
      const uint32_t coded_bits = (uint32_t)cbf_luma | (do_chroma ? (uint32_t)(cbf_cb << 1) | (uint32_t)(cbf_cr << 2) : 0u);
#pragma clang loop unroll(disable)
      for (int k = 0; k < 3; k++) {
        if (!((coded_bits >> k) & 1u)) continue;
        const int lg = k == 0 ? t : tc;
        int16_t* dst = k == 0 ? coef_y + zu * 16 : (k == 1 ? coef_cb : coef_cr) + zc * 4;
        ts_bits |= (uint32_t)residual_coding(s, lg, k, 1 /* not intra: the up-right diagonal scan */) << k;
        flush_coef(s, dst, 1 << (2 * lg));
      }
      fill_tu_maps(s, zu, tu_units,
                   (uint32_t)((cbf_luma ? UF_CBF_LUMA : 0) | ((do_chroma && cbf_cb) ? UF_CBF_CB : 0) | ((do_chroma && cbf_cr) ? UF_CBF_CR : 0) | bypass |
                              ((ts_bits & 1u) ? UF_TS_LUMA : 0)),
                   (uint32_t)(1u | ((ts_bits & 2u) ? 64u : 0u) | ((ts_bits & 4u) ? 128u : 0u)), (uint32_t)((log2cb << 4) | t));
      qn += tu_units;
    }
  }
  // prediction block edges inside the coding unit (only those on the 8x8 luma grid get filtered; the deblocking kernel checks that)
  if (s.deblock && (vx >= 0 || hy >= 0)) {
    PC_VEC_BEGIN
      uint32_t add = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t z = (uint32_t)lane * 4u + (uint32_t)k;
        const int ux = (int)compact1by1(z) - ux0, uy = (int)compact1by1(z >> 1) - uy0;
        if ((uint32_t)ux < (uint32_t)nw && (uint32_t)uy < (uint32_t)nw) {
          if (ux == vx) add |= (uint32_t)UF_VEDGE << (8 * k);
          if (uy == hy) add |= (uint32_t)UF_HEDGE << (8 * k);
        }
      }
      PC_L(s.m_flags) |= add;
    PC_VEC_END
  }
  set_qp_y(s);
  map_fill(s.m_qp, zb, n_units, (uint32_t)(uint8_t)(int8_t)s.cur_qp_y);
  s.last_qp_y = s.cur_qp_y;
}
#endif   // HIPDEC_PARSE_INTER

// ---- 7.3.8.5 coding_unit + 7.3.8.8 transform_tree, stackless over the z-ordered unit index -------
PC_DEV void coding_unit(PS& s, int zb /*unit z-index of the CU inside the CTB*/, int log2cb, int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr)
{
  const int ux0 = (int)compact1by1((uint32_t)zb), uy0 = (int)compact1by1((uint32_t)zb >> 1);
  const int n_units = 1 << (2 * (log2cb - 2));
  s.cu_tq_bypass = 0;
  if (s.tools & TOOL_TQBYPASS) s.cu_tq_bypass = decode_bin(s, s.ctxA, A_CU_TQ_BYPASS);
#if HIPDEC_PARSE_INTER
  if (s.is_p) {   // cu_skip_flag (context: the left / above neighbours' flags, 9.3.4.2.2), pred_mode_flag
    int inc = 0;
    if (ux0 > 0) inc += (int)(map_get(s.m_ipmc, (int)interleave4((uint32_t)ux0 - 1, (uint32_t)uy0)) >> 7);
    else if (s.ctb_avail & AV_LEFT) inc += (int)((pc_rdlane(s.p_left, uy0) >> 23) & 1u);
    if (uy0 > 0) inc += (int)(map_get(s.m_ipmc, (int)interleave4((uint32_t)ux0, (uint32_t)uy0 - 1)) >> 7);
    else if (s.ctb_avail & AV_UP) inc += (int)((pc_rdlane(s.up, 13) >> ux0) & 1u);
    const int cu_skip = decode_bin(s, s.ctxC, C_SKIP_FLAG + inc);
    int inter = 1;
    if (!cu_skip) inter = decode_bin(s, s.ctxC, C_PRED_MODE) ? 0 : 1;
    if (inter) { inter_coding_unit(s, zb, log2cb, cu_skip, coef_y, coef_cb, coef_cr); return; }
  }
#endif
  int part_nxn = 0;
  if (log2cb == s.log2_min_cb) part_nxn = decode_bin(s, s.ctxA, A_PART_MODE) ? 0 : 1;
  if (part_nxn && log2cb == 3 && s.log2_min_tb > 2) { s.err = DEV_ERR_SYNTAX; part_nxn = 0; }
  if ((s.pcm & 255u) && !part_nxn && log2cb >= (int)((s.pcm >> 24) & 15u) && log2cb <= (int)(s.pcm >> 28)) {
    if (decode_terminate(s)) { pcm_coding_unit(s, zb, log2cb, coef_y, coef_cb, coef_cr); return; }   // pcm_flag
  }
  set_qp_y(s);
  // CU-level map fill (contiguous in z-order)
  map_fill(s.m_size, zb, n_units, (uint32_t)(log2cb << 4));
  map_fill(s.m_flags, zb, n_units, (uint32_t)(s.cu_tq_bypass ? UF_BYPASS : 0));
  map_fill(s.m_ipm, zb, n_units, 1u);
  // intra prediction modes 7.3.8.5 / 8.4.2
  const int n_part = part_nxn ? 4 : 1;
  const int pu_units = n_units / n_part;                    // units per PU (contiguous quadrant)
  const int pu_w = 1 << (log2cb - 2 - (part_nxn ? 1 : 0));  // PU width in units
  uint32_t prev_flags = 0;
  for (int k = 0; k < n_part; k++) prev_flags |= (uint32_t)decode_bin(s, s.ctxA, A_PREV_INTRA_LUMA) << k;
  for (int k = 0; k < n_part; k++) {
    int mpm_idx = 0, rem = 0;
    if ((prev_flags >> k) & 1u) { if (decode_bypass(s)) mpm_idx = decode_bypass(s) ? 2 : 1; }
    else rem = decode_bypass_bits(s, 5);
    const int ux = ux0 + (k & 1) * pu_w, uy = uy0 + (k >> 1) * pu_w;
    int cand_a = 1, cand_b = 1;
    if (ux > 0) cand_a = (int)(map_get(s.m_ipm, (int)interleave4((uint32_t)ux - 1, (uint32_t)uy)) & 63u);
    else if (s.ctb_avail & AV_LEFT) cand_a = (int)((pc_rdlane(s.p_left, uy) >> 8) & 63u);
    if (uy > 0) cand_b = (int)(map_get(s.m_ipm, (int)interleave4((uint32_t)ux, (uint32_t)uy - 1)) & 63u);  // above CTB row: INTRA_DC (8.4.2)
    int c0, c1, c2;
    if (cand_a == cand_b) {
      if (cand_a < 2) { c0 = 0; c1 = 1; c2 = 26; }
      else { c0 = cand_a; c1 = 2 + ((cand_a + 29) & 31); c2 = 2 + ((cand_a - 2 + 1) & 31); }
    } else {
      c0 = cand_a; c1 = cand_b;
      if (cand_a != 0 && cand_b != 0) c2 = 0; else if (cand_a != 1 && cand_b != 1) c2 = 1; else c2 = 26;
    }
    int mode;
    if ((prev_flags >> k) & 1u) mode = mpm_idx == 0 ? c0 : (mpm_idx == 1 ? c1 : c2);
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
    map_fill(s.m_ipm, zb + k * pu_units, pu_units, (uint32_t)mode);
  }
  // intra_chroma_pred_mode: one per coding unit, or one per partition of an NxN coding unit when ChromaArrayType is 3 (7.3.8.5)
  const int c444 = pc_is444(s);
  int chroma_mode = 1;
  if (s.chroma_format_idc) {
    const int n_cp = (c444 && part_nxn) ? 4 : 1, cp_units = n_units / n_cp;
    for (int k = 0; k < n_cp; k++) {
      int icpm = 4;
      if (decode_bin(s, s.ctxA, A_INTRA_CHROMA)) icpm = decode_bypass_bits(s, 2);
      const int lm = (int)(map_get(s.m_ipm, zb + k * cp_units) & 63u);
      if (icpm == 4) chroma_mode = lm;
      else { const int m = icpm == 0 ? 0 : icpm == 1 ? 26 : icpm == 2 ? 10 : 1; chroma_mode = (m == lm) ? 34 : m; }
      if (pc_is422(s)) chroma_mode = (int)c_map422[chroma_mode];   // 8.4.3, Table 8-3: the 4:2:2 sampling grid is not square
      map_fill(s.m_ipmc, zb + k * cp_units, cp_units, (uint32_t)chroma_mode);
    }
  } else map_fill(s.m_ipmc, zb, n_units, 1u);

  // ---- transform tree ----
  const int max_trafo_depth = s.max_th_depth_intra + part_nxn;
  uint32_t cbf_cb_bits = 0, cbf_cr_bits = 0;  // bit d = cbf at trafoDepth d along the current path; 4:2:2: bit 8 + d = the flag of the lower chroma block
  const int c422 = pc_is422(s);
  int q = 0;
  while (q < n_units && !s.err) {
    int t;  // log2 size of the node that starts at q
    if (q == 0) t = log2cb; else { t = 2 + ((pc_ffs((uint32_t)q) - 1) >> 1); if (t > log2cb) t = log2cb; }
    for (;;) {
      const int depth = log2cb - t;
      int split;
      if (t <= s.log2_max_tb && t > s.log2_min_tb && depth < max_trafo_depth && !(part_nxn && depth == 0))
        split = decode_bin(s, s.ctxA, A_SPLIT_TRANSFORM + 5 - t);
      else split = (t > s.log2_max_tb || (part_nxn && depth == 0)) ? 1 : 0;
      if (s.chroma_format_idc) {
        const uint32_t bit = 1u << depth, pbit = depth ? (1u << (depth - 1)) : 0;
        if (t > 2 || c444) {
          int cb = 0, cr = 0;
          const int cc = depth == 4 ? A_CBF_CHROMA4 : A_CBF_CHROMA + depth;   // depth 4 only occurs with ChromaArrayType 3
          // ChromaArrayType 2: a second flag, for the lower chroma block, where the chroma is coded (a leaf, or the 8x8 node above four 4x4 leaves)
          const int two = c422 && (!split || t == 3);
          int cb2 = 0, cr2 = 0;
          if (depth == 0 || (cbf_cb_bits & pbit)) { cb = decode_bin(s, s.ctxA, cc); if (two) cb2 = decode_bin(s, s.ctxA, cc); }
          if (depth == 0 || (cbf_cr_bits & pbit)) { cr = decode_bin(s, s.ctxA, cc); if (two) cr2 = decode_bin(s, s.ctxA, cc); }
          cbf_cb_bits = (cbf_cb_bits & ~(bit * 0x101u)) | (cb ? bit : 0) | (cb2 ? bit << 8 : 0);
          cbf_cr_bits = (cbf_cr_bits & ~(bit * 0x101u)) | (cr ? bit : 0) | (cr2 ? bit << 8 : 0);
        } else {  // 4x4 luma: inherits the parent's flags (7.4.9.8)
          cbf_cb_bits = (cbf_cb_bits & ~(bit * 0x101u)) | ((cbf_cb_bits & (pbit * 0x101u)) << 1);
          cbf_cr_bits = (cbf_cr_bits & ~(bit * 0x101u)) | ((cbf_cr_bits & (pbit * 0x101u)) << 1);
        }
      }
      if (!split) break;
      t--;
    }
    // leaf transform unit at unit index zb + q, size 1 << t
    const int depth = log2cb - t;
    const int zu = zb + q;
    const int tu_units = 1 << (2 * (t - 2));
    const int cbf_luma = decode_bin(s, s.ctxA, A_CBF_LUMA + (depth == 0 ? 1 : 0));
    const int cbf_cb = (int)((cbf_cb_bits >> depth) & 1u), cbf_cr = (int)((cbf_cr_bits >> depth) & 1u);
    const int cbf_cb2 = (int)((cbf_cb_bits >> (8 + depth)) & 1u), cbf_cr2 = (int)((cbf_cr_bits >> (8 + depth)) & 1u);   // 4:2:2 only
    if ((cbf_luma | cbf_cb | cbf_cr | cbf_cb2 | cbf_cr2) && (s.tools & TOOL_CUQPD) && !s.is_cu_qp_delta_coded) parse_cu_qp_delta(s);
    const int luma_mode = (int)(map_get(s.m_ipm, zu) & 63u);
    int do_chroma = 0, zc = zu, tc = t - 1;
    if (c444) { do_chroma = 1; tc = t; chroma_mode = (int)map_get(s.m_ipmc, zu); }   // chroma blocks coincide with the luma blocks
    else if (s.chroma_format_idc) {
      if (t > 2) do_chroma = 1;
      else if ((q & 3) == 3) { do_chroma = 1; zc = zb + (q & ~3); tc = 2; }
    }
    // one residual_coding instance for all blocks of the unit (keeps the hot code small): luma, Cb, Cr - 4:2:2: luma, Cb upper, Cb lower, Cr upper,
    // Cr lower (k = 3 / 4 are the lower blocks of Cb / Cr; their coefficients follow the upper block's)
    uint32_t ts_bits = 0;
    const uint32_t coded_bits = (uint32_t)cbf_luma | (do_chroma ? (uint32_t)(cbf_cb << 1) | (uint32_t)(cbf_cr << 2) | (uint32_t)(cbf_cb2 << 3) | (uint32_t)(cbf_cr2 << 4) : 0u);
    const int c_mult = c444 ? 16 : (c422 ? 8 : 4);    // chroma samples per 4x4 luma unit
#pragma clang loop unroll(disable)
    for (int k0 = 0; k0 < (c422 ? 5 : 3); k0++) {
      const int k = c422 ? (k0 == 0 ? 0 : (k0 == 1 ? 1 : (k0 == 2 ? 3 : (k0 == 3 ? 2 : 4)))) : k0;   // coding order: both Cb blocks before the Cr blocks
      if (!((coded_bits >> k) & 1u)) continue;
      const int c = k == 0 ? 0 : 1 + ((k - 1) & 1), low = k >= 3;
      const int lg = c == 0 ? t : tc;
      int16_t* dst = c == 0 ? coef_y + zu * 16 : (c == 1 ? coef_cb : coef_cr) + zc * c_mult + (low << (2 * lg));
#ifdef HIPDEC_PARSE_CYCLES
      const unsigned long long pc_t0 = __builtin_readcyclecounter();
#endif
      ts_bits |= (uint32_t)residual_coding(s, lg, c, c == 0 ? luma_mode : chroma_mode) << k;
#ifdef HIPDEC_PARSE_CYCLES
      const unsigned long long pc_t1 = __builtin_readcyclecounter();
#endif
      flush_coef(s, dst, 1 << (2 * lg));
#ifdef HIPDEC_PARSE_CYCLES
      { const unsigned long long pc_t2 = __builtin_readcyclecounter(); s.c_resid += pc_t1 - pc_t0; s.c_flush += pc_t2 - pc_t1; s.n_resid++; }
#endif
    }
    const int ts_y = (int)(ts_bits & 1u), ts_cb = (int)((ts_bits >> 1) & 1u), ts_cr = (int)((ts_bits >> 2) & 1u);
    // TU-level map fill: size, cbf, transform-skip, deblocking edges (8.7.2.2 / 8.7.2.3)
    fill_tu_maps(s, zu, tu_units,
                 (uint32_t)((cbf_luma ? UF_CBF_LUMA : 0) | ((do_chroma && cbf_cb) ? UF_CBF_CB : 0) | ((do_chroma && cbf_cr) ? UF_CBF_CR : 0) |
                            (s.cu_tq_bypass ? UF_BYPASS : 0) | (ts_y ? UF_TS_LUMA : 0)),
                 (uint32_t)(luma_mode | (ts_cb ? 64 : 0) | (ts_cr ? 128 : 0)), (uint32_t)((log2cb << 4) | t));
    if (c422 && do_chroma) {
      // 4:2:2: the flags of the LOWER chroma blocks live in the unit next to the one that carries the upper blocks' (index ^ 1: the 2nd unit of a
      // block of 8x8 and up, the 3rd of a quad of 4x4 luma blocks): its cbf_cb / cbf_cr bits and the transform-skip bits of its mode byte
      const int z2 = zu ^ 1;
      const uint32_t f2 = (map_get(s.m_flags, z2) & ~(uint32_t)(UF_CBF_CB | UF_CBF_CR)) | (cbf_cb2 ? UF_CBF_CB : 0u) | (cbf_cr2 ? UF_CBF_CR : 0u);
      const uint32_t m2 = (map_get(s.m_ipm, z2) & 63u) | (((ts_bits >> 3) & 1u) ? 64u : 0u) | (((ts_bits >> 4) & 1u) ? 128u : 0u);
      map_fill(s.m_flags, z2, 1, f2);
      map_fill(s.m_ipm, z2, 1, m2);
    }
    q += tu_units;
  }
  set_qp_y(s);
  map_fill(s.m_qp, zb, n_units, (uint32_t)(uint8_t)(int8_t)s.cur_qp_y);
  s.last_qp_y = s.cur_qp_y;
}

// ---- 7.3.8.3 sao: parameters are kept in lanes 0..8 of s.sao as SaoParams dwords -----------------
//   dword 3c+0: type | band_or_class << 8 | offset[0] << 16     3c+1: offset[1] | offset[2] << 16     3c+2: offset[3]
PC_DEV void parse_sao(PS& s, int allow_left, int allow_up)
{
  int merge_left = 0, merge_up = 0;
  if (allow_left) merge_left = decode_bin(s, s.ctxA, A_SAO_MERGE);
  if (allow_up && !merge_left) merge_up = decode_bin(s, s.ctxA, A_SAO_MERGE);
  const int ncomp = s.chroma_format_idc ? 3 : 1;
  if (merge_left) { PC_VEC_BEGIN PC_L(s.sao) = PC_L(s.sao_left); PC_VEC_END return; }
  if (merge_up) { PC_VEC_BEGIN PC_L(s.sao) = lane < 9 ? PC_L(s.up) : 0u; PC_VEC_END return; }
  PC_VEC_BEGIN PC_L(s.sao) = 0u; PC_VEC_END
  int type1 = 0, cls1 = 0;
  for (int c = 0; c < ncomp; c++) {
    const int on = c == 0 ? s.sao_luma : s.sao_chroma;
    if (!on) continue;
    int type;
    if (c == 2) type = type1;
    else { type = 0; if (decode_bin(s, s.ctxA, A_SAO_TYPE)) type = decode_bypass(s) ? 2 : 1; }
    if (c == 1) type1 = type;
    if (!type) continue;
    const int bd = c ? s.bit_depth_chroma : s.bit_depth_luma;
    const int c_max = (1 << ((bd < 10 ? bd : 10) - 5)) - 1;
    int a[4], sg[4] = {0, 0, 1, 1};
    for (int i = 0; i < 4; i++) { int v = 0; while (v < c_max && decode_bypass(s)) v++; a[i] = v; }
    int cls;
    if (type == 1) {
      for (int i = 0; i < 4; i++) sg[i] = a[i] ? decode_bypass(s) : 0;
      cls = decode_bypass_bits(s, 5);
    } else {
      if (c == 2) cls = cls1; else cls = decode_bypass_bits(s, 2);
    }
    if (c == 1) cls1 = cls;
    const int sh = bd - (bd < 10 ? bd : 10);
    uint32_t o[4];
    for (int i = 0; i < 4; i++) o[i] = (uint32_t)(uint16_t)(int16_t)((sg[i] ? -a[i] : a[i]) << sh);
    pc_wrlane(s.sao, 3 * c + 0, (uint32_t)type | ((uint32_t)cls << 8) | (o[0] << 16));
    pc_wrlane(s.sao, 3 * c + 1, o[1] | (o[2] << 16));
    pc_wrlane(s.sao, 3 * c + 2, o[3]);
  }
}

// ---- platform glue for the cross-wave protocol ----------------------------------------------------
// Cross-wave hand-off (WPP): what the row below needs from a CTB — its SAO parameters, the size bytes of
// its bottom unit row and, once per row, the context snapshot — is tiny, so it travels as write-through
// (sc1) dword stores into a per-CTB record, drained with s_waitcnt and followed by ONE relaxed agent-scope
// progress store; the consumer polls that word and reads the record with sc1 loads.  No release / acquire
// fence, i.e. no L2 write-back and no L1 invalidate per CTB (cdna guide, Guideline 16, form R1).  The bulk
// outputs (unit maps, coefficients) are plain stores: only later kernels read them.
#if defined(HIPDEC_HOST_EMU)
// the emulation runs substreams one after the other in order, so a dependency is always satisfied
PC_DEV int pc_wait_progress(const uint32_t* word, uint32_t need, const int32_t*) { return *word >= need ? 0 : DEV_ERR_TIMEOUT; }
PC_DEV void pc_publish(uint32_t* word, uint32_t v) { *word = v; }
PC_DEV void pc_report(int32_t* status, int32_t code) { if (*status == 0) *status = code; }
PC_DEV void pc_store_wt(uint32_t* p, uint32_t v) { *p = v; }
PC_DEV uint32_t pc_load_wt(const uint32_t* p) { return *p; }
PC_DEV void pc_store16_wt(uint32_t* p, const uint32_t* v) { memcpy(p, v, 16); }
PC_DEV void pc_load16_wt(uint32_t* v, const uint32_t* p) { memcpy(v, p, 16); }
#define PC_GATHER(r, idx) ((r).v[(idx) & 63])   /* value of lane idx; only inside PC_VEC_BEGIN .. PC_VEC_END, r not written there */
PC_DEV void pc_drain() {}
// returned (completed-before-continuing) atomics of the pool scheduler; the emulation is single threaded
PC_DEV uint32_t pc_atomic_exch(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = v; return o; }
PC_DEV uint32_t pc_atomic_add(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
PC_DEV uint32_t pc_atomic_cas(uint32_t* p, uint32_t expect, uint32_t v) { const uint32_t o = *p; if (o == expect) *p = v; return o; }
PC_DEV void pc_idle() {}
#define PC_POOL_SPINS 1   /* an empty queue means "nothing runnable now": the single emulated wave returns */
#else
PC_DEV int pc_wait_progress(const uint32_t* word, uint32_t need, const int32_t* status)
{
  // A blocked row waits milliseconds (its predecessor is busy with a whole CTB), so poll rarely: every poll
  // is an L2 round trip plus issue slots taken from the waves that are decoding.  Bounded: ~2^20 polls of
  // ~8k cycles each; a failing substream releases the waiters through the status word.
  int err = 0;
  uint32_t spins = 0;
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    if (spins < 4) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(127);
    if (++spins > (1u << 20) || ((spins & 7u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { err = DEV_ERR_TIMEOUT; break; }
  }
  return err;
}
PC_DEV void pc_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
PC_DEV void pc_publish(uint32_t* word, uint32_t v)
{
  pc_drain();   // this wave is the only writer of the record
  if (threadIdx.x == 0) __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
PC_DEV void pc_report(int32_t* status, int32_t code) { if (threadIdx.x == 0) atomicCAS((int*)status, 0, code); }
PC_DEV void pc_store_wt(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PC_DEV uint32_t pc_load_wt(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes per lane, write-through / L1-bypassing (sc1): a relaxed agent-scope atomic lowers to sc1 only up to 8 bytes, and narrow sc1
// stores are one fabric write per LANE (microarch guide), so the parked record travels as 21 x 16 B instead of 448 x 4 B
typedef uint32_t PcU4 __attribute__((ext_vector_type(4)));
PC_DEV void pc_store16_wt(uint32_t* p, const uint32_t* v)
{
  const PcU4 d = *(const PcU4*)v;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
}
PC_DEV void pc_load16_wt(uint32_t* v, const uint32_t* p)
{
  PcU4 d;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(d) : "v"(p) : "memory");
  *(PcU4*)v = d;
}
#define PC_GATHER(r, idx) ((uint32_t)__shfl((int)(r), (int)(idx)))
// returned atomics by lane 0, result broadcast: the wave continues only after the operation was performed at the
// device-wide coherence point, which gives the store->load ordering the suspend / wake-up handshake relies on
// (the empty asm consumes the broadcast result so that the compiler can neither drop the return path — a no-return
// atomic is fire-and-forget — nor move later memory operations above the wait for it)
PC_DEV uint32_t pc_atomic_done(uint32_t o) { const uint32_t r = pc_uni(o); asm volatile("" :: "s"(r) : "memory"); return r; }
PC_DEV uint32_t pc_atomic_exch(uint32_t* p, uint32_t v)
{
  uint32_t o = 0;
  if (threadIdx.x == 0) o = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return pc_atomic_done(o);
}
PC_DEV uint32_t pc_atomic_add(uint32_t* p, uint32_t v)
{
  uint32_t o = 0;
  if (threadIdx.x == 0) o = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return pc_atomic_done(o);
}
PC_DEV uint32_t pc_atomic_cas(uint32_t* p, uint32_t expect, uint32_t v)
{
  uint32_t o = 0;
  if (threadIdx.x == 0) { o = expect; __hip_atomic_compare_exchange_strong(p, &o, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  return pc_atomic_done(o);
}
PC_DEV void pc_idle() { __builtin_amdgcn_s_sleep(64); }
#define PC_POOL_SPINS (1u << 22)
#endif

PC_DEV uint32_t uload32(const void* p) { return pc_uni(*(const uint32_t*)p); }
PC_DEV uint32_t pc_load_wt_uni(const uint32_t* p) { return pc_uni(pc_load_wt(p)); }
PC_DEV uint64_t uload64(const void* p) { return (uint64_t)uload32(p) | ((uint64_t)uload32((const uint8_t*)p + 4) << 32); }

// ---- pool scheduler: ready queue --------------------------------------------------------------------------
// Rows (substreams) are tasks.  A task that finds its WPP predecessor not far enough ahead saves its parser state
// and leaves; the predecessor's wave re-queues it when its progress reaches the recorded need.  Arbitration of
// "who continues the row" is one CAS on waitneed[row]; all handshake words use returned atomics (see above).
PC_DEV void pool_push(const ParseArgs& A, uint32_t sub)
{
  const uint32_t t = pc_atomic_add(A.qctl + 1, 1u);
  PC_VEC_BEGIN if (lane == 0) pc_store_wt(A.queue + (t & (A.queue_cap - 1u)), sub + 1u); PC_VEC_END
}
// arms `sub` to be woken when its predecessor reaches `need`; returns 1 if the caller may run it right away
PC_DEV int pool_arm(const ParseArgs& A, uint32_t sub, uint32_t dep, uint32_t need)
{
  pc_atomic_exch(A.waitneed + sub, need);
  const uint32_t p = pc_atomic_add(A.progress + dep, 0u);
  if (p >= need) return pc_atomic_cas(A.waitneed + sub, need, 0u) == need;   // lost the race: the producer queued it
  return 0;
}
// the predecessor side: after publishing progress `done` of substream `sub`
PC_DEV void pool_wake_dependent(const ParseArgs& A, int32_t dependent, uint32_t done)
{
  if (dependent < 0) return;
  const uint32_t w = pc_atomic_add(A.waitneed + dependent, 0u);
  if (w != 0 && done >= w && pc_atomic_cas(A.waitneed + dependent, w, 0u) == w) pool_push(A, (uint32_t)dependent);
}

// ---- parked row state ------------------------------------------------------------------------------------------------------------
// What a row needs to continue on another wave: the context variables (one byte each: p' | valMps << 6), the left neighbour column,
// the left CTB's SAO parameters and 11 scalars — 84 dwords.  The image is assembled in LDS and leaves as 21 write-through 16-byte
// stores (round 2 parked six whole registers: 448 dword stores, each a fabric write of its own: profiles/pmc_traffic.json 4.7 B/px).
//   dwords  0..47  ctxA | ctxB | ctxC, byte = lane        48..63  p_left lanes 0..15        64..72  sao_left        73..83  scalars
enum : int { PARK_CTX = 0, PARK_LEFT = 48, PARK_SAO = 64, PARK_SCALARS = 73, PARK_USED = 84 };
static_assert((int)PARK_USED <= (int)SAVE_DWORDS && PARK_USED % 4 == 0, "parked record");
PC_DEV uint32_t ctx_pack(uint32_t v) { return (v & 63u) | ((v >> 16) << 6); }
PC_DEV uint32_t ctx_unpack(uint32_t b) { return (b & 63u) | ((b >> 6) << 16); }
PC_DEV void stage_contexts(PS& s)   // -> bytes 0 .. 191 of the LDS image
{
  uint8_t* pb = (uint8_t*)s.L->park;
#if HIPDEC_PARSE_LDS_CTX
  PC_LDS_SYNC();
  PC_VEC_BEGIN
    for (int g = 0; g < 3; g++) { const uint32_t w = s.L->ctx[g * 64 + lane]; pb[g * 64 + lane] = (uint8_t)(((w >> 2) & 63u) | ((w >> 16) << 6)); }
  PC_VEC_END
#else
  PC_VEC_BEGIN
    pb[lane] = (uint8_t)ctx_pack(PC_L(s.ctxA)); pb[64 + lane] = (uint8_t)ctx_pack(PC_L(s.ctxB)); pb[128 + lane] = (uint8_t)ctx_pack(PC_L(s.ctxC));
  PC_VEC_END
#endif
}
PC_DEV void unstage_contexts(PS& s)
{
  const uint8_t* pb = (const uint8_t*)s.L->park;
#if HIPDEC_PARSE_LDS_CTX
  PC_VEC_BEGIN
    for (int g = 0; g < 3; g++) { const uint32_t b = pb[g * 64 + lane]; s.L->ctx[g * 64 + lane] = ((b & 63u) << 2) | ((b >> 6) << 16); }
  PC_VEC_END
  PC_LDS_SYNC();
#else
  PC_VEC_BEGIN
    PC_L(s.ctxA) = ctx_unpack(pb[lane]); PC_L(s.ctxB) = ctx_unpack(pb[64 + lane]); PC_L(s.ctxC) = ctx_unpack(pb[128 + lane]);
  PC_VEC_END
#endif
}
PC_DEV void park_flush(PS& s, uint32_t* dst, int dwords)   // LDS image -> HBM, write-through; the caller drains
{
  PC_LDS_SYNC();
  PC_VEC_BEGIN if (lane * 4 < dwords) pc_store16_wt(dst + lane * 4, s.L->park + lane * 4); PC_VEC_END
}
PC_DEV void park_fetch(PS& s, const uint32_t* src, int dwords)
{
  PC_LDS_SYNC();
  PC_VEC_BEGIN if (lane * 4 < dwords) pc_load16_wt(s.L->park + lane * 4, src + lane * 4); PC_VEC_END
  PC_LDS_SYNC();
}
// resume_word: where the CTB index to resume at goes (a suspended row), or nullptr (the END state of a finished substream, which the first
// substream of a dependent slice segment continues: contexts, left-neighbour column, SAO parameters, QP state)
PC_DEV void save_row_state(PS& s, uint32_t* resume_word, uint32_t* saved, uint32_t k)
{
  stage_contexts(s);
  uint32_t* pd = s.L->park;
  PC_VEC_BEGIN
    uint32_t v = 0;
    if (lane == 0) v = s.range; else if (lane == 1) v = s.value; else if (lane == 2) v = s.bits_needed;
    else if (lane == 3) v = s.pos; else if (lane == 4) v = s.end; else if (lane == 5) v = (uint32_t)s.zeros;
    else if (lane == 6) v = (uint32_t)s.last_qp_y; else if (lane == 7) v = (uint32_t)s.qpy_pred; else if (lane == 8) v = (uint32_t)s.cur_qp_y;
    else if (lane == 9) v = (uint32_t)s.is_cu_qp_delta_coded; else if (lane == 10) v = (uint32_t)s.cu_qp_delta_val;
    if (lane < 16) pd[PARK_LEFT + lane] = PC_L(s.p_left);
    if (lane < 9) pd[PARK_SAO + lane] = PC_L(s.sao_left);
    if (lane < 11) pd[PARK_SCALARS + lane] = v;
  PC_VEC_END
  park_flush(s, saved, PARK_USED);
  PC_VEC_BEGIN if (lane == 0 && resume_word) pc_store_wt(resume_word, k); PC_VEC_END
  pc_drain();
}
// the inverse; `all` = 0 leaves the arithmetic decoder alone (a dependent slice segment starts its own)
PC_DEV void load_row_state(PS& s, const uint32_t* saved, int all)
{
  park_fetch(s, saved, PARK_USED);
  unstage_contexts(s);
  const uint32_t* pd = s.L->park;
  VReg sc;
  PC_VEC_BEGIN
    PC_L(s.p_left) = lane < 16 ? pd[PARK_LEFT + lane] : 0u;
    PC_L(s.sao_left) = lane < 9 ? pd[PARK_SAO + lane] : 0u;
    PC_L(sc) = lane < 11 ? pd[PARK_SCALARS + lane] : 0u;
  PC_VEC_END
  PC_LDS_SYNC();
  if (all) {
    s.range = pc_vec(pc_rdlane(sc, 0)); s.value = pc_vec(pc_rdlane(sc, 1)); s.bits_needed = pc_vec(pc_rdlane(sc, 2));
    s.pos = pc_rdlane(sc, 3); s.end = pc_rdlane(sc, 4); s.zeros = (int32_t)pc_rdlane(sc, 5); s.win_base = 0xfffff000u; s.fast_limit = 0;
  }
  s.last_qp_y = (int32_t)pc_rdlane(sc, 6); s.qpy_pred = (int32_t)pc_rdlane(sc, 7); s.cur_qp_y = (int32_t)pc_rdlane(sc, 8);
  s.is_cu_qp_delta_coded = (int32_t)pc_rdlane(sc, 9); s.cu_qp_delta_val = (int32_t)pc_rdlane(sc, 10);
}

enum : int { PARSE_DONE = 0, PARSE_SUSPENDED = -1 };   // > 0: device error code

// One CABAC substream (slice segment / tile / WPP row).  Static mode (A.pool == 0): start to finish, waiting for the
// predecessor row in place.  Pool mode: from the row's resume point until it finishes or has to wait.
PC_DEV int parse_substream(const ParseArgs& A, uint32_t sub_idx, int same_wave_dep, uint32_t start_lag, Lds* lds)
{
  PS s;
#ifdef HIPDEC_PARSE_CYCLES
  const unsigned long long pc_row0 = __builtin_readcyclecounter();
#endif
  const Substream* subp = A.subs + sub_idx;
  const uint32_t sub_pic = uload32(&subp->pic), byte_start = uload32(&subp->byte_start), byte_end = uload32(&subp->byte_end);
  const uint32_t first_ctb_ts = uload32(&subp->first_ctb_ts), num_ctbs = uload32(&subp->num_ctbs), slice_idx = uload32(&subp->slice_idx);
  const int32_t dep_sub = (int32_t)uload32(&subp->dep_sub);
  const uint32_t dep_len = uload32(&subp->dep_len);
  const uint32_t sflags = uload32(&subp->wpp_sync);  // wpp_sync | has_dependent << 8 | last_in_slice_segment << 16
  const int wpp_sync = (int)(sflags & 255u), has_dependent = (int)((sflags >> 8) & 255u), last_in_slice_segment = (int)((sflags >> 16) & 255u);
  const int32_t dependent = (int32_t)uload32(&subp->dependent);
  const uint32_t pool = A.pool;
  const PicParams* P = A.pics + sub_pic;

  s.L = lds; s.err = 0;
  s.width = (int32_t)uload32(&P->width); s.height = (int32_t)uload32(&P->height);
  s.log2_ctb = (int32_t)uload32(&P->log2_ctb); s.log2_min_cb = (int32_t)uload32(&P->log2_min_cb);
  s.log2_min_tb = (int32_t)uload32(&P->log2_min_tb); s.log2_max_tb = (int32_t)uload32(&P->log2_max_tb);
  s.max_th_depth_intra = (int32_t)uload32(&P->max_th_depth_intra);
  s.chroma_format_idc = (int32_t)uload32(&P->chroma_format_idc);
  if (!HIPDEC_PARSE_CHROMA_GENERAL && s.chroma_format_idc >= 2) s.err = DEV_ERR_SYNTAX;   // (the host launches the general build for such batches)
  s.bit_depth_luma = (int32_t)uload32(&P->bit_depth_luma); s.bit_depth_chroma = (int32_t)uload32(&P->bit_depth_chroma);
  s.log2_min_cu_qp_delta_size = (int32_t)uload32(&P->log2_min_cu_qp_delta_size);
  const int ctb_w = (int32_t)uload32(&P->ctb_w);
  {
    const uint32_t t0 = uload32(&P->sao_enabled);        // sao_enabled, sign_data_hiding, transform_skip_enabled, cu_qp_delta_enabled
    const uint32_t t1 = uload32(&P->transquant_bypass_enabled);
    s.tools = (((t0 >> 8) & 255u) ? TOOL_SDH : 0u) | (((t0 >> 16) & 255u) ? TOOL_TS : 0u) | (((t0 >> 24) & 255u) ? TOOL_CUQPD : 0u) |
              ((t1 & 255u) ? TOOL_TQBYPASS : 0u);
    s.pcm = uload32(&P->pcm_enabled);
  }
  uint8_t* const arena = A.arena;
  s.bs = arena + uload64(&P->off_bitstream);
  {
    const SliceParams* sl = (const SliceParams*)(arena + uload64(&P->off_slices)) + slice_idx;
    s.slice_qp_y = (int32_t)uload32(&sl->slice_qp_y);
    const uint32_t w1 = uload32(&sl->cb_qp_offset);        // cb, cr, pps_cb, pps_cr
    const uint32_t w2 = uload32(&sl->beta_offset_div2);    // beta, tc, deblocking_disabled, sao_luma
    const uint32_t w3 = uload32(&sl->sao_chroma);          // sao_chroma, lf_across_slices, slice_addr_rs
    (void)w1;
    s.deblock = ((w2 >> 16) & 255u) ? 0 : 1;
    s.sao_luma = (int)((w2 >> 24) & 255u);
    s.sao_chroma = (int)(w3 & 255u);
#if HIPDEC_PARSE_INTER
    const uint32_t w4 = uload32(&sl->is_p);                // is_p, num_ref_idx, max_merge_cand, init_type
    s.is_p = (int)(w4 & 255u); s.num_ref_idx = (int)((w4 >> 8) & 255u); s.max_merge_cand = (int)((w4 >> 16) & 255u); s.init_type = (int)(w4 >> 24);
    const uint32_t w6 = uload32(&sl->is_b);                // is_b, num_ref_idx_l1, mvd_l1_zero, tmvp
    s.is_b = (int)(w6 & 255u); s.num_ref_idx_l1 = (int)((w6 >> 8) & 255u); s.mvd_l1_zero = (int)((w6 >> 16) & 255u);
    const uint32_t w5 = uload32(&P->is_inter);             // is_inter, amp_enabled, max_th_depth_inter, log2_par_mrg_level
    s.amp = (int)((w5 >> 8) & 255u); s.max_th_depth_inter = (int)((w5 >> 16) & 255u);
    s.msyn = nullptr;
#else
    if (uload32(&sl->is_p) & 255u) s.err = DEV_ERR_SYNTAX;   // (the host launches the inter build for batches with P slices)
#endif
  }
  s.is_cu_qp_delta_coded = 0; s.cu_qp_delta_val = 0; s.qpy_pred = s.slice_qp_y; s.last_qp_y = s.slice_qp_y; s.cur_qp_y = s.slice_qp_y;
  s.cu_tq_bypass = 0;

  // Per-picture base pointers are NOT kept live across the CTB body (the CABAC state machine needs every register it can get:
  // held here, they were spilled to scratch and reloaded once per CTB anyway); each use site below loads its offset from
  // PicParams again — a handful of cache-resident dword loads per CTB.
  const int units_log2 = 2 * (s.log2_ctb - 2);
  const int units = 1 << units_log2;
  const int uw = 1 << (s.log2_ctb - 2);  // units per CTB side
  const int ctb_size = 1 << s.log2_ctb;
  const int n_mincb_log2 = 2 * (s.log2_ctb - s.log2_min_cb);

  load_tables(s);
  PC_VEC_BEGIN
    PC_L(s.m_size) = 0; PC_L(s.m_flags) = 0; PC_L(s.m_ipm) = 0; PC_L(s.m_ipmc) = 0; PC_L(s.m_qp) = 0;
    PC_L(s.p_left) = 0; PC_L(s.up) = 0; PC_L(s.sao) = 0; PC_L(s.sao_left) = 0;
#if HIPDEC_PARSE_LDS_CTX
    for (int g = 0; g < 3; g++) s.L->ctx[g * 64 + lane] = 0;
    if (lane < 16) s.L->vctx[lane] = (uint32_t)__builtin_offsetof(Lds, ctx);
#else
    PC_L(s.ctxA) = 0; PC_L(s.ctxB) = 0; PC_L(s.ctxC) = 0;
#endif
    PC_L(s.win) = 0; PC_L(s.win_next) = 0;
  PC_VEC_END
  uint32_t k0 = 0;
  uint32_t* saved = A.saved + (size_t)sub_idx * SAVE_DWORDS;
  if (pool) k0 = pc_load_wt_uni(A.resume_k + sub_idx);
  if (k0 == 0) cabac_start(s, byte_start, byte_end);
  else load_row_state(s, saved, 1);   // resume a suspended row

  for (uint32_t k = k0; k < num_ctbs && !s.err; k++) {
    if (pool && A.yield_ctbs && k > k0 && (k - k0) % A.yield_ctbs == 0) {   // test knob: forced yield every N CTBs
      save_row_state(s, A.resume_k + sub_idx, saved, k);
      pool_push(A, sub_idx);
      return PARSE_SUSPENDED;
    }
    const uint8_t* const ts_to_rs = arena + uload64(&P->off_ctb_ts_to_rs);   // uint16_t per CTB
    const int ctb_rs = (int)(uload32(ts_to_rs + ((first_ctb_ts + k) & ~1u) * 2u) >> (((first_ctb_ts + k) & 1u) * 16u)) & 0xffff;
    const int cx = ctb_rs % ctb_w, cy = ctb_rs / ctb_w;
    const uint32_t ci = uload32((const CtbInfo*)(arena + uload64(&P->off_ctb_info)) + ctb_rs);  // slice_idx | avail << 16 | tile_id << 24
    s.x_ctb = cx << s.log2_ctb; s.y_ctb = cy << s.log2_ctb; s.ctb_avail = (int)((ci >> 16) & 255u);

    // ---- WPP dependency on the CTB row above ----
    if (dep_sub >= 0 && !same_wave_dep) {   // (a predecessor decoded earlier by this very wave is complete)
      // What CTB k needs from the row above is the hand-off record of the CTB directly above it (SAO parameters for sao_merge_up,
      // the CB sizes for split_cu_flag's context): k + 1 finished CTBs.  Only the row's first CTB needs two (9.3.1: the context
      // tables are those stored after the second CTB above).  The top-right CTB is an intra-PREDICTION dependency — that is the
      // reconstruction kernel's wavefront, not the parser's.  (A larger start distance decouples the rows in static mode.)
      uint32_t need = k == 0 ? start_lag : k + 1;
      if (need > dep_len || wpp_sync == 2) need = dep_len;   // (a dependent slice segment continues the END of its predecessor: all of it)
      if (!pool) {
#ifdef HIPDEC_PARSE_CYCLES
        const unsigned long long pc_w0 = __builtin_readcyclecounter();
#endif
        const int e = pc_wait_progress(A.progress + dep_sub, need, A.status);
#ifdef HIPDEC_PARSE_CYCLES
        s.c_wait += __builtin_readcyclecounter() - pc_w0;
#endif
        if (e) { s.err = e; break; }
      } else if (pc_load_wt_uni(A.progress + dep_sub) < need) {
        // suspend: save the row's state, record what it waits for (with two CTBs of hysteresis), re-check
        if (k > 0) save_row_state(s, A.resume_k + sub_idx, saved, k);
        uint32_t wake = need + A.wake_hyst;
        if (wake > dep_len) wake = dep_len;
        if (!pool_arm(A, sub_idx, (uint32_t)dep_sub, wake)) return PARSE_SUSPENDED;
      }
    }
    // ---- context initialisation / synchronisation (9.3.1) ----
    if (k == 0) {
      if (wpp_sync == 2 && dep_sub >= 0) {
        // first CTB of a dependent slice segment (9.3.1, 9.3.2.4): everything but the arithmetic decoder continues where the preceding slice
        // segment ended — context variables, the CTB to the left (same slice: available), its SAO parameters, qPY_PREV (8.6.1)
        load_row_state(s, A.saved + (size_t)dep_sub * SAVE_DWORDS, 0);
      } else if (wpp_sync && dep_sub >= 0) {
        park_fetch(s, (const uint32_t*)(A.ctx_store + (size_t)dep_sub * CTX_STORE), CTX_STORE / 4);
        unstage_contexts(s);
        PC_LDS_SYNC();
      } else init_contexts(s);
    }
    // ---- hand-off record of the CTB above ----
    if (s.ctb_avail & AV_UP) {
      const uint32_t* src = (const uint32_t*)(arena + uload64(&P->off_handoff)) + (size_t)(ctb_rs - ctb_w) * HANDOFF_DWORDS;   // HANDOFF_DWORDS per CTB (raster)
      PC_VEC_BEGIN
        PC_L(s.up) = lane < (HIPDEC_PARSE_INTER ? 14 : 13) ? pc_load_wt(src + lane) : 0u;
      PC_VEC_END
    }
    // ---- coding_tree_unit ----
    if (s.sao_luma || s.sao_chroma) {
      parse_sao(s, (s.ctb_avail & AV_LEFT) && (k > 0 || wpp_sync == 2), (s.ctb_avail & AV_UP) ? 1 : 0);
    } else {
      PC_VEC_BEGIN PC_L(s.sao) = 0u; PC_VEC_END
    }
    if (!(s.tools & TOOL_CUQPD)) { s.is_cu_qp_delta_coded = 0; s.cu_qp_delta_val = 0; s.qpy_pred = s.last_qp_y; }

    int16_t* coef_y = (int16_t*)(arena + uload64(&P->off_coeff[0])) + (size_t)ctb_rs * ctb_size * ctb_size;
    const int cc_shift = pc_is444(s) ? 0 : (pc_is422(s) ? 1 : 2);   // chroma samples per CTB = luma samples >> cc_shift
    int16_t* coef_cb = (int16_t*)(arena + uload64(&P->off_coeff[1])) + (size_t)ctb_rs * ((ctb_size * ctb_size) >> cc_shift);
    int16_t* coef_cr = (int16_t*)(arena + uload64(&P->off_coeff[2])) + (size_t)ctb_rs * ((ctb_size * ctb_size) >> cc_shift);

#if HIPDEC_PARSE_INTER
    s.msyn = (MotionSyntax*)(arena + uload64(&P->off_msyn)) + ((size_t)ctb_rs << units_log2);
#endif
    // coding quadtree, stackless over the z-ordered min-CB index
    const int n_mincb = 1 << n_mincb_log2;
    const int mincb_units_log2 = 2 * (s.log2_min_cb - 2);
    int p = 0;
    while (p < n_mincb && !s.err) {
      int lg;  // log2 size of the node starting at p
      if (p == 0) lg = s.log2_ctb; else { lg = s.log2_min_cb + ((pc_ffs((uint32_t)p) - 1) >> 1); if (lg > s.log2_ctb) lg = s.log2_ctb; }
      const int zb = p << mincb_units_log2;
      const int ux = (int)compact1by1((uint32_t)zb), uy = (int)compact1by1((uint32_t)zb >> 1);
      const int x0 = s.x_ctb + (ux << 2), y0 = s.y_ctb + (uy << 2);
      if (x0 >= s.width || y0 >= s.height) { p += 1 << (2 * (lg - s.log2_min_cb)); continue; }
      for (;;) {
        const int size = 1 << lg;
        int split;
        if (x0 + size <= s.width && y0 + size <= s.height && lg > s.log2_min_cb) {
          const int depth = s.log2_ctb - lg;
          int inc = 0;
          const int l = left_cb_log2(s, ux, uy), u = up_cb_log2(s, ux, uy);
          if (l && s.log2_ctb - l > depth) inc++;
          if (u && s.log2_ctb - u > depth) inc++;
          split = decode_bin(s, s.ctxA, A_SPLIT_CU + inc);
        } else split = lg > s.log2_min_cb;
        if ((s.tools & TOOL_CUQPD) && lg >= s.log2_min_cu_qp_delta_size) {
          s.is_cu_qp_delta_coded = 0; s.cu_qp_delta_val = 0;
          derive_qp_pred(s, ux, uy);
        }
        if (!split) break;
        lg--;
      }
      if (!(s.tools & TOOL_CUQPD)) s.qpy_pred = s.last_qp_y;
#ifdef HIPDEC_PARSE_CYCLES
      const unsigned long long pc_c0 = __builtin_readcyclecounter();
#endif
      coding_unit(s, zb, lg, coef_y, coef_cb, coef_cr);
#ifdef HIPDEC_PARSE_CYCLES
      s.c_cu += __builtin_readcyclecounter() - pc_c0;
#endif
      p += 1 << (2 * (lg - s.log2_min_cb));
    }

    // end_of_slice_segment_flag / end_of_subset_one_bit
    const int last = (k + 1 == num_ctbs);
    const int eos = decode_terminate(s);
    if (last) {
      if (last_in_slice_segment) { if (!eos) s.err = DEV_ERR_TERMINATE; }
      else { if (eos || !decode_terminate(s)) s.err = DEV_ERR_TERMINATE; }
    } else if (eos) s.err = DEV_ERR_TERMINATE;

    // ---- publish the CTB: unit maps, SAO parameters, WPP context table ----
    {
      const size_t base = (size_t)ctb_rs << units_log2;
      const int nl = units >> 2;
      uint32_t* sao_dst = (uint32_t*)(arena + uload64(&P->off_sao)) + (size_t)ctb_rs * 9;
      uint8_t* const g_size = arena + uload64(&P->off_u_size);
      uint8_t* const g_flags = arena + uload64(&P->off_u_flags);
      uint8_t* const g_ipm = arena + uload64(&P->off_u_ipm);
      uint8_t* const g_ipmc = arena + uload64(&P->off_u_ipmc);
      uint8_t* const g_qp = arena + uload64(&P->off_u_qp);
      PC_VEC_BEGIN
        if (lane < nl) {
          ((uint32_t*)(g_size + base))[lane] = PC_L(s.m_size);
          ((uint32_t*)(g_flags + base))[lane] = PC_L(s.m_flags);
          ((uint32_t*)(g_ipm + base))[lane] = PC_L(s.m_ipm);
          ((uint32_t*)(g_ipmc + base))[lane] = PC_L(s.m_ipmc);
          ((uint32_t*)(g_qp + base))[lane] = PC_L(s.m_qp);
        }
        if (lane < 9) sao_dst[lane] = PC_L(s.sao);
        // this CTB becomes the left neighbour of the next one
        PC_L(s.sao_left) = PC_L(s.sao);
      PC_VEC_END
      {   // the rightmost unit column, one unit row per lane (what split_cu_flag's context and the MPM candidate A look at)
        VReg col;
        PC_VEC_BEGIN
          const uint32_t z = interleave4((uint32_t)uw - 1u, (uint32_t)lane & 15u);
          const uint32_t sz = (PC_GATHER(s.m_size, z >> 2) >> ((z & 3u) * 8u)) & 255u, im = (PC_GATHER(s.m_ipm, z >> 2) >> ((z & 3u) * 8u)) & 255u;
#if HIPDEC_PARSE_INTER
          const uint32_t ic = (PC_GATHER(s.m_ipmc, z >> 2) >> ((z & 3u) * 8u)) & 255u;   // bit 7: the unit is skipped (cu_skip_flag's context)
          PC_L(col) = (lane < uw) ? (sz | (im << 8) | (ic << 16)) : 0u;
#else
          PC_L(col) = (lane < uw) ? (sz | (im << 8)) : 0u;
#endif
        PC_VEC_END
        PC_VEC_BEGIN PC_L(s.p_left) = PC_L(col); PC_VEC_END
      }
      // hand-off record for the CTB below: lanes 0..8 SAO, 9..12 bottom-row size bytes
      {
        VReg rec;
        PC_VEC_BEGIN PC_L(rec) = PC_L(s.sao); PC_VEC_END
        for (int j = 0; j < (uw + 3) / 4; j++) {
          uint32_t w = 0;
          for (int b = 0; b < 4 && 4 * j + b < uw; b++) w |= map_get(s.m_size, (int)interleave4((uint32_t)(4 * j + b), (uint32_t)uw - 1)) << (8 * b);
          pc_wrlane(rec, 9 + j, w);
        }
#if HIPDEC_PARSE_INTER
        {   // lane 13: cu_skip_flag of the bottom unit row, one bit per unit column
          uint32_t sk = 0;
          for (int j = 0; j < uw; j++) sk |= (map_get(s.m_ipmc, (int)interleave4((uint32_t)j, (uint32_t)uw - 1)) >> 7) << j;
          pc_wrlane(rec, 13, sk);
        }
#endif
        uint32_t* dst = (uint32_t*)(arena + uload64(&P->off_handoff)) + (size_t)ctb_rs * HANDOFF_DWORDS;
        PC_VEC_BEGIN
          if (lane < (HIPDEC_PARSE_INTER ? 14 : 13)) pc_store_wt(dst + lane, PC_L(rec));
        PC_VEC_END
      }
      if (has_dependent == 2 && k + 1 == num_ctbs) save_row_state(s, nullptr, saved, num_ctbs);   // (the registers already hold this CTB as "the previous one")
      if (has_dependent == 1 && k == 1) {
        stage_contexts(s);
        park_flush(s, (uint32_t*)(A.ctx_store + (size_t)sub_idx * CTX_STORE), CTX_STORE / 4);
      }
    }
    if (!pool) { if (has_dependent) pc_publish(A.progress + sub_idx, k + 1); else pc_drain(); }
    else {
      pc_drain();
      if (has_dependent) { pc_atomic_exch(A.progress + sub_idx, k + 1); pool_wake_dependent(A, dependent, k + 1); }
    }
  }
#if defined(HIPDEC_PARSE_CYCLES) && !defined(HIPDEC_HOST_EMU)
  if (threadIdx.x == 0 && sub_pic == 0)
    printf("cyc sub %u ctbs %u total %llu wait %llu cu %llu resid %llu flush %llu nresid %u\n", sub_idx, num_ctbs, __builtin_readcyclecounter() - pc_row0, s.c_wait, s.c_cu, s.c_resid,
           s.c_flush, s.n_resid);
#endif
  if (s.err) pc_report(A.status, s.err | (int32_t)(sub_idx << 8));
  return s.err;
}

// Driver of one parser wave, both modes, with ONE call site of the (fully inlined) substream parser.
//   static mode: substreams first, first + stride, ... < end of the wave's table entry, each start to finish;
//   pool mode  : start-up share of the task list, then ready rows from the queue until every row is finished.
PC_DEV void parse_wave(const ParseArgs& A, uint32_t wave_idx, Lds* lds)
{
  const uint32_t pool = A.pool;
  uint32_t sub = 0, stride = 1, end = 0, lag = 2;
  if (!pool) {
    sub = uload32(&A.waves[wave_idx].first); stride = uload32(&A.waves[wave_idx].stride);
    end = uload32(&A.waves[wave_idx].end); lag = uload32(&A.waves[wave_idx].start_lag);
  } else {
    // start-up: rows without a predecessor are ready; the others are armed to be woken at distance 2.  Only the first
    // waves to RUN (low tickets) share this work: a wave that is not resident yet must not own a share, or the resident
    // ones would starve waiting for rows nobody queued.
    const uint32_t n_init = A.num_waves < 64u ? A.num_waves : 64u;
    for (uint32_t s0 = wave_idx; wave_idx < n_init && s0 < A.num_subs; s0 += n_init) {
      const int32_t dep = (int32_t)uload32(&A.subs[s0].dep_sub);
      if (dep < 0) pool_push(A, s0);
      else {
        uint32_t need = 2u;
        const uint32_t dep_len = uload32(&A.subs[s0].dep_len);
        if (need > dep_len || (uload32(&A.subs[s0].wpp_sync) & 255u) == 2u) need = dep_len;
        if (pool_arm(A, s0, (uint32_t)dep, need)) pool_push(A, s0);
      }
    }
  }
#if defined(HIPDEC_POOL_TRACE) && !defined(HIPDEC_HOST_EMU)
  unsigned long long tr_start = 0, tr_last = 0, tr_idle = 0, tr_tasks = 0, tr_busy = 0, tr_now0 = wall_clock64();
#endif
  for (;;) {
    uint32_t* slot = A.queue;
#if defined(HIPDEC_POOL_TRACE) && !defined(HIPDEC_HOST_EMU)
    tr_now0 = wall_clock64();
#endif
    if (!pool) { if (sub >= end) return; }
    else {
#if defined(HIPDEC_POOL_TRACE) && !defined(HIPDEC_HOST_EMU)
      const unsigned long long tr_now = wall_clock64();   // 100 MHz
      if (tr_tasks == 0 && tr_idle == 0) tr_start = tr_now;
      if (threadIdx.x == 0 && A.trace) {   // (kept current: the wave returns from several places)
        unsigned long long* tr = A.trace + (size_t)wave_idx * 32;
        tr[0] = tr_start; tr[1] = tr_last; tr[2] = tr_now; tr[3] = tr_idle; tr[4] = tr_tasks; tr[5] = tr_busy;
      }
#endif
      if (pc_load_wt_uni(A.qctl + 2) >= A.num_subs) return;                      // every row finished
      if (pc_load_wt_uni((const uint32_t*)A.status) != 0) return;               // a row failed: stop the batch
      const uint32_t h = pc_atomic_add(A.qctl + 0, 1u);
      slot = A.queue + (h & (A.queue_cap - 1u));
      uint32_t v = 0, spins = 0;
      for (;;) {
        v = pc_load_wt_uni(slot);
        if (v != 0) break;
        if (++spins >= PC_POOL_SPINS) break;
        if ((spins & 15u) == 0 && (pc_load_wt_uni(A.qctl + 2) >= A.num_subs || pc_load_wt_uni((const uint32_t*)A.status) != 0)) break;
        pc_idle();
      }
      if (v == 0) {
#if defined(HIPDEC_HOST_EMU)
        pc_atomic_add(A.qctl + 0, (uint32_t)-1);   // single emulated wave: give the ticket back, nothing is runnable now
#endif
        if (pc_load_wt_uni(A.qctl + 2) >= A.num_subs || pc_load_wt_uni((const uint32_t*)A.status) != 0) return;
        pc_report(A.status, DEV_ERR_TIMEOUT | (int32_t)0x20000000);   // queue empty (emulation) / starved for seconds (device): loud
        return;
      }
      PC_VEC_BEGIN if (lane == 0) pc_store_wt(slot, 0u); PC_VEC_END
      sub = v - 1u;
    }
#if defined(HIPDEC_POOL_TRACE) && !defined(HIPDEC_HOST_EMU)
    const unsigned long long tr_t0 = wall_clock64();
    if (pool) {
      tr_idle += tr_t0 - tr_now0;
      if (tr_start == 0) tr_start = tr_now0;
      const unsigned long long bk = (tr_t0 - tr_start) >> 22;   // waiting for work, by 42 ms bucket of the wave's life (booked where the wait ended)
      if (threadIdx.x == 0 && A.trace && bk < 24) A.trace[(size_t)wave_idx * 32 + 8 + bk] += tr_t0 - tr_now0;
    }
#endif
    const int r = parse_substream(A, sub, !pool && stride == 1, lag, lds);
#if defined(HIPDEC_POOL_TRACE) && !defined(HIPDEC_HOST_EMU)
    tr_last = wall_clock64(); tr_busy += tr_last - tr_t0; tr_tasks++;
#endif
    if (!pool) { if (r) return; sub += stride; }
    else if (r == PARSE_DONE) pc_atomic_add(A.qctl + 2, 1u);
    else if (r > 0) return;
  }
}

}  // namespace pcore
}  // namespace hipdec
