// issue_model.hip — what does one gfx950 CU issue per cycle?  (dev tool; round 6)
//
// The CABAC parser (k_parse) is an instruction-issue problem: 19 SALU + 24 VALU + 6.5 branch wave-instructions per pixel, nothing waiting for
// memory.  Its "roofline" is the CU's issue model, which the two guides do not state for integer VALU / v_readlane / SALU / branch mixes.  This tool
// measures it: straight-line bodies of ONE instruction kind (or a fixed mix), independent or dependent, at 1 / 2 / 4 / 8 waves per SIMD on every CU.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_model.hip -o /tmp/issue_model && /tmp/issue_model
//
// Output per (mode, waves per SIMD): cycles per instruction seen by ONE wave (s_memtime around the loop, wave 0 of block 0) and aggregate
// instructions per cycle per CU (total instructions / wall time / 256 CUs / measured clock).  The clock is measured by the same kernel
// (s_memtime ticks per s_memrealtime tick of 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)

enum Mode : int {
  M_VALU_INDEP = 0,     // 32 x v_add_u32 over 8 registers
  M_VALU_DEP,           // 32 x v_add_u32, one chain
  M_SALU_INDEP,         // 32 x s_add_u32 over 8 registers
  M_SALU_DEP,           // 32 x s_add_u32, one chain
  M_MIX_INDEP,          // 16 x (v_add_u32, s_add_u32), independent
  M_READLANE_INDEP,     // 32 x v_readlane_b32 to 8 SGPRs, constant lane
  M_READLANE_SALU_RT,   // v_readlane -> s_and -> v_readlane ... round trip (the context fetch chain), 16 pairs
  M_VALU_TO_SGPR_RT,    // v_cmp -> s_and_b64 (vcc) -> v_cndmask ... : VALU -> SGPR -> VALU chain, 16 triples
  M_BRANCH_TAKEN,       // 16 x s_branch to the next label
  M_CBRANCH_NOT_TAKEN,  // 16 x (s_cmp ; s_cbranch_scc1 never taken)
  M_VCC_BRANCH_TAKEN,   // 16 x (v_cmp ; s_cbranch_vccnz taken)
  M_VCC_BRANCH_NOT,     // 16 x (v_cmp ; s_cbranch_vccz not taken)
  M_MPS_BIN,            // 8 x the parser's context-coded MPS bin without renormalisation (the statement of decode_bin, state never changing)
  M_MIX_2V1S,           // 32 x (2 VALU + 1 SALU) independent
  M_MIX_1V2S,           // 32 x (1 VALU + 2 SALU) independent
  M_VALU_MOV_SGPR,      // 32 x v_mov_b32 v, s (reads SGPR)
  M_CNDMASK_SGPRMASK,   // 32 x v_cndmask_b32_e64 with an SGPR-pair mask
  M_DPP_MOV,            // 32 x v_mov_b32_dpp row_shr:1 (chain)
  M_BPERMUTE,           // 8 x ds_bpermute_b32 + wait (chain)
  M_SNOP,               // 32 x s_nop 0
  M_V1C,                // 32 x v_add_u32 v0, v0, 1 (one chain, inline constant)
  M_V2C,                // 2 chains, inline constant
  M_V8C,                // 8 chains, inline constant
  M_V2VG,               // 2 chains, VGPR operand
  M_V2SG,               // 2 chains, SGPR operand
  M_V3S1,               // 24 x (V V V S)
  M_V4S1,               // 16 x (V V V V S) ... 80 instructions
  M_XOR2,               // v_xor_b32 two chains VGPR operands
  M_RL_ADD,             // 16 x (v_readlane const lane ; v_add_u32 inline)
  M_CMP_CND,            // 16 x (v_cmp_lt_u32 vcc ; v_cndmask vcc)
  M_CMP64_S,            // 32 x v_cmp_eq_u32_e64 sgpr-pair, sgpr, vgpr
  M_MAD24_S,            // 32 x v_mad_i32_i24 v, v, s, v (2 chains)
  M_BFE_S,              // 32 x v_bfe_u32 v, s, v, 8 (2 chains)
  M_PKSUB_S,            // 32 x v_pk_sub_u16 v, s, 1 clamp
  M_V1S1_SG,            // 16 x (v_add_u32 v,v,s ; s_add_u32)
  M_V1B1,               // 16 x (v_add_u32 inline ; s_cbranch_scc1 not taken)
  M_S1B1,               // 16 x (s_add_u32 ; s_cbranch_scc1 not taken): SALU + branch
  M_V1S1B1,             // 10 x (V S B) + 2
  M_WRLANE,             // 32 x v_writelane_b32 v, s, const
  M_LDS_BIN,            // 8 x the LDS-context form of the MPS bin (state + LPS row from LDS, everything else pure VALU, 2 v_cmp + branches)
  M_DSR_SAME,           // 32 x ds_read_b32, all lanes one address, one wait at the end
  M_DSW_SAME,           // 32 x ds_write_b32, all lanes one address
  M_DSW_LANE0,          // 32 x ds_write_b32, lane 0 in range, the others out of range (dropped)
  M_DSR_CHAIN,          // 16 x (ds_read_b32 same address -> wait -> v_and) dependent chain
  M_SNOP15,             // 32 x s_nop 15
  M_COUNT
};
static const char* kNames[M_COUNT] = {
  "VALU v_add_u32 independent", "VALU v_add_u32 dependent chain", "SALU s_add_u32 independent", "SALU s_add_u32 dependent chain",
  "mix VALU,SALU 1:1 independent", "v_readlane_b32 independent", "v_readlane -> s_and -> v_readlane round trip (per pair)",
  "v_cmp -> s_and_b64 -> v_cndmask chain (per triple)", "s_branch taken (to next label)", "s_cmp + s_cbranch not taken (per pair)",
  "v_cmp + s_cbranch_vccnz taken (per pair)", "v_cmp + s_cbranch_vccz not taken (per pair)", "context-coded MPS bin, decode_bin statement (per bin)",
  "mix 2 VALU + 1 SALU independent (per instr)", "mix 1 VALU + 2 SALU independent (per instr)", "v_mov_b32 v, s", "v_cndmask_b32 SGPR mask",
  "v_mov_b32 dpp row_shr:1 chain", "ds_bpermute_b32 + lgkmcnt(0) chain", "s_nop 0",
  "v_add_u32 inline const, 1 chain", "v_add_u32 inline const, 2 chains", "v_add_u32 inline const, 8 chains", "v_add_u32 VGPR operand, 2 chains",
  "v_add_u32 SGPR operand, 2 chains", "mix V V V S", "mix V V V V S", "v_xor_b32 VGPR operands, 2 chains", "v_readlane const ; v_add inline (per instr)",
  "v_cmp vcc ; v_cndmask vcc (per instr)", "v_cmp_eq_u32_e64 sgprpair, s, v", "v_mad_i32_i24 v, v, s, v, 2 chains", "v_bfe_u32 v, s, v, 8",
  "v_pk_sub_u16 v, s, 1 clamp", "mix (v_add v,v,s ; s_add)", "mix (v_add inline ; s_cbranch not taken)", "mix (s_add ; s_cbranch not taken)",
  "mix (V S B)", "v_writelane_b32 v, s, const",
  "LDS-context MPS bin (per bin)", "ds_read_b32 one address x32, one wait", "ds_write_b32 one address x32", "ds_write_b32 lane 0 only (others out of range)",
  "ds_read_b32 -> wait -> v_and chain (per pair)", "s_nop 15",
};
static const int kUnits[M_COUNT] = {32, 32, 32, 32, 32, 32, 16, 16, 16, 16, 16, 16, 8, 96, 96, 32, 32, 32, 8, 32,
                                    32, 32, 32, 32, 32, 96, 80, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 30, 32, 8, 32, 32, 32, 16, 32};         // reported units per body
static const int kInstrs[M_COUNT] = {32, 32, 32, 32, 32, 32, 32, 48, 16, 32, 32, 32, 8 * 17, 96, 96, 32, 32, 64, 16, 32,
                                     32, 32, 32, 32, 32, 96, 80, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 30, 32, 8 * 22, 33, 33, 33, 48, 32};  // instructions per body

template <int MODE>
__global__ __launch_bounds__(256) void k_issue(int iters, uint64_t* out, uint32_t* sink)
{
  uint32_t v0 = threadIdx.x, v1 = 1, v2 = (MODE == M_MPS_BIN || MODE == M_LDS_BIN) ? 0xff00u : 2u, v3 = 3, v4 = (MODE == M_MPS_BIN) ? 0u : 4u, v5 = 5, v6 = 6, v7 = 7;
  uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane(iters), s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
  asm volatile("" : "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7));
  asm volatile("" : "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
  __shared__ uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = 0;
  __syncthreads();
  const uint32_t lbase = (uint32_t)(threadIdx.x >> 6) * 1024u;   // each wave its own KB: [0,256) contexts, [256,512) LPS rows (all zero), [512, 576) ctx addresses
  uint64_t rt0, rt1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0));
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if constexpr (MODE == M_VALU_INDEP) {
      asm volatile(REP4("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t")
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s1));
    } else if constexpr (MODE == M_VALU_DEP) {
      asm volatile(REP32("v_add_u32 %0, %0, %1\n\t") : "+v"(v0) : "s"(s1));
    } else if constexpr (MODE == M_SALU_INDEP) {
      asm volatile(REP4("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\ts_add_u32 %4, %4, 1\n\ts_add_u32 %5, %5, 1\n\ts_add_u32 %6, %6, 1\n\ts_add_u32 %7, %7, 1\n\t")
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
    } else if constexpr (MODE == M_SALU_DEP) {
      asm volatile(REP32("s_add_u32 %0, %0, 1\n\t") : "+s"(s1) :: "scc");
    } else if constexpr (MODE == M_MIX_INDEP) {
      asm volatile(REP4("v_add_u32 %0, %0, 1\n\ts_add_u32 %4, %4, 1\n\tv_add_u32 %1, %1, 1\n\ts_add_u32 %5, %5, 1\n\tv_add_u32 %2, %2, 1\n\ts_add_u32 %6, %6, 1\n\tv_add_u32 %3, %3, 1\n\ts_add_u32 %7, %7, 1\n\t")
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4) :: "scc");
    } else if constexpr (MODE == M_READLANE_INDEP) {
      asm volatile(REP4("v_readlane_b32 %0, %8, 3\n\tv_readlane_b32 %1, %8, 5\n\tv_readlane_b32 %2, %8, 7\n\tv_readlane_b32 %3, %8, 9\n\tv_readlane_b32 %4, %8, 11\n\tv_readlane_b32 %5, %8, 13\n\tv_readlane_b32 %6, %8, 15\n\tv_readlane_b32 %7, %8, 17\n\t")
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(v0));
    } else if constexpr (MODE == M_READLANE_SALU_RT) {
      // lane select written by SALU: no hazard wait states needed (only a VALU-written SGPR needs 4)
      asm volatile(REP16("v_readlane_b32 %0, %1, %0\n\ts_and_b32 %0, %0, 63\n\t") : "+s"(s1) : "v"(v0) : "scc");
    } else if constexpr (MODE == M_VALU_TO_SGPR_RT) {
      asm volatile(REP16("v_cmp_lt_u32_e64 vcc, %0, %1\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\t") : "+v"(v0) : "v"(v1), "v"(v2) : "vcc", "scc");
    } else if constexpr (MODE == M_BRANCH_TAKEN) {
      asm volatile(REP16("s_branch 1f\n1:\n\t") ::: "memory");
    } else if constexpr (MODE == M_CBRANCH_NOT_TAKEN) {
      asm volatile(REP16("s_cmp_eq_u32 %0, -1\n\ts_cbranch_scc1 9f\n\t") "9:\n\t" :: "s"(s1) : "scc");
    } else if constexpr (MODE == M_VCC_BRANCH_TAKEN) {
      asm volatile(REP16("v_cmp_ne_u32_e32 vcc, -1, %0\n\ts_cbranch_vccnz 1f\n\ts_nop 0\n1:\n\t") :: "v"(v1) : "vcc");
    } else if constexpr (MODE == M_VCC_BRANCH_NOT) {
      asm volatile(REP16("v_cmp_ne_u32_e32 vcc, -1, %0\n\ts_cbranch_vccz 9f\n\t") "9:\n\t" :: "v"(v1) : "vcc");
    } else if constexpr (MODE == M_MPS_BIN) {
      // decode_bin's statement, MPS without renormalisation: R stays >= 0x8000 because the LPS width read is 0 (t_lps register = 0) and value 0 < R.
      // 15 instructions: head_q(2) cmp nop readlane nop(3 states) readlane s_lshr nop bfe mad cmp cbranch | pk_sub cmp cbranch cndmask
      uint32_t r, st, row; uint64_t mask; uint32_t vt, vl, vn;
      asm volatile(REP8(
        "v_lshrrev_b32 %[vt], 10, %[R]\n\tv_and_b32 %[vt], 24, %[vt]\n\t"
        "v_cmp_eq_u32_e64 %[mask], %[c], %[lane]\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[st], %[grp], %[c]\n\t"
        "s_nop 3\n\t"
        "v_readlane_b32 %[row], %[tl], %[st]\n\t"
        "s_lshr_b32 %[r], %[st], 16\n\t"
        "s_nop 0\n\t"
        "v_bfe_u32 %[vl], %[row], %[vt], 8\n\t"
        "v_mad_i32_i24 %[R], %[vl], %[m128], %[R]\n\t"
        "v_cmp_lt_u32_e32 vcc, %[val], %[R]\n\t"
        "s_cbranch_vccz 7f\n\t"
        "v_pk_sub_u16 %[vn], %[st], 1 clamp\n\t"
        "v_cmp_gt_u32_e32 vcc, 0x8000, %[R]\n\t"
        "s_cbranch_vccnz 7f\n\t"
        "v_cndmask_b32_e64 %[grp], %[grp], %[vn], %[mask]\n\t") "7:\n\t"
        : [grp] "+v"(v1), [R] "+v"(v2), [r] "=&s"(r), [st] "=&s"(st), [row] "=&s"(row), [mask] "=&s"(mask), [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn)
        : [c] "s"(s3), [tl] "v"(v4), [lane] "v"((uint32_t)threadIdx.x), [val] "v"(v5), [m128] "s"(0xffffff80u)
        : "vcc", "scc");
      s2 += r;
    } else if constexpr (MODE == M_MIX_2V1S) {
      asm volatile(REP32("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\t") : "+v"(v0), "+v"(v1), "+s"(s1) :: "scc");
    } else if constexpr (MODE == M_MIX_1V2S) {
      asm volatile(REP32("v_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\t") : "+v"(v0), "+s"(s1), "+s"(s2) :: "scc");
    } else if constexpr (MODE == M_VALU_MOV_SGPR) {
      asm volatile(REP4("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8\n\t")
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s1));
    } else if constexpr (MODE == M_CNDMASK_SGPRMASK) {
      uint64_t m = 0x5555555555555555ull;
      asm volatile("" : "+s"(m));
      asm volatile(REP4("v_cndmask_b32_e64 %0, %0, %8, %9\n\tv_cndmask_b32_e64 %1, %1, %8, %9\n\tv_cndmask_b32_e64 %2, %2, %8, %9\n\tv_cndmask_b32_e64 %3, %3, %8, %9\n\tv_cndmask_b32_e64 %4, %4, %8, %9\n\tv_cndmask_b32_e64 %5, %5, %8, %9\n\tv_cndmask_b32_e64 %6, %6, %8, %9\n\tv_cndmask_b32_e64 %7, %7, %8, %9\n\t")
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(s1), "s"(m));
    } else if constexpr (MODE == M_DPP_MOV) {
      asm volatile(REP32("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t") : "+v"(v0));
    } else if constexpr (MODE == M_BPERMUTE) {
      asm volatile(REP8("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(v0) : "v"(v1));
    } else if constexpr (MODE == M_SNOP) {
      asm volatile(REP32("s_nop 0\n\t"));
    } else if constexpr (MODE == M_V1C) {
      asm volatile(REP32("v_add_u32 %0, %0, 1\n\t") : "+v"(v0));
    } else if constexpr (MODE == M_V2C) {
      asm volatile(REP16("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\t") : "+v"(v0), "+v"(v1));
    } else if constexpr (MODE == M_V8C) {
      asm volatile(REP4("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1\n\tv_add_u32 %4, %4, 1\n\tv_add_u32 %5, %5, 1\n\tv_add_u32 %6, %6, 1\n\tv_add_u32 %7, %7, 1\n\t")
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    } else if constexpr (MODE == M_V2VG) {
      asm volatile(REP16("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2\n\t") : "+v"(v0), "+v"(v1) : "v"(v6));
    } else if constexpr (MODE == M_V2SG) {
      asm volatile(REP16("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2\n\t") : "+v"(v0), "+v"(v1) : "s"(s1));
    } else if constexpr (MODE == M_V3S1) {
      asm volatile(REP8(REP2("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t") "v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+s"(s1) :: "scc");
    } else if constexpr (MODE == M_V4S1) {
      asm volatile(REP16("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1\n\ts_add_u32 %4, %4, 1\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s1) :: "scc");
    } else if constexpr (MODE == M_XOR2) {
      asm volatile(REP16("v_xor_b32 %0, %0, %2\n\tv_xor_b32 %1, %1, %2\n\t") : "+v"(v0), "+v"(v1) : "v"(v6));
    } else if constexpr (MODE == M_RL_ADD) {
      asm volatile(REP16("v_readlane_b32 %1, %2, 5\n\tv_add_u32 %0, %0, 1\n\t") : "+v"(v0), "+s"(s1) : "v"(v6));
    } else if constexpr (MODE == M_CMP_CND) {
      asm volatile(REP16("v_cmp_lt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\t") : "+v"(v0) : "v"(v1), "v"(v2) : "vcc");
    } else if constexpr (MODE == M_CMP64_S) {
      uint64_t m0, m1;
      asm volatile(REP16("v_cmp_eq_u32_e64 %0, %2, %3\n\tv_cmp_eq_u32_e64 %1, %2, %3\n\t") : "=&s"(m0), "=&s"(m1) : "s"(s1), "v"(v0));
      s2 += (uint32_t)m0 + (uint32_t)m1;
    } else if constexpr (MODE == M_MAD24_S) {
      asm volatile(REP16("v_mad_i32_i24 %0, %2, %3, %0\n\tv_mad_i32_i24 %1, %2, %3, %1\n\t") : "+v"(v0), "+v"(v1) : "v"(v6), "s"(s1));
    } else if constexpr (MODE == M_BFE_S) {
      asm volatile(REP16("v_bfe_u32 %0, %2, %0, 8\n\tv_bfe_u32 %1, %2, %1, 8\n\t") : "+v"(v0), "+v"(v1) : "s"(s1));
    } else if constexpr (MODE == M_PKSUB_S) {
      asm volatile(REP16("v_pk_sub_u16 %0, %2, 1 clamp\n\tv_pk_sub_u16 %1, %2, 1 clamp\n\t") : "+v"(v0), "+v"(v1) : "s"(s1));
    } else if constexpr (MODE == M_V1S1_SG) {
      asm volatile(REP16("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 1\n\t") : "+v"(v0), "+s"(s1) : "s"(s2) : "scc");
    } else if constexpr (MODE == M_V1B1) {
      asm volatile("s_cmp_eq_u32 %1, -1\n\t" REP16("v_add_u32 %0, %0, 1\n\ts_cbranch_scc1 9f\n\t") "9:\n\t" : "+v"(v0) : "s"(s1) : "scc");
    } else if constexpr (MODE == M_S1B1) {
      asm volatile(REP16("s_add_u32 %0, %0, 0\n\ts_cbranch_scc1 9f\n\t") "9:\n\t" : "+s"(s1) :: "scc");
    } else if constexpr (MODE == M_V1S1B1) {
      asm volatile(REP8("v_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 0\n\ts_cbranch_scc1 9f\n\t") REP2("v_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 0\n\ts_cbranch_scc1 9f\n\t") "9:\n\t" : "+v"(v0), "+s"(s1) :: "scc");
    } else if constexpr (MODE == M_WRLANE) {
      asm volatile(REP16("v_writelane_b32 %0, %2, 3\n\tv_writelane_b32 %1, %2, 5\n\t") : "+v"(v0), "+v"(v1) : "s"(s1));
    } else if constexpr (MODE == M_LDS_BIN) {
      // state word 0 everywhere: p' = 0, valMps 0; LPS row 0 -> R never changes, MPS always, no renormalisation (R = 0xff00 >= 0x8000)
      uint32_t vst, vcn, vt, vr, vrow, vb, vl, vn;
      uint32_t vc = lbase, vj = lbase + 512u + 60u, vm128 = 0xffffff80u;
      asm volatile("" : "+v"(vc), "+v"(vj), "+v"(vm128));
      asm volatile(REP8(
        "ds_read_b32 %[vst], %[vc]\n\t"
        "v_add_u32 %[vj], -4, %[vj]\n\t"
        "ds_read_b32 %[vcn], %[vj]\n\t"
        "v_lshrrev_b32 %[vt], 10, %[R]\n\tv_and_b32 %[vt], 24, %[vt]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_and_b32 %[vr], 0xfc, %[vst]\n\t"
        "ds_read_b32 %[vrow], %[vr] offset:256\n\t"
        "v_lshrrev_b32 %[vb], 16, %[vst]\n\t"
        "v_lshl_or_b32 %[acc], %[acc], 1, %[vb]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[vl], %[vrow], %[vt], 8\n\t"
        "v_mad_i32_i24 %[R], %[vl], %[m128], %[R]\n\t"
        "v_cmp_lt_u32_e32 vcc, %[val], %[R]\n\t"
        "s_cbranch_vccz 7f\n\t"
        "v_pk_sub_u16 %[vn], %[vst], 4 clamp\n\t"
        "v_cmp_gt_u32_e32 vcc, 0x8000, %[R]\n\t"
        "s_cbranch_vccnz 7f\n\t"
        "ds_write_b32 %[vc], %[vn]\n\t"
        "v_mov_b32 %[vc], %[vcn]\n\t"
        "s_add_u32 %[j], %[j], -1\n\t"
        "s_cbranch_scc0 7f\n\t") "7:\n\t"
        : [R] "+v"(v2), [acc] "+v"(v3), [vc] "+v"(vc), [vj] "+v"(vj), [j] "+s"(s4),
          [vst] "=&v"(vst), [vcn] "=&v"(vcn), [vt] "=&v"(vt), [vr] "=&v"(vr), [vrow] "=&v"(vrow), [vb] "=&v"(vb), [vl] "=&v"(vl), [vn] "=&v"(vn)
        : [val] "v"(v5), [m128] "v"(vm128) : "vcc", "scc", "memory");
      s4 = 1000;
    } else if constexpr (MODE == M_DSR_SAME) {
      uint32_t a = lbase; asm volatile("" : "+v"(a));
      asm volatile(REP4("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\tds_read_b32 %3, %8 offset:12\n\tds_read_b32 %4, %8 offset:16\n\tds_read_b32 %5, %8 offset:20\n\tds_read_b32 %6, %8 offset:24\n\tds_read_b32 %7, %8 offset:28\n\t") "s_waitcnt lgkmcnt(0)\n\t"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a) : "memory");
    } else if constexpr (MODE == M_DSW_SAME) {
      uint32_t a = lbase; asm volatile("" : "+v"(a));
      asm volatile(REP32("ds_write_b32 %0, %1\n\t") "s_waitcnt lgkmcnt(0)\n\t" :: "v"(a), "v"(v1) : "memory");
    } else if constexpr (MODE == M_DSW_LANE0) {
      uint32_t a = (threadIdx.x & 63) ? 0xfffffff0u : lbase; asm volatile("" : "+v"(a));
      asm volatile(REP32("ds_write_b32 %0, %1\n\t") "s_waitcnt lgkmcnt(0)\n\t" :: "v"(a), "v"(v1) : "memory");
    } else if constexpr (MODE == M_SNOP15) {
      asm volatile(REP32("s_nop 15\n\t"));
    } else if constexpr (MODE == M_DSR_CHAIN) {
      uint32_t a = lbase; asm volatile("" : "+v"(a));
      asm volatile(REP16("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xfc, %0\n\t") : "+v"(a) :: "memory");
      v0 += a;
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1));
  const uint32_t acc = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
  if (acc == 0x12345678u) sink[0] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = rt1 - rt0; }
}

typedef void (*KFn)(int, uint64_t*, uint32_t*);
template <int M> struct Table { static void fill(KFn* t) { t[M] = k_issue<M>; Table<M + 1>::fill(t); } };
template <> struct Table<M_COUNT> { static void fill(KFn*) {} };

int main(int argc, char** argv)
{
  int iters = argc > 1 ? atoi(argv[1]) : 4000;
  int first = argc > 2 ? atoi(argv[2]) : 0;
  KFn table[M_COUNT]; Table<0>::fill(table);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  uint64_t* out; uint32_t* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 64);
  printf("# %s, %d CUs; body = straight-line asm, %d iterations per wave; blocks of 256 threads (one wave per SIMD), W blocks per CU\n", prop.gcnArchName, cus, iters);
  printf("# %-58s %3s %12s %12s %10s %12s %12s\n", "mode", "W", "cyc/unit", "cyc/instr", "clock GHz", "instr/cyc/CU", "instr/ns/CU");
  for (int m = 0; m < M_COUNT; m++) {
    if (m < first) continue;
    for (int w : {1, 2, 4, 8}) {
      const int blocks = cus * w;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(table[m], dim3(blocks), dim3(256), 0, 0, 64, out, sink);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(table[m], dim3(blocks), dim3(256), 0, 0, iters, out, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      uint64_t h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      const double clock_ghz = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;   // s_memrealtime: 100 MHz
      const double total_instr = (double)blocks * 4.0 * iters * kInstrs[m];
      const double ipc_cu = total_instr / (ms * 1e-3) / cus / (clock_ghz * 1e9);
      printf("  %-58s %3d %12.2f %12.2f %10.3f %12.3f %12.3f\n", kNames[m], w, (double)h[0] / iters / kUnits[m], (double)h[0] / iters / kInstrs[m], clock_ghz, ipc_cu, total_instr / (ms * 1e6) / cus);
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
  }
  return 0;
}
