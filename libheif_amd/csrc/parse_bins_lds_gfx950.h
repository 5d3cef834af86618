// parse_bins_lds_gfx950.h — the hand-scheduled gfx950 statements of the CABAC parser for the build with LDS-resident contexts
// (HIPDEC_PARSE_LDS_CTX, the throughput kernel k_parse_occ8): the context-coded bin, the sig_coeff_flag run, the coeff_abs_level_greater1 run and the
// unary context-coded prefix.  Included by parse_core.h inside namespace hipdec::pcore, device builds only (PC_ASM_BINS); the register-file twin is
// parse_bins_gfx950.h (latency kernel k_parse, the inter / general builds), the C++ twin decode_bin_cxx & co. in parse_core.h.
//
// Why (profiles/r06_issue_model_*.txt, MI355X): at 8 waves per SIMD a VALU instruction that touches the scalar register file (SGPR operand, VCC,
// v_readlane / v_writelane, v_cmp, v_cndmask) issues at 0.9 per cycle and CU, one that only touches VGPRs and inline constants at 1.76, and k_parse
// ran at exactly 0.9 VALU per cycle and CU: the register-file form of an MPS bin has 9 scalar-file VALU instructions and 2 pure ones (~51 SIMD cycles
// of issue), this form 2 and 10 (~31) plus four LDS instructions on the otherwise idle LDS pipe.
//
// Conventions of every statement here
//   * the context variable is  (p' << 2) | valMps << 16  (p' = 62 - pStateIdx): its low half is the byte offset of the variable's rangeTabLps row
//     in Lds::tlps and of its LPS successor in Lds::tnext; the MPS transition is one v_pk_sub_u16 ... 4 clamp;
//   * LDS addresses are byte offsets inside Lds, which sits at LDS address 0 (the kernel's only __shared__ object; parse_wave checks it), wave-uniform
//     in VGPRs, so every ds_read is a broadcast and every ds_write stores the same dword from all lanes;
//   * LDS instructions of one wave complete in order, so a read behind a write of the same variable sees it, and counted lgkmcnt waits are exact
//     once the statement has drained what the compiler left in flight (the first wait of every statement is lgkmcnt(0));
//   * range / value / bits_needed stay wave-uniform vector values as in the register-file build; the byte refill still reads the window register.
#define PC_LDS_HEAD_Q   "v_lshrrev_b32 %[vt], 10, %[R]\n\tv_and_b32 %[vt], 24, %[vt]\n\t"
// LPS tail: value -= R, renormalise by clz(lps), the variable after the LPS from tnext (valMps flips where its bit 16 is set), the bin flips
#define PC_LDS_LPS                                                                                                                 \
  "ds_read_b32 %[vn], %[vr] offset:%c[tnext]\n\t"                                                                                   \
  "v_sub_u32 %[val], %[val], %[R]\n\t"                                                                                              \
  "v_ffbh_u32 %[vt], %[vl]\n\t"                                                                                                     \
  "v_add_u32 %[vt], -16, %[vt]\n\t"                                                                                                 \
  "v_lshlrev_b32 %[R], %[vt], %[vl]\n\t"                                                                                            \
  "v_add_u32 %[vt], -7, %[vt]\n\t"                                                                                                  \
  "v_lshlrev_b32 %[val], %[vt], %[val]\n\t"                                                                                         \
  "v_add_u32 %[bits], %[vt], %[bits]\n\t"                                                                                           \
  "v_and_b32 %[vt], 0x10000, %[vst]\n\t"                                                                                            \
  "v_xor_b32 %[vb], 1, %[vb]\n\t"                                                                                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                        \
  "v_xor_b32 %[vn], %[vn], %[vt]\n\t"
// byte refill from the window register; branches to SLOW when the fast window is exhausted
#define PC_LDS_REFILL(SLOW)                                                                                                        \
  "s_cmp_lt_u32 %[pos], %[flim]\n\t"                                                                                                \
  "s_cbranch_scc0 " SLOW "\n\t"                                                                                                     \
  "s_lshr_b32 %[st], %[pos], 2\n\t"                                                                                                 \
  "s_lshl_b32 %[row], %[pos], 3\n\t"                                                                                                \
  "v_readlane_b32 %[st], %[win], %[st]\n\t"                                                                                         \
  "s_add_u32 %[pos], %[pos], 1\n\t"                                                                                                 \
  "s_nop 0\n\t"                                                                                                                     \
  "s_lshr_b32 %[st], %[st], %[row]\n\t"                                                                                             \
  "s_and_b32 %[st], %[st], 0xff\n\t"                                                                                                \
  "v_lshl_add_u32 %[val], %[st], %[bits], %[val]\n\t"                                                                               \
  "v_add_u32 %[bits], -8, %[bits]\n\t"
// the part of a bin every statement shares: state in %[vst] (being loaded) -> range update, MPS / LPS decision.  Falls through on an MPS
// with %[vn] = the variable after the MPS and VCC = "renormalise"; LPS_LABEL is taken for an LPS.
#ifdef HIPDEC_EXP_BIN_LATENCY   // measurement build: ~64 cycles of pure latency per bin (is the pool latency- or issue-bound?)
#define PC_LDS_EXP "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#else
#define PC_LDS_EXP
#endif
#define PC_LDS_CORE(LPS_LABEL)                                                                                                     \
  PC_LDS_EXP                                                                                                                        \
  "v_and_b32 %[vr], 0xfc, %[vst]\n\t"                                                                                               \
  "ds_read_b32 %[vrow], %[vr] offset:%c[tlps]\n\t"                                                                                  \
  "v_lshrrev_b32 %[vb], 16, %[vst]\n\t"                                                                                             \
  "v_pk_sub_u16 %[vn], %[vst], 4 clamp\n\t"                                                                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                        \
  "v_bfe_u32 %[vl], %[vrow], %[vt], 8\n\t"                                                                                          \
  "v_mad_i32_i24 %[R], %[vl], %[m128], %[R]\n\t"                                                                                    \
  "v_cmp_lt_u32_e32 vcc, %[val], %[R]\n\t"                                                                                          \
  "s_cbranch_vccz " LPS_LABEL "\n\t"                                                                                                \
  "v_cmp_gt_u32_e32 vcc, 0x8000, %[R]\n\t"
#define PC_LDS_MPS_SHIFT "v_lshlrev_b32 %[R], 1, %[R]\n\tv_lshlrev_b32 %[val], 1, %[val]\n\tv_add_u32 %[bits], 1, %[bits]\n"
#define PC_LDS_OFFSETS [tlps] "i"(__builtin_offsetof(Lds, tlps)), [tnext] "i"(__builtin_offsetof(Lds, tnext)), [vctx] "i"(__builtin_offsetof(Lds, vctx))
PC_DEV uint32_t pc_lds_ctx_addr(CtxGroup grp, int lane) { return (uint32_t)__builtin_offsetof(Lds, ctx) + 4u * (uint32_t)(grp.base + lane); }

PC_DEV int decode_bin(PS& s, CtxRef grp, int ctx_lane)
{
  uint32_t r, st, row, flag;
  uint32_t vst, vrow, vt, vl, vn, vr, vb;
  const uint32_t va = pc_lds_ctx_addr(grp, ctx_lane);   // (a literal for most call sites: the compiler materialises it with one move)
  uint32_t pos = pc_uni(s.pos);
  const uint32_t flim = pc_uni(s.fast_limit);
  asm volatile(
    "ds_read_b32 %[vst], %[va]\n\t"
    "s_mov_b32 %[flag], 0\n\t"
    PC_LDS_HEAD_Q
    "s_waitcnt lgkmcnt(0)\n\t"
    PC_LDS_CORE("1f")
    "s_cbranch_vccz 6f\n\t"                      // MPS without renormalisation: done
    PC_LDS_MPS_SHIFT
    "s_branch 2f\n"
    "1:\n\t"
    PC_LDS_LPS
    "2:\n\t"
    "v_cmp_lt_i32_e32 vcc, -1, %[bits]\n\t"
    "s_cbranch_vccz 6f\n\t"
    PC_LDS_REFILL("3f")
    "s_branch 6f\n"
    "3:\n\t"
    "s_mov_b32 %[flag], 1\n"
    "6:\n\t"
    "ds_write_b32 %[va], %[vn]\n\t"
    "v_readfirstlane_b32 %[r], %[vb]\n\t"
    : [R] "+v"(s.range), [val] "+v"(s.value), [bits] "+v"(s.bits_needed), [pos] "+s"(pos),
      [r] "=&s"(r), [st] "=&s"(st), [row] "=&s"(row), [flag] "=&s"(flag),
      [vst] "=&v"(vst), [vrow] "=&v"(vrow), [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn), [vr] "=&v"(vr), [vb] "=&v"(vb)
    : [va] "v"(va), [win] "v"(s.win), [flim] "s"(flim), [m128] "v"(0xffffff80u), PC_LDS_OFFSETS
    : "vcc", "scc", "memory");
  s.pos = pc_uni(pos);
  if (__builtin_expect(pc_uni(flag) != 0u, 0)) refill_byte(s);
  return (int)pc_uni(r);
}

// sig_coeff_flag of the scan positions n_start .. 1 of one sub-block (bit k of the result = position k).  The context variables' addresses of the 16
// positions are in Lds::vctx (residual_coding stores them beside computing vctx); the run reads the next position's address while it decodes the
// current one and collects the bins in a vector register: ONE crossing to the scalar side per run.
#define PC_LDS_SIG_ITER(P, CA, CB, NEXT)                                                                                           \
  "1" P "0:\n\t"                                                                                                                    \
  "ds_read_b32 %[vst], %[" CA "]\n\t"                                                                                               \
  "v_add_u32 %[vj], -4, %[vj]\n\t"                                                                                                  \
  "ds_read_b32 %[" CB "], %[vj] offset:%c[vctx]\n\t"                                                                                \
  PC_LDS_HEAD_Q                                                                                                                     \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                                        \
  PC_LDS_CORE("1" P "1f")                                                                                                           \
  "s_cbranch_vccnz 1" P "2f\n"                                                                                                      \
  "1" P "4:\n\t"                                                                                                                    \
  "ds_write_b32 %[" CA "], %[vn]\n\t"                                                                                               \
  "v_lshl_or_b32 %[vacc], %[vacc], 1, %[vb]\n\t"                                                                                    \
  "s_add_u32 %[j], %[j], -1\n\t"                                                                                                    \
  "s_cbranch_scc1 " NEXT "\n\t"                                                                                                     \
  "s_branch 190f\n"                                                                                                                 \
  "1" P "1:\n\t"                                                                                                                    \
  PC_LDS_LPS                                                                                                                        \
  "s_branch 1" P "3f\n"                                                                                                             \
  "1" P "2:\n\t"                                                                                                                    \
  PC_LDS_MPS_SHIFT                                                                                                                  \
  "1" P "3:\n\t"                                                                                                                    \
  "v_cmp_lt_i32_e32 vcc, -1, %[bits]\n\t"                                                                                           \
  "s_cbranch_vccz 1" P "4b\n\t"                                                                                                     \
  PC_LDS_REFILL("1" P "5f")                                                                                                         \
  "s_branch 1" P "4b\n"                                                                                                             \
  "1" P "5:\n\t"                                                                                                                    \
  "ds_write_b32 %[" CA "], %[vn]\n\t"                                                                                               \
  "v_lshl_or_b32 %[vacc], %[vacc], 1, %[vb]\n\t"                                                                                    \
  "s_add_u32 %[j], %[j], -1\n\t"                                                                                                    \
  "s_mov_b32 %[flag], 1\n\t"                                                                                                        \
  "s_branch 190f\n"

PC_DEV uint32_t decode_sig_run(PS& s, const VReg& vctx, int n_start)
{
  (void)vctx;
  uint32_t vacc = 0;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vacc));
  int32_t j = __builtin_amdgcn_readfirstlane(n_start - 1);   // the position after the current one; the run ends when it leaves 0 .. 15
  for (;;) {
    uint32_t flag, st, row, t;
    uint32_t vst, vrow, vt, vl, vn, vr, vb, vca, vcb, vj;
    uint32_t pos = pc_uni(s.pos);
    const uint32_t flim = pc_uni(s.fast_limit);
    asm volatile(
      "s_lshl_b32 %[t], %[j], 2\n\t"
      "s_add_u32 %[t], %[t], 4\n\t"
      "s_mov_b32 %[flag], 0\n\t"
      "v_mov_b32 %[vj], %[t]\n\t"                               // byte offset of the current position inside vctx
      "ds_read_b32 %[vca], %[vj] offset:%c[vctx]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      PC_LDS_SIG_ITER("0", "vca", "vcb", "110f")
      PC_LDS_SIG_ITER("1", "vcb", "vca", "100b")
      "190:\n\t"
      : [R] "+v"(s.range), [val] "+v"(s.value), [bits] "+v"(s.bits_needed), [pos] "+s"(pos), [j] "+s"(j), [vacc] "+v"(vacc),
        [flag] "=&s"(flag), [st] "=&s"(st), [row] "=&s"(row), [t] "=&s"(t),
        [vst] "=&v"(vst), [vrow] "=&v"(vrow), [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn), [vr] "=&v"(vr), [vb] "=&v"(vb),
        [vca] "=&v"(vca), [vcb] "=&v"(vcb), [vj] "=&v"(vj)
      : [win] "v"(s.win), [flim] "s"(flim), [m128] "v"(0xffffff80u), PC_LDS_OFFSETS
      : "vcc", "scc", "memory");
    s.pos = pc_uni(pos);
    if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)flag) != 0, 0)) refill_byte(s);
    j = __builtin_amdgcn_readfirstlane(j);
    if (j < 0) break;
  }
  return pc_uni(vacc) << 1;
}

// coeff_abs_level_greater1_flag of one sub-block: n (1 .. 8) flags from the highest significant position down, ctxInc = min(greater1Ctx, 3) with
// greater1Ctx (g) reset by a 1 and counted up by 0s (9.3.4.2.6).  Returns the flags MSB-first (the first decoded flag in bit n - 1).  The context
// state machine runs on the vector side: g' = bin ? 0 : g + (g != 0)  =  (g + min(g, 1)) * (bin ^ 1).
PC_DEV uint32_t decode_g1_run(PS& s, int base_lane, int n, int& g_io)
{
  uint32_t vgb = 0, vg = 0;
  int32_t m = __builtin_amdgcn_readfirstlane(n - 1);
  const uint32_t vbase = pc_lds_ctx_addr(PS::ctxC, base_lane);
  asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, %2" : "=&v"(vgb), "=&v"(vg) : "s"((uint32_t)__builtin_amdgcn_readfirstlane(g_io)));
  for (;;) {
    uint32_t flag, st, row;
    uint32_t vst, vrow, vt, vl, vn, vr, vb, vc, vt2;
    uint32_t pos = pc_uni(s.pos);
    const uint32_t flim = pc_uni(s.fast_limit);
    asm volatile(
      "s_mov_b32 %[flag], 0\n\t"
      "s_waitcnt lgkmcnt(0)\n"
      "200:\n\t"
      "v_min_u32 %[vc], 3, %[vg]\n\t"
      "v_lshl_add_u32 %[vc], %[vc], 2, %[vbase]\n\t"
      "ds_read_b32 %[vst], %[vc]\n\t"
      PC_LDS_HEAD_Q
      "v_min_u32 %[vt2], 1, %[vg]\n\t"
      "v_add_u32 %[vg], %[vg], %[vt2]\n\t"                      // g + (g != 0): what g becomes behind a 0 flag
      "s_waitcnt lgkmcnt(0)\n\t"
      PC_LDS_CORE("201f")
      "s_cbranch_vccnz 202f\n"
      "204:\n\t"
      "ds_write_b32 %[vc], %[vn]\n\t"
      "v_lshl_or_b32 %[vgb], %[vgb], 1, %[vb]\n\t"
      "v_xor_b32 %[vt2], 1, %[vb]\n\t"
      "v_mul_u32_u24 %[vg], %[vg], %[vt2]\n\t"                  // ... and 0 behind a 1
      "s_add_u32 %[m], %[m], -1\n\t"
      "s_cbranch_scc1 200b\n\t"
      "s_branch 290f\n"
      "201:\n\t"
      PC_LDS_LPS
      "s_branch 203f\n"
      "202:\n\t"
      PC_LDS_MPS_SHIFT
      "203:\n\t"
      "v_cmp_lt_i32_e32 vcc, -1, %[bits]\n\t"
      "s_cbranch_vccz 204b\n\t"
      PC_LDS_REFILL("205f")
      "s_branch 204b\n"
      "205:\n\t"
      "ds_write_b32 %[vc], %[vn]\n\t"
      "v_lshl_or_b32 %[vgb], %[vgb], 1, %[vb]\n\t"
      "v_xor_b32 %[vt2], 1, %[vb]\n\t"
      "v_mul_u32_u24 %[vg], %[vg], %[vt2]\n\t"
      "s_add_u32 %[m], %[m], -1\n\t"
      "s_mov_b32 %[flag], 1\n"
      "290:\n\t"
      : [R] "+v"(s.range), [val] "+v"(s.value), [bits] "+v"(s.bits_needed), [pos] "+s"(pos), [m] "+s"(m), [vgb] "+v"(vgb), [vg] "+v"(vg),
        [flag] "=&s"(flag), [st] "=&s"(st), [row] "=&s"(row),
        [vst] "=&v"(vst), [vrow] "=&v"(vrow), [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn), [vr] "=&v"(vr), [vb] "=&v"(vb), [vc] "=&v"(vc),
        [vt2] "=&v"(vt2)
      : [vbase] "v"(vbase), [win] "v"(s.win), [flim] "s"(flim), [m128] "v"(0xffffff80u), PC_LDS_OFFSETS
      : "vcc", "scc", "memory");
    s.pos = pc_uni(pos);
    if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)flag) != 0, 0)) refill_byte(s);
    m = __builtin_amdgcn_readfirstlane(m);
    if (m < 0) break;
  }
  g_io = (int)pc_uni(vg);
  return pc_uni(vgb);
}

// A unary context-coded prefix (last_sig_coeff_x / y_prefix): bins with context lane base + (i >> shift) while they are 1, at most `max` of them;
// returns the number of 1s.  The count lives on both sides: i (scalar) bounds the loop, vi (vector) addresses the context.
PC_DEV int decode_unary_ctx_run(PS& s, CtxRef grp, int base_lane, int shift_, int max_)
{
  const uint32_t mx = (uint32_t)__builtin_amdgcn_readfirstlane(max_);
  if (mx == 0) return 0;
  uint32_t i = 0, vi = 0;
  const uint32_t vbase = pc_lds_ctx_addr(grp, base_lane);
  uint32_t vsh;
  asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, %2" : "=&v"(vi), "=&v"(vsh) : "s"((uint32_t)__builtin_amdgcn_readfirstlane(shift_)));
  for (;;) {
    uint32_t flag, st, row;
    uint32_t vst, vrow, vt, vl, vn, vr, vb, vc;
    uint32_t pos = pc_uni(s.pos);
    const uint32_t flim = pc_uni(s.fast_limit);
    asm volatile(
      "s_mov_b32 %[flag], 0\n\t"
      "s_waitcnt lgkmcnt(0)\n"
      "300:\n\t"
      "v_lshrrev_b32 %[vc], %[vsh], %[vi]\n\t"
      "v_lshl_add_u32 %[vc], %[vc], 2, %[vbase]\n\t"
      "ds_read_b32 %[vst], %[vc]\n\t"
      PC_LDS_HEAD_Q
      "s_waitcnt lgkmcnt(0)\n\t"
      PC_LDS_CORE("301f")
      "s_cbranch_vccnz 302f\n"
      "304:\n\t"
      "ds_write_b32 %[vc], %[vn]\n\t"
      "v_cmp_eq_u32_e32 vcc, 0, %[vb]\n\t"
      "s_cbranch_vccnz 390f\n\t"                 // a 0 bin ends the prefix
      "s_add_u32 %[i], %[i], 1\n\t"
      "v_add_u32 %[vi], 1, %[vi]\n\t"
      "s_cmp_lt_u32 %[i], %[mx]\n\t"
      "s_cbranch_scc1 300b\n\t"
      "s_branch 390f\n"
      "301:\n\t"
      PC_LDS_LPS
      "s_branch 303f\n"
      "302:\n\t"
      PC_LDS_MPS_SHIFT
      "303:\n\t"
      "v_cmp_lt_i32_e32 vcc, -1, %[bits]\n\t"
      "s_cbranch_vccz 304b\n\t"
      PC_LDS_REFILL("305f")
      "s_branch 304b\n"
      "305:\n\t"                                   // slow refill: finish this bin's bookkeeping, leave with flag = 1 (+ 2 when the prefix is complete)
      "ds_write_b32 %[vc], %[vn]\n\t"
      "s_mov_b32 %[flag], 1\n\t"
      "v_cmp_eq_u32_e32 vcc, 0, %[vb]\n\t"
      "s_cbranch_vccnz 306f\n\t"
      "s_add_u32 %[i], %[i], 1\n\t"
      "v_add_u32 %[vi], 1, %[vi]\n\t"
      "s_cmp_lt_u32 %[i], %[mx]\n\t"
      "s_cbranch_scc1 390f\n"
      "306:\n\t"
      "s_mov_b32 %[flag], 3\n"
      "390:\n\t"
      : [R] "+v"(s.range), [val] "+v"(s.value), [bits] "+v"(s.bits_needed), [pos] "+s"(pos), [i] "+s"(i), [vi] "+v"(vi),
        [flag] "=&s"(flag), [st] "=&s"(st), [row] "=&s"(row),
        [vst] "=&v"(vst), [vrow] "=&v"(vrow), [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn), [vr] "=&v"(vr), [vb] "=&v"(vb), [vc] "=&v"(vc)
      : [vbase] "v"(vbase), [vsh] "v"(vsh), [mx] "s"(mx), [win] "v"(s.win), [flim] "s"(flim), [m128] "v"(0xffffff80u), PC_LDS_OFFSETS
      : "vcc", "scc", "memory");
    s.pos = pc_uni(pos);
    i = pc_uni(i);
    const uint32_t f = pc_uni(flag);
    if (__builtin_expect(f == 0u, 1)) break;
    refill_byte(s);
    if (f & 2u) break;
  }
  return (int)i;
}
