// parse_bins_gfx950.h — the hand-scheduled gfx950 statements of the CABAC parser, in one reviewable file: the context-coded bin
// (decode_bin), the sig_coeff_flag run and the coeff_abs_level_greater1 run of residual_coding.  Included by parse_core.h inside
// namespace hipdec::pcore, device builds only (PC_ASM_BINS); the host emulation and -DHIPDEC_PARSE_CXX_BINS builds compile the C++
// forms of the same arithmetic that follow the #include there (decode_bin_cxx & co.), and the GPU parity suite runs these.
// Not a stand-alone header: it uses PS, UReg and the pc_* primitives parse_core.h defines above the #include.
// ---- hand-scheduled gfx950 forms of the two hot primitives ---------------------------------------------------------------------------
// The compiler's code for decode_bin_cxx inside the sig_coeff_flag loop is 37 instructions per MPS bin (profiles/r03a_*): four hazard
// s_nops, a v_mov per SGPR->VGPR hand-over, three instructions for the saturating state increment, shift / add / compare / branch of the
// renormalisation although an MPS whose range stays >= 256 shifts nothing, phi copies around the byte refill.  Written by hand the same
// bin is 18 instructions with every hazard slot filled by useful work:
//   * the wait states a VALU-written SGPR needs before it can be a lane select (4) or a VALU operand (2) are filled with the qRangeIdx
//     extraction, the write-back mask compare and (in the run) the read of the NEXT position's context index;
//   * R - (lps << 7) is one v_mad_i32_i24 on the scaled range, the MPS state update one v_pk_sub_u16 ... clamp;
//   * an MPS that needs no renormalisation skips shift, bit count and refill test with one compare + branch;
//   * the byte refill reads the 256-byte window register directly; only a window change / an emulation-prevention candidate / the end
//     of the substream leaves the statement (flag bit 1 of the result), where refill_byte() does it the general way.
// The statements are opaque to the compiler: all wait states are inside the strings (cdna guide section 5.7).  The host emulation runs
// decode_bin_cxx — the same arithmetic — and the GPU parity suite runs these (build with -DHIPDEC_PARSE_CXX_BINS for the C++ form on the device).
#define PC_ASM_HEAD_Q   "v_lshrrev_b32 %[vt], 10, %[R]\n\tv_and_b32 %[vt], 24, %[vt]\n\t"
// LPS tail shared by both statements: value -= R, renormalise by clz(lps), next state from t_next (valMps flips at pStateIdx 0)
#define PC_ASM_LPS(BIN_FIX)                                                                                                        \
  "v_sub_u32 %[val], %[val], %[R]\n\t"                                                                                              \
  "v_ffbh_u32 %[vt], %[vl]\n\t"                                                                                                     \
  "v_add_u32 %[vt], -16, %[vt]\n\t"                                                                                                 \
  "v_lshlrev_b32 %[R], %[vt], %[vl]\n\t"                                                                                            \
  "v_add_u32 %[vt], -7, %[vt]\n\t"                                                                                                  \
  "v_lshlrev_b32 %[val], %[vt], %[val]\n\t"                                                                                         \
  "v_add_u32 %[bits], %[vt], %[bits]\n\t"                                                                                           \
  "v_readlane_b32 %[row], %[tn], %[st]\n\t"                                                                                         \
  BIN_FIX                                                                                                                            \
  "s_and_b32 %[st], %[st], 0x10000\n\t"                                                                                             \
  "s_and_b32 %[row], %[row], 0x1003f\n\t"                                                                                           \
  "s_xor_b32 %[row], %[row], %[st]\n\t"                                                                                             \
  "v_mov_b32 %[vn], %[row]\n\t"
// byte refill from the window register; branches to SLOW when the fast window is exhausted
#define PC_ASM_REFILL(SLOW)                                                                                                        \
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

PC_DEV int decode_bin(PS& s, VReg& grp, int ctx_lane)
{
  uint32_t r, st, row;
  uint64_t mask;
  uint32_t vt, vl, vn;
  const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane(ctx_lane);
  uint32_t pos = pc_uni(s.pos);
  const uint32_t flim = pc_uni(s.fast_limit);
  asm volatile(
    PC_ASM_HEAD_Q
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
    "s_cbranch_vccnz 5f\n\t"
    PC_ASM_LPS("s_xor_b32 %[r], %[r], 1\n\t")
    "s_branch 2f\n"
    "1:\n\t"                                    // MPS with renormalisation: exactly one shift
    "v_lshlrev_b32 %[R], 1, %[R]\n\t"
    "v_lshlrev_b32 %[val], 1, %[val]\n\t"
    "v_add_u32 %[bits], 1, %[bits]\n"
    "2:\n\t"
    "v_cmp_lt_i32_e32 vcc, -1, %[bits]\n\t"
    "s_cbranch_vccz 6f\n\t"
    PC_ASM_REFILL("3f")
    "s_branch 6f\n"
    "3:\n\t"
    "s_or_b32 %[r], %[r], 2\n\t"
    "s_branch 6f\n"
    "5:\n\t"                                    // MPS
    "v_pk_sub_u16 %[vn], %[st], 1 clamp\n\t"
    "v_cmp_gt_u32_e32 vcc, 0x8000, %[R]\n\t"
    "s_cbranch_vccnz 1b\n"
    "6:\n\t"
    "v_cndmask_b32_e64 %[grp], %[grp], %[vn], %[mask]\n\t"
    : [grp] "+v"(grp), [R] "+v"(s.range), [val] "+v"(s.value), [bits] "+v"(s.bits_needed), [pos] "+s"(pos),
      [r] "=&s"(r), [st] "=&s"(st), [row] "=&s"(row), [mask] "=&s"(mask), [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn)
    : [c] "s"(c), [tl] "v"(s.t_lps), [tn] "v"(s.t_next), [lane] "v"((uint32_t)threadIdx.x), [win] "v"(s.win), [flim] "s"(flim),
      [m128] "s"(0xffffff80u)
    : "vcc", "scc");
  s.pos = pc_uni(pos);
  r = pc_uni(r);
  if (__builtin_expect(r > 1u, 0)) { refill_byte(s); r &= 1u; }
  return (int)r;
}

// sig_coeff_flag of the scan positions n_start .. 1 of one sub-block (bit k of the result = position k); lane k of vctx is the
// position's context variable (lane of group B).  One statement per run: the loop, the context read of the next position and the
// refills stay inside; it is left early only for a refill the window register cannot serve.
#define PC_ASM_SIG_ITER(P, CA, CB, NEXT)                                                                                           \
  "1" P "0:\n\t"                                                                                                                    \
  "v_readlane_b32 %[st], %[grp], %[" CA "]\n\t"                                                                                     \
  PC_ASM_HEAD_Q                                                                                                                     \
  "v_cmp_eq_u32_e64 %[mask], %[" CA "], %[lane]\n\t"                                                                                \
  "v_readlane_b32 %[" CB "], %[vx], %[j]\n\t"                                                                                       \
  "v_readlane_b32 %[row], %[tl], %[st]\n\t"                                                                                         \
  "s_lshr_b32 %[t], %[st], 16\n\t"                                                                                                  \
  "s_lshl1_add_u32 %[acc], %[acc], %[t]\n\t"                                                                                        \
  "v_bfe_u32 %[vl], %[row], %[vt], 8\n\t"                                                                                           \
  "v_mad_i32_i24 %[R], %[vl], %[m128], %[R]\n\t"                                                                                    \
  "v_cmp_lt_u32_e32 vcc, %[val], %[R]\n\t"                                                                                          \
  "s_cbranch_vccz 1" P "1f\n\t"                                                                                                     \
  "v_pk_sub_u16 %[vn], %[st], 1 clamp\n\t"                                                                                          \
  "v_cmp_gt_u32_e32 vcc, 0x8000, %[R]\n\t"                                                                                          \
  "s_cbranch_vccnz 1" P "2f\n"                                                                                                      \
  "1" P "4:\n\t"                                                                                                                    \
  "v_cndmask_b32_e64 %[grp], %[grp], %[vn], %[mask]\n\t"                                                                            \
  "s_add_u32 %[j], %[j], -1\n\t"                                                                                                    \
  "s_cbranch_scc1 " NEXT "\n\t"                                                                                                     \
  "s_branch 190f\n"                                                                                                                 \
  "1" P "1:\n\t"                                                                                                                    \
  PC_ASM_LPS("s_xor_b32 %[acc], %[acc], 1\n\t")                                                                                     \
  "s_branch 1" P "3f\n"                                                                                                             \
  "1" P "2:\n\t"                                                                                                                    \
  "v_lshlrev_b32 %[R], 1, %[R]\n\t"                                                                                                 \
  "v_lshlrev_b32 %[val], 1, %[val]\n\t"                                                                                             \
  "v_add_u32 %[bits], 1, %[bits]\n"                                                                                                 \
  "1" P "3:\n\t"                                                                                                                    \
  "v_cmp_lt_i32_e32 vcc, -1, %[bits]\n\t"                                                                                           \
  "s_cbranch_vccz 1" P "4b\n\t"                                                                                                     \
  PC_ASM_REFILL("1" P "5f")                                                                                                         \
  "s_branch 1" P "4b\n"                                                                                                             \
  "1" P "5:\n\t"                                                                                                                    \
  "v_cndmask_b32_e64 %[grp], %[grp], %[vn], %[mask]\n\t"                                                                            \
  "s_add_u32 %[j], %[j], -1\n\t"                                                                                                    \
  "s_mov_b32 %[flag], 1\n\t"                                                                                                        \
  "s_branch 190f\n"

PC_DEV uint32_t decode_sig_run(PS& s, const VReg& vctx, int n_start)
{
  uint32_t acc = 0;
  int32_t j = __builtin_amdgcn_readfirstlane(n_start - 1);   // the position after the current one; the run ends when it leaves 0 .. 15
  for (;;) {
    uint32_t flag, ca, cb, st, row, t;
    uint64_t mask;
    uint32_t vt, vl, vn;
    uint32_t pos = pc_uni(s.pos);
    const uint32_t flim = pc_uni(s.fast_limit);
    asm volatile(
      "s_add_u32 %[t], %[j], 1\n\t"
      "s_mov_b32 %[flag], 0\n\t"
      "v_readlane_b32 %[ca], %[vx], %[t]\n\t"
      "s_nop 3\n\t"
      PC_ASM_SIG_ITER("0", "ca", "cb", "110f")
      PC_ASM_SIG_ITER("1", "cb", "ca", "100b")
      "190:\n\t"
      : [grp] "+v"(s.ctxB), [R] "+v"(s.range), [val] "+v"(s.value), [bits] "+v"(s.bits_needed), [pos] "+s"(pos), [j] "+s"(j), [acc] "+s"(acc),
        [flag] "=&s"(flag), [ca] "=&s"(ca), [cb] "=&s"(cb), [st] "=&s"(st), [row] "=&s"(row), [t] "=&s"(t), [mask] "=&s"(mask),
        [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn)
      : [vx] "v"(vctx), [tl] "v"(s.t_lps), [tn] "v"(s.t_next), [lane] "v"((uint32_t)threadIdx.x), [win] "v"(s.win), [flim] "s"(flim),
        [m128] "s"(0xffffff80u)
      : "vcc", "scc");
    s.pos = pc_uni(pos);
    acc = pc_uni(acc);
    if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)flag) != 0, 0)) refill_byte(s);
    j = __builtin_amdgcn_readfirstlane(j);
    if (j < 0) break;
  }
  return acc << 1;
}

// coeff_abs_level_greater1_flag of one sub-block: n (1 .. 8) flags from the highest significant position down, ctxInc = min(greater1Ctx, 3)
// with greater1Ctx (g) reset by a 1 and counted up by 0s (9.3.4.2.6).  Returns the flags MSB-first (the first decoded flag in bit n - 1).
// One statement per run, the context state machine inside on the scalar side in the engine's hazard slots.
PC_DEV uint32_t decode_g1_run(PS& s, int base_lane, int n, int& g_io)
{
  uint32_t gb = 0;
  int32_t m = __builtin_amdgcn_readfirstlane(n - 1);
  uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane(g_io);
  const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(base_lane);
  for (;;) {
    uint32_t flag, c, st, row, b, g2;
    uint64_t mask;
    uint32_t vt, vl, vn;
    uint32_t pos = pc_uni(s.pos);
    const uint32_t flim = pc_uni(s.fast_limit);
    asm volatile(
      "s_mov_b32 %[flag], 0\n\t"
      "s_nop 1\n"
      "200:\n\t"
      "s_min_u32 %[c], %[g], 3\n\t"
      "s_add_u32 %[c], %[base], %[c]\n\t"
      "v_cmp_eq_u32_e64 %[mask], %[c], %[lane]\n\t"
      "v_readlane_b32 %[st], %[grp], %[c]\n\t"
      PC_ASM_HEAD_Q
      "s_cmp_lg_u32 %[g], 0\n\t"
      "s_addc_u32 %[g2], %[g], 0\n\t"
      "v_readlane_b32 %[row], %[tl], %[st]\n\t"
      "s_lshr_b32 %[b], %[st], 16\n\t"
      "s_lshl1_add_u32 %[gb], %[gb], %[b]\n\t"
      "v_bfe_u32 %[vl], %[row], %[vt], 8\n\t"
      "v_mad_i32_i24 %[R], %[vl], %[m128], %[R]\n\t"
      "v_cmp_lt_u32_e32 vcc, %[val], %[R]\n\t"
      "s_cbranch_vccz 201f\n\t"
      "v_pk_sub_u16 %[vn], %[st], 1 clamp\n\t"
      "v_cmp_gt_u32_e32 vcc, 0x8000, %[R]\n\t"
      "s_cbranch_vccnz 202f\n"
      "204:\n\t"
      "v_cndmask_b32_e64 %[grp], %[grp], %[vn], %[mask]\n\t"
      "s_cmp_eq_u32 %[b], 0\n\t"
      "s_cselect_b32 %[g], %[g2], 0\n\t"
      "s_add_u32 %[m], %[m], -1\n\t"
      "s_cbranch_scc1 200b\n\t"
      "s_branch 290f\n"
      "201:\n\t"
      PC_ASM_LPS("s_xor_b32 %[gb], %[gb], 1\n\ts_xor_b32 %[b], %[b], 1\n\t")
      "s_branch 203f\n"
      "202:\n\t"
      "v_lshlrev_b32 %[R], 1, %[R]\n\t"
      "v_lshlrev_b32 %[val], 1, %[val]\n\t"
      "v_add_u32 %[bits], 1, %[bits]\n"
      "203:\n\t"
      "v_cmp_lt_i32_e32 vcc, -1, %[bits]\n\t"
      "s_cbranch_vccz 204b\n\t"
      PC_ASM_REFILL("205f")
      "s_branch 204b\n"
      "205:\n\t"
      "v_cndmask_b32_e64 %[grp], %[grp], %[vn], %[mask]\n\t"
      "s_cmp_eq_u32 %[b], 0\n\t"
      "s_cselect_b32 %[g], %[g2], 0\n\t"
      "s_add_u32 %[m], %[m], -1\n\t"
      "s_mov_b32 %[flag], 1\n"
      "290:\n\t"
      : [grp] "+v"(s.ctxC), [R] "+v"(s.range), [val] "+v"(s.value), [bits] "+v"(s.bits_needed), [pos] "+s"(pos), [m] "+s"(m), [gb] "+s"(gb), [g] "+s"(g),
        [flag] "=&s"(flag), [c] "=&s"(c), [st] "=&s"(st), [row] "=&s"(row), [b] "=&s"(b), [g2] "=&s"(g2), [mask] "=&s"(mask),
        [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn)
      : [base] "s"(base), [tl] "v"(s.t_lps), [tn] "v"(s.t_next), [lane] "v"((uint32_t)threadIdx.x), [win] "v"(s.win), [flim] "s"(flim),
        [m128] "s"(0xffffff80u)
      : "vcc", "scc");
    s.pos = pc_uni(pos);
    gb = pc_uni(gb); g = pc_uni(g);
    if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)flag) != 0, 0)) refill_byte(s);
    m = __builtin_amdgcn_readfirstlane(m);
    if (m < 0) break;
  }
  g_io = (int)g;
  return gb;
}

// A unary context-coded prefix (last_sig_coeff_x / y_prefix): bins with context lane base + (i >> shift) while they are 1, at most `max`
// of them; returns the number of 1s.  One statement per run (the wrapper code around a decode_bin() per bin was ~13 instructions a bin).
PC_DEV int decode_unary_ctx_run(PS& s, VReg& grp, int base_lane, int shift_, int max_)
{
  const uint32_t mx = (uint32_t)__builtin_amdgcn_readfirstlane(max_);
  if (mx == 0) return 0;
  uint32_t i = 0;
  const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(base_lane), sh = (uint32_t)__builtin_amdgcn_readfirstlane(shift_);
  for (;;) {
    uint32_t flag, c, st, row, b;
    uint64_t mask;
    uint32_t vt, vl, vn;
    uint32_t pos = pc_uni(s.pos);
    const uint32_t flim = pc_uni(s.fast_limit);
    asm volatile(
      "s_mov_b32 %[flag], 0\n\t"
      "s_nop 1\n"
      "300:\n\t"
      "s_lshr_b32 %[c], %[i], %[sh]\n\t"
      "s_add_u32 %[c], %[base], %[c]\n\t"
      "v_cmp_eq_u32_e64 %[mask], %[c], %[lane]\n\t"
      "v_readlane_b32 %[st], %[grp], %[c]\n\t"
      PC_ASM_HEAD_Q
      "s_nop 1\n\t"
      "v_readlane_b32 %[row], %[tl], %[st]\n\t"
      "s_lshr_b32 %[b], %[st], 16\n\t"
      "s_nop 0\n\t"
      "v_bfe_u32 %[vl], %[row], %[vt], 8\n\t"
      "v_mad_i32_i24 %[R], %[vl], %[m128], %[R]\n\t"
      "v_cmp_lt_u32_e32 vcc, %[val], %[R]\n\t"
      "s_cbranch_vccz 301f\n\t"
      "v_pk_sub_u16 %[vn], %[st], 1 clamp\n\t"
      "v_cmp_gt_u32_e32 vcc, 0x8000, %[R]\n\t"
      "s_cbranch_vccnz 302f\n"
      "304:\n\t"
      "v_cndmask_b32_e64 %[grp], %[grp], %[vn], %[mask]\n\t"
      "s_cmp_eq_u32 %[b], 0\n\t"
      "s_cbranch_scc1 390f\n\t"                  // a 0 bin ends the prefix
      "s_add_u32 %[i], %[i], 1\n\t"
      "s_cmp_lt_u32 %[i], %[mx]\n\t"
      "s_cbranch_scc1 300b\n\t"
      "s_branch 390f\n"
      "301:\n\t"
      PC_ASM_LPS("s_xor_b32 %[b], %[b], 1\n\t")
      "s_branch 303f\n"
      "302:\n\t"
      "v_lshlrev_b32 %[R], 1, %[R]\n\t"
      "v_lshlrev_b32 %[val], 1, %[val]\n\t"
      "v_add_u32 %[bits], 1, %[bits]\n"
      "303:\n\t"
      "v_cmp_lt_i32_e32 vcc, -1, %[bits]\n\t"
      "s_cbranch_vccz 304b\n\t"
      PC_ASM_REFILL("305f")
      "s_branch 304b\n"
      "305:\n\t"                                   // slow refill: finish this bin's bookkeeping, leave with flag = 1 (+ 2 when the prefix is complete)
      "v_cndmask_b32_e64 %[grp], %[grp], %[vn], %[mask]\n\t"
      "s_mov_b32 %[flag], 1\n\t"
      "s_cmp_eq_u32 %[b], 0\n\t"
      "s_cbranch_scc1 306f\n\t"
      "s_add_u32 %[i], %[i], 1\n\t"
      "s_cmp_lt_u32 %[i], %[mx]\n\t"
      "s_cbranch_scc1 390f\n"
      "306:\n\t"
      "s_mov_b32 %[flag], 3\n"
      "390:\n\t"
      : [grp] "+v"(grp), [R] "+v"(s.range), [val] "+v"(s.value), [bits] "+v"(s.bits_needed), [pos] "+s"(pos), [i] "+s"(i),
        [flag] "=&s"(flag), [c] "=&s"(c), [st] "=&s"(st), [row] "=&s"(row), [b] "=&s"(b), [mask] "=&s"(mask),
        [vt] "=&v"(vt), [vl] "=&v"(vl), [vn] "=&v"(vn)
      : [base] "s"(base), [sh] "s"(sh), [mx] "s"(mx), [tl] "v"(s.t_lps), [tn] "v"(s.t_next), [lane] "v"((uint32_t)threadIdx.x), [win] "v"(s.win), [flim] "s"(flim),
        [m128] "s"(0xffffff80u)
      : "vcc", "scc");
    s.pos = pc_uni(pos);
    i = pc_uni(i);
    const uint32_t f = pc_uni(flag);
    if (__builtin_expect(f == 0u, 1)) break;
    refill_byte(s);
    if (f & 2u) break;
  }
  return (int)i;
}
