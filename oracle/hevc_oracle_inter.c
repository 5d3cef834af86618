/*
 * hevc_oracle_inter.c — CPU ORACLE, inter prediction part (test infrastructure, NOT product code; #included by hevc_oracle.c).
 *
 * Spec-literal restatement of what a P slice adds to the intra decoding process (ITU-T H.265): picture order count and reference
 * picture set (8.3.1, 8.3.2), reference picture list construction (8.3.4), prediction block availability (6.4.2), merge mode
 * (8.5.3.2.2 - 8.5.3.2.5, spatial and zero candidates), motion vector prediction (8.5.3.2.6 - 8.5.3.2.8, spatial candidates with
 * scaling), fractional sample interpolation (8.5.3.3.3) and the default weighted sample prediction (8.5.3.3.4.2).  It stands in for
 * libde265's handling of the samples libheif's sequence tracks push through the same plugin (libheif/sequences/track_visual.cc:200-280,
 * libheif/plugins/decoder_libde265.cc:360, :417-419).
 *
 * Scope (what the HIP path implements too): P slices, uni-prediction from list 0 with any number of short-term reference pictures,
 * all partition modes incl. AMP, skip / merge / AMVP, parallel merge level, 4:0:0 and 4:2:0, 8 - 12 bit.  Refused loudly: B slices,
 * temporal motion vector prediction, weighted prediction, long-term reference pictures, constrained intra prediction in P slices.
 * PARITY: unpinned - no fixture of the reference holds inter-coded pictures; the generator's lossless round trips pin the syntax.
 */

/* ---- 8.3.1 picture order count, 8.3.2 reference picture set, 8.3.4 RefPicList0 ---------------------------------------------------- */
typedef struct {
  int num_neg, num_pos;
  int delta_s0[17], delta_s1[17];
  uint8_t used_s0[17], used_s1[17];
} StRps;

static int dpb_find(Dec* d, int poc)
{
  for (int i = 0; i < d->n_dpb; i++) if (d->dpb[i].valid && d->dpb[i].poc == poc) return i;
  return -1;
}

static void dpb_free_entry(RefPic* r)
{
  for (int c = 0; c < 3; c++) { free(r->plane[c]); r->plane[c] = NULL; }
  r->valid = 0;
}

/* at the first slice segment of a picture: POC of the picture, then the RPS decides which pictures stay and which one(s) list 0 holds */
static void inter_begin_picture(Dec* d, int nal_type, int poc_lsb, const StRps* rps)
{
  const SPS* s = d->s;
  int irap = nal_type >= 16 && nal_type <= 23, idr = nal_type == 19 || nal_type == 20;
  if (idr || (irap && d->first_picture)) { d->poc = idr ? 0 : poc_lsb; d->prev_tid0_lsb = idr ? 0 : poc_lsb; d->prev_tid0_msb = 0; }
  else {
    int MaxLsb = 1 << s->log2_max_poc_lsb, msb;
    if (poc_lsb < d->prev_tid0_lsb && d->prev_tid0_lsb - poc_lsb >= MaxLsb / 2) msb = d->prev_tid0_msb + MaxLsb;
    else if (poc_lsb > d->prev_tid0_lsb && poc_lsb - d->prev_tid0_lsb > MaxLsb / 2) msb = d->prev_tid0_msb - MaxLsb;
    else msb = d->prev_tid0_msb;
    d->poc = msb + poc_lsb;
    /* prevTid0Pic: TemporalId 0 and not RASL / RADL / a sub-layer non-reference picture (here: every odd nal_unit_type below 16, and IRAPs) */
    if (irap || (nal_type <= 9 && (nal_type & 1))) { d->prev_tid0_lsb = poc_lsb; d->prev_tid0_msb = msb; }
  }
  d->first_picture = 0;
  /* 8.3.2: pictures that are in no subset of the RPS are no longer "used for reference" (IDR: none is) */
  d->n_st_curr_before = d->n_st_curr_after = 0;
  uint8_t keep[MAX_DPB]; memset(keep, 0, sizeof(keep));
  if (!idr) {
    for (int i = 0; i < rps->num_neg; i++) {
      int k = dpb_find(d, d->poc + rps->delta_s0[i]);
      if (k >= 0) keep[k] = 1;
      if (rps->used_s0[i]) { if (k < 0) fail(d, "reference picture with POC %d is missing", d->poc + rps->delta_s0[i]); d->st_curr_before[d->n_st_curr_before++] = k; }
    }
    for (int i = 0; i < rps->num_pos; i++) {
      int k = dpb_find(d, d->poc + rps->delta_s1[i]);
      if (k >= 0) keep[k] = 1;
      if (rps->used_s1[i]) { if (k < 0) fail(d, "reference picture with POC %d is missing", d->poc + rps->delta_s1[i]); d->st_curr_after[d->n_st_curr_after++] = k; }
    }
  }
  for (int i = 0; i < d->n_dpb; i++) if (d->dpb[i].valid && !keep[i]) dpb_free_entry(&d->dpb[i]);
}

/* 8.3.4 (P slices): RefPicListTemp0 = StCurrBefore, StCurrAfter (no long-term pictures here), repeated; optional list_entry_l0 */
static void build_ref_list0(Dec* d, SliceHdr* h, const int* list_entry /* NULL: no modification */)
{
  int total = d->n_st_curr_before + d->n_st_curr_after;
  if (total == 0) fail(d, "P slice without a reference picture");
  int temp[32], n = 0, want = Max(h->num_ref_idx_l0_active, total);
  while (n < want) {
    for (int i = 0; i < d->n_st_curr_before && n < want; i++) temp[n++] = d->st_curr_before[i];
    for (int i = 0; i < d->n_st_curr_after && n < want; i++) temp[n++] = d->st_curr_after[i];
  }
  for (int i = 0; i < h->num_ref_idx_l0_active; i++) {
    int e = list_entry ? list_entry[i] : i;
    if (e < 0 || e >= want) fail(d, "list_entry_l0 out of range");
    h->ref_list0[i] = (int8_t)temp[e];
    h->ref_poc0[i] = d->dpb[temp[e]].poc;
  }
}

/* ---- 6.4.2 prediction block availability ------------------------------------------------------------------------------------------- */
typedef struct { int xCb, yCb, nCbS, xPb, yPb, nPbW, nPbH, partIdx; } PbGeom;

static int pb_available(Dec* d, const PbGeom* g, int xN, int yN)
{
  int sameCb = (g->xCb <= xN && g->yCb <= yN && g->xCb + g->nCbS > xN && g->yCb + g->nCbS > yN);
  int av;
  if (!sameCb) av = available_z(d, g->xPb, g->yPb, xN, yN);
  else if ((g->nPbW << 1) == g->nCbS && (g->nPbH << 1) == g->nCbS && g->partIdx == 1 && g->yCb + g->nPbH <= yN && g->xCb + g->nPbW > xN) av = 0;
  else av = 1;
  if (av && d->m_pred[(yN >> 2) * d->mw + (xN >> 2)] == 0) av = 0;    /* MODE_INTRA */
  return av;
}

typedef struct { int mv[2]; int ref_idx; } Motion;   /* P slices: predFlagL0 is 1 for every inter block */

static Motion motion_at(Dec* d, int x, int y)
{
  int idx = (y >> 2) * d->mw + (x >> 2);
  Motion m = {{d->mf_mv[2 * idx], d->mf_mv[2 * idx + 1]}, d->mf_ref[idx]};
  return m;
}
static int same_motion(const Motion* a, const Motion* b) { return a->mv[0] == b->mv[0] && a->mv[1] == b->mv[1] && a->ref_idx == b->ref_idx; }

/* ---- 8.5.3.2.2 - 8.5.3.2.5 merge mode (spatial candidates, then zero candidates; no temporal candidate: TMVP is refused) ---------- */
static Motion derive_merge(Dec* d, const PbGeom* g0, int PartMode, int merge_idx)
{
  const PPS* p = d->p;
  PbGeom g = *g0;
  int plevel = p->log2_parallel_merge_level;
  if (plevel > 2 && g.nCbS == 8) { g.xPb = g.xCb; g.yPb = g.yCb; g.nPbW = g.nPbH = g.nCbS; g.partIdx = 0; PartMode = PART_2Nx2N; }   /* singleMCLFlag */
  int xPb = g.xPb, yPb = g.yPb, nPbW = g.nPbW, nPbH = g.nPbH;
#define SAME_MER(xn, yn) ((xPb >> plevel) == ((xn) >> plevel) && (yPb >> plevel) == ((yn) >> plevel))
  Motion cand[6]; int n = 0;
  Motion A1 = {{0, 0}, 0}, B1 = A1, B0 = A1, A0 = A1, B2 = A1;
  /* availableN: 6.4.2 availability minus the merge-estimation-region / second-partition exclusions; availableFlagN: after the pruning against the
     neighbours compared with.  The comparisons read availableN of the other candidate, NOT its flag (B0 is compared with a B1 that was itself
     pruned against A1); only the "all four present" rule of B2 counts flags */
  int xA1 = xPb - 1, yA1 = yPb + nPbH - 1;
  int avA1 = pb_available(d, &g, xA1, yA1);
  if (SAME_MER(xA1, yA1) || (g.partIdx == 1 && (PartMode == PART_Nx2N || PartMode == PART_nLx2N || PartMode == PART_nRx2N))) avA1 = 0;
  int fA1 = avA1;
  if (avA1) A1 = motion_at(d, xA1, yA1);
  if (fA1) cand[n++] = A1;
  int xB1 = xPb + nPbW - 1, yB1 = yPb - 1;
  int avB1 = pb_available(d, &g, xB1, yB1);
  if (SAME_MER(xB1, yB1) || (g.partIdx == 1 && (PartMode == PART_2NxN || PartMode == PART_2NxnU || PartMode == PART_2NxnD))) avB1 = 0;
  if (avB1) B1 = motion_at(d, xB1, yB1);
  int fB1 = avB1 && !(avA1 && same_motion(&A1, &B1));
  if (fB1) cand[n++] = B1;
  int xB0 = xPb + nPbW, yB0 = yPb - 1;
  int avB0 = pb_available(d, &g, xB0, yB0);
  if (SAME_MER(xB0, yB0)) avB0 = 0;
  if (avB0) B0 = motion_at(d, xB0, yB0);
  int fB0 = avB0 && !(avB1 && same_motion(&B1, &B0));
  if (fB0) cand[n++] = B0;
  int xA0 = xPb - 1, yA0 = yPb + nPbH;
  int avA0 = pb_available(d, &g, xA0, yA0);
  if (SAME_MER(xA0, yA0)) avA0 = 0;
  if (avA0) A0 = motion_at(d, xA0, yA0);
  int fA0 = avA0 && !(avA1 && same_motion(&A1, &A0));
  if (fA0) cand[n++] = A0;
  int xB2 = xPb - 1, yB2 = yPb - 1;
  int avB2 = pb_available(d, &g, xB2, yB2);
  if (SAME_MER(xB2, yB2)) avB2 = 0;
  if (avB2) B2 = motion_at(d, xB2, yB2);
  int fB2 = avB2 && !(avA1 && same_motion(&A1, &B2)) && !(avB1 && same_motion(&B1, &B2)) && fA0 + fA1 + fB0 + fB1 != 4;
  if (fB2) cand[n++] = B2;
#undef SAME_MER
  if (n > d->sh->max_num_merge_cand) n = d->sh->max_num_merge_cand;   /* (slice-level MaxNumMergeCand caps the list: 8.5.3.2.2 step 8 onward fills, never trims
                                                                          spatial candidates below five - the cap only matters when merge_idx addresses them) */
  /* 8.5.3.2.5 zero candidates: refIdxL0 = zeroIdx while below the number of reference pictures, then 0 */
  int zeroIdx = 0;
  while (n < d->sh->max_num_merge_cand) {
    Motion z = {{0, 0}, zeroIdx < d->sh->num_ref_idx_l0_active ? zeroIdx : 0};
    cand[n++] = z; zeroIdx++;
  }
  if (merge_idx >= n) fail(d, "merge_idx out of range");
  return cand[merge_idx];
}

/* ---- 8.5.3.2.6 - 8.5.3.2.8 luma motion vector prediction (spatial candidates, no temporal one) ----------------------------------- */
static void scale_mv(int* mv, int td, int tb)
{
  td = Clip3(-128, 127, td); tb = Clip3(-128, 127, tb);
  int tx = (16384 + (Abs(td) >> 1)) / td;
  int dsf = Clip3(-4096, 4095, (tb * tx + 32) >> 6);
  for (int k = 0; k < 2; k++) {
    int v = dsf * mv[k];
    mv[k] = Clip3(-32768, 32767, (v < 0 ? -1 : 1) * ((Abs(v) + 127) >> 8));
  }
}

static void derive_mvp(Dec* d, const PbGeom* g, int refIdx, int mvp_flag, int* mvp)
{
  int xPb = g->xPb, yPb = g->yPb, nPbW = g->nPbW, nPbH = g->nPbH;
  int curPoc = d->poc, tgtPoc = d->sh->ref_poc0[refIdx];
  int xA[2] = {xPb - 1, xPb - 1}, yA[2] = {yPb + nPbH, yPb + nPbH - 1};
  int avA[2];
  for (int k = 0; k < 2; k++) avA[k] = pb_available(d, g, xA[k], yA[k]);
  int isScaled = avA[0] || avA[1];
  int flagA = 0, mvA[2] = {0, 0};
  for (int k = 0; k < 2 && !flagA; k++)
    if (avA[k]) {
      Motion m = motion_at(d, xA[k], yA[k]);
      if (d->mf_poc[(yA[k] >> 2) * d->mw + (xA[k] >> 2)] == tgtPoc) { flagA = 1; mvA[0] = m.mv[0]; mvA[1] = m.mv[1]; }
    }
  for (int k = 0; k < 2 && !flagA; k++)
    if (avA[k]) {
      Motion m = motion_at(d, xA[k], yA[k]);
      int nbPoc = d->mf_poc[(yA[k] >> 2) * d->mw + (xA[k] >> 2)];
      flagA = 1; mvA[0] = m.mv[0]; mvA[1] = m.mv[1];
      if (nbPoc != tgtPoc) scale_mv(mvA, curPoc - nbPoc, curPoc - tgtPoc);
    }
  int xB[3] = {xPb + nPbW, xPb + nPbW - 1, xPb - 1}, yB[3] = {yPb - 1, yPb - 1, yPb - 1};
  int avB[3];
  for (int k = 0; k < 3; k++) avB[k] = pb_available(d, g, xB[k], yB[k]);
  int flagB = 0, mvB[2] = {0, 0};
  for (int k = 0; k < 3 && !flagB; k++)
    if (avB[k] && d->mf_poc[(yB[k] >> 2) * d->mw + (xB[k] >> 2)] == tgtPoc) {
      Motion m = motion_at(d, xB[k], yB[k]);
      flagB = 1; mvB[0] = m.mv[0]; mvB[1] = m.mv[1];
    }
  if (!isScaled && flagB) { flagA = 1; mvA[0] = mvB[0]; mvA[1] = mvB[1]; }
  if (!isScaled) {
    flagB = 0;
    for (int k = 0; k < 3 && !flagB; k++)
      if (avB[k]) {
        Motion m = motion_at(d, xB[k], yB[k]);
        int nbPoc = d->mf_poc[(yB[k] >> 2) * d->mw + (xB[k] >> 2)];
        flagB = 1; mvB[0] = m.mv[0]; mvB[1] = m.mv[1];
        if (nbPoc != tgtPoc) scale_mv(mvB, curPoc - nbPoc, curPoc - tgtPoc);
      }
  }
  int list[3][2], n = 0;
  if (flagA) { list[n][0] = mvA[0]; list[n][1] = mvA[1]; n++; }
  if (flagB && !(flagA && mvA[0] == mvB[0] && mvA[1] == mvB[1])) { list[n][0] = mvB[0]; list[n][1] = mvB[1]; n++; }
  while (n < 2) { list[n][0] = 0; list[n][1] = 0; n++; }
  mvp[0] = list[mvp_flag][0]; mvp[1] = list[mvp_flag][1];
}

/* ---- 8.5.3.3 decoding process for inter sample prediction (uni-prediction from list 0, default weights) --------------------------- */
static const int8_t fL[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t fC[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

static void mc_block(Dec* d, const RefPic* ref, int cIdx, int xP, int yP, int w, int h, int mvx, int mvy)
{
  /* (xP, yP), w, h in samples of component cIdx; mv in quarter luma samples = eighth chroma samples for 4:2:0 */
  const SPS* s = d->s;
  int W = cIdx ? d->Wc : d->W, H = cIdx ? d->Hc : d->H;
  int bitDepth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
  int shift1 = Min(4, bitDepth - 8), shift2 = 6, shift3 = Max(2, 14 - bitDepth);
  int fbits = cIdx ? 3 : 2, taps = cIdx ? 4 : 8, before = cIdx ? 1 : 3;
  int xFrac = mvx & ((1 << fbits) - 1), yFrac = mvy & ((1 << fbits) - 1);
  int xInt0 = xP + (mvx >> fbits), yInt0 = yP + (mvy >> fbits);
  const uint16_t* rp = ref->plane[cIdx];
  uint16_t* dst = d->rec[cIdx];
  int maxv = (1 << bitDepth) - 1;
  int wshift = 14 - bitDepth, woff = wshift > 0 ? 1 << (wshift - 1) : 0;
#define REF(x, y) ((int)rp[(size_t)Clip3(0, H - 1, (y)) * W + Clip3(0, W - 1, (x))])
#define COEF_H(i) (cIdx ? fC[xFrac][i] : fL[xFrac][i])
#define COEF_V(i) (cIdx ? fC[yFrac][i] : fL[yFrac][i])
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int xi = xInt0 + x, yi = yInt0 + y, v;
      if (!xFrac && !yFrac) v = REF(xi, yi) << shift3;
      else if (!yFrac) { int a = 0; for (int i = 0; i < taps; i++) a += COEF_H(i) * REF(xi + i - before, yi); v = a >> shift1; }
      else if (!xFrac) { int a = 0; for (int i = 0; i < taps; i++) a += COEF_V(i) * REF(xi, yi + i - before); v = a >> shift1; }
      else {
        int a = 0;
        for (int j = 0; j < taps; j++) {
          int t = 0;
          for (int i = 0; i < taps; i++) t += COEF_H(i) * REF(xi + i - before, yi + j - before);
          a += COEF_V(j) * (t >> shift1);
        }
        v = a >> shift2;
      }
      dst[(size_t)(yP + y) * W + xP + x] = (uint16_t)Clip3(0, maxv, (v + woff) >> wshift);   /* 8.5.3.3.4.2, predFlagL0 only */
    }
#undef REF
#undef COEF_H
#undef COEF_V
}

static void predict_pu(Dec* d, int xPb, int yPb, int nPbW, int nPbH, const Motion* m)
{
  const RefPic* ref = &d->dpb[d->sh->ref_list0[m->ref_idx]];
  mc_block(d, ref, 0, xPb, yPb, nPbW, nPbH, m->mv[0], m->mv[1]);
  if (d->s->chroma_format_idc == 1) {
    mc_block(d, ref, 1, xPb / 2, yPb / 2, nPbW / 2, nPbH / 2, m->mv[0], m->mv[1]);
    mc_block(d, ref, 2, xPb / 2, yPb / 2, nPbW / 2, nPbH / 2, m->mv[0], m->mv[1]);
  }
}

static void store_motion(Dec* d, int xPb, int yPb, int nPbW, int nPbH, const Motion* m)
{
  for (int y = yPb >> 2; y < (yPb + nPbH) >> 2; y++)
    for (int x = xPb >> 2; x < (xPb + nPbW) >> 2; x++) {
      int idx = y * d->mw + x;
      d->mf_mv[2 * idx] = (int16_t)m->mv[0]; d->mf_mv[2 * idx + 1] = (int16_t)m->mv[1];
      d->mf_ref[idx] = (int8_t)m->ref_idx;
      d->mf_poc[idx] = d->sh->ref_poc0[m->ref_idx];
    }
}

/* ---- 7.3.8.6 prediction_unit, 7.3.8.9 mvd_coding ------------------------------------------------------------------------------------ */
static int decode_egk_bypass(Dec* d, int k)
{
  int v = 0;
  while (decode_bypass(d)) { v += 1 << k; k++; if (k > 20) fail(d, "Exp-Golomb prefix too long"); }
  while (k--) v += decode_bypass(d) << k;
  return v;
}

static void parse_mvd(Dec* d, int* mvd)
{
  int gt0[2], gt1[2] = {0, 0};
  gt0[0] = decode_decision(d, CTX_MVD_GT0);
  gt0[1] = decode_decision(d, CTX_MVD_GT0);
  if (gt0[0]) gt1[0] = decode_decision(d, CTX_MVD_GT1);
  if (gt0[1]) gt1[1] = decode_decision(d, CTX_MVD_GT1);
  for (int k = 0; k < 2; k++) {
    mvd[k] = 0;
    if (gt0[k]) {
      int a = 1;
      if (gt1[k]) a = decode_egk_bypass(d, 1) + 2;
      if (a > 32768) fail(d, "mvd out of range");
      mvd[k] = decode_bypass(d) ? -a : a;
    }
  }
}

/* returns merge_flag */
static int prediction_unit(Dec* d, const PbGeom* g, int PartMode, int cu_skip)
{
  const SliceHdr* h = d->sh;
  int merge_flag = 1, merge_idx = 0;
  Motion m;
  if (!cu_skip) merge_flag = decode_decision(d, CTX_MERGE_FLAG);
  if (merge_flag) {
    if (h->max_num_merge_cand > 1) {   /* TR, cMax = MaxNumMergeCand - 1: first bin context coded, the others bypass */
      if (decode_decision(d, CTX_MERGE_IDX)) { merge_idx = 1; while (merge_idx < h->max_num_merge_cand - 1 && decode_bypass(d)) merge_idx++; }
    }
    m = derive_merge(d, g, PartMode, merge_idx);
  } else {
    int ref_idx = 0;
    if (h->num_ref_idx_l0_active > 1) {   /* TR, cMax = num_ref_idx_l0_active_minus1: bins 0 and 1 context coded, the rest bypass */
      int cmax = h->num_ref_idx_l0_active - 1;
      while (ref_idx < cmax) {
        int b = ref_idx < 2 ? decode_decision(d, CTX_REF_IDX + ref_idx) : decode_bypass(d);
        if (!b) break;
        ref_idx++;
      }
    }
    int mvd[2], mvp[2];
    parse_mvd(d, mvd);
    int mvp_flag = decode_decision(d, CTX_MVP_FLAG);
    derive_mvp(d, g, ref_idx, mvp_flag, mvp);
    for (int k = 0; k < 2; k++) {   /* 8.5.3.2.1: uLX = (mvpLX + mvdLX + 2^16) % 2^16, wrapped into 16 bits */
      int u = (mvp[k] + mvd[k] + 65536) & 65535;
      m.mv[k] = u >= 32768 ? u - 65536 : u;
    }
    m.ref_idx = ref_idx;
  }
  store_motion(d, g->xPb, g->yPb, g->nPbW, g->nPbH, &m);
  predict_pu(d, g->xPb, g->yPb, g->nPbW, g->nPbH, &m);
  return merge_flag;
}

/* partitions of a coding unit (7.4.9.5, Table 7-10): geometry of partition partIdx */
static int part_geometry(int PartMode, int xCb, int yCb, int nCbS, int partIdx, PbGeom* g)
{
  int q = nCbS / 4, hf = nCbS / 2;
  int x = 0, y = 0, w = nCbS, h = nCbS, n = 1;
  switch (PartMode) {
    case PART_2Nx2N: break;
    case PART_2NxN: n = 2; h = hf; y = partIdx * hf; break;
    case PART_Nx2N: n = 2; w = hf; x = partIdx * hf; break;
    case PART_NxN: n = 4; w = h = hf; x = (partIdx & 1) * hf; y = (partIdx >> 1) * hf; break;
    case PART_2NxnU: n = 2; h = partIdx ? nCbS - q : q; y = partIdx ? q : 0; break;
    case PART_2NxnD: n = 2; h = partIdx ? q : nCbS - q; y = partIdx ? nCbS - q : 0; break;
    case PART_nLx2N: n = 2; w = partIdx ? nCbS - q : q; x = partIdx ? q : 0; break;
    case PART_nRx2N: n = 2; w = partIdx ? q : nCbS - q; x = partIdx ? nCbS - q : 0; break;
  }
  g->xCb = xCb; g->yCb = yCb; g->nCbS = nCbS; g->xPb = xCb + x; g->yPb = yCb + y; g->nPbW = w; g->nPbH = h; g->partIdx = partIdx;
  return n;
}

/* part_mode of an inter coding unit (9.3.3.7 binarisation, Table 9-43; context assignment Table 9-46) */
static int parse_part_mode_inter(Dec* d, int log2CbSize)
{
  const SPS* s = d->s;
  if (decode_decision(d, CTX_PART_MODE)) return PART_2Nx2N;
  if (log2CbSize == s->log2_min_cb) {
    if (decode_decision(d, CTX_PART_MODE_INTER + 0)) return PART_2NxN;
    if (log2CbSize == 3) return PART_Nx2N;
    if (decode_decision(d, CTX_PART_MODE_INTER + 1)) return PART_Nx2N;
    return PART_NxN;
  }
  if (!s->amp_enabled_flag) return decode_decision(d, CTX_PART_MODE_INTER + 0) ? PART_2NxN : PART_Nx2N;
  if (decode_decision(d, CTX_PART_MODE_INTER + 0)) {
    if (decode_decision(d, CTX_PART_MODE_INTER + 2)) return PART_2NxN;
    return decode_bypass(d) ? PART_2NxnD : PART_2NxnU;
  }
  if (decode_decision(d, CTX_PART_MODE_INTER + 2)) return PART_Nx2N;
  return decode_bypass(d) ? PART_nRx2N : PART_nLx2N;
}

/* prediction block edges inside the coding unit are deblocking edges too (8.7.2.3); only those on the 8x8 grid get filtered */
static void mark_pu_edges(Dec* d, int xCb, int yCb, int nCbS, int PartMode)
{
  if (d->sh->slice_deblocking_filter_disabled_flag) return;
  int vx = -1, hy = -1, q = nCbS / 4;
  switch (PartMode) {
    case PART_2NxN: hy = nCbS / 2; break;
    case PART_Nx2N: vx = nCbS / 2; break;
    case PART_NxN: vx = hy = nCbS / 2; break;
    case PART_2NxnU: hy = q; break;
    case PART_2NxnD: hy = nCbS - q; break;
    case PART_nLx2N: vx = q; break;
    case PART_nRx2N: vx = nCbS - q; break;
    default: break;
  }
  if (vx >= 0) for (int y = 0; y < nCbS; y += 4) d->m_flags[((yCb + y) >> 2) * d->mw + ((xCb + vx) >> 2)] |= 0x20;
  if (hy >= 0) for (int x = 0; x < nCbS; x += 4) d->m_flags[((yCb + hy) >> 2) * d->mw + ((xCb + x) >> 2)] |= 0x40;
}

/* a coding unit without a transform tree (cu_skip_flag, or rqt_root_cbf 0): for the maps it is covered by transform blocks of
   min(CB size, 32) without coefficients, whose only deblocking edges are the coding unit's own left / top edge */
static void mark_cu_no_residual(Dec* d, CuCtx* cu, int x0, int y0, int log2CbSize)
{
  mark_tu(d, cu, x0, y0, log2CbSize, 0, 0, 0);
  if (log2CbSize > 5) {
    int nu = 1 << (log2CbSize - 2);
    for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) d->m_log2_tb[((y0 >> 2) + j) * d->mw + (x0 >> 2) + i] = 5;
  }
}

/* 8.7.2.4 boundary filtering strength of the edge between units idxP and idxQ (luma sample position (x, y) of q0; dir 0: vertical edge) */
static int edge_bs(Dec* d, int idxP, int idxQ, int x, int y, int dir)
{
  if (!d->m_pred) return 2;                                     /* intra picture */
  if (d->m_pred[idxP] == 0 || d->m_pred[idxQ] == 0) return 2;
  int tbq = 1 << d->m_log2_tb[idxQ];
  int tu_edge = dir == 0 ? (x & (tbq - 1)) == 0 : (y & (tbq - 1)) == 0;      /* transform blocks are aligned to their size */
  if (tu_edge && ((d->m_flags[idxP] | d->m_flags[idxQ]) & 1)) return 1;     /* a block with non-zero luma coefficient levels */
  if (d->mf_poc[idxP] != d->mf_poc[idxQ]) return 1;                          /* different reference PICTURES (not indices) */
  if (Abs(d->mf_mv[2 * idxP] - d->mf_mv[2 * idxQ]) >= 4 || Abs(d->mf_mv[2 * idxP + 1] - d->mf_mv[2 * idxQ + 1]) >= 4) return 1;
  return 0;
}
