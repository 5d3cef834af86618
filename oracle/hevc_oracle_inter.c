/*
 * hevc_oracle_inter.c — CPU ORACLE, inter prediction part (test infrastructure, NOT product code; #included by hevc_oracle.c).
 *
 * Spec-literal restatement of what P and B slices add to the intra decoding process (ITU-T H.265): picture order count and reference
 * picture set (8.3.1, 8.3.2), reference picture list construction (8.3.4), prediction block availability (6.4.2), merge mode
 * (8.5.3.2.2 - 8.5.3.2.5: spatial, temporal, combined bi-predictive and zero candidates), motion vector prediction (8.5.3.2.6 - 8.5.3.2.9:
 * spatial candidates with scaling, the collocated candidate), fractional sample interpolation (8.5.3.3.3), default and explicit
 * weighted sample prediction (8.5.3.3.4.2, 8.5.3.3.4.3).  It stands in for libde265's handling of the samples libheif's sequence
 * tracks push through the same plugin (libheif/sequences/track_visual.cc:200-280, libheif/plugins/decoder_libde265.cc:360, :417-419) -
 * the streams libheif's own x265 plugin writes for its "lowdelay" (P pictures: TMVP, weighted prediction, several reference
 * pictures) and "unrestricted" (B pictures) GOP structures (libheif/plugins/encoder_x265.cc:875-888).
 *
 * Scope: P and B slices, short-term and long-term reference pictures (8.3.2; the motion vector rules of 8.5.3.2.7 / 8.5.3.2.9 for them), all
 * partition modes incl. AMP, skip / merge / AMVP, parallel merge level, TMVP, explicit weighted prediction, scaling lists and constrained
 * intra prediction in P / B pictures, 8 - 12 bit; 4:0:0 and 4:2:0 on the HIP path (this oracle also decodes 4:2:2 / 4:4:4 inter pictures).
 * PARITY: unpinned - no fixture of the reference holds inter-coded pictures; the generator's lossless round trips pin the syntax.
 */

/* ---- 8.3.1 picture order count, 8.3.2 reference picture set, 8.3.4 RefPicList0 ---------------------------------------------------- */
typedef struct {
  int num_neg, num_pos;
  int delta_s0[17], delta_s1[17];
  uint8_t used_s0[17], used_s1[17];
  /* long-term part of the slice header (7.3.6.1): PocLsbLt, UsedByCurrPicLt, delta_poc_msb_present_flag, DeltaPocMsbCycleLt (7-52: accumulated) */
  int num_lt;
  int lt_poc_lsb[33], lt_msb_cycle[33];
  uint8_t lt_used[33], lt_msb_present[33];
} StRps;

static int dpb_find(Dec* d, int poc)
{
  for (int i = 0; i < d->n_dpb; i++) if (d->dpb[i].valid && d->dpb[i].poc == poc) return i;
  return -1;
}

static void dpb_free_entry(RefPic* r)
{
  for (int c = 0; c < 3; c++) { free(r->plane[c]); r->plane[c] = NULL; }
  free(r->m_pred); free(r->mf_mv); free(r->mf_ref); free(r->mf_poc); free(r->mf_lt);
  r->m_pred = NULL; r->mf_mv = NULL; r->mf_ref = NULL; r->mf_poc = NULL; r->mf_lt = NULL;
  r->valid = 0; r->is_lt = 0;
}

/* the motion field of the picture that has just been decoded stays with it in the DPB: the collocated picture of later slices (8.5.3.2.8) */
static void dpb_store_motion(Dec* d, RefPic* r)
{
  size_t mn = (size_t)d->mw * d->mh;
  r->m_pred = (uint8_t*)xcalloc(d, mn, 1); r->mf_mv = (int16_t*)xcalloc(d, mn * 4, sizeof(int16_t));
  r->mf_ref = (int8_t*)xcalloc(d, mn * 2, 1); r->mf_poc = (int32_t*)xcalloc(d, mn * 2, sizeof(int32_t));
  memcpy(r->m_pred, d->m_pred, mn); memcpy(r->mf_mv, d->mf_mv, mn * 4 * sizeof(int16_t));
  memcpy(r->mf_ref, d->mf_ref, mn * 2); memcpy(r->mf_poc, d->mf_poc, mn * 2 * sizeof(int32_t));
  r->mf_lt = (uint8_t*)xcalloc(d, mn * 2, 1); memcpy(r->mf_lt, d->mf_lt, mn * 2);
}

/* at the first slice segment of a picture: POC of the picture, then the RPS decides which pictures stay and which one(s) list 0 holds */
static void inter_begin_picture(Dec* d, int nal_type, int temporal_id, int poc_lsb, const StRps* rps)
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
    /* prevTid0Pic (8.3.1): TemporalId 0 and not a RASL / RADL / sub-layer non-reference picture: TRAIL_R 1, TSA_R 3, STSA_R 5 and the IRAPs */
    if (temporal_id == 0 && (irap || (nal_type <= 5 && (nal_type & 1)))) { d->prev_tid0_lsb = poc_lsb; d->prev_tid0_msb = msb; }
  }
  d->first_picture = 0;
  /* 8.3.2: first the long-term subsets - a candidate is ANY reference picture of the DPB, named by its POC LSBs or, with
     delta_poc_msb_present_flag, by its whole POC; everything in RefPicSetLtCurr / LtFoll is marked "used for long-term reference" -, then the
     short-term subsets among the pictures that are still short-term reference pictures; pictures in no subset are no longer reference pictures */
  d->n_st_curr_before = d->n_st_curr_after = d->n_lt_curr = 0;
  uint8_t keep[MAX_DPB]; memset(keep, 0, sizeof(keep));
  if (!idr) {
    int MaxLsb = 1 << s->log2_max_poc_lsb;
    for (int i = 0; i < rps->num_lt; i++) {
      int pocLt = rps->lt_poc_lsb[i], k = -1;
      if (rps->lt_msb_present[i]) pocLt += d->poc - rps->lt_msb_cycle[i] * MaxLsb - (d->poc & (MaxLsb - 1));
      for (int j = 0; j < d->n_dpb && k < 0; j++)
        if (d->dpb[j].valid && (rps->lt_msb_present[i] ? d->dpb[j].poc == pocLt : (d->dpb[j].poc & (MaxLsb - 1)) == pocLt)) k = j;
      if (k >= 0) { keep[k] = 1; d->dpb[k].is_lt = 1; }
      if (rps->lt_used[i]) { if (k < 0) fail(d, "long-term reference picture with POC (LSBs) %d is missing", pocLt); if (d->n_lt_curr < 16) d->lt_curr[d->n_lt_curr++] = k; }
    }
    for (int i = 0; i < rps->num_neg; i++) {
      int k = dpb_find(d, d->poc + rps->delta_s0[i]);
      if (k >= 0 && d->dpb[k].is_lt) k = -1;
      if (k >= 0) keep[k] = 1;
      if (rps->used_s0[i]) { if (k < 0) fail(d, "reference picture with POC %d is missing", d->poc + rps->delta_s0[i]); d->st_curr_before[d->n_st_curr_before++] = k; }
    }
    for (int i = 0; i < rps->num_pos; i++) {
      int k = dpb_find(d, d->poc + rps->delta_s1[i]);
      if (k >= 0 && d->dpb[k].is_lt) k = -1;
      if (k >= 0) keep[k] = 1;
      if (rps->used_s1[i]) { if (k < 0) fail(d, "reference picture with POC %d is missing", d->poc + rps->delta_s1[i]); d->st_curr_after[d->n_st_curr_after++] = k; }
    }
  }
  for (int i = 0; i < d->n_dpb; i++) if (d->dpb[i].valid && !keep[i]) dpb_free_entry(&d->dpb[i]);
}

/* 8.3.4: RefPicListTemp0 = StCurrBefore, StCurrAfter, LtCurr; RefPicListTemp1 = StCurrAfter, StCurrBefore, LtCurr, repeated up to the list
   size; optional list_entry_lX */
static void build_ref_list(Dec* d, SliceHdr* h, int X, const int* list_entry /* NULL: no modification */)
{
  int total = d->n_st_curr_before + d->n_st_curr_after + d->n_lt_curr;   /* NumPicTotalCurr */
  if (total == 0) fail(d, "P / B slice without a reference picture");
  int active = X ? h->num_ref_idx_l1_active : h->num_ref_idx_l0_active;
  int temp[32], n = 0, want = Max(active, total);
  const int* first = X ? d->st_curr_after : d->st_curr_before; int n_first = X ? d->n_st_curr_after : d->n_st_curr_before;
  const int* second = X ? d->st_curr_before : d->st_curr_after; int n_second = X ? d->n_st_curr_before : d->n_st_curr_after;
  while (n < want) {
    for (int i = 0; i < n_first && n < want; i++) temp[n++] = first[i];
    for (int i = 0; i < n_second && n < want; i++) temp[n++] = second[i];
    for (int i = 0; i < d->n_lt_curr && n < want; i++) temp[n++] = d->lt_curr[i];
  }
  for (int i = 0; i < active; i++) {
    int e = list_entry ? list_entry[i] : i;
    if (e < 0 || e >= want) fail(d, "list_entry_l%d out of range", X);
    h->ref_list[X][i] = (int8_t)temp[e];
    h->ref_poc[X][i] = d->dpb[temp[e]].poc;
    h->ref_is_lt[X][i] = (uint8_t)d->dpb[temp[e]].is_lt;
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

typedef struct { int mv[2][2]; int ref_idx[2]; int pred_flag[2]; } Motion;   /* [list][component]; ref_idx -1 when the list is not used */

static Motion motion_at(Dec* d, int x, int y)
{
  int idx = (y >> 2) * d->mw + (x >> 2);
  Motion m;
  for (int X = 0; X < 2; X++) {
    m.mv[X][0] = d->mf_mv[4 * idx + 2 * X]; m.mv[X][1] = d->mf_mv[4 * idx + 2 * X + 1];
    m.ref_idx[X] = d->mf_ref[2 * idx + X]; m.pred_flag[X] = m.ref_idx[X] >= 0;
  }
  return m;
}
static int same_motion(const Motion* a, const Motion* b)
{
  for (int X = 0; X < 2; X++) {
    if (a->pred_flag[X] != b->pred_flag[X]) return 0;
    if (a->pred_flag[X] && (a->mv[X][0] != b->mv[X][0] || a->mv[X][1] != b->mv[X][1] || a->ref_idx[X] != b->ref_idx[X])) return 0;
  }
  return 1;
}
static Motion motion_none(void) { Motion m; memset(&m, 0, sizeof(m)); m.ref_idx[0] = m.ref_idx[1] = -1; return m; }

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

/* ---- 8.5.3.2.8 temporal luma motion vector prediction, 8.5.3.2.9 collocated motion vectors ------------------------------------------ */
/* the motion the collocated picture stored for the unit covering (x, y) & ~15: the "compressed" motion field of 16x16 granularity */
static int collocated_mv(Dec* d, const RefPic* col, int xCol, int yCol, int refIdxLX, int X, int* mvLXCol)
{
  const SliceHdr* h = d->sh;
  int idx = (((yCol >> 4) << 4) >> 2) * d->mw + (((xCol >> 4) << 4) >> 2);
  if (!col->m_pred || col->m_pred[idx] == 0) return 0;                       /* colPb is intra coded */
  int f0 = col->mf_ref[2 * idx] >= 0, f1 = col->mf_ref[2 * idx + 1] >= 0, L;
  if (!f0) L = 1;
  else if (!f1) L = 0;
  else {
    /* both lists used: NoBackwardPredFlag (no reference picture of the current slice follows the current picture) takes the list the
       prediction is derived for, otherwise the list collocated_from_l0_flag names */
    int no_backward = 1;
    for (int Y = 0; Y < 2; Y++) {
      int n = Y ? h->num_ref_idx_l1_active : h->num_ref_idx_l0_active;
      if (Y == 1 && h->slice_type != 0) n = 0;
      for (int i = 0; i < n; i++) if (h->ref_poc[Y][i] > d->poc) no_backward = 0;
    }
    L = no_backward ? X : h->collocated_from_l0;
  }
  int mv[2] = {col->mf_mv[4 * idx + 2 * L], col->mf_mv[4 * idx + 2 * L + 1]};
  int colPocDiff = col->poc - col->mf_poc[2 * idx + L];
  int currPocDiff = d->poc - h->ref_poc[X][refIdxLX];
  /* LongTermRefPic(currPic, currPb, refIdxLX, LX) != LongTermRefPic(ColPic, colPb, refIdxCol, listCol): no candidate; a long-term target takes the
     collocated vector as it is (the flag of the collocated block is the marking its reference had when ColPic was decoded) */
  int colLt = col->mf_lt ? col->mf_lt[2 * idx + L] : 0, curLt = h->ref_is_lt[X][refIdxLX];
  if (colLt != curLt) return 0;
  if (!curLt && colPocDiff != currPocDiff) {
    if (colPocDiff == 0) return 0;   /* cannot happen in a conforming stream (a picture never references itself) */
    scale_mv(mv, colPocDiff, currPocDiff);
  }
  mvLXCol[0] = mv[0]; mvLXCol[1] = mv[1];
  return 1;
}

static int temporal_mv(Dec* d, int xPb, int yPb, int nPbW, int nPbH, int refIdxLX, int X, int* mvLXCol)
{
  const SliceHdr* h = d->sh; const SPS* s = d->s;
  if (!h->slice_temporal_mvp) return 0;
  const RefPic* col = &d->dpb[h->ref_list[(h->slice_type == 0 && !h->collocated_from_l0) ? 1 : 0][h->collocated_ref_idx]];
  int xBr = xPb + nPbW, yBr = yPb + nPbH;
  if ((yPb >> s->log2_ctb) == (yBr >> s->log2_ctb) && yBr < s->pic_height && xBr < s->pic_width &&
      collocated_mv(d, col, xBr, yBr, refIdxLX, X, mvLXCol)) return 1;
  return collocated_mv(d, col, xPb + (nPbW >> 1), yPb + (nPbH >> 1), refIdxLX, X, mvLXCol);
}

/* ---- 8.5.3.2.2 - 8.5.3.2.5 merge mode: spatial candidates, the temporal candidate, combined bi-predictive candidates (B), zero candidates */
static Motion derive_merge(Dec* d, const PbGeom* g0, int PartMode, int merge_idx)
{
  const PPS* p = d->p; const SliceHdr* h = d->sh;
  PbGeom g = *g0;
  const int nOrigPbW = g0->nPbW, nOrigPbH = g0->nPbH;
  int plevel = p->log2_parallel_merge_level;
  if (plevel > 2 && g.nCbS == 8) { g.xPb = g.xCb; g.yPb = g.yCb; g.nPbW = g.nPbH = g.nCbS; g.partIdx = 0; PartMode = PART_2Nx2N; }   /* singleMCLFlag */
  int xPb = g.xPb, yPb = g.yPb, nPbW = g.nPbW, nPbH = g.nPbH;
#define SAME_MER(xn, yn) ((xPb >> plevel) == ((xn) >> plevel) && (yPb >> plevel) == ((yn) >> plevel))
  Motion cand[8]; int n = 0;
  Motion A1 = motion_none(), B1 = A1, B0 = A1, A0 = A1, B2 = A1;
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
  /* the temporal candidate: refIdxLXCol = 0 in each list the slice has (8.5.3.2.2 steps 3 - 5) */
  if (h->slice_temporal_mvp) {
    Motion c = motion_none();
    for (int X = 0; X < (h->slice_type == 0 ? 2 : 1); X++)
      if (temporal_mv(d, xPb, yPb, nPbW, nPbH, 0, X, c.mv[X])) { c.pred_flag[X] = 1; c.ref_idx[X] = 0; }
    if (c.pred_flag[0] || c.pred_flag[1]) cand[n++] = c;
  }
  const int MaxNumMergeCand = h->max_num_merge_cand;
  if (n > MaxNumMergeCand) n = MaxNumMergeCand;   /* (merge_idx < MaxNumMergeCand: entries beyond it are never addressed, and the steps below only
                                                      run while the list is shorter) */
  /* 8.5.3.2.4 combined bi-predictive candidates (B slices) */
  if (h->slice_type == 0 && n > 1 && n < MaxNumMergeCand) {
    static const uint8_t l0Cand[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1Cand[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
    const int numOrig = n;
    for (int combIdx = 0; combIdx < numOrig * (numOrig - 1) && n < MaxNumMergeCand; combIdx++) {
      const Motion* a = &cand[l0Cand[combIdx]]; const Motion* b = &cand[l1Cand[combIdx]];
      if (a->pred_flag[0] && b->pred_flag[1] &&
          (h->ref_poc[0][a->ref_idx[0]] != h->ref_poc[1][b->ref_idx[1]] || a->mv[0][0] != b->mv[1][0] || a->mv[0][1] != b->mv[1][1])) {
        Motion c = motion_none();
        c.pred_flag[0] = c.pred_flag[1] = 1;
        c.ref_idx[0] = a->ref_idx[0]; c.mv[0][0] = a->mv[0][0]; c.mv[0][1] = a->mv[0][1];
        c.ref_idx[1] = b->ref_idx[1]; c.mv[1][0] = b->mv[1][0]; c.mv[1][1] = b->mv[1][1];
        cand[n++] = c;
      }
    }
  }
  /* 8.5.3.2.5 zero candidates: the reference index counts up while below the number of reference pictures (B: of the shorter list), then 0 */
  int numRefIdx = h->slice_type == 0 ? Min(h->num_ref_idx_l0_active, h->num_ref_idx_l1_active) : h->num_ref_idx_l0_active;
  int zeroIdx = 0;
  while (n < MaxNumMergeCand) {
    Motion z = motion_none();
    int r = zeroIdx < numRefIdx ? zeroIdx : 0;
    z.pred_flag[0] = 1; z.ref_idx[0] = r;
    if (h->slice_type == 0) { z.pred_flag[1] = 1; z.ref_idx[1] = r; }
    cand[n++] = z; zeroIdx++;
  }
  if (merge_idx >= n) fail(d, "merge_idx out of range");
  Motion m = cand[merge_idx];
  if (m.pred_flag[0] && m.pred_flag[1] && nOrigPbW + nOrigPbH == 12) { m.pred_flag[1] = 0; m.ref_idx[1] = -1; m.mv[1][0] = m.mv[1][1] = 0; }   /* 8x4 / 4x8: uni-prediction */
  return m;
}

/* ---- 8.5.3.2.6 - 8.5.3.2.7 luma motion vector prediction for list X: spatial candidates (the neighbour's own list X first, then its other
        list; unscaled when it points at the target picture, scaled otherwise), then the temporal candidate ------------------------------- */
static int nb_mv_same_poc(Dec* d, int x, int y, int X, int tgtPoc, int* mv)
{
  int idx = (y >> 2) * d->mw + (x >> 2);
  for (int k = 0; k < 2; k++) {
    int L = k ? 1 - X : X;
    if (d->mf_ref[2 * idx + L] >= 0 && d->mf_poc[2 * idx + L] == tgtPoc) { mv[0] = d->mf_mv[4 * idx + 2 * L]; mv[1] = d->mf_mv[4 * idx + 2 * L + 1]; return 1; }
  }
  return 0;
}
static int nb_mv_scaled(Dec* d, int x, int y, int X, int tgtPoc, int tgtLt, int* mv)
{
  int idx = (y >> 2) * d->mw + (x >> 2);
  for (int k = 0; k < 2; k++) {
    int L = k ? 1 - X : X;
    /* 8.5.3.2.7 (7): the neighbour's vector counts when its reference picture and the target are both long-term or both short-term pictures;
       it is scaled only between short-term pictures */
    if (d->mf_ref[2 * idx + L] >= 0 && d->mf_lt[2 * idx + L] == tgtLt) {
      mv[0] = d->mf_mv[4 * idx + 2 * L]; mv[1] = d->mf_mv[4 * idx + 2 * L + 1];
      int nbPoc = d->mf_poc[2 * idx + L];
      if (!tgtLt && nbPoc != tgtPoc) scale_mv(mv, d->poc - nbPoc, d->poc - tgtPoc);
      return 1;
    }
  }
  return 0;
}

static void derive_mvp(Dec* d, const PbGeom* g, int X, int refIdx, int mvp_flag, int* mvp)
{
  int xPb = g->xPb, yPb = g->yPb, nPbW = g->nPbW, nPbH = g->nPbH;
  int tgtPoc = d->sh->ref_poc[X][refIdx], tgtLt = d->sh->ref_is_lt[X][refIdx];
  int xA[2] = {xPb - 1, xPb - 1}, yA[2] = {yPb + nPbH, yPb + nPbH - 1};
  int avA[2];
  for (int k = 0; k < 2; k++) avA[k] = pb_available(d, g, xA[k], yA[k]);
  int isScaled = avA[0] || avA[1];
  int flagA = 0, mvA[2] = {0, 0};
  for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) flagA = nb_mv_same_poc(d, xA[k], yA[k], X, tgtPoc, mvA);
  for (int k = 0; k < 2 && !flagA; k++) if (avA[k]) flagA = nb_mv_scaled(d, xA[k], yA[k], X, tgtPoc, tgtLt, mvA);
  int xB[3] = {xPb + nPbW, xPb + nPbW - 1, xPb - 1}, yB[3] = {yPb - 1, yPb - 1, yPb - 1};
  int avB[3];
  for (int k = 0; k < 3; k++) avB[k] = pb_available(d, g, xB[k], yB[k]);
  int flagB = 0, mvB[2] = {0, 0};
  for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) flagB = nb_mv_same_poc(d, xB[k], yB[k], X, tgtPoc, mvB);
  if (!isScaled && flagB) { flagA = 1; mvA[0] = mvB[0]; mvA[1] = mvB[1]; }
  if (!isScaled) {
    flagB = 0;
    for (int k = 0; k < 3 && !flagB; k++) if (avB[k]) flagB = nb_mv_scaled(d, xB[k], yB[k], X, tgtPoc, tgtLt, mvB);
  }
  int list[3][2], n = 0;
  if (flagA) { list[n][0] = mvA[0]; list[n][1] = mvA[1]; n++; }
  if (flagB && !(flagA && mvA[0] == mvB[0] && mvA[1] == mvB[1])) { list[n][0] = mvB[0]; list[n][1] = mvB[1]; n++; }
  if (n < 2) {   /* the temporal candidate is only derived when the spatial ones left a place (8.5.3.2.6 step 2) */
    int mvCol[2];
    if (temporal_mv(d, xPb, yPb, nPbW, nPbH, refIdx, X, mvCol)) { list[n][0] = mvCol[0]; list[n][1] = mvCol[1]; n++; }
  }
  while (n < 2) { list[n][0] = 0; list[n][1] = 0; n++; }
  mvp[0] = list[mvp_flag][0]; mvp[1] = list[mvp_flag][1];
}

/* ---- 8.5.3.3 decoding process for inter sample prediction: 14-bit prediction sample arrays per list (8.5.3.3.3), then the default
        (8.5.3.3.4.2) or explicit (8.5.3.3.4.3) weighted sample prediction ------------------------------------------------------------------ */
static const int8_t fL[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t fC[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

static void mc_block(Dec* d, const RefPic* ref, int cIdx, int xP, int yP, int w, int h, int mvx, int mvy, int16_t* out /* w x h */)
{
  /* (xP, yP), w, h in samples of component cIdx; luma: mv in quarter samples; chroma: in eighth samples (the caller applies 8.5.3.2.10) */
  const SPS* s = d->s;
  int W = cIdx ? d->Wc : d->W, H = cIdx ? d->Hc : d->H;
  int bitDepth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
  int shift1 = Min(4, bitDepth - 8), shift2 = 6, shift3 = Max(2, 14 - bitDepth);
  int fbits = cIdx ? 3 : 2, taps = cIdx ? 4 : 8, before = cIdx ? 1 : 3;
  int xFrac = mvx & ((1 << fbits) - 1), yFrac = mvy & ((1 << fbits) - 1);
  int xInt0 = xP + (mvx >> fbits), yInt0 = yP + (mvy >> fbits);
  const uint16_t* rp = ref->plane[cIdx];
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
      out[y * w + x] = (int16_t)v;
    }
#undef REF
#undef COEF_H
#undef COEF_V
}

static void predict_component(Dec* d, int cIdx, int xP, int yP, int w, int h, const Motion* m)
{
  const SPS* s = d->s; const SliceHdr* sh = d->sh;
  int W = cIdx ? d->Wc : d->W;
  int bitDepth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma, maxv = (1 << bitDepth) - 1;
  int16_t* pred[2] = {NULL, NULL};
  for (int X = 0; X < 2; X++)
    if (m->pred_flag[X]) {
      pred[X] = (int16_t*)xcalloc(d, (size_t)w * h, sizeof(int16_t));
      /* 8.5.3.2.10: mvCLX = mvLX * 2 / SubWidthC (SubHeightC), in units of 1/8 chroma sample: the luma vector itself where the chroma plane is
         subsampled in that direction, twice it where it is not (4:2:2 vertically, 4:4:4 both ways: only the even eighths occur then) */
      int mvx = m->mv[X][0], mvy = m->mv[X][1];
      if (cIdx) { if (s->chroma_format_idc == 3) mvx *= 2; if (s->chroma_format_idc != 1) mvy *= 2; }
      mc_block(d, &d->dpb[sh->ref_list[X][m->ref_idx[X]]], cIdx, xP, yP, w, h, mvx, mvy, pred[X]);
    }
  uint16_t* dst = d->rec[cIdx];
  int shift1 = 14 - bitDepth;
  if (!sh->weighted) {   /* 8.5.3.3.4.2 */
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        int v;
        if (pred[0] && pred[1]) v = (pred[0][y * w + x] + pred[1][y * w + x] + (1 << shift1)) >> (shift1 + 1);
        else { const int16_t* q = pred[0] ? pred[0] : pred[1]; v = (q[y * w + x] + (shift1 > 0 ? 1 << (shift1 - 1) : 0)) >> shift1; }
        dst[(size_t)(yP + y) * W + xP + x] = (uint16_t)Clip3(0, maxv, v);
      }
  } else {               /* 8.5.3.3.4.3 */
    int log2WD = (cIdx ? sh->chroma_log2_wd : sh->luma_log2_wd) + shift1;
    int wgt[2] = {0, 0}, off[2] = {0, 0};
    for (int X = 0; X < 2; X++)
      if (m->pred_flag[X]) { wgt[X] = sh->wp_weight[X][m->ref_idx[X]][cIdx]; off[X] = sh->wp_offset[X][m->ref_idx[X]][cIdx] << (bitDepth - 8); }
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        int v;
        if (pred[0] && pred[1]) v = (pred[0][y * w + x] * wgt[0] + pred[1][y * w + x] * wgt[1] + ((off[0] + off[1] + 1) << log2WD)) >> (log2WD + 1);
        else {
          int X = pred[0] ? 0 : 1;
          int q = pred[X][y * w + x];
          v = log2WD >= 1 ? ((q * wgt[X] + (1 << (log2WD - 1))) >> log2WD) + off[X] : q * wgt[X] + off[X];
        }
        dst[(size_t)(yP + y) * W + xP + x] = (uint16_t)Clip3(0, maxv, v);
      }
  }
  free(pred[0]); free(pred[1]);
}

static void predict_pu(Dec* d, int xPb, int yPb, int nPbW, int nPbH, const Motion* m)
{
  predict_component(d, 0, xPb, yPb, nPbW, nPbH, m);
  if (d->s->chroma_format_idc) {
    const int sw = d->s->chroma_format_idc == 3 ? 1 : 2, shh = d->s->chroma_format_idc == 1 ? 2 : 1;   /* SubWidthC, SubHeightC */
    predict_component(d, 1, xPb / sw, yPb / shh, nPbW / sw, nPbH / shh, m);
    predict_component(d, 2, xPb / sw, yPb / shh, nPbW / sw, nPbH / shh, m);
  }
}

static void store_motion(Dec* d, int xPb, int yPb, int nPbW, int nPbH, const Motion* m)
{
  for (int y = yPb >> 2; y < (yPb + nPbH) >> 2; y++)
    for (int x = xPb >> 2; x < (xPb + nPbW) >> 2; x++) {
      int idx = y * d->mw + x;
      for (int X = 0; X < 2; X++) {
        int used = m->pred_flag[X];
        d->mf_mv[4 * idx + 2 * X] = (int16_t)(used ? m->mv[X][0] : 0); d->mf_mv[4 * idx + 2 * X + 1] = (int16_t)(used ? m->mv[X][1] : 0);
        d->mf_ref[2 * idx + X] = (int8_t)(used ? m->ref_idx[X] : -1);
        d->mf_poc[2 * idx + X] = used ? d->sh->ref_poc[X][m->ref_idx[X]] : 0;
        d->mf_lt[2 * idx + X] = (uint8_t)(used ? d->sh->ref_is_lt[X][m->ref_idx[X]] : 0);
      }
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
static int prediction_unit(Dec* d, const PbGeom* g, int PartMode, int cu_skip, int ctDepth)
{
  const SliceHdr* h = d->sh;
  int merge_flag = 1, merge_idx = 0;
  Motion m = motion_none();
  if (!cu_skip) merge_flag = decode_decision(d, CTX_MERGE_FLAG);
  if (merge_flag) {
    if (h->max_num_merge_cand > 1) {   /* TR, cMax = MaxNumMergeCand - 1: first bin context coded, the others bypass */
      if (decode_decision(d, CTX_MERGE_IDX)) { merge_idx = 1; while (merge_idx < h->max_num_merge_cand - 1 && decode_bypass(d)) merge_idx++; }
    }
    m = derive_merge(d, g, PartMode, merge_idx);
  } else {
    /* inter_pred_idc (9.3.3.8 / Table 9-41): PRED_L0 0, PRED_L1 1, PRED_BI 2; blocks of 8x4 / 4x8 cannot be bi-predicted */
    int idc = 0;
    if (h->slice_type == 0) {
      if (g->nPbW + g->nPbH != 12 && decode_decision(d, CTX_INTER_PRED_IDC + ctDepth)) idc = 2;
      else idc = decode_decision(d, CTX_INTER_PRED_IDC + 4);
    }
    int mvd[2][2] = {{0, 0}, {0, 0}}, ref_idx[2] = {-1, -1}, mvp_flag[2] = {0, 0};
    for (int X = 0; X < 2; X++) {
      if (!(idc == 2 || idc == X)) continue;
      int active = X ? h->num_ref_idx_l1_active : h->num_ref_idx_l0_active;
      ref_idx[X] = 0;
      if (active > 1) {   /* TR, cMax = num_ref_idx_lX_active_minus1: bins 0 and 1 context coded, the rest bypass */
        int cmax = active - 1, r = 0;
        while (r < cmax) {
          int b = r < 2 ? decode_decision(d, CTX_REF_IDX + r) : decode_bypass(d);
          if (!b) break;
          r++;
        }
        ref_idx[X] = r;
      }
      if (X == 1 && h->mvd_l1_zero_flag && idc == 2) { mvd[1][0] = mvd[1][1] = 0; }
      else parse_mvd(d, mvd[X]);
      mvp_flag[X] = decode_decision(d, CTX_MVP_FLAG);
    }
    for (int X = 0; X < 2; X++) {
      if (ref_idx[X] < 0) continue;
      int mvp[2];
      derive_mvp(d, g, X, ref_idx[X], mvp_flag[X], mvp);
      m.pred_flag[X] = 1; m.ref_idx[X] = ref_idx[X];
      for (int k = 0; k < 2; k++) {   /* 8.5.3.2.1: uLX = (mvpLX + mvdLX + 2^16) % 2^16, wrapped into 16 bits */
        int u = (mvp[k] + mvd[X][k] + 65536) & 65535;
        m.mv[X][k] = u >= 32768 ? u - 65536 : u;
      }
    }
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
static int mv_far(const int16_t* a, const int16_t* b) { return Abs(a[0] - b[0]) >= 4 || Abs(a[1] - b[1]) >= 4; }
static int edge_bs(Dec* d, int idxP, int idxQ, int x, int y, int dir)
{
  if (!d->m_pred) return 2;                                     /* intra picture */
  if (d->m_pred[idxP] == 0 || d->m_pred[idxQ] == 0) return 2;
  int tbq = 1 << d->m_log2_tb[idxQ];
  int tu_edge = dir == 0 ? (x & (tbq - 1)) == 0 : (y & (tbq - 1)) == 0;      /* transform blocks are aligned to their size */
  if (tu_edge && ((d->m_flags[idxP] | d->m_flags[idxQ]) & 1)) return 1;     /* a block with non-zero luma coefficient levels */
  /* different reference PICTURES (not indices, not lists) or a different number of motion vectors */
  const int8_t* rP = d->mf_ref + 2 * idxP; const int8_t* rQ = d->mf_ref + 2 * idxQ;
  const int32_t* pP = d->mf_poc + 2 * idxP; const int32_t* pQ = d->mf_poc + 2 * idxQ;
  const int16_t* vP = d->mf_mv + 4 * idxP; const int16_t* vQ = d->mf_mv + 4 * idxQ;
  int nP = (rP[0] >= 0) + (rP[1] >= 0), nQ = (rQ[0] >= 0) + (rQ[1] >= 0);
  if (nP != nQ) return 1;
  if (nP == 1) {
    int LP = rP[0] >= 0 ? 0 : 1, LQ = rQ[0] >= 0 ? 0 : 1;
    if (pP[LP] != pQ[LQ]) return 1;
    return mv_far(vP + 2 * LP, vQ + 2 * LQ);
  }
  if (!((pP[0] == pQ[0] && pP[1] == pQ[1]) || (pP[0] == pQ[1] && pP[1] == pQ[0]))) return 1;
  if (pP[0] != pP[1]) {   /* two different reference pictures: compare the vectors that point at the same one */
    if (pP[0] == pQ[0]) return mv_far(vP, vQ) || mv_far(vP + 2, vQ + 2);
    return mv_far(vP, vQ + 2) || mv_far(vP + 2, vQ);
  }
  /* both vectors of both blocks point at the same picture: either pairing may match */
  return (mv_far(vP, vQ) || mv_far(vP + 2, vQ + 2)) && (mv_far(vP, vQ + 2) || mv_far(vP + 2, vQ));
}
