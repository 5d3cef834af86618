/*
 * hevc_testenc_inter.c — TEST-ONLY stream generator, P and B pictures (#included by hevc_testenc.c; see there).
 *
 * Emits the syntax a P / B slice adds (7.3.8.5 cu_skip_flag / pred_mode_flag / inter part_mode, 7.3.8.6 prediction_unit incl. inter_pred_idc and list 1, 7.3.8.9 mvd_coding,
 * rqt_root_cbf) with pseudo-random decisions, while driving the oracle's own state machine (hevc_oracle_inter.c: candidate derivation,
 * interpolation, motion field) so that the residual it codes is the difference to exactly the prediction a decoder will form.
 * The syntax side is independent of the oracle's parser; lossless (cu_transquant_bypass) sequences make decoded == source a hard check.
 */

static void enc_mvd(Enc* e, const int* mvd)
{
  int a[2] = {Abs(mvd[0]), Abs(mvd[1])};
  EV_D(CTX_MVD_GT0, a[0] > 0);
  EV_D(CTX_MVD_GT0, a[1] > 0);
  if (a[0] > 0) EV_D(CTX_MVD_GT1, a[0] > 1);
  if (a[1] > 0) EV_D(CTX_MVD_GT1, a[1] > 1);
  for (int k = 0; k < 2; k++) {
    if (!a[k]) continue;
    if (a[k] > 1) {   /* abs_mvd_minus2: EG1 */
      int v = a[k] - 2, kk = 1;
      while (v >= (1 << kk)) { EV_B(1); v -= 1 << kk; kk++; }
      EV_B(0);
      EV_BB(v, kk);
    }
    EV_B(mvd[k] < 0);
  }
}

static void enc_part_mode_inter(Enc* e, int log2CbSize, int PartMode)
{
  const SPS* s = e->d->s;
  if (PartMode == PART_2Nx2N) { EV_D(CTX_PART_MODE, 1); return; }
  EV_D(CTX_PART_MODE, 0);
  if (log2CbSize == s->log2_min_cb) {
    if (PartMode == PART_2NxN) { EV_D(CTX_PART_MODE_INTER + 0, 1); return; }
    EV_D(CTX_PART_MODE_INTER + 0, 0);
    if (log2CbSize == 3) return;                       /* Nx2N */
    EV_D(CTX_PART_MODE_INTER + 1, PartMode == PART_Nx2N);
    return;
  }
  if (!s->amp_enabled_flag) { EV_D(CTX_PART_MODE_INTER + 0, PartMode == PART_2NxN); return; }
  int horizontal = PartMode == PART_2NxN || PartMode == PART_2NxnU || PartMode == PART_2NxnD;
  EV_D(CTX_PART_MODE_INTER + 0, horizontal);
  int symmetric = PartMode == PART_2NxN || PartMode == PART_Nx2N;
  EV_D(CTX_PART_MODE_INTER + 2, symmetric);
  if (!symmetric) EV_B(PartMode == PART_2NxnD || PartMode == PART_nRx2N);
}

/* a coding unit of a P / B slice that is not intra coded; ev_skip / ev_pred: the events of its cu_skip_flag / pred_mode_flag */
static void enc_inter_coding_unit(Enc* e, CuCtx* cu, int x0, int y0, int log2CbSize, int cqtDepth, int cu_skip, int ev_skip, int ev_pred)
{
  Dec* d = e->d; const SPS* s = d->s;
  const hevc_testenc_params* prm = &e->prm;
  int nCbS = 1 << log2CbSize;
  int u0x = x0 >> 2, u0y = y0 >> 2, nu = nCbS >> 2;
  d->cu_pred_inter = 1;
  set_qp_y(d);
  for (int j = 0; j < nu; j++)
    for (int i = 0; i < nu; i++) {
      int idx = (u0y + j) * d->mw + u0x + i;
      d->m_log2_cb[idx] = (uint8_t)log2CbSize;
      d->m_ctdepth[idx] = (uint8_t)cqtDepth;
      d->m_flags[idx] = (uint8_t)(d->cu_transquant_bypass_flag ? 0x08 : 0);
      d->m_decoded[idx] = 1;
      d->m_ipm[idx] = 1; d->m_ipmc[idx] = 1;
      d->m_pred[idx] = (uint8_t)(cu_skip ? 2 : 1);
    }
  int PartMode = PART_2Nx2N;
  int ev_first = e->nev;      /* events of part_mode .. the prediction units (withdrawn if the unit turns into a skipped one) */
  if (!cu_skip) {
    int modes[8], nm = 0;
    modes[nm++] = PART_2Nx2N; modes[nm++] = PART_2NxN; modes[nm++] = PART_Nx2N;
    if (log2CbSize == s->log2_min_cb) { if (log2CbSize > 3) modes[nm++] = PART_NxN; }
    else if (s->amp_enabled_flag) { modes[nm++] = PART_2NxnU; modes[nm++] = PART_2NxnD; modes[nm++] = PART_nLx2N; modes[nm++] = PART_nRx2N; }
    PartMode = rnd_pct(e, 40) ? PART_2Nx2N : modes[rnd(e) % nm];
    enc_part_mode_inter(e, log2CbSize, PartMode);
  }
  PbGeom g;
  int nParts = part_geometry(PartMode, x0, y0, nCbS, 0, &g), merge0 = 0, ev_merge_idx0 = -1, n_merge_idx0 = 0;
  int max_cand = d->sh->max_num_merge_cand, nref = d->sh->num_ref_idx_l0_active;
  for (int k = 0; k < nParts; k++) {
    part_geometry(PartMode, x0, y0, nCbS, k, &g);
    int merge = cu_skip || rnd_pct(e, prm->inter_merge_pct);
    Motion m = motion_none();
    if (!cu_skip) EV_D(CTX_MERGE_FLAG, merge);
    if (k == 0) merge0 = merge;
    if (merge) {
      int merge_idx = (int)(rnd(e) % (unsigned)max_cand);
      if (rnd_pct(e, 50)) merge_idx = 0;
      if (k == 0) ev_merge_idx0 = e->nev;
      if (max_cand > 1) {
        EV_D(CTX_MERGE_IDX, merge_idx > 0);
        if (merge_idx > 0) for (int i = 1; i < max_cand - 1; i++) { EV_B(merge_idx > i); if (merge_idx <= i) break; }
      }
      if (k == 0) n_merge_idx0 = e->nev - ev_merge_idx0;
      m = derive_merge(d, &g, PartMode, merge_idx);
    } else {
      /* inter_pred_idc of a B slice: PRED_L0 0, PRED_L1 1, PRED_BI 2 (not for 8x4 / 4x8 blocks) */
      int idc = 0;
      if (d->sh->slice_type == 0) {
        int can_bi = g.nPbW + g.nPbH != 12;
        idc = can_bi && rnd_pct(e, prm->inter_bi_pct) ? 2 : (int)(rnd(e) & 1);
        if (can_bi) EV_D(CTX_INTER_PRED_IDC + cqtDepth, idc == 2);
        if (idc != 2) EV_D(CTX_INTER_PRED_IDC + 4, idc);
      }
      for (int X = 0; X < 2; X++) {
        if (!(idc == 2 || idc == X)) continue;
        int nrefX = X ? d->sh->num_ref_idx_l1_active : nref;
        int ref_idx = nrefX > 1 ? (int)(rnd(e) % (unsigned)nrefX) : 0;
        if (nrefX > 1) {
          int cmax = nrefX - 1;
          for (int i = 0; i < cmax; i++) {
            int b = ref_idx > i;
            if (i < 2) EV_D(CTX_REF_IDX + i, b); else EV_B(b);
            if (!b) break;
          }
        }
        int mvp_flag = (int)(rnd(e) & 1), mvp[2];
        derive_mvp(d, &g, X, ref_idx, mvp_flag, mvp);
        int mv[2];
        unsigned r = rnd(e) % 100;
        if (r < 30) { mv[0] = mvp[0]; mv[1] = mvp[1]; }                                          /* mvd 0 */
        else if (r < 85) {   /* around the global motion: towards a picture after this one the scene moves the other way */
          int sgn = d->sh->ref_poc[X][ref_idx] > d->poc ? -1 : 1;
          mv[0] = sgn * prm->global_mv_x + (int)(rnd(e) % 9) - 4; mv[1] = sgn * prm->global_mv_y + (int)(rnd(e) % 9) - 4;
        }
        else if (r < 97) { mv[0] = (int)(rnd(e) % 257) - 128; mv[1] = (int)(rnd(e) % 257) - 128; }
        else { mv[0] = (int)(rnd(e) % (unsigned)(8 * d->W + 1)) - 4 * d->W; mv[1] = (int)(rnd(e) % (unsigned)(8 * d->H + 1)) - 4 * d->H; }   /* far outside: padding */
        int mvd[2];
        for (int c = 0; c < 2; c++) { mv[c] = Clip3(-32768, 32767, mv[c]); mvd[c] = Clip3(-32768, 32767, mv[c] - mvp[c]); }
        if (X == 1 && idc == 2 && d->sh->mvd_l1_zero_flag) mvd[0] = mvd[1] = 0;                   /* MvdL1 is inferred: not coded */
        else enc_mvd(e, mvd);
        EV_D(CTX_MVP_FLAG, mvp_flag);
        for (int c = 0; c < 2; c++) { int u = (mvp[c] + mvd[c] + 65536) & 65535; m.mv[X][c] = u >= 32768 ? u - 65536 : u; }
        m.ref_idx[X] = ref_idx; m.pred_flag[X] = 1;
      }
    }
    store_motion(d, g.xPb, g.yPb, g.nPbW, g.nPbH, &m);
    predict_pu(d, g.xPb, g.yPb, g.nPbW, g.nPbH, &m);
  }
  cu->inter = 1; cu->PartMode = PartMode; cu->chroma_mode = 1;
  int coded = 0;
  if (!cu_skip) {
    int root_inferred = PartMode == PART_2Nx2N && merge0;
    int ev_root = root_inferred ? -1 : EV_D(CTX_RQT_ROOT_CBF, 1);
    int ev_tree = e->nev;
    e->cu_any_cbf = 0;
    cu->IntraSplitFlag = 0;
    cu->MaxTrafoDepth = s->max_transform_hierarchy_depth_inter;
    int sv_coded = d->IsCuQpDeltaCoded, sv_delta = d->CuQpDeltaVal, sv_qp = d->cur_qp_y;
    enc_transform_tree(e, cu, x0, y0, x0, y0, log2CbSize, 0, 0, -1, -1, 0, 0);
    coded = e->cu_any_cbf;
    if (!coded) {
      /* every block came out empty: the transform tree is withdrawn (no cu_qp_delta was coded in it) */
      e->nev = ev_tree;
      d->IsCuQpDeltaCoded = sv_coded; d->CuQpDeltaVal = sv_delta; d->cur_qp_y = sv_qp;
      if (ev_root >= 0) e->ev[ev_root].val = 0;
      else {
        /* a 2Nx2N merged unit has no rqt_root_cbf to say so: it becomes a skipped unit - cu_skip_flag 1, pred_mode_flag / part_mode /
           merge_flag withdrawn, its merge_idx bins kept (7.3.8.5, 7.3.8.6) */
        e->ev[ev_skip].val = 1;
        e->ev[ev_pred].kind = EV_NONE;
        for (int i = ev_first; i < ev_tree; i++) if (i < ev_merge_idx0 || i >= ev_merge_idx0 + n_merge_idx0) e->ev[i].kind = EV_NONE;
        for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) d->m_pred[(u0y + j) * d->mw + u0x + i] = 2;
      }
    }
  }
  if (!coded) mark_cu_no_residual(d, cu, x0, y0, log2CbSize);
  mark_pu_edges(d, x0, y0, nCbS, PartMode);
  set_qp_y(d);
  for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) d->m_qp[(u0y + j) * d->mw + u0x + i] = (int8_t)d->cur_qp_y;
  d->last_qp_y = d->cur_qp_y;
  d->cu_pred_inter = 0;
}
