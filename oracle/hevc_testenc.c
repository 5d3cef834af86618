/*
 * hevc_testenc.c — TEST-ONLY HEVC intra bitstream generator (test infrastructure, NOT product).
 *
 * There is no HEVC encoder in the container and the reference ships none (x265/kvazaar are
 * external plugins), so the synthetic streams of BASELINE.json's configs are produced
 * here.  It is a real (if crude) encoder: quadtree / intra-mode / transform-split decisions, forward
 * transform + quantisation, CABAC encoding (H.265 9.3.4.x, encoder side), in-loop reconstruction so
 * that prediction uses decoded samples.  It reuses the oracle decoder's state machine and helpers by
 * including its translation unit, so every decision is replayed exactly as a decoder will see it.
 * What it emits matches what libheif expects from an encoder plugin (libheif/codecs/hevc_enc.cc:53-82):
 * VPS, SPS, PPS and slice NALs, here already in the plugin framing [u32 BE length][NAL].
 */
#include "hevc_oracle.c"
#include "hevc_testenc.h"
#include <math.h>

/* ------------------------------------------------------------------------------------------ */
/* bit writer                                                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  uint8_t* p;
  size_t cap, nbits;
} BW;

static void bw_put(BW* w, unsigned bit)
{
  if ((w->nbits >> 3) >= w->cap) {
    size_t nc = w->cap ? w->cap * 2 : 4096;
    w->p = (uint8_t*)realloc(w->p, nc);
    memset(w->p + w->cap, 0, nc - w->cap);
    w->cap = nc;
  }
  if (bit) w->p[w->nbits >> 3] |= (uint8_t)(0x80 >> (w->nbits & 7));
  w->nbits++;
}
static void bw_u(BW* w, unsigned v, int n) { for (int i = n - 1; i >= 0; i--) bw_put(w, (v >> i) & 1); }
static void bw_ue(BW* w, unsigned v)
{
  unsigned x = v + 1; int len = 0;
  while ((x >> len) > 1) len++;
  for (int i = 0; i < len; i++) bw_put(w, 0);
  bw_u(w, x, len + 1);
}
static void bw_se(BW* w, int v) { bw_ue(w, v > 0 ? (unsigned)(2 * v - 1) : (unsigned)(-2 * v)); }
/* 7.3.4 scaling_list_data() with pseudo-random content that exercises every branch of the syntax: lists copied from the
 * default, lists copied from an earlier matrix (incl. its DC), explicit DPCM-coded lists with DC coefficients.  The lists the
 * decoder will derive are obtained by parsing the bits just written with the oracle's own parser. */
static void write_scaling_list_data(Dec* d, BW* w, ScalingList* sl, unsigned seed)
{
  const size_t start = w->nbits;
  unsigned st = seed * 2654435761u + 12345u;
  init_scans();
  for (int sizeId = 0; sizeId < 4; sizeId++)
    for (int matrixId = 0; matrixId < 6; matrixId += (sizeId == 3) ? 3 : 1) {
      st = st * 1664525u + 1013904223u;
      const unsigned choice = (st >> 24) % 4;
      const int step = sizeId == 3 ? 3 : 1;
      if (choice == 0) { bw_u(w, 0, 1); bw_ue(w, 0); }                                   /* the default list */
      else if (choice == 1 && matrixId >= step) { bw_u(w, 0, 1); bw_ue(w, 1 + ((st >> 8) % (unsigned)(matrixId / step))); }  /* copy an earlier one */
      else {
        bw_u(w, 1, 1);
        const int n = sizeId == 0 ? 16 : 64;
        int next = 8;
        if (sizeId > 1) { int dc = 4 + (int)((st >> 12) % 40); bw_se(w, dc - 8); next = dc; }
        int v = 10 + (int)((st >> 4) % 20);
        for (int i = 0; i < n; i++) {
          st = st * 1664525u + 1013904223u;
          v += (int)((st >> 20) % 9) - 3 + (i > n / 2);          /* a rising random walk like real perceptual lists */
          if (v < 1) v = 1;
          if (v > 255) v = 255;
          int delta = v - next;
          if (delta > 127) delta -= 256;
          if (delta < -128) delta += 256;
          bw_se(w, delta);
          next = v;
        }
      }
    }
  BR b; memset(&b, 0, sizeof(b));
  b.d = d; b.p = w->p; b.pos = start; b.nbits = w->nbits;
  parse_scaling_list_data(d, &b, sl);
  if (b.pos != w->nbits) fail(d, "testenc: scaling list writer / parser disagree");
}

static void bw_trailing(BW* w) { bw_put(w, 1); while (w->nbits & 7) bw_put(w, 0); }
static void bw_free(BW* w) { free(w->p); memset(w, 0, sizeof(*w)); }

/* ------------------------------------------------------------------------------------------ */
/* events + CABAC encoder (9.3.4.x encoding process)                                          */
/* ------------------------------------------------------------------------------------------ */
enum { EV_DECISION, EV_BYPASS, EV_BYPASS_BITS, EV_TERMINATE, EV_PCM, EV_NONE /* withdrawn */ };
typedef struct { uint8_t kind; int16_t ctx; int32_t val; int32_t n; int32_t cond; } Event;

typedef struct {
  BW bw;
  uint32_t low, range;
  int firstBitFlag, bitsOutstanding;
} CabacEnc;

static void ce_init(CabacEnc* e) { e->low = 0; e->range = 510; e->firstBitFlag = 1; e->bitsOutstanding = 0; }
static void ce_putbit(CabacEnc* e, unsigned b)
{
  if (e->firstBitFlag) e->firstBitFlag = 0; else bw_put(&e->bw, b);
  while (e->bitsOutstanding > 0) { bw_put(&e->bw, 1 - b); e->bitsOutstanding--; }
}
static void ce_renorm(CabacEnc* e)
{
  while (e->range < 256) {
    if (e->low < 256) ce_putbit(e, 0);
    else if (e->low >= 512) { e->low -= 512; ce_putbit(e, 1); }
    else { e->low -= 256; e->bitsOutstanding++; }
    e->range <<= 1; e->low <<= 1;
  }
}
static void ce_decision(CabacEnc* e, uint8_t* ctx, int bin)
{
  int pState = *ctx >> 1, valMps = *ctx & 1;
  unsigned rLps = hevc_cabac_range_lps[pState][(e->range >> 6) & 3];
  e->range -= rLps;
  if (bin != valMps) {
    e->low += e->range; e->range = rLps;
    if (pState == 0) valMps = 1 - valMps;
    pState = hevc_cabac_next_lps[pState];
  } else pState = hevc_cabac_next_mps[pState];
  *ctx = (uint8_t)((pState << 1) | valMps);
  ce_renorm(e);
}
static void ce_bypass(CabacEnc* e, int bin)
{
  e->low <<= 1;
  if (bin) e->low += e->range;
  if (e->low >= 1024) { ce_putbit(e, 1); e->low -= 1024; }
  else if (e->low < 512) ce_putbit(e, 0);
  else { e->low -= 512; e->bitsOutstanding++; }
}
static void ce_flush(CabacEnc* e)
{
  e->range = 2;
  ce_renorm(e);
  ce_putbit(e, (e->low >> 9) & 1);
  bw_u(&e->bw, ((e->low >> 7) & 3) | 1, 2);
}
static void ce_terminate(CabacEnc* e, int bin)
{
  e->range -= 2;
  if (bin) { e->low += e->range; ce_flush(e); }
  else ce_renorm(e);
}

/* ------------------------------------------------------------------------------------------ */
/* encoder state                                                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  Dec* d;                 /* the oracle decoder's state machine, driven by the encoder */
  hevc_testenc_params prm;
  const uint16_t* src[3]; /* coded-size source planes */
  uint64_t rng;
  Event* ev; int nev, capev;
  uint8_t* pcm_blob; size_t pcm_bits, pcm_cap; /* raw bits referenced by EV_PCM */
  CabacEnc ce;
  uint8_t ctx[MAXCTX], ctx_wpp[MAXCTX];
  int qg_delta;           /* chosen delta for the current quantisation group */
  int cu_any_cbf;         /* a transform block of the current coding unit has coefficients */
  uint16_t* src_alloc[3];
  int frame_idx;
} Enc;

/* one picture of a sequence in coding order: its type and its reference picture set (POCs, closest first) */
typedef struct {
  int poc, slice_type /* 2 I, 1 P, 0 B */, nal_type;
  int n_neg, n_pos, neg_poc[16], pos_poc[16];
  uint8_t neg_used[16], pos_used[16];
  int n_lt, lt_poc[4];                      /* long-term reference pictures (hevc_testenc_params::long_term_ref): always used by the picture */
} PicPlan;

static uint32_t rnd(Enc* e)
{
  e->rng = e->rng * 6364136223846793005ULL + 1442695040888963407ULL;
  return (uint32_t)(e->rng >> 33);
}
static int rnd_pct(Enc* e, int pct) { return (int)(rnd(e) % 100) < pct; }

static int ev_add(Enc* e, int kind, int ctx, int val, int n, int cond)
{
  if (e->nev == e->capev) { e->capev = e->capev ? e->capev * 2 : 4096; e->ev = (Event*)realloc(e->ev, sizeof(Event) * e->capev); }
  Event* v = &e->ev[e->nev];
  v->kind = (uint8_t)kind; v->ctx = (int16_t)ctx; v->val = val; v->n = n; v->cond = cond;
  return e->nev++;
}
#define EV_D(ctx, bin) ev_add(e, EV_DECISION, (ctx), (bin), 0, -1)
#define EV_B(bin) ev_add(e, EV_BYPASS, 0, (bin), 0, -1)
#define EV_BB(val, n) ev_add(e, EV_BYPASS_BITS, 0, (val), (n), -1)

static void flush_events(Enc* e)
{
  for (int i = 0; i < e->nev; i++) {
    Event* v = &e->ev[i];
    if (v->cond >= 0 && e->ev[v->cond].val == 0) continue; /* parent cbf is 0: flag not coded */
    switch (v->kind) {
      case EV_NONE: break;
      case EV_DECISION: ce_decision(&e->ce, &e->ctx[v->ctx], v->val); break;
      case EV_BYPASS: ce_bypass(&e->ce, v->val); break;
      case EV_BYPASS_BITS: for (int k = v->n - 1; k >= 0; k--) ce_bypass(&e->ce, (v->val >> k) & 1); break;
      case EV_TERMINATE: ce_terminate(&e->ce, v->val); break;
      case EV_PCM: {
        /* pcm_flag was a terminate(1) (flushed); now pcm_alignment_zero_bits + samples, then re-init */
        while (e->ce.bw.nbits & 7) bw_put(&e->ce.bw, 0);
        for (int k = 0; k < v->n; k++) {
          size_t bp = (size_t)v->val + k;
          bw_put(&e->ce.bw, (e->pcm_blob[bp >> 3] >> (7 - (bp & 7))) & 1);
        }
        ce_init(&e->ce);
        break;
      }
    }
  }
  e->nev = 0; e->pcm_bits = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* forward transform + quantisation (encoder-side freedom; any level array is a legal stream)  */
/* ------------------------------------------------------------------------------------------ */
static void forward_quant(Enc* e, int32_t* lev, const int32_t* res, int n, int qP, int bit_depth, int trType,
                          int transform_skip, int bypass)
{
  static const int quantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564};
  int log2n = 0; while ((1 << log2n) < n) log2n++;
  (void)e;
  if (bypass) { memcpy(lev, res, sizeof(int32_t) * n * n); return; }
  int64_t coef[32 * 32];
  int transformShift = 15 - bit_depth - log2n;
  if (transform_skip) {
    for (int i = 0; i < n * n; i++) coef[i] = transformShift >= 0 ? (int64_t)res[i] << transformShift : (int64_t)res[i] >> -transformShift;
  } else {
    int64_t tmp[32 * 32];
    int s1 = log2n + bit_depth - 9, s2 = log2n + 6;
    for (int k = 0; k < n; k++)       /* vertical: tmp[k][x] = sum_j M[k][j] res[j][x] */
      for (int x = 0; x < n; x++) {
        int64_t s = 0;
        for (int j = 0; j < n; j++) s += (int64_t)(trType ? g_dst[k][j] : g_dct[k * (32 / n)][j]) * res[j * n + x];
        tmp[k * n + x] = s1 > 0 ? (s + ((int64_t)1 << (s1 - 1))) >> s1 : s;
      }
    for (int k = 0; k < n; k++)       /* horizontal: coef[y][k] = sum_j tmp[y][j] M[k][j] */
      for (int y = 0; y < n; y++) {
        int64_t s = 0;
        for (int j = 0; j < n; j++) s += tmp[y * n + j] * (trType ? g_dst[k][j] : g_dct[k * (32 / n)][j]);
        coef[y * n + k] = (s + ((int64_t)1 << (s2 - 1))) >> s2;
      }
  }
  int qbits = 14 + qP / 6 + transformShift;
  int64_t add = ((int64_t)171 << qbits) >> 9;
  for (int i = 0; i < n * n; i++) {
    int64_t a = coef[i] < 0 ? -coef[i] : coef[i];
    int64_t l = (a * quantScales[qP % 6] + add) >> qbits;
    if (l > 32767) l = 32767;
    lev[i] = (int32_t)(coef[i] < 0 ? -l : l);
  }
}

/* sign-data-hiding parity fix so that the hidden sign decodes correctly (7.3.8.11 / 9.3.4.x) */
static void sdh_fix(int32_t* lev, int log2n, int scanIdx)
{
  int n = 1 << log2n;
  const uint8_t* scanSB = log2n > 2 ? g_scan[log2n - 2][scanIdx] : NULL;
  const uint8_t* scanPos = g_scan[2][scanIdx];
  int nsb = 1 << (2 * (log2n - 2));
  for (int i = 0; i < nsb; i++) {
    int xS = scanSB ? (scanSB[i] & 15) : 0, yS = scanSB ? (scanSB[i] >> 4) : 0;
    int first = 16, last = -1, sum = 0;
    for (int k = 0; k < 16; k++) {
      int xC = (xS << 2) + (scanPos[k] & 15), yC = (yS << 2) + (scanPos[k] >> 4);
      int v = lev[yC * n + xC];
      if (v) { if (first == 16) first = k; last = k; sum += v < 0 ? -v : v; }
    }
    if (last - first > 3) {
      int xC = (xS << 2) + (scanPos[first] & 15), yC = (yS << 2) + (scanPos[first] >> 4);
      int32_t* pv = &lev[yC * n + xC];
      int neg = *pv < 0;
      if ((sum & 1) != neg) { if (*pv < 0) { if (*pv > -32767) (*pv)--; else (*pv)++; } else { if (*pv < 32767) (*pv)++; else (*pv)--; } }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* residual_coding writer (mirror of the decoder's residual_coding)                            */
/* ------------------------------------------------------------------------------------------ */
static void emit_last_prefix(Enc* e, int base, int ctxOffset, int ctxShift, int cMax, int p)
{
  for (int b = 0; b < p; b++) EV_D(base + ctxOffset + (b >> ctxShift), 1);
  if (p < cMax) EV_D(base + ctxOffset + (p >> ctxShift), 0);
}
static void split_last(int v, int* prefix, int* suffix, int* nbits)
{
  if (v < 4) { *prefix = v; *suffix = 0; *nbits = 0; return; }
  int k = 0; while ((v >> (k + 1)) != 0) k++;
  int bit = (v >> (k - 1)) & 1;
  *prefix = 2 * k + bit; *nbits = k - 1; *suffix = v - ((2 + bit) << (k - 1));
}
static void emit_remaining(Enc* e, int value, int rice)
{
  int prefix, suffix, nb;
  if (value < (4 << rice)) {
    prefix = value >> rice;
    for (int i = 0; i < prefix; i++) EV_B(1);
    EV_B(0);
    if (rice) EV_BB(value & ((1 << rice) - 1), rice);
    return;
  }
  /* value = (((1 << (prefix-3)) + 2) << rice) + suffix, suffix < 1 << (prefix-3+rice) */
  int v = value >> rice, p3 = 0;
  while (v >= ((1 << (p3 + 1)) + 2)) p3++;
  prefix = p3 + 3;
  nb = p3 + rice;
  suffix = value - (((1 << p3) + 2) << rice);
  for (int i = 0; i < prefix; i++) EV_B(1);
  EV_B(0);
  if (nb) EV_BB(suffix, nb);
}

static void emit_residual(Enc* e, const int32_t* lev, int log2TrafoSize, int cIdx, int predModeIntra, int transform_skip)
{
  Dec* d = e->d; const PPS* p = d->p;
  int nTbS = 1 << log2TrafoSize;
  if (p->transform_skip_enabled_flag && !d->cu_transquant_bypass_flag && log2TrafoSize <= 2)
    EV_D(CTX_TRANSFORM_SKIP + (cIdx ? 1 : 0), transform_skip);
  int scanIdx = 0;
  if (log2TrafoSize == 2 || (log2TrafoSize == 3 && (cIdx == 0 || d->s->chroma_format_idc == 3))) {
    if (predModeIntra >= 6 && predModeIntra <= 14) scanIdx = 2;
    else if (predModeIntra >= 22 && predModeIntra <= 30) scanIdx = 1;
  }
  const uint8_t* scanSB = log2TrafoSize > 2 ? g_scan[log2TrafoSize - 2][scanIdx] : NULL;
  const uint8_t* scanPos = g_scan[2][scanIdx];
  int nsb = 1 << (2 * (log2TrafoSize - 2));
  int lastSubBlock = -1, lastScanPos = -1;
  for (int i = nsb - 1; i >= 0 && lastSubBlock < 0; i--) {
    int xS = scanSB ? (scanSB[i] & 15) : 0, yS = scanSB ? (scanSB[i] >> 4) : 0;
    for (int n = 15; n >= 0; n--) {
      int xC = (xS << 2) + (scanPos[n] & 15), yC = (yS << 2) + (scanPos[n] >> 4);
      if (lev[yC * nTbS + xC]) { lastSubBlock = i; lastScanPos = n; break; }
    }
  }
  int xSl = scanSB ? (scanSB[lastSubBlock] & 15) : 0, ySl = scanSB ? (scanSB[lastSubBlock] >> 4) : 0;
  int LastX = (xSl << 2) + (scanPos[lastScanPos] & 15), LastY = (ySl << 2) + (scanPos[lastScanPos] >> 4);
  int codeX = LastX, codeY = LastY;
  if (scanIdx == 2) { codeX = LastY; codeY = LastX; }
  int ctxOffset, ctxShift;
  if (cIdx == 0) { ctxOffset = 3 * (log2TrafoSize - 2) + ((log2TrafoSize - 1) >> 2); ctxShift = (log2TrafoSize + 1) >> 2; }
  else { ctxOffset = 15; ctxShift = log2TrafoSize - 2; }
  int cMax = (log2TrafoSize << 1) - 1;
  int px, sx, nx, py, sy, ny;
  split_last(codeX, &px, &sx, &nx); split_last(codeY, &py, &sy, &ny);
  emit_last_prefix(e, CTX_LAST_X, ctxOffset, ctxShift, cMax, px);
  emit_last_prefix(e, CTX_LAST_Y, ctxOffset, ctxShift, cMax, py);
  if (px > 3) EV_BB(sx, nx);
  if (py > 3) EV_BB(sy, ny);

  uint8_t csbf_map[8][8];
  memset(csbf_map, 0, sizeof(csbf_map));
  int sbw = 1 << (log2TrafoSize - 2);
  int greater1Ctx_carry = 1, first_subblock_with_g1 = 1;
  for (int i = lastSubBlock; i >= 0; i--) {
    int xS = scanSB ? (scanSB[i] & 15) : 0, yS = scanSB ? (scanSB[i] >> 4) : 0;
    int32_t v16[16]; int any = 0;
    for (int n = 0; n < 16; n++) {
      int xC = (xS << 2) + (scanPos[n] & 15), yC = (yS << 2) + (scanPos[n] >> 4);
      v16[n] = lev[yC * nTbS + xC];
      if (v16[n]) any = 1;
    }
    int inferSbDcSigCoeffFlag = 0, csbf;
    if (i < lastSubBlock && i > 0) {
      int csbfCtx = 0;
      if (xS < sbw - 1) csbfCtx += csbf_map[xS + 1][yS];
      if (yS < sbw - 1) csbfCtx += csbf_map[xS][yS + 1];
      csbf = any;
      EV_D(CTX_CODED_SUB_BLOCK + Min(csbfCtx, 1) + (cIdx ? 2 : 0), csbf);
      inferSbDcSigCoeffFlag = 1;
    } else csbf = 1;
    csbf_map[xS][yS] = (uint8_t)csbf;
    int prevCsbf = 0;
    if (xS < sbw - 1) prevCsbf += csbf_map[xS + 1][yS];
    if (yS < sbw - 1) prevCsbf += 2 * csbf_map[xS][yS + 1];
    int nStart = (i == lastSubBlock) ? lastScanPos - 1 : 15;
    for (int n = nStart; n >= 0; n--) {
      int xP = scanPos[n] & 15, yP = scanPos[n] >> 4;
      int xC = (xS << 2) + xP, yC = (yS << 2) + yP;
      if (csbf && (n > 0 || !inferSbDcSigCoeffFlag)) {
        int sigCtx;
        if (log2TrafoSize == 2) {
          static const uint8_t ctxIdxMap[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
          sigCtx = ctxIdxMap[(yC << 2) + xC];
        } else if (xC + yC == 0) sigCtx = 0;
        else {
          if (prevCsbf == 0) sigCtx = (xP + yP == 0) ? 2 : (xP + yP < 3) ? 1 : 0;
          else if (prevCsbf == 1) sigCtx = (yP == 0) ? 2 : (yP == 1) ? 1 : 0;
          else if (prevCsbf == 2) sigCtx = (xP == 0) ? 2 : (xP == 1) ? 1 : 0;
          else sigCtx = 2;
          if (cIdx == 0) {
            if (xS > 0 || yS > 0) sigCtx += 3;
            if (log2TrafoSize == 3) sigCtx += (scanIdx == 0) ? 9 : 15; else sigCtx += 21;
          } else { if (log2TrafoSize == 3) sigCtx += 9; else sigCtx += 12; }
        }
        int sig = v16[n] != 0;
        EV_D(CTX_SIG_COEFF + (cIdx == 0 ? sigCtx : 27 + sigCtx), sig);
        if (sig) inferSbDcSigCoeffFlag = 0;
      }
    }
    if (!csbf) continue;
    int firstSigScanPos = 16, lastSigScanPos = -1, numGreater1Flag = 0, lastGreater1ScanPos = -1;
    int ctxSet = 0, greater1Ctx = 1, first_in_sb = 1;
    uint8_t g1[16], g2[16];
    memset(g1, 0, 16); memset(g2, 0, 16);
    for (int n = 15; n >= 0; n--) {
      if (!v16[n]) continue;
      int a = v16[n] < 0 ? -v16[n] : v16[n];
      if (numGreater1Flag < 8) {
        if (first_in_sb) {
          ctxSet = (i == 0 || cIdx > 0) ? 0 : 2;
          if (!first_subblock_with_g1 && greater1Ctx_carry == 0) ctxSet++;
          greater1Ctx = 1; first_in_sb = 0; first_subblock_with_g1 = 0;
        }
        g1[n] = a > 1;
        EV_D(CTX_GREATER1 + ctxSet * 4 + Min(3, greater1Ctx) + (cIdx ? 16 : 0), g1[n]);
        if (g1[n]) greater1Ctx = 0; else if (greater1Ctx > 0) greater1Ctx++;
        greater1Ctx_carry = greater1Ctx;
        numGreater1Flag++;
        if (g1[n] && lastGreater1ScanPos == -1) lastGreater1ScanPos = n;
      }
      if (lastSigScanPos == -1) lastSigScanPos = n;
      firstSigScanPos = n;
    }
    int signHidden = d->cu_transquant_bypass_flag ? 0 : (lastSigScanPos - firstSigScanPos > 3);
    if (lastGreater1ScanPos != -1) {
      int a = v16[lastGreater1ScanPos] < 0 ? -v16[lastGreater1ScanPos] : v16[lastGreater1ScanPos];
      g2[lastGreater1ScanPos] = a > 2;
      EV_D(CTX_GREATER2 + ctxSet + (cIdx ? 4 : 0), g2[lastGreater1ScanPos]);
    }
    for (int n = 15; n >= 0; n--)
      if (v16[n] && (!p->sign_data_hiding_enabled_flag || !signHidden || n != firstSigScanPos)) EV_B(v16[n] < 0);
    int numSigCoeff = 0, cRiceParam = 0;
    for (int n = 15; n >= 0; n--) {
      if (!v16[n]) continue;
      int a = v16[n] < 0 ? -v16[n] : v16[n];
      int baseLevel = 1 + g1[n] + g2[n];
      if (baseLevel == ((numSigCoeff < 8) ? ((n == lastGreater1ScanPos) ? 3 : 2) : 1)) {
        emit_remaining(e, a - baseLevel, cRiceParam);
        if (a > 3 * (1 << cRiceParam)) cRiceParam = Min(cRiceParam + 1, 4);
      }
      numSigCoeff++;
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* analysis + syntax emission, mirroring the decoder's control flow                            */
/* ------------------------------------------------------------------------------------------ */
static int block_variance(Enc* e, int x0, int y0, int n)
{
  Dec* d = e->d;
  int64_t s = 0, s2 = 0; int cnt = 0;
  for (int y = y0; y < Min(y0 + n, d->H); y++)
    for (int x = x0; x < Min(x0 + n, d->W); x++) { int v = e->src[0][y * d->W + x]; s += v; s2 += (int64_t)v * v; cnt++; }
  if (!cnt) return 0;
  int64_t var = (s2 - s * s / cnt) / cnt;
  return (int)(var >> (2 * (d->s->bit_depth_luma - 8)));
}

/* one transform block: predict, quantise, reconstruct; returns cbf and leaves levels in lev */
static int analyse_tb(Enc* e, int x0c, int y0c, int log2n, int cIdx, int mode, int32_t* lev, int* ts_out)
{
  Dec* d = e->d; const SPS* s = d->s; const PPS* p = d->p;
  int n = 1 << log2n;
  int stride = cIdx ? d->Wc : d->W;
  int bit_depth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
  if (!d->cu_pred_inter) intra_predict_block(d, x0c, y0c, log2n, cIdx, mode);   /* inter: rec holds the motion-compensated prediction */
  int32_t res[32 * 32];
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++)
      res[y * n + x] = (int)e->src[cIdx][(y0c + y) * stride + x0c + x] - (int)d->rec[cIdx][(y0c + y) * stride + x0c + x];
  int ts = 0;
  if (p->transform_skip_enabled_flag && !d->cu_transquant_bypass_flag && log2n == 2) ts = rnd_pct(e, 25);
  int qP;
  if (cIdx == 0) qP = d->cur_qp_y + 6 * (s->bit_depth_luma - 8);
  else {
    int QpBdOffsetC = 6 * (s->bit_depth_chroma - 8);
    int off = cIdx == 1 ? p->pps_cb_qp_offset + d->sh->slice_cb_qp_offset : p->pps_cr_qp_offset + d->sh->slice_cr_qp_offset;
    int qPi = Clip3(-QpBdOffsetC, 57, d->cur_qp_y + off);
    qP = (s->chroma_format_idc == 1 ? hevc_chroma_qp_420(qPi) : Min(qPi, 51)) + QpBdOffsetC;
  }
  forward_quant(e, lev, res, n, qP, bit_depth, cIdx == 0 && n == 4 && !d->cu_pred_inter, ts, d->cu_transquant_bypass_flag);
  if (e->prm.zero_residual_pct && rnd_pct(e, e->prm.zero_residual_pct)) memset(lev, 0, sizeof(int32_t) * n * n);
  int cbf = 0;
  for (int i = 0; i < n * n; i++) if (lev[i]) { cbf = 1; break; }
  if (cbf && p->sign_data_hiding_enabled_flag && !d->cu_transquant_bypass_flag) {
    int scanIdx = 0;
    if (log2n == 2 || (log2n == 3 && (cIdx == 0 || s->chroma_format_idc == 3))) {
      if (mode >= 6 && mode <= 14) scanIdx = 2; else if (mode >= 22 && mode <= 30) scanIdx = 1;
    }
    sdh_fix(lev, log2n, scanIdx);
  }
  *ts_out = ts;
  if (cbf) e->cu_any_cbf = 1;
  reconstruct_tb(d, x0c, y0c, log2n, cIdx, mode, cbf, lev, ts);
  return cbf;
}

typedef struct { int cbf_cb, cbf_cr; } ChromaCbf;

static void enc_cu_qp_delta(Enc* e)
{
  Dec* d = e->d;
  if (!d->p->cu_qp_delta_enabled_flag || d->IsCuQpDeltaCoded) return;
  int delta = e->qg_delta;
  int a = delta < 0 ? -delta : delta;
  /* prefix TU cMax 5, suffix EG0 */
  int pre = Min(a, 5);
  for (int i = 0; i < pre; i++) EV_D(CTX_CU_QP_DELTA + (i == 0 ? 0 : 1), 1);
  if (pre < 5) EV_D(CTX_CU_QP_DELTA + (pre == 0 ? 0 : 1), 0);
  else {
    int v = a - 5, k = 0;
    while (v >= (1 << k)) { EV_B(1); v -= 1 << k; k++; }
    EV_B(0);
    if (k) EV_BB(v, k);
  }
  if (a) EV_B(delta < 0);
  d->IsCuQpDeltaCoded = 1;
  d->CuQpDeltaVal = delta;
  set_qp_y(d);
}

/* returns chroma cbfs of the subtree; parent_slots = event indices of the parent's cbf_cb/cbf_cr */
static ChromaCbf enc_transform_tree(Enc* e, CuCtx* cu, int x0, int y0, int xBase, int yBase, int log2TrafoSize,
                                    int trafoDepth, int blkIdx, int slot_cb, int slot_cr, int pre_cb, int pre_cr)
{
  Dec* d = e->d; const SPS* s = d->s;
  int ChromaArrayType = s->chroma_format_idc;
  ChromaCbf out = {0, 0};
  int split;
  int can_signal = log2TrafoSize <= s->log2_max_tb && log2TrafoSize > s->log2_min_tb && trafoDepth < cu->MaxTrafoDepth &&
                   !(cu->IntraSplitFlag && trafoDepth == 0);
  if (can_signal) {
    split = e->prm.stress ? rnd_pct(e, 50) : (block_variance(e, x0, y0, 1 << log2TrafoSize) > 200 && rnd_pct(e, 60));
    EV_D(CTX_SPLIT_TRANSFORM + 5 - log2TrafoSize, split);
  } else {
    int interSplit = cu->inter && s->max_transform_hierarchy_depth_inter == 0 && cu->PartMode != PART_2Nx2N && trafoDepth == 0;
    split = (log2TrafoSize > s->log2_max_tb || (cu->IntraSplitFlag && trafoDepth == 0) || interSplit) ? 1 : 0;
  }

  int my_cb = -1, my_cr = -1; /* event slots of this node's chroma cbfs */
  int32_t levCb[16 * 16], levCr[16 * 16];
  int tsCb = 0, tsCr = 0;
  int chroma_here_early = 0;
  if (ChromaArrayType == 2) {
    /* 4:2:2: the chroma of a transform unit is two square blocks one above the other (k = 2 c + t: Cb 0, Cb 1, Cr 0, Cr 1), each with its own
       coded-block flag where the chroma is coded: at a leaf, or at the 8x8 node above four 4x4 luma leaves */
    int two = !split || log2TrafoSize == 3;
    int sl[4] = {-1, -1, -1, -1};
    if (log2TrafoSize > 2)
      for (int c = 0; c < 2; c++)
        for (int t = 0; t < (two ? 2 : 1); t++)
          sl[2 * c + t] = ev_add(e, EV_DECISION, CTX_CBF_CHROMA + trafoDepth, 0, 0, trafoDepth == 0 ? -1 : (c ? slot_cr : slot_cb));
    int x1 = x0 + (1 << (log2TrafoSize - 1)), y1 = y0 + (1 << (log2TrafoSize - 1));
    if (split && log2TrafoSize > 3) {
      for (int k = 0; k < 4; k++) {
        ChromaCbf c = enc_transform_tree(e, cu, (k & 1) ? x1 : x0, (k & 2) ? y1 : y0, x0, y0, log2TrafoSize - 1, trafoDepth + 1, k, sl[0], sl[2], 0, 0);
        out.cbf_cb |= c.cbf_cb; out.cbf_cr |= c.cbf_cr;
      }
      e->ev[sl[0]].val = out.cbf_cb; e->ev[sl[2]].val = out.cbf_cr;
      return out;
    }
    /* chroma blocks of this node: 4x4 pairs under an 8x8 split node, else blocks of half the luma size */
    int log2C = split ? 2 : log2TrafoSize - 1, nC = 1 << log2C;
    int32_t (*lv)[16 * 16] = (int32_t (*)[16 * 16])malloc(sizeof(int32_t) * 4 * 16 * 16);
    int32_t* lY = (int32_t*)malloc(sizeof(int32_t) * 32 * 32);
    int ts4[4] = {0, 0, 0, 0}, cbf4[4] = {0, 0, 0, 0};
    int pending = d->p->cu_qp_delta_enabled_flag && !d->IsCuQpDeltaCoded;
    int saved_delta = d->CuQpDeltaVal, saved_qp = d->cur_qp_y;
    if (split) {   /* log2TrafoSize == 3: chroma first (its flags are coded here), then the four luma leaves; child 3 carries the chroma residuals */
      if (pending) { d->CuQpDeltaVal = e->qg_delta; set_qp_y(d); }
      for (int k = 0; k < 4; k++) cbf4[k] = analyse_tb(e, x0 / 2, y0 + (k & 1) * nC, log2C, 1 + (k >> 1), cu->chroma_mode, lv[k], &ts4[k]);
      if (pending) { d->CuQpDeltaVal = saved_delta; d->cur_qp_y = saved_qp; }
      for (int k = 0; k < 4; k++) e->ev[sl[k]].val = cbf4[k];
      int anyC = cbf4[0] | cbf4[1] | cbf4[2] | cbf4[3];
      for (int k = 0; k < 4; k++) {
        int xx = (k & 1) ? x1 : x0, yy = (k & 2) ? y1 : y0, tsY;
        int mode = d->m_ipm[(yy >> 2) * d->mw + (xx >> 2)];
        int pend = d->p->cu_qp_delta_enabled_flag && !d->IsCuQpDeltaCoded;
        int sv_delta = d->CuQpDeltaVal, sv_qp = d->cur_qp_y;
        if (pend) { d->CuQpDeltaVal = e->qg_delta; set_qp_y(d); }
        int cbfY = analyse_tb(e, xx, yy, 2, 0, mode, lY, &tsY);
        if (pend) { d->CuQpDeltaVal = sv_delta; d->cur_qp_y = sv_qp; }
        EV_D(CTX_CBF_LUMA + 0, cbfY);
        if (cbfY || anyC) enc_cu_qp_delta(e);
        if (cbfY) emit_residual(e, lY, 2, 0, mode, tsY);
        if (k == 3)
          for (int q = 0; q < 4; q++) if (cbf4[q]) emit_residual(e, lv[q], 2, 1 + (q >> 1), cu->chroma_mode, ts4[q]);
        mark_tu(d, cu, xx, yy, 2, cbfY, k == 3 ? (cbf4[0] | cbf4[1]) : 0, k == 3 ? (cbf4[2] | cbf4[3]) : 0);
      }
      out.cbf_cb = cbf4[0] | cbf4[1]; out.cbf_cr = cbf4[2] | cbf4[3];
      free(lv); free(lY);
      return out;
    }
    /* leaf with log2TrafoSize >= 3 */
    int tsY = 0;
    int mode = d->m_ipm[(y0 >> 2) * d->mw + (x0 >> 2)];
    if (pending) { d->CuQpDeltaVal = e->qg_delta; set_qp_y(d); }
    int cbfY = analyse_tb(e, x0, y0, log2TrafoSize, 0, mode, lY, &tsY);
    for (int k = 0; k < 4; k++) cbf4[k] = analyse_tb(e, x0 / 2, y0 + (k & 1) * nC, log2C, 1 + (k >> 1), cu->chroma_mode, lv[k], &ts4[k]);
    if (pending) { d->CuQpDeltaVal = saved_delta; d->cur_qp_y = saved_qp; }
    for (int k = 0; k < 4; k++) e->ev[sl[k]].val = cbf4[k];
    /* 7.3.8.8: an inter coding unit's root leaf without chroma coefficients - all four flags of 4:2:2 - does not code cbf_luma (inferred 1) */
    if (!cu->inter || trafoDepth != 0 || cbf4[0] || cbf4[1] || cbf4[2] || cbf4[3]) EV_D(CTX_CBF_LUMA + (trafoDepth == 0 ? 1 : 0), cbfY);
    if (cbfY || cbf4[0] || cbf4[1] || cbf4[2] || cbf4[3]) enc_cu_qp_delta(e);
    if (cbfY) emit_residual(e, lY, log2TrafoSize, 0, mode, tsY);
    for (int q = 0; q < 4; q++) if (cbf4[q]) emit_residual(e, lv[q], log2C, 1 + (q >> 1), cu->chroma_mode, ts4[q]);
    mark_tu(d, cu, x0, y0, log2TrafoSize, cbfY, cbf4[0] | cbf4[1], cbf4[2] | cbf4[3]);
    out.cbf_cb = cbf4[0] | cbf4[1]; out.cbf_cr = cbf4[2] | cbf4[3];
    free(lv); free(lY);
    return out;
  }
  if (ChromaArrayType == 3) {
    /* 4:4:4: every node carries its chroma cbfs and every leaf its own chroma blocks, the size of the luma block */
    int cc = trafoDepth == 4 ? CTX_CBF_CHROMA4 : CTX_CBF_CHROMA + trafoDepth;
    my_cb = ev_add(e, EV_DECISION, cc, 0, 0, trafoDepth == 0 ? -1 : slot_cb);
    my_cr = ev_add(e, EV_DECISION, cc, 0, 0, trafoDepth == 0 ? -1 : slot_cr);
    if (split) {
      int x1 = x0 + (1 << (log2TrafoSize - 1)), y1 = y0 + (1 << (log2TrafoSize - 1));
      for (int k = 0; k < 4; k++) {
        ChromaCbf c = enc_transform_tree(e, cu, (k & 1) ? x1 : x0, (k & 2) ? y1 : y0, x0, y0, log2TrafoSize - 1, trafoDepth + 1, k,
                                         my_cb, my_cr, 0, 0);
        out.cbf_cb |= c.cbf_cb; out.cbf_cr |= c.cbf_cr;
      }
      e->ev[my_cb].val = out.cbf_cb; e->ev[my_cr].val = out.cbf_cr;
      return out;
    }
    int32_t* lev = (int32_t*)malloc(sizeof(int32_t) * 3 * 32 * 32);
    int32_t *lY = lev, *lCb = lev + 1024, *lCr = lev + 2048;
    int tsY = 0;
    int mode = d->m_ipm[(y0 >> 2) * d->mw + (x0 >> 2)], cmode = d->m_ipmc[(y0 >> 2) * d->mw + (x0 >> 2)];
    int pending = d->p->cu_qp_delta_enabled_flag && !d->IsCuQpDeltaCoded;
    int saved_delta = d->CuQpDeltaVal, saved_qp = d->cur_qp_y;
    if (pending) { d->CuQpDeltaVal = e->qg_delta; set_qp_y(d); }
    int cbfY = analyse_tb(e, x0, y0, log2TrafoSize, 0, mode, lY, &tsY);
    out.cbf_cb = analyse_tb(e, x0, y0, log2TrafoSize, 1, cmode, lCb, &tsCb);
    out.cbf_cr = analyse_tb(e, x0, y0, log2TrafoSize, 2, cmode, lCr, &tsCr);
    e->ev[my_cb].val = out.cbf_cb; e->ev[my_cr].val = out.cbf_cr;
    if (pending) { d->CuQpDeltaVal = saved_delta; d->cur_qp_y = saved_qp; }
    if (!cu->inter || trafoDepth != 0 || out.cbf_cb || out.cbf_cr) EV_D(CTX_CBF_LUMA + (trafoDepth == 0 ? 1 : 0), cbfY);   /* (7.3.8.8, as below) */
    if (cbfY || out.cbf_cb || out.cbf_cr) enc_cu_qp_delta(e);
    if (cbfY) emit_residual(e, lY, log2TrafoSize, 0, mode, tsY);
    if (out.cbf_cb) emit_residual(e, lCb, log2TrafoSize, 1, cmode, tsCb);
    if (out.cbf_cr) emit_residual(e, lCr, log2TrafoSize, 2, cmode, tsCr);
    mark_tu(d, cu, x0, y0, log2TrafoSize, cbfY, out.cbf_cb, out.cbf_cr);
    free(lev);
    return out;
  }
  if (log2TrafoSize > 2 && ChromaArrayType != 0) {
    my_cb = ev_add(e, EV_DECISION, CTX_CBF_CHROMA + trafoDepth, 0, 0, trafoDepth == 0 ? -1 : slot_cb);
    my_cr = ev_add(e, EV_DECISION, CTX_CBF_CHROMA + trafoDepth, 0, 0, trafoDepth == 0 ? -1 : slot_cr);
    if (split && log2TrafoSize == 3) {
      /* the 4x4 chroma blocks of this 8x8 node are coded with child blkIdx 3 but their cbf lives
         here: analyse chroma first (chroma prediction never depends on luma) */
      chroma_here_early = 1;
      /* if a cu_qp_delta is still pending it will be coded (in child 0) before these residuals */
      int pend = d->p->cu_qp_delta_enabled_flag && !d->IsCuQpDeltaCoded;
      int sv_delta = d->CuQpDeltaVal, sv_qp = d->cur_qp_y;
      if (pend) { d->CuQpDeltaVal = e->qg_delta; set_qp_y(d); }
      out.cbf_cb = analyse_tb(e, x0 / 2, y0 / 2, 2, 1, cu->chroma_mode, levCb, &tsCb);
      out.cbf_cr = analyse_tb(e, x0 / 2, y0 / 2, 2, 2, cu->chroma_mode, levCr, &tsCr);
      if (pend) { d->CuQpDeltaVal = sv_delta; d->cur_qp_y = sv_qp; }
    }
  }
  if (split) {
    int x1 = x0 + (1 << (log2TrafoSize - 1)), y1 = y0 + (1 << (log2TrafoSize - 1));
    if (chroma_here_early) {
      e->ev[my_cb].val = out.cbf_cb; e->ev[my_cr].val = out.cbf_cr;
      /* children are 4x4 luma leaves; child 3 emits the chroma residuals */
      for (int k = 0; k < 4; k++) {
        int xx = (k & 1) ? x1 : x0, yy = (k & 2) ? y1 : y0;
        /* leaf: cbf_luma always coded for intra */
        int32_t levY[16]; int tsY;
        int mode = d->m_ipm[(yy >> 2) * d->mw + (xx >> 2)];
        /* cu_qp_delta is needed before the first residual of the QG: decide from chroma cbf early */
        int cbfChroma = out.cbf_cb || out.cbf_cr;
        /* luma analysis needs the final QP; if a delta will be coded in this TU it is coded before
           the residual, so apply it now when (luma cbf || chroma cbf) — luma cbf is unknown until
           quantised, so pre-apply the delta whenever it is still pending: harmless if nothing is
           coded afterwards because QpY is restored below */
        int pending = d->p->cu_qp_delta_enabled_flag && !d->IsCuQpDeltaCoded;
        int saved_delta = d->CuQpDeltaVal, saved_qp = d->cur_qp_y;
        if (pending) { d->CuQpDeltaVal = e->qg_delta; set_qp_y(d); }
        int cbfY = analyse_tb(e, xx, yy, 2, 0, mode, levY, &tsY);
        if (pending) { d->CuQpDeltaVal = saved_delta; d->cur_qp_y = saved_qp; }
        EV_D(CTX_CBF_LUMA + 0, cbfY); /* trafoDepth > 0 here */
        if (cbfY || cbfChroma) enc_cu_qp_delta(e);
        if (cbfY) emit_residual(e, levY, 2, 0, mode, tsY);
        if (k == 3) {
          if (out.cbf_cb) emit_residual(e, levCb, 2, 1, cu->chroma_mode, tsCb);
          if (out.cbf_cr) emit_residual(e, levCr, 2, 2, cu->chroma_mode, tsCr);
        }
        mark_tu(d, cu, xx, yy, 2, cbfY, k == 3 ? out.cbf_cb : 0, k == 3 ? out.cbf_cr : 0);
      }
      return out;
    }
    ChromaCbf c0 = enc_transform_tree(e, cu, x0, y0, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 0, my_cb, my_cr, 0, 0);
    ChromaCbf c1 = enc_transform_tree(e, cu, x1, y0, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 1, my_cb, my_cr, 0, 0);
    ChromaCbf c2 = enc_transform_tree(e, cu, x0, y1, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 2, my_cb, my_cr, 0, 0);
    ChromaCbf c3 = enc_transform_tree(e, cu, x1, y1, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 3, my_cb, my_cr, 0, 0);
    out.cbf_cb = c0.cbf_cb | c1.cbf_cb | c2.cbf_cb | c3.cbf_cb;
    out.cbf_cr = c0.cbf_cr | c1.cbf_cr | c2.cbf_cr | c3.cbf_cr;
    if (my_cb >= 0) { e->ev[my_cb].val = out.cbf_cb; e->ev[my_cr].val = out.cbf_cr; }
    return out;
  }
  /* leaf transform unit with log2TrafoSize >= 3 (4x4 leaves only occur under an 8x8 split node,
     handled above) or a 4x4 leaf when chroma is absent */
  (void)xBase; (void)yBase; (void)blkIdx; (void)pre_cb; (void)pre_cr;
  int32_t levY[32 * 32];
  int tsY = 0;
  int mode = d->m_ipm[(y0 >> 2) * d->mw + (x0 >> 2)];
  int pending = d->p->cu_qp_delta_enabled_flag && !d->IsCuQpDeltaCoded;
  int saved_delta = d->CuQpDeltaVal, saved_qp = d->cur_qp_y;
  if (pending) { d->CuQpDeltaVal = e->qg_delta; set_qp_y(d); }
  int cbfY = analyse_tb(e, x0, y0, log2TrafoSize, 0, mode, levY, &tsY);
  if (log2TrafoSize > 2 && ChromaArrayType != 0) {
    out.cbf_cb = analyse_tb(e, x0 / 2, y0 / 2, log2TrafoSize - 1, 1, cu->chroma_mode, levCb, &tsCb);
    out.cbf_cr = analyse_tb(e, x0 / 2, y0 / 2, log2TrafoSize - 1, 2, cu->chroma_mode, levCr, &tsCr);
    e->ev[my_cb].val = out.cbf_cb; e->ev[my_cr].val = out.cbf_cr;
  }
  if (pending) { d->CuQpDeltaVal = saved_delta; d->cur_qp_y = saved_qp; }
  /* 7.3.8.8: an inter coding unit's root leaf without chroma coefficients does not code cbf_luma (inferred 1; the caller turns a unit
     whose blocks all came out empty into rqt_root_cbf 0 / a skipped unit) */
  if (!cu->inter || trafoDepth != 0 || out.cbf_cb || out.cbf_cr) EV_D(CTX_CBF_LUMA + (trafoDepth == 0 ? 1 : 0), cbfY);
  if (cbfY || out.cbf_cb || out.cbf_cr) enc_cu_qp_delta(e);
  if (cbfY) emit_residual(e, levY, log2TrafoSize, 0, mode, tsY);
  if (out.cbf_cb) emit_residual(e, levCb, log2TrafoSize - 1, 1, cu->chroma_mode, tsCb);
  if (out.cbf_cr) emit_residual(e, levCr, log2TrafoSize - 1, 2, cu->chroma_mode, tsCr);
  mark_tu(d, cu, x0, y0, log2TrafoSize, cbfY, out.cbf_cb, out.cbf_cr);
  return out;
}

static int pick_luma_mode(Enc* e, int xPb, int yPb, int nPb)
{
  Dec* d = e->d;
  if (e->prm.stress) return (int)(rnd(e) % 35);
  static const uint8_t cands[] = {0, 1, 10, 26, 2, 6, 14, 18, 22, 30, 34};
  int best = 0; int64_t bestSad = -1;
  int nc = (int)sizeof(cands) + 2;
  /* prediction at PB size is only defined up to 32; evaluate on the (clamped) first transform block */
  int log2n = 0; while ((1 << log2n) < Min(nPb, 1 << d->s->log2_max_tb)) log2n++;
  int n = 1 << log2n;
  for (int k = 0; k < nc; k++) {
    int m = k < (int)sizeof(cands) ? cands[k] : (int)(rnd(e) % 35);
    intra_predict_block(d, xPb, yPb, log2n, 0, m);
    int64_t sad = 0;
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++) {
        int df = (int)e->src[0][(yPb + y) * d->W + xPb + x] - (int)d->rec[0][(yPb + y) * d->W + xPb + x];
        sad += df < 0 ? -df : df;
      }
    if (bestSad < 0 || sad < bestSad) { bestSad = sad; best = m; }
  }
  return best;
}

#include "hevc_testenc_inter.c"

static void enc_coding_unit(Enc* e, int x0, int y0, int log2CbSize, int cqtDepth)
{
  Dec* d = e->d; const SPS* s = d->s; const PPS* p = d->p;
  int nCbS = 1 << log2CbSize;
  CuCtx cu; memset(&cu, 0, sizeof(cu));
  cu.xCb = x0; cu.yCb = y0; cu.log2CbSize = log2CbSize;
  d->cu_transquant_bypass_flag = 0;
  if (p->transquant_bypass_enabled_flag) {
    d->cu_transquant_bypass_flag = rnd_pct(e, e->prm.lossless_pct);
    EV_D(CTX_CU_TQ_BYPASS, d->cu_transquant_bypass_flag);
  }
  d->cu_pred_inter = 0;
  if (d->sh->slice_type != 2) {   /* P / B slice: cu_skip_flag, pred_mode_flag (7.3.8.5) */
    int ctxInc = 0;
    if (available_z(d, x0, y0, x0 - 1, y0) && d->m_pred[(y0 >> 2) * d->mw + ((x0 - 1) >> 2)] == 2) ctxInc++;
    if (available_z(d, x0, y0, x0, y0 - 1) && d->m_pred[((y0 - 1) >> 2) * d->mw + (x0 >> 2)] == 2) ctxInc++;
    unsigned r = rnd(e) % 100;
    int cu_skip = (int)r < e->prm.inter_skip_pct, intra = !cu_skip && (int)r < e->prm.inter_skip_pct + e->prm.inter_intra_pct;
    int ev_skip = EV_D(CTX_SKIP_FLAG + ctxInc, cu_skip), ev_pred = -1;
    if (!cu_skip) ev_pred = EV_D(CTX_PRED_MODE, intra);
    if (!intra) { enc_inter_coding_unit(e, &cu, x0, y0, log2CbSize, cqtDepth, cu_skip, ev_skip, ev_pred); return; }
  }
  int PartMode = 0;
  if (log2CbSize == s->log2_min_cb) {
    int want = e->prm.stress ? rnd_pct(e, 50) : (block_variance(e, x0, y0, nCbS) > 100 && rnd_pct(e, 50));
    if (log2CbSize == 3 && s->log2_min_tb > 2) want = 0;
    PartMode = want;
    EV_D(CTX_PART_MODE, PartMode ? 0 : 1);
  }
  int pcm_flag = 0;
  if (PartMode == 0 && s->pcm_enabled_flag && log2CbSize >= s->log2_min_pcm_cb && log2CbSize <= s->log2_max_pcm_cb) {
    pcm_flag = rnd_pct(e, e->prm.pcm_pct);
    ev_add(e, EV_TERMINATE, 0, pcm_flag, 0, -1);
  }
  set_qp_y(d);
  int u0x = x0 >> 2, u0y = y0 >> 2, nu = nCbS >> 2;
  for (int j = 0; j < nu; j++)
    for (int i = 0; i < nu; i++) {
      int idx = (u0y + j) * d->mw + u0x + i;
      d->m_log2_cb[idx] = (uint8_t)log2CbSize;
      d->m_ctdepth[idx] = (uint8_t)cqtDepth;
      d->m_flags[idx] = (uint8_t)((d->cu_transquant_bypass_flag ? 0x08 : 0) | (pcm_flag ? 0x10 : 0));
      d->m_decoded[idx] = 1;
      d->m_ipm[idx] = 1;
      if (d->m_pred) { d->m_pred[idx] = 0; d->mf_ref[2 * idx] = d->mf_ref[2 * idx + 1] = -1; d->mf_poc[2 * idx] = d->mf_poc[2 * idx + 1] = 0; memset(d->mf_mv + 4 * idx, 0, 4 * sizeof(int16_t)); }
    }
  if (pcm_flag) {
    size_t start = e->pcm_bits;
    for (int cIdx = 0; cIdx < (s->chroma_format_idc ? 3 : 1); cIdx++) {
      int csw = cIdx ? d->subw : 1, csh = cIdx ? d->subh : 1;
      int n = nCbS / csw, nh = nCbS / csh, xs = x0 / csw, ys = y0 / csh;
      int depth = cIdx ? s->pcm_bit_depth_chroma : s->pcm_bit_depth_luma;
      int bd = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
      int stride = cIdx ? d->Wc : d->W;
      for (int y = 0; y < nh; y++)
        for (int x = 0; x < n; x++) {
          unsigned v = e->src[cIdx][(ys + y) * stride + xs + x] >> (bd - depth);
          d->rec[cIdx][(ys + y) * stride + xs + x] = (uint16_t)(v << (bd - depth));
          for (int b = depth - 1; b >= 0; b--) {
            if ((e->pcm_bits >> 3) >= e->pcm_cap) {
              size_t nc = e->pcm_cap ? e->pcm_cap * 2 : 65536;
              e->pcm_blob = (uint8_t*)realloc(e->pcm_blob, nc);
              e->pcm_cap = nc;
            }
            if ((e->pcm_bits & 7) == 0) e->pcm_blob[e->pcm_bits >> 3] = 0;
            if ((v >> b) & 1) e->pcm_blob[e->pcm_bits >> 3] |= (uint8_t)(0x80 >> (e->pcm_bits & 7));
            e->pcm_bits++;
          }
        }
    }
    ev_add(e, EV_PCM, 0, (int)start, (int)(e->pcm_bits - start), -1);
    mark_tu(d, &cu, x0, y0, log2CbSize, 0, 0, 0);
    for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) {
      int idx = (u0y + j) * d->mw + u0x + i;
      d->m_ipmc[idx] = 1; d->m_qp[idx] = (int8_t)d->cur_qp_y;
    }
    d->last_qp_y = d->cur_qp_y;
    return;
  }
  int pbOffset = PartMode == 1 ? nCbS / 2 : nCbS;
  int nPart = PartMode == 1 ? 2 : 1;
  int prev_flag[4], mpm_idx[4], rem_mode[4], modes[4];
  /* decide + derive in partition order (later partitions see earlier ones as neighbours) */
  for (int j = 0; j < nPart; j++)
    for (int i = 0; i < nPart; i++) {
      int k = j * 2 + i;
      int xPb = x0 + i * pbOffset, yPb = y0 + j * pbOffset;
      int candA = cand_mode(d, xPb, yPb, xPb - 1, yPb, 0);
      int candB = cand_mode(d, xPb, yPb, xPb, yPb - 1, 1);
      int cl[3];
      if (candA == candB) {
        if (candA < 2) { cl[0] = 0; cl[1] = 1; cl[2] = 26; }
        else { cl[0] = candA; cl[1] = 2 + ((candA + 29) % 32); cl[2] = 2 + ((candA - 2 + 1) % 32); }
      } else {
        cl[0] = candA; cl[1] = candB;
        if (candA != 0 && candB != 0) cl[2] = 0; else if (candA != 1 && candB != 1) cl[2] = 1; else cl[2] = 26;
      }
      int mode = pick_luma_mode(e, xPb, yPb, pbOffset);
      if (!e->prm.stress && rnd_pct(e, 30)) mode = cl[rnd(e) % 3]; /* exercise the MPM path */
      modes[k] = mode;
      prev_flag[k] = 0; mpm_idx[k] = 0; rem_mode[k] = 0;
      for (int q = 0; q < 3; q++) if (cl[q] == mode) { prev_flag[k] = 1; mpm_idx[k] = q; break; }
      if (!prev_flag[k]) {
        int t;
        if (cl[0] > cl[1]) { t = cl[0]; cl[0] = cl[1]; cl[1] = t; }
        if (cl[0] > cl[2]) { t = cl[0]; cl[0] = cl[2]; cl[2] = t; }
        if (cl[1] > cl[2]) { t = cl[1]; cl[1] = cl[2]; cl[2] = t; }
        int r = mode;
        for (int q = 2; q >= 0; q--) if (r > cl[q]) r--;
        rem_mode[k] = r;
      }
      int pu = pbOffset >> 2;
      for (int jj = 0; jj < pu; jj++) for (int ii = 0; ii < pu; ii++)
        d->m_ipm[((yPb >> 2) + jj) * d->mw + (xPb >> 2) + ii] = (uint8_t)mode;
    }
  for (int j = 0; j < nPart; j++) for (int i = 0; i < nPart; i++) EV_D(CTX_PREV_INTRA_LUMA, prev_flag[j * 2 + i]);
  for (int j = 0; j < nPart; j++)
    for (int i = 0; i < nPart; i++) {
      int k = j * 2 + i;
      if (prev_flag[k]) { if (mpm_idx[k] == 0) EV_B(0); else { EV_B(1); EV_B(mpm_idx[k] == 2); } }
      else EV_BB(rem_mode[k], 5);
    }
  (void)modes;
  int chroma_mode = 1;
  for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) d->m_ipmc[(u0y + j) * d->mw + u0x + i] = 1;
  if (s->chroma_format_idc) {
    /* one intra_chroma_pred_mode per CU, or one per partition for an NxN CU of a 4:4:4 picture (7.3.8.5) */
    int ncp = (s->chroma_format_idc == 3 && PartMode == 1) ? 2 : 1;
    int cpb = ncp == 2 ? nCbS / 2 : nCbS;
    for (int j = 0; j < ncp; j++)
      for (int i = 0; i < ncp; i++) {
        int icpm = e->prm.stress ? (int)(rnd(e) % 5) : (rnd_pct(e, 70) ? 4 : (int)(rnd(e) % 4));
        if (icpm == 4) EV_D(CTX_INTRA_CHROMA, 0);
        else { EV_D(CTX_INTRA_CHROMA, 1); EV_BB(icpm, 2); }
        int xP = x0 + i * cpb, yP = y0 + j * cpb;
        int lm = d->m_ipm[(yP >> 2) * d->mw + (xP >> 2)];
        static const uint8_t tab[4] = {0, 26, 10, 1};
        int m = icpm == 4 ? lm : ((tab[icpm] == lm) ? 34 : tab[icpm]);
        if (s->chroma_format_idc == 2) {   /* Table 8-3 */
          static const uint8_t map422[35] = {0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 11, 13, 15, 16, 18, 19, 20, 21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27,
                                             28, 28, 29, 29, 30, 31};
          m = map422[m];
        }
        if (i == 0 && j == 0) chroma_mode = m;
        for (int jj = 0; jj < (cpb >> 2); jj++) for (int ii = 0; ii < (cpb >> 2); ii++)
          d->m_ipmc[((yP >> 2) + jj) * d->mw + (xP >> 2) + ii] = (uint8_t)m;
      }
  }
  cu.chroma_mode = chroma_mode;
  cu.IntraSplitFlag = PartMode == 1;
  cu.MaxTrafoDepth = s->max_transform_hierarchy_depth_intra + cu.IntraSplitFlag;
  enc_transform_tree(e, &cu, x0, y0, x0, y0, log2CbSize, 0, 0, -1, -1, 0, 0);
  set_qp_y(d);
  for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) d->m_qp[(u0y + j) * d->mw + u0x + i] = (int8_t)d->cur_qp_y;
  d->last_qp_y = d->cur_qp_y;
}

static void enc_coding_quadtree(Enc* e, int x0, int y0, int log2CbSize, int cqtDepth)
{
  Dec* d = e->d; const SPS* s = d->s; const PPS* p = d->p;
  int split;
  if (x0 + (1 << log2CbSize) <= d->W && y0 + (1 << log2CbSize) <= d->H && log2CbSize > s->log2_min_cb) {
    int ctxInc = 0;
    if (available_z(d, x0, y0, x0 - 1, y0) && d->m_ctdepth[(y0 >> 2) * d->mw + ((x0 - 1) >> 2)] > cqtDepth) ctxInc++;
    if (available_z(d, x0, y0, x0, y0 - 1) && d->m_ctdepth[((y0 - 1) >> 2) * d->mw + (x0 >> 2)] > cqtDepth) ctxInc++;
    if (e->prm.stress) split = rnd_pct(e, 55);
    else {
      int var = block_variance(e, x0, y0, 1 << log2CbSize);
      int thr = log2CbSize >= 6 ? 30 : log2CbSize == 5 ? 80 : 200;
      split = var > thr;
    }
    EV_D(CTX_SPLIT_CU + ctxInc, split);
  } else split = log2CbSize > s->log2_min_cb;
  if (p->cu_qp_delta_enabled_flag && log2CbSize >= s->log2_ctb - p->diff_cu_qp_delta_depth) {
    d->IsCuQpDeltaCoded = 0; d->CuQpDeltaVal = 0;
    derive_qp_pred(d, x0, y0);
    /* choose the delta of this quantisation group so that the resulting QpY stays in [1, 50] */
    int want = (int)(rnd(e) % 7) - 3;
    int q = d->qPY_PRED + want;
    if (q < 1 || q > 50) want = 0;
    e->qg_delta = rnd_pct(e, 60) ? want : 0;
  }
  if (split) {
    int x1 = x0 + (1 << (log2CbSize - 1)), y1 = y0 + (1 << (log2CbSize - 1));
    enc_coding_quadtree(e, x0, y0, log2CbSize - 1, cqtDepth + 1);
    if (x1 < d->W) enc_coding_quadtree(e, x1, y0, log2CbSize - 1, cqtDepth + 1);
    if (y1 < d->H) enc_coding_quadtree(e, x0, y1, log2CbSize - 1, cqtDepth + 1);
    if (x1 < d->W && y1 < d->H) enc_coding_quadtree(e, x1, y1, log2CbSize - 1, cqtDepth + 1);
  } else enc_coding_unit(e, x0, y0, log2CbSize, cqtDepth);
}

static void enc_sao(Enc* e, int rx, int ry)
{
  Dec* d = e->d; const SPS* s = d->s;
  int ctb = ry * d->ctbW + rx;
  int merge_left = 0, merge_up = 0;
  if (rx > 0) {
    int leftCtbInSliceSeg = d->CtbAddrInRs > d->sh->SliceAddrRs;
    int leftCtbInTile = d->TileId[d->CtbAddrInTs] == d->TileId[d->CtbAddrRsToTs[d->CtbAddrInRs - 1]];
    if (leftCtbInSliceSeg && leftCtbInTile) { merge_left = rnd_pct(e, 25); EV_D(CTX_SAO_MERGE, merge_left); }
  }
  if (ry > 0 && !merge_left) {
    int upCtbInSliceSeg = (d->CtbAddrInRs - d->ctbW) >= d->sh->SliceAddrRs;
    int upCtbInTile = d->TileId[d->CtbAddrInTs] == d->TileId[d->CtbAddrRsToTs[d->CtbAddrInRs - d->ctbW]];
    if (upCtbInSliceSeg && upCtbInTile) { merge_up = rnd_pct(e, 25); EV_D(CTX_SAO_MERGE, merge_up); }
  }
  if (merge_left || merge_up) {
    int src = merge_left ? ctb - 1 : ctb - d->ctbW;
    memcpy(&d->sao_type[ctb * 3], &d->sao_type[src * 3], 3);
    memcpy(&d->sao_bc[ctb * 3], &d->sao_bc[src * 3], 3);
    memcpy(&d->sao_off[ctb * 12], &d->sao_off[src * 12], 12 * sizeof(int16_t));
    return;
  }
  for (int cIdx = 0; cIdx < (s->chroma_format_idc ? 3 : 1); cIdx++) {
    int on = cIdx == 0 ? d->sh->slice_sao_luma_flag : d->sh->slice_sao_chroma_flag;
    d->sao_type[ctb * 3 + cIdx] = 0; d->sao_bc[ctb * 3 + cIdx] = 0;
    for (int i = 0; i < 4; i++) d->sao_off[(ctb * 3 + cIdx) * 4 + i] = 0;
    if (!on) continue;
    int type;
    if (cIdx == 2) type = d->sao_type[ctb * 3 + 1];
    else {
      int r = (int)(rnd(e) % 100);
      type = r < 35 ? 0 : r < 60 ? 1 : 2;
      if (type == 0) EV_D(CTX_SAO_TYPE, 0); else { EV_D(CTX_SAO_TYPE, 1); EV_B(type == 2); }
    }
    d->sao_type[ctb * 3 + cIdx] = (uint8_t)type;
    if (!type) continue;
    int bitDepth = cIdx ? s->bit_depth_chroma : s->bit_depth_luma;
    int cMax = (1 << (Min(bitDepth, 10) - 5)) - 1;
    int absv[4], sign[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      absv[i] = e->prm.stress ? (int)(rnd(e) % (cMax + 1)) : (int)(rnd(e) % 4);
      for (int k = 0; k < absv[i]; k++) EV_B(1);
      if (absv[i] < cMax) EV_B(0);
    }
    if (type == 1) {
      for (int i = 0; i < 4; i++) if (absv[i]) { sign[i] = (int)(rnd(e) & 1); EV_B(sign[i]); }
      int band = (int)(rnd(e) % 32);
      d->sao_bc[ctb * 3 + cIdx] = (uint8_t)band;
      EV_BB(band, 5);
    } else {
      if (cIdx == 0) { int c = (int)(rnd(e) % 4); d->sao_bc[ctb * 3] = (uint8_t)c; EV_BB(c, 2); }
      else if (cIdx == 1) { int c = (int)(rnd(e) % 4); d->sao_bc[ctb * 3 + 1] = (uint8_t)c; EV_BB(c, 2); }
      else d->sao_bc[ctb * 3 + 2] = d->sao_bc[ctb * 3 + 1];
      sign[0] = sign[1] = 0; sign[2] = sign[3] = 1;
    }
    int log2OffsetScale = bitDepth - Min(bitDepth, 10);
    for (int i = 0; i < 4; i++)
      d->sao_off[(ctb * 3 + cIdx) * 4 + i] = (int16_t)((sign[i] ? -absv[i] : absv[i]) * (1 << log2OffsetScale));
  }
}

/* ------------------------------------------------------------------------------------------ */
/* NAL assembly                                                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t* p; size_t n, cap; } Bytes;
static void by_push(Bytes* b, const uint8_t* s, size_t n)
{
  if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 1024; b->p = (uint8_t*)realloc(b->p, b->cap); }
  memcpy(b->p + b->n, s, n); b->n += n;
}
/* escapes payload (adds emulation prevention) and appends [len][hdr0 hdr1 payload] */
static size_t escape_into(Bytes* out, const uint8_t* s, size_t n)
{
  size_t start = out->n; int zeros = 0;
  for (size_t i = 0; i < n; i++) {
    if (zeros >= 2 && s[i] <= 3) { uint8_t e3 = 3; by_push(out, &e3, 1); zeros = 0; }
    by_push(out, &s[i], 1);
    zeros = s[i] == 0 ? zeros + 1 : 0;
  }
  return out->n - start;
}
static void put_nal(Bytes* out, int nal_type, const uint8_t* rbsp, size_t n)
{
  Bytes tmp = {0, 0, 0};
  uint8_t hdr[2] = {(uint8_t)(nal_type << 1), 1};
  by_push(&tmp, hdr, 2);
  escape_into(&tmp, rbsp, n);
  uint8_t len[4] = {(uint8_t)(tmp.n >> 24), (uint8_t)(tmp.n >> 16), (uint8_t)(tmp.n >> 8), (uint8_t)tmp.n};
  by_push(out, len, 4);
  by_push(out, tmp.p, tmp.n);
  free(tmp.p);
}

static void write_ptl(BW* w, int bit_depth, int chroma)
{
  int idc = chroma != 1 ? 4 : bit_depth > 8 ? 2 : 1;   /* 4:0:0, 4:2:2 and 4:4:4 are format-range-extension profiles */
  bw_u(w, 0, 2); bw_u(w, 0, 1); bw_u(w, idc, 5);
  for (int i = 0; i < 32; i++) bw_put(w, i == idc || (idc == 1 && i == 2));
  bw_u(w, 1, 1); bw_u(w, 0, 1); bw_u(w, 0, 1); bw_u(w, 1, 1);
  for (int i = 0; i < 43; i++) bw_put(w, 0);
  bw_put(w, 0);
  bw_u(w, 186, 8);
}

/* parameter sets (filled as decoder structs, then written as VPS / SPS / PPS NAL units into `stream`) */
static void enc_parameter_sets(Enc* e, Bytes* pstream)
{
  Dec* d = e->d;
  const hevc_testenc_params* prm = &e->prm;
#define stream (*pstream)
  /* ---- parameter sets (filled as decoder structs, then written) ---- */
  SPS* s = &d->sps[0]; PPS* p = &d->pps[0];
  int minCb = 1 << prm->log2_min_cb;
  s->chroma_format_idc = prm->chroma_format_idc;
  s->pic_width = (prm->width + minCb - 1) / minCb * minCb;
  s->pic_height = (prm->height + minCb - 1) / minCb * minCb;
  int subcw = (prm->chroma_format_idc == 1 || prm->chroma_format_idc == 2) ? 2 : 1, subch = prm->chroma_format_idc == 1 ? 2 : 1;
  if ((s->pic_width - prm->width) % subcw || (s->pic_height - prm->height) % subch) fail(d, "odd picture size needs 4:0:0 or 4:4:4");
  s->conf_win_right = (s->pic_width - prm->width) / subcw;
  s->conf_win_bottom = (s->pic_height - prm->height) / subch;
  s->bit_depth_luma = s->bit_depth_chroma = prm->bit_depth;
  s->log2_max_poc_lsb = 8;
  s->log2_min_cb = prm->log2_min_cb; s->log2_ctb = prm->log2_ctb;
  s->log2_min_tb = prm->log2_min_tb; s->log2_max_tb = prm->log2_max_tb;
  s->max_transform_hierarchy_depth_inter = prm->max_transform_hierarchy_depth_inter;
  s->max_transform_hierarchy_depth_intra = prm->max_transform_hierarchy_depth_intra;
  s->scaling_list_enabled_flag = prm->scaling_list ? 1 : 0;
  if (prm->chroma_format_idc < 0 || prm->chroma_format_idc > 3) fail(d, "chroma_format_idc must be 0 .. 3");
  scaling_list_default(&s->sl); scaling_list_default(&p->sl);
  s->amp_enabled_flag = prm->amp ? 1 : 0; s->sao_enabled_flag = prm->sao;
  s->pcm_enabled_flag = prm->pcm_pct > 0;
  if (s->pcm_enabled_flag) {
    s->pcm_bit_depth_luma = prm->bit_depth - 1; s->pcm_bit_depth_chroma = prm->bit_depth;
    s->log2_min_pcm_cb = Max(3, prm->log2_min_cb); s->log2_max_pcm_cb = Min(5, prm->log2_ctb);
    if (s->log2_min_pcm_cb > s->log2_max_pcm_cb) s->log2_min_pcm_cb = s->log2_max_pcm_cb;
    s->pcm_loop_filter_disabled_flag = prm->pcm_loop_filter_disabled;
  }
  s->strong_intra_smoothing_enabled_flag = prm->strong_intra_smoothing;
  s->colour_primaries = 2; s->transfer_characteristics = 2; s->matrix_coeffs = 2; s->video_full_range_flag = 0;
  s->PicWidthInCtbsY = (s->pic_width + (1 << s->log2_ctb) - 1) >> s->log2_ctb;
  s->PicHeightInCtbsY = (s->pic_height + (1 << s->log2_ctb) - 1) >> s->log2_ctb;
  if (s->log2_max_tb > Min(5, s->log2_ctb) || s->log2_min_tb >= s->log2_min_cb || s->log2_min_cb > s->log2_ctb ||
      s->log2_ctb < 4 || s->log2_ctb > 6 || s->log2_min_tb < 2 || s->log2_max_tb < s->log2_min_tb)
    fail(d, "inconsistent block size parameters");
  s->valid = 1;
  p->sps_id = 0;
  p->sign_data_hiding_enabled_flag = prm->sign_data_hiding;
  p->init_qp_minus26 = 0;
  p->constrained_intra_pred_flag = prm->constrained_intra_pred ? 1 : 0;
  p->transform_skip_enabled_flag = prm->transform_skip;
  p->cu_qp_delta_enabled_flag = prm->cu_qp_delta;
  p->diff_cu_qp_delta_depth = prm->cu_qp_delta ? Min(prm->diff_cu_qp_delta_depth, s->log2_ctb - s->log2_min_cb) : 0;
  p->pps_cb_qp_offset = prm->cb_qp_offset; p->pps_cr_qp_offset = prm->cr_qp_offset;
  p->transquant_bypass_enabled_flag = prm->lossless_pct > 0;
  p->num_tile_columns = Max(1, prm->tile_cols); p->num_tile_rows = Max(1, prm->tile_rows);
  if (p->num_tile_columns > s->PicWidthInCtbsY) p->num_tile_columns = s->PicWidthInCtbsY;
  if (p->num_tile_rows > s->PicHeightInCtbsY) p->num_tile_rows = s->PicHeightInCtbsY;
  p->tiles_enabled_flag = p->num_tile_columns * p->num_tile_rows > 1;
  p->uniform_spacing_flag = 1;
  p->loop_filter_across_tiles_enabled_flag = prm->loop_filter_across_tiles;
  p->entropy_coding_sync_enabled_flag = prm->wpp;
  p->pps_loop_filter_across_slices_enabled_flag = prm->loop_filter_across_slices;
  p->deblocking_filter_control_present_flag = 1;
  p->deblocking_filter_override_enabled_flag = 0;
  p->pps_deblocking_filter_disabled_flag = prm->deblock_disable;
  p->pps_beta_offset_div2 = prm->beta_offset_div2; p->pps_tc_offset_div2 = prm->tc_offset_div2;
  p->log2_parallel_merge_level = prm->parallel_merge_level >= 2 ? Min(prm->parallel_merge_level, s->log2_ctb) : 2;
  p->num_ref_idx_l0_default_active = 1; p->num_ref_idx_l1_default_active = 1;
  p->weighted_pred_flag = p->weighted_bipred_flag = d->seq_mode && prm->weighted_pred ? 1 : 0;
  p->cabac_init_present_flag = prm->cabac_init_present ? 1 : 0;
  p->lists_modification_present_flag = prm->lists_modification ? 1 : 0;
  p->valid = 1;
  d->s = s; d->p = p;

  BW w; memset(&w, 0, sizeof(w));
  /* VPS 7.3.2.1 */
  bw_u(&w, 0, 4); bw_u(&w, 3, 2); bw_u(&w, 0, 6); bw_u(&w, 0, 3); bw_u(&w, 1, 1); bw_u(&w, 0xffff, 16);
  write_ptl(&w, prm->bit_depth, prm->chroma_format_idc);
  /* vps_max_dec_pic_buffering_minus1, vps_max_num_reorder_pics (B pictures are coded after the anchor that follows them), vps_max_latency_increase_plus1 */
  const int dpb_minus1 = d->seq_mode ? Min(15, Max(1, prm->inter_num_refs) + (prm->b_frames > 0 ? 2 : 0)) : 0, reorder = d->seq_mode && prm->b_frames > 0 ? 1 : 0;
  bw_u(&w, 1, 1); bw_ue(&w, dpb_minus1); bw_ue(&w, reorder); bw_ue(&w, 0);
  bw_u(&w, 0, 6); bw_ue(&w, 0); bw_u(&w, 0, 1); bw_u(&w, 0, 1);
  bw_trailing(&w);
  put_nal(&stream, 32, w.p, w.nbits >> 3);
  bw_free(&w);
  /* SPS 7.3.2.2 */
  bw_u(&w, 0, 4); bw_u(&w, 0, 3); bw_u(&w, 1, 1);
  write_ptl(&w, prm->bit_depth, prm->chroma_format_idc);
  bw_ue(&w, 0); bw_ue(&w, s->chroma_format_idc);
  if (s->chroma_format_idc == 3) bw_u(&w, 0, 1);   /* separate_colour_plane_flag */
  bw_ue(&w, s->pic_width); bw_ue(&w, s->pic_height);
  int cw = s->conf_win_right || s->conf_win_bottom;
  bw_u(&w, cw, 1);
  if (cw) { bw_ue(&w, 0); bw_ue(&w, s->conf_win_right); bw_ue(&w, 0); bw_ue(&w, s->conf_win_bottom); }
  bw_ue(&w, s->bit_depth_luma - 8); bw_ue(&w, s->bit_depth_chroma - 8);
  bw_ue(&w, s->log2_max_poc_lsb - 4);
  bw_u(&w, 1, 1); bw_ue(&w, dpb_minus1); bw_ue(&w, reorder); bw_ue(&w, 0);
  bw_ue(&w, s->log2_min_cb - 3); bw_ue(&w, s->log2_ctb - s->log2_min_cb);
  bw_ue(&w, s->log2_min_tb - 2); bw_ue(&w, s->log2_max_tb - s->log2_min_tb);
  bw_ue(&w, s->max_transform_hierarchy_depth_inter); bw_ue(&w, s->max_transform_hierarchy_depth_intra);
  bw_u(&w, s->scaling_list_enabled_flag, 1);
  if (s->scaling_list_enabled_flag) {   /* scaling_list: 1 = default lists, 2 = explicit lists in the SPS, 3 = in the PPS */
    bw_u(&w, prm->scaling_list == 2, 1);
    if (prm->scaling_list == 2) write_scaling_list_data(d, &w, &s->sl, (unsigned)prm->seed + 17u);
  }
  bw_u(&w, s->amp_enabled_flag, 1); bw_u(&w, s->sao_enabled_flag, 1); bw_u(&w, s->pcm_enabled_flag, 1);
  if (s->pcm_enabled_flag) {
    bw_u(&w, s->pcm_bit_depth_luma - 1, 4); bw_u(&w, s->pcm_bit_depth_chroma - 1, 4);
    bw_ue(&w, s->log2_min_pcm_cb - 3); bw_ue(&w, s->log2_max_pcm_cb - s->log2_min_pcm_cb);
    bw_u(&w, s->pcm_loop_filter_disabled_flag, 1);
  }
  s->sps_temporal_mvp_enabled_flag = d->seq_mode && prm->temporal_mvp ? 1 : 0;
  bw_ue(&w, 0);                                   /* num_short_term_ref_pic_sets */
  s->long_term_ref_pics_present_flag = d->seq_mode && prm->long_term_ref ? 1 : 0;
  s->num_long_term_ref_pics_sps = prm->long_term_ref == 3 ? 2 : 0;   /* two candidates in the SPS: LSBs 5 (never used) and 0 (the IDR picture) */
  bw_u(&w, s->long_term_ref_pics_present_flag, 1);
  if (s->long_term_ref_pics_present_flag) {
    bw_ue(&w, s->num_long_term_ref_pics_sps);
    for (int i = 0; i < s->num_long_term_ref_pics_sps; i++) {
      s->lt_ref_pic_poc_lsb_sps[i] = i == 0 ? 5 : 0; s->used_by_curr_pic_lt_sps_flag[i] = 1;
      bw_u(&w, s->lt_ref_pic_poc_lsb_sps[i], s->log2_max_poc_lsb); bw_u(&w, 1, 1);
    }
  }
  bw_u(&w, s->sps_temporal_mvp_enabled_flag, 1); bw_u(&w, s->strong_intra_smoothing_enabled_flag, 1);
  if (prm->vui_matrix >= 0) {
    s->colour_primaries = prm->vui_primaries; s->transfer_characteristics = prm->vui_transfer;
    s->matrix_coeffs = prm->vui_matrix; s->video_full_range_flag = prm->vui_full_range;
    bw_u(&w, 1, 1);
    bw_u(&w, 0, 1); bw_u(&w, 0, 1);
    bw_u(&w, 1, 1); bw_u(&w, 5, 3); bw_u(&w, s->video_full_range_flag, 1); bw_u(&w, 1, 1);
    bw_u(&w, s->colour_primaries, 8); bw_u(&w, s->transfer_characteristics, 8); bw_u(&w, s->matrix_coeffs, 8);
    bw_u(&w, 0, 1); bw_u(&w, 0, 1); bw_u(&w, 0, 1); bw_u(&w, 0, 1); bw_u(&w, 0, 1); bw_u(&w, 0, 1); bw_u(&w, 0, 1);
  } else bw_u(&w, 0, 1);
  bw_u(&w, 0, 1);
  bw_trailing(&w);
  put_nal(&stream, 33, w.p, w.nbits >> 3);
  bw_free(&w);
  /* PPS 7.3.2.3 */
  p->dependent_slice_segments_enabled_flag = prm->dependent_segments > 1;
  p->output_flag_present_flag = d->seq_mode && prm->hidden_poc > 0;
  bw_ue(&w, 0); bw_ue(&w, 0); bw_u(&w, p->dependent_slice_segments_enabled_flag, 1); bw_u(&w, p->output_flag_present_flag, 1); bw_u(&w, 0, 3);
  bw_u(&w, p->sign_data_hiding_enabled_flag, 1); bw_u(&w, p->cabac_init_present_flag, 1); bw_ue(&w, 0); bw_ue(&w, 0);
  bw_se(&w, p->init_qp_minus26); bw_u(&w, p->constrained_intra_pred_flag, 1); bw_u(&w, p->transform_skip_enabled_flag, 1);
  bw_u(&w, p->cu_qp_delta_enabled_flag, 1);
  if (p->cu_qp_delta_enabled_flag) bw_ue(&w, p->diff_cu_qp_delta_depth);
  bw_se(&w, p->pps_cb_qp_offset); bw_se(&w, p->pps_cr_qp_offset); bw_u(&w, 0, 1);
  bw_u(&w, p->weighted_pred_flag, 1); bw_u(&w, p->weighted_bipred_flag, 1);
  bw_u(&w, p->transquant_bypass_enabled_flag, 1); bw_u(&w, p->tiles_enabled_flag, 1); bw_u(&w, p->entropy_coding_sync_enabled_flag, 1);
  if (p->tiles_enabled_flag) {
    bw_ue(&w, p->num_tile_columns - 1); bw_ue(&w, p->num_tile_rows - 1); bw_u(&w, 1, 1);
    bw_u(&w, p->loop_filter_across_tiles_enabled_flag, 1);
  } else p->loop_filter_across_tiles_enabled_flag = 1;
  bw_u(&w, p->pps_loop_filter_across_slices_enabled_flag, 1);
  bw_u(&w, 1, 1); bw_u(&w, 0, 1); bw_u(&w, p->pps_deblocking_filter_disabled_flag, 1);
  if (!p->pps_deblocking_filter_disabled_flag) { bw_se(&w, p->pps_beta_offset_div2); bw_se(&w, p->pps_tc_offset_div2); }
  p->pps_scaling_list_data_present_flag = prm->scaling_list == 3;
  bw_u(&w, p->pps_scaling_list_data_present_flag, 1);
  if (p->pps_scaling_list_data_present_flag) write_scaling_list_data(d, &w, &p->sl, (unsigned)prm->seed + 29u);
  bw_u(&w, p->lists_modification_present_flag, 1); bw_ue(&w, p->log2_parallel_merge_level - 2); bw_u(&w, 0, 1); bw_u(&w, 0, 1);
  bw_trailing(&w);
  put_nal(&stream, 34, w.p, w.nbits >> 3);
  bw_free(&w);
#undef stream
}

/* one picture of the coding-order plan (enc_run): an IDR intra picture, a P or a B picture */
static void enc_picture(Enc* e, const uint16_t* const planes[3], const PicPlan* plan, Bytes* pstream)
{
  Dec* d = e->d;
  const hevc_testenc_params* prm = &e->prm;
  const SPS* s = d->s; const PPS* p = d->p;
  uint16_t** src = e->src_alloc;
  BW w; memset(&w, 0, sizeof(w));
#define stream (*pstream)
  const int frame_idx = plan->poc;
  const int is_p = plan->slice_type != 2;      /* a P or B picture */
  const int is_b = plan->slice_type == 0;
  const int nal_type = plan->nal_type;
  e->frame_idx = frame_idx;
  setup_picture(d);
  StRps rps; memset(&rps, 0, sizeof(rps));
  rps.num_neg = plan->n_neg; rps.num_pos = plan->n_pos;
  for (int i = 0; i < plan->n_neg; i++) { rps.delta_s0[i] = plan->neg_poc[i] - plan->poc; rps.used_s0[i] = plan->neg_used[i]; }
  for (int i = 0; i < plan->n_pos; i++) { rps.delta_s1[i] = plan->pos_poc[i] - plan->poc; rps.used_s1[i] = plan->pos_used[i]; }
  rps.num_lt = plan->n_lt;
  for (int i = 0; i < plan->n_lt; i++) {
    const int MaxLsb = 1 << s->log2_max_poc_lsb;
    rps.lt_poc_lsb[i] = plan->lt_poc[i] & (MaxLsb - 1); rps.lt_used[i] = 1;
    rps.lt_msb_present[i] = prm->long_term_ref == 2;
    /* pocLt = PocLsbLt + PicOrderCntVal - DeltaPocMsbCycleLt * MaxLsb - (PicOrderCntVal & (MaxLsb - 1))  (8.3.2) */
    rps.lt_msb_cycle[i] = ((frame_idx - (frame_idx & (MaxLsb - 1))) - (plan->lt_poc[i] - rps.lt_poc_lsb[i])) / MaxLsb;
  }
  if (d->seq_mode) inter_begin_picture(d, nal_type, 0, frame_idx & 255, &rps);
  for (int c = 0; c < (s->chroma_format_idc ? 3 : 1); c++) {
    int W = c ? d->Wc : d->W, H = c ? d->Hc : d->H;
    int csw = c ? d->subw : 1, csh = c ? d->subh : 1;
    int sw = (prm->width + csw - 1) / csw, sh_ = (prm->height + csh - 1) / csh;
    src[c] = (uint16_t*)xcalloc(d, (size_t)W * H, sizeof(uint16_t));
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) src[c][y * W + x] = planes[c][(size_t)Min(y, sh_ - 1) * sw + Min(x, sw - 1)];
    e->src[c] = src[c];
  }
  uint8_t* seg_dependent = NULL;   /* per slice segment: 1 = dependent slice segment */
  /* slices: boundaries in tile-scan CTB addresses */
  int nsl = Max(1, prm->num_slices);
  int* slice_start = (int*)xcalloc(d, nsl + 1, sizeof(int));
  {
    int cnt = 0;
    if (p->tiles_enabled_flag) { /* slices hold complete tiles */
      int ntiles = p->num_tile_columns * p->num_tile_rows;
      if (nsl > ntiles) nsl = ntiles;
      int* tile_first = (int*)xcalloc(d, ntiles + 1, sizeof(int));
      int nt = 0;
      for (int ts = 0; ts < d->nCtb; ts++) if (ts == 0 || d->TileId[ts] != d->TileId[ts - 1]) tile_first[nt++] = ts;
      for (int k = 0; k < nsl; k++) slice_start[cnt++] = tile_first[(k * ntiles) / nsl];
      free(tile_first);
    } else {
      for (int k = 0; k < nsl; k++) {
        int a = (int)(((int64_t)k * d->nCtb) / nsl);
        if (p->entropy_coding_sync_enabled_flag) a = a / d->ctbW * d->ctbW; /* WPP: start at row starts */
        if (cnt && a <= slice_start[cnt - 1]) continue;
        slice_start[cnt++] = a;
      }
    }
    nsl = cnt; slice_start[nsl] = d->nCtb;
  }
  /* slice segments: every slice in dependent_segments parts; all but a slice's first are dependent slice segments (7.3.6.1).  With WPP the
     parts start at CTB row starts; with tiles the slices (complete tiles) are not split any further */
  if (prm->dependent_segments > 1 && !p->tiles_enabled_flag) {
    int* seg = (int*)xcalloc(d, (size_t)nsl * prm->dependent_segments + 2, sizeof(int));
    uint8_t* dep = (uint8_t*)xcalloc(d, (size_t)nsl * prm->dependent_segments + 2, 1);
    int cnt = 0;
    for (int si = 0; si < nsl; si++) {
      const int a0 = slice_start[si], a1 = slice_start[si + 1];
      for (int k = 0; k < prm->dependent_segments; k++) {
        int a = a0 + (int)(((int64_t)k * (a1 - a0)) / prm->dependent_segments);
        if (p->entropy_coding_sync_enabled_flag) a = a / d->ctbW * d->ctbW;
        if (a < a0) a = a0;
        if (cnt && a <= seg[cnt - 1]) continue;
        seg[cnt] = a; dep[cnt] = k > 0; cnt++;
      }
    }
    seg[cnt] = d->nCtb;
    free(slice_start);
    slice_start = seg; seg_dependent = dep; nsl = cnt;
  }
  int CtbSizeY = 1 << s->log2_ctb;
  SliceHdr cur_hdr; memset(&cur_hdr, 0, sizeof(cur_hdr));
  int list_mod[2] = {0, 0}, list_entries[2][16];
  uint8_t wp_lf[2][16], wp_cf[2][16]; int wp_dw[2][16][3], wp_do[2][16][3];   /* pred_weight_table as coded (flags, deltas) */
  memset(wp_lf, 0, sizeof(wp_lf)); memset(wp_cf, 0, sizeof(wp_cf)); memset(wp_dw, 0, sizeof(wp_dw)); memset(wp_do, 0, sizeof(wp_do));
  for (int si = 0; si < nsl; si++) {
    const int is_dep = seg_dependent ? seg_dependent[si] : 0;
    SliceHdr hdr; memset(&hdr, 0, sizeof(hdr));
    if (is_dep) goto segment_data;   /* a dependent slice segment continues the slice: its header fields are those of cur_hdr */
    hdr.first_slice_segment_in_pic_flag = si == 0;
    hdr.slice_segment_address = d->CtbAddrTsToRs[slice_start[si]];
    hdr.slice_type = plan->slice_type;
    if (is_p) {
      int total = d->n_st_curr_before + d->n_st_curr_after + d->n_lt_curr;
      hdr.num_ref_idx_l0_active = (si & 1) ? Min(15, total + 1) : total;   /* one more than there are pictures: the list wraps around (8.3.4) */
      if (is_b) hdr.num_ref_idx_l1_active = (si & 1) ? total : Min(15, total + 1);
      hdr.max_num_merge_cand = prm->max_merge_cand >= 1 && prm->max_merge_cand <= 5 ? prm->max_merge_cand : 5;
      hdr.cabac_init_flag = p->cabac_init_present_flag ? (int)((si + frame_idx) & 1) : 0;
      hdr.slice_temporal_mvp = s->sps_temporal_mvp_enabled_flag;
      hdr.mvd_l1_zero_flag = is_b && prm->mvd_l1_zero ? (int)((si + frame_idx) & 1) ^ 1 : 0;
      hdr.collocated_from_l0 = 1;
    }
    hdr.slice_sao_luma_flag = s->sao_enabled_flag; hdr.slice_sao_chroma_flag = s->sao_enabled_flag && s->chroma_format_idc;
    hdr.slice_qp_delta = prm->qp - 26 + (si % 3) - (si ? 1 : 0) * 0;
    if (26 + hdr.slice_qp_delta < 1) hdr.slice_qp_delta = -25;
    if (26 + hdr.slice_qp_delta > 50) hdr.slice_qp_delta = 24;
    hdr.slice_deblocking_filter_disabled_flag = p->pps_deblocking_filter_disabled_flag;
    hdr.slice_beta_offset_div2 = p->pps_beta_offset_div2; hdr.slice_tc_offset_div2 = p->pps_tc_offset_div2;
    hdr.slice_loop_filter_across_slices_enabled_flag = p->pps_loop_filter_across_slices_enabled_flag;
    int lf_flag_present = p->pps_loop_filter_across_slices_enabled_flag &&
                          (hdr.slice_sao_luma_flag || hdr.slice_sao_chroma_flag || !hdr.slice_deblocking_filter_disabled_flag);
    if (lf_flag_present) hdr.slice_loop_filter_across_slices_enabled_flag = (si & 1) ? 0 : 1;
    hdr.SliceAddrRs = hdr.slice_segment_address;
    hdr.SliceQpY = 26 + p->init_qp_minus26 + hdr.slice_qp_delta;
    if (d->nslices == d->capslices) { d->capslices = d->capslices ? d->capslices * 2 : 8; d->slices = (SliceHdr*)realloc(d->slices, sizeof(SliceHdr) * d->capslices); }
    if (is_p) {
      int total = d->n_st_curr_before + d->n_st_curr_after + d->n_lt_curr;
      for (int X = 0; X < (is_b ? 2 : 1); X++) {
        list_mod[X] = p->lists_modification_present_flag && total > 1 && ((si + frame_idx + X) % 3 != 0);
        for (int i = 0; i < (X ? hdr.num_ref_idx_l1_active : hdr.num_ref_idx_l0_active); i++) list_entries[X][i] = (int)(rnd(e) % (unsigned)total);
        build_ref_list(d, &hdr, X, list_mod[X] ? list_entries[X] : NULL);
      }
      if (hdr.slice_temporal_mvp) {   /* the collocated picture: any entry of either list */
        if (is_b) hdr.collocated_from_l0 = (int)(rnd(e) & 1);
        int n = hdr.collocated_from_l0 ? hdr.num_ref_idx_l0_active : hdr.num_ref_idx_l1_active;
        hdr.collocated_ref_idx = (int)(rnd(e) % (unsigned)n);
      }
      hdr.weighted = is_b ? p->weighted_bipred_flag : p->weighted_pred_flag;
      if (hdr.weighted) {   /* a random pred_weight_table; entries without a flag keep the default weight */
        int nc = s->chroma_format_idc ? 3 : 1;
        hdr.luma_log2_wd = (int)(rnd(e) % 8); hdr.chroma_log2_wd = nc == 3 ? (int)(rnd(e) % 8) : hdr.luma_log2_wd;
        for (int X = 0; X < (is_b ? 2 : 1); X++)
          for (int i = 0; i < (X ? hdr.num_ref_idx_l1_active : hdr.num_ref_idx_l0_active); i++) {
            wp_lf[X][i] = (uint8_t)rnd_pct(e, 60); wp_cf[X][i] = (uint8_t)(nc == 3 && rnd_pct(e, 60));
            hdr.wp_weight[X][i][0] = (int16_t)(1 << hdr.luma_log2_wd); hdr.wp_offset[X][i][0] = 0;
            hdr.wp_weight[X][i][1] = hdr.wp_weight[X][i][2] = (int16_t)(1 << hdr.chroma_log2_wd); hdr.wp_offset[X][i][1] = hdr.wp_offset[X][i][2] = 0;
            if (wp_lf[X][i]) {
              wp_dw[X][i][0] = (int)(rnd(e) % 17) - 8; wp_do[X][i][0] = (int)(rnd(e) % 41) - 20;
              if (rnd_pct(e, 10)) { wp_dw[X][i][0] = rnd_pct(e, 50) ? -128 : 127; wp_do[X][i][0] = rnd_pct(e, 50) ? -128 : 127; }   /* range ends */
              hdr.wp_weight[X][i][0] = (int16_t)((1 << hdr.luma_log2_wd) + wp_dw[X][i][0]); hdr.wp_offset[X][i][0] = (int16_t)wp_do[X][i][0];
            }
            if (wp_cf[X][i])
              for (int j = 1; j < 3; j++) {
                wp_dw[X][i][j] = (int)(rnd(e) % 17) - 8; wp_do[X][i][j] = (int)(rnd(e) % 81) - 40;
                if (rnd_pct(e, 10)) { wp_dw[X][i][j] = rnd_pct(e, 50) ? -128 : 127; wp_do[X][i][j] = rnd_pct(e, 50) ? -512 : 511; }
                int wgt = (1 << hdr.chroma_log2_wd) + wp_dw[X][i][j];
                hdr.wp_weight[X][i][j] = (int16_t)wgt;
                hdr.wp_offset[X][i][j] = (int16_t)Clip3(-128, 127, (128 + wp_do[X][i][j] - ((128 * wgt) >> hdr.chroma_log2_wd)));
              }
          }
      }
    }
    d->slices[d->nslices] = hdr; d->sh = &d->slices[d->nslices]; d->sh_idx = d->nslices; d->nslices++;
    cur_hdr = hdr;
  segment_data:
    if (is_dep) { hdr = cur_hdr; hdr.first_slice_segment_in_pic_flag = 0; hdr.slice_segment_address = d->CtbAddrTsToRs[slice_start[si]]; d->sh = &d->slices[d->nslices - 1]; }
    const int lf_present = p->pps_loop_filter_across_slices_enabled_flag &&
                           (hdr.slice_sao_luma_flag || hdr.slice_sao_chroma_flag || !hdr.slice_deblocking_filter_disabled_flag);

    /* substreams */
    Bytes subs = {0, 0, 0};
    uint32_t* sizes = (uint32_t*)xcalloc(d, d->nCtb + 1, sizeof(uint32_t));
    int nsub = 0;
    memset(&e->ce.bw, 0, sizeof(BW));
    ce_init(&e->ce);
    d->CtbAddrInTs = slice_start[si];
    d->CtbAddrInRs = d->CtbAddrTsToRs[d->CtbAddrInTs];
    int first_ctb_in_segment = 1;
    for (;;) {
      int xCtb = (d->CtbAddrInRs % d->ctbW) << s->log2_ctb, yCtb = (d->CtbAddrInRs / d->ctbW) << s->log2_ctb;
      int tile_first = (d->CtbAddrInTs == 0) || d->TileId[d->CtbAddrInTs] != d->TileId[d->CtbAddrInTs - 1];
      int row_first = 0;
      if (p->entropy_coding_sync_enabled_flag)
        row_first = (d->CtbAddrInRs % d->ctbW == 0) || d->TileId[d->CtbAddrInTs] != d->TileId[d->CtbAddrRsToTs[d->CtbAddrInRs - 1]];
      d->ctb_slice_addr[d->CtbAddrInRs] = d->sh->SliceAddrRs;
      d->ctb_slice_idx[d->CtbAddrInRs] = d->sh_idx;
      if (first_ctb_in_segment && is_dep && !tile_first && !row_first) {
        first_ctb_in_segment = 0;   /* 9.3.1: the contexts (and qPY_PREV, 8.6.1) continue from the end of the previous slice segment */
      } else if (first_ctb_in_segment || tile_first || row_first) {
        /* the encoder keeps its own context array (e->ctx); derive via the decoder's initialiser */
        if (tile_first || !row_first) { cabac_init_contexts(d); memcpy(e->ctx, d->c.ctx, MAXCTX); }
        else {
          if (available_z(d, xCtb, yCtb, xCtb + CtbSizeY, yCtb - CtbSizeY)) memcpy(e->ctx, e->ctx_wpp, MAXCTX);
          else { cabac_init_contexts(d); memcpy(e->ctx, d->c.ctx, MAXCTX); }
        }
        d->last_qp_y = d->sh->SliceQpY;
        first_ctb_in_segment = 0;
      }
      if (d->sh->slice_sao_luma_flag || d->sh->slice_sao_chroma_flag) enc_sao(e, xCtb >> s->log2_ctb, yCtb >> s->log2_ctb);
      if (!p->cu_qp_delta_enabled_flag) { d->IsCuQpDeltaCoded = 0; d->CuQpDeltaVal = 0; derive_qp_pred(d, xCtb, yCtb); e->qg_delta = 0; }
      enc_coding_quadtree(e, xCtb, yCtb, s->log2_ctb, 0);
      flush_events(e);
      int last_in_slice = d->CtbAddrInTs + 1 == slice_start[si + 1];
      ce_terminate(&e->ce, last_in_slice); /* end_of_slice_segment_flag */
      if (p->entropy_coding_sync_enabled_flag && d->CtbAddrInRs % d->ctbW >= 1) {
        int leftRs = d->CtbAddrInRs - 1, leftTs = d->CtbAddrRsToTs[leftRs];
        if (d->TileId[leftTs] == d->TileId[d->CtbAddrInTs] &&
            ((leftRs % d->ctbW == 0) || d->TileId[leftTs] != d->TileId[d->CtbAddrRsToTs[leftRs - 1]]))
          memcpy(e->ctx_wpp, e->ctx, MAXCTX);
      }
      d->CtbAddrInTs++;
      if (last_in_slice) {
        while (e->ce.bw.nbits & 7) bw_put(&e->ce.bw, 0); /* the flush's final 1 is rbsp_stop_one_bit */
        by_push(&subs, e->ce.bw.p, e->ce.bw.nbits >> 3);
        sizes[nsub++] = (uint32_t)(e->ce.bw.nbits >> 3);
        bw_free(&e->ce.bw);
        break;
      }
      d->CtbAddrInRs = d->CtbAddrTsToRs[d->CtbAddrInTs];
      int new_tile = p->tiles_enabled_flag && d->TileId[d->CtbAddrInTs] != d->TileId[d->CtbAddrInTs - 1];
      int new_row = p->entropy_coding_sync_enabled_flag &&
                    (d->CtbAddrInRs % d->ctbW == 0 || d->TileId[d->CtbAddrInTs] != d->TileId[d->CtbAddrRsToTs[d->CtbAddrInRs - 1]]);
      if (new_tile || new_row) {
        ce_terminate(&e->ce, 1); /* end_of_subset_one_bit; its flush ends with the alignment '1' */
        while (e->ce.bw.nbits & 7) bw_put(&e->ce.bw, 0);
        by_push(&subs, e->ce.bw.p, e->ce.bw.nbits >> 3);
        sizes[nsub++] = (uint32_t)(e->ce.bw.nbits >> 3);
        bw_free(&e->ce.bw);
        memset(&e->ce.bw, 0, sizeof(BW));
        ce_init(&e->ce);
      }
    }
    /* entry points count escaped bytes: escape each substream on its own (a substream never ends
       with a zero byte, so this equals escaping the concatenation) */
    Bytes esc = {0, 0, 0};
    uint32_t* esz = (uint32_t*)xcalloc(d, nsub + 1, sizeof(uint32_t));
    {
      size_t off = 0;
      for (int k = 0; k < nsub; k++) { esz[k] = (uint32_t)escape_into(&esc, subs.p + off, sizes[k]); off += sizes[k]; }
    }
    /* slice segment header 7.3.6.1 */
    memset(&w, 0, sizeof(w));
    bw_u(&w, hdr.first_slice_segment_in_pic_flag, 1);
    if (nal_type >= 16 && nal_type <= 23) bw_u(&w, 0, 1); /* no_output_of_prior_pics_flag (IRAP only) */
    bw_ue(&w, 0);
    if (!hdr.first_slice_segment_in_pic_flag) {
      if (p->dependent_slice_segments_enabled_flag) bw_u(&w, is_dep, 1);
      bw_u(&w, hdr.slice_segment_address, ceil_log2(d->nCtb));
    }
    if (!is_dep) {
      bw_ue(&w, hdr.slice_type);
      if (p->output_flag_present_flag) bw_u(&w, frame_idx != prm->hidden_poc, 1);   /* pic_output_flag */
      if (nal_type != 19 && nal_type != 20) {   /* 7.3.6.1: POC lsb, the picture's RPS coded in the slice header (idx == num_short_term_ref_pic_sets == 0: no inter-RPS flag) */
        bw_u(&w, frame_idx & 255, s->log2_max_poc_lsb);
        bw_u(&w, 0, 1);                                   /* short_term_ref_pic_set_sps_flag */
        bw_ue(&w, rps.num_neg); bw_ue(&w, rps.num_pos);
        for (int i = 0, prev = 0; i < rps.num_neg; i++) { bw_ue(&w, prev - rps.delta_s0[i] - 1); bw_u(&w, rps.used_s0[i], 1); prev = rps.delta_s0[i]; }
        for (int i = 0, prev = 0; i < rps.num_pos; i++) { bw_ue(&w, rps.delta_s1[i] - prev - 1); bw_u(&w, rps.used_s1[i], 1); prev = rps.delta_s1[i]; }
        if (s->long_term_ref_pics_present_flag) {   /* 7.3.6.1: the long-term pictures of the RPS */
          const int from_sps = s->num_long_term_ref_pics_sps > 0 ? rps.num_lt : 0;   /* mode 3: every entry names a candidate of the SPS */
          if (s->num_long_term_ref_pics_sps > 0) bw_ue(&w, from_sps);
          bw_ue(&w, rps.num_lt - from_sps);
          for (int i = 0, prev_cycle = 0; i < rps.num_lt; i++) {
            if (i < from_sps) { if (s->num_long_term_ref_pics_sps > 1) bw_u(&w, 1, ceil_log2(s->num_long_term_ref_pics_sps)); }   /* candidate 1: LSBs 0 */
            else { bw_u(&w, rps.lt_poc_lsb[i], s->log2_max_poc_lsb); bw_u(&w, rps.lt_used[i], 1); }
            bw_u(&w, rps.lt_msb_present[i], 1);
            if (rps.lt_msb_present[i]) bw_ue(&w, rps.lt_msb_cycle[i] - ((i == 0 || i == from_sps) ? 0 : prev_cycle));
            prev_cycle = rps.lt_msb_cycle[i];
          }
        }
        if (s->sps_temporal_mvp_enabled_flag) bw_u(&w, hdr.slice_temporal_mvp, 1);
      }
      if (s->sao_enabled_flag) { bw_u(&w, hdr.slice_sao_luma_flag, 1); if (s->chroma_format_idc) bw_u(&w, hdr.slice_sao_chroma_flag, 1); }
      if (is_p) {
        bw_u(&w, 1, 1); bw_ue(&w, hdr.num_ref_idx_l0_active - 1);                               /* num_ref_idx_active_override_flag */
        if (is_b) bw_ue(&w, hdr.num_ref_idx_l1_active - 1);
        int total = d->n_st_curr_before + d->n_st_curr_after + d->n_lt_curr;
        if (p->lists_modification_present_flag && total > 1)
          for (int X = 0; X < (is_b ? 2 : 1); X++) {
            bw_u(&w, list_mod[X], 1);
            if (list_mod[X]) for (int i = 0; i < (X ? hdr.num_ref_idx_l1_active : hdr.num_ref_idx_l0_active); i++) bw_u(&w, list_entries[X][i], ceil_log2(total));
          }
        if (is_b) bw_u(&w, hdr.mvd_l1_zero_flag, 1);
        if (p->cabac_init_present_flag) bw_u(&w, hdr.cabac_init_flag, 1);
        if (hdr.slice_temporal_mvp) {
          if (is_b) bw_u(&w, hdr.collocated_from_l0, 1);
          if ((hdr.collocated_from_l0 && hdr.num_ref_idx_l0_active > 1) || (!hdr.collocated_from_l0 && hdr.num_ref_idx_l1_active > 1)) bw_ue(&w, hdr.collocated_ref_idx);
        }
        if (hdr.weighted) {   /* 7.3.6.3 */
          int nc = s->chroma_format_idc ? 3 : 1;
          bw_ue(&w, hdr.luma_log2_wd);
          if (nc == 3) bw_se(&w, hdr.chroma_log2_wd - hdr.luma_log2_wd);
          for (int X = 0; X < (is_b ? 2 : 1); X++) {
            int n = X ? hdr.num_ref_idx_l1_active : hdr.num_ref_idx_l0_active;
            for (int i = 0; i < n; i++) bw_u(&w, wp_lf[X][i], 1);
            if (nc == 3) for (int i = 0; i < n; i++) bw_u(&w, wp_cf[X][i], 1);
            for (int i = 0; i < n; i++) {
              if (wp_lf[X][i]) { bw_se(&w, wp_dw[X][i][0]); bw_se(&w, wp_do[X][i][0]); }
              if (wp_cf[X][i]) for (int j = 1; j < 3; j++) { bw_se(&w, wp_dw[X][i][j]); bw_se(&w, wp_do[X][i][j]); }
            }
          }
        }
        bw_ue(&w, 5 - hdr.max_num_merge_cand);
      }
      bw_se(&w, hdr.slice_qp_delta);
      if (lf_present) bw_u(&w, hdr.slice_loop_filter_across_slices_enabled_flag, 1);
    }
    if (p->tiles_enabled_flag || p->entropy_coding_sync_enabled_flag) {
      bw_ue(&w, nsub - 1);
      if (nsub > 1) {
        uint32_t mx = 0;
        for (int k = 0; k < nsub - 1; k++) if (esz[k] - 1 > mx) mx = esz[k] - 1;
        int len = 1; while (len < 32 && (mx >> len)) len++;
        bw_ue(&w, len - 1);
        for (int k = 0; k < nsub - 1; k++) bw_u(&w, esz[k] - 1, len);
      }
    }
    bw_trailing(&w); /* byte_alignment(): same bit pattern */
    Bytes nal = {0, 0, 0};
    uint8_t nh[2] = {(uint8_t)(nal_type << 1), 1};
    by_push(&nal, nh, 2);
    escape_into(&nal, w.p, w.nbits >> 3);
    by_push(&nal, esc.p, esc.n);
    uint8_t len4[4] = {(uint8_t)(nal.n >> 24), (uint8_t)(nal.n >> 16), (uint8_t)(nal.n >> 8), (uint8_t)nal.n};
    by_push(&stream, len4, 4);
    by_push(&stream, nal.p, nal.n);
    bw_free(&w); free(nal.p); free(esc.p); free(esz); free(subs.p); free(sizes);
  }
  free(slice_start); free(seg_dependent);
  for (int c = 0; c < 3; c++) { free(src[c]); src[c] = NULL; e->src[c] = NULL; }
  if (d->seq_mode) {   /* the decoded picture (deblocked, SAO applied) becomes a reference picture, as in the decoder */
    int nc = s->chroma_format_idc ? 3 : 1;
    deblock_picture(d);
    uint16_t* fin[3] = {0, 0, 0};
    for (int c = 0; c < nc; c++) fin[c] = (uint16_t*)xcalloc(d, c ? (size_t)d->Wc * d->Hc : (size_t)d->W * d->H, sizeof(uint16_t));
    sao_picture(d, d->rec, fin);
    int slot = -1;
    for (int i = 0; i < MAX_DPB; i++) if (!d->dpb[i].valid) { slot = i; break; }
    if (slot < 0) fail(d, "decoded picture buffer is full");
    if (slot >= d->n_dpb) d->n_dpb = slot + 1;
    for (int c = 0; c < nc; c++) d->dpb[slot].plane[c] = fin[c];
    d->dpb[slot].poc = d->poc; d->dpb[slot].valid = 1;
    dpb_store_motion(d, &d->dpb[slot]);
  }
  release_picture(d);
#undef stream
}

static int enc_run(const hevc_testenc_params* prm, int n_frames, const uint16_t* const* planes, uint8_t** out, size_t* out_sizes,
                   char* errbuf, size_t errbuf_len, int seq_mode)
{
  init_scans(); init_dct();
  Enc E; memset(&E, 0, sizeof(E));
  Enc* e = &E;
  e->prm = *prm;
  e->rng = 0x9E3779B97F4A7C15ULL ^ ((uint64_t)prm->seed * 0x100000001B3ULL);
  Dec* d = (Dec*)calloc(1, sizeof(Dec));
  e->d = d;
  d->keep_taps = 0;
  d->seq_mode = seq_mode; d->first_picture = 1;
  Bytes stream = {0, 0, 0};
  for (int f = 0; f < n_frames; f++) { out[f] = NULL; out_sizes[f] = 0; }
  if (setjmp(d->jb)) {
    if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "testenc: %s", d->err);
    for (int c = 0; c < 3; c++) free(e->src_alloc[c]);
    free(stream.p); free(e->ev); free(e->pcm_blob); bw_free(&e->ce.bw);
    for (int f = 0; f < n_frames; f++) { free(out[f]); out[f] = NULL; }
    free_dec(d);
    return -1;
  }
  enc_parameter_sets(e, &stream);
  /* ---- coding order: frame 0 an IDR picture; with b_frames = b every (b + 1)-th picture is a P picture (an "anchor") and the b pictures
     before it are B pictures coded after it; the pictures behind the last anchor are P pictures.  A picture's own references: the anchors
     before it (inter_num_refs of them; the farthest of three or more is kept in the RPS but not used), for a B picture also the anchor
     after it and, with b_ref, the B picture before it.  Its RPS additionally keeps every decoded picture a later picture still needs. */
  PicPlan* plan = (PicPlan*)xcalloc(d, (size_t)n_frames, sizeof(PicPlan));
  {
    const int b = seq_mode ? Max(0, prm->b_frames) : 0, step = b + 1;
    int n = 0;
    plan[n].poc = 0; plan[n].slice_type = 2; plan[n].nal_type = 19; n++;
    for (int a = step; a - step < n_frames - 1; a += step) {
      if (a < n_frames) {
        plan[n].poc = a; plan[n].slice_type = 1; plan[n].nal_type = 1; n++;
        for (int q = a - b; q < a; q++) { plan[n].poc = q; plan[n].slice_type = 0; plan[n].nal_type = prm->b_ref ? 1 : 0; n++; }
      } else
        for (int q = a - b; q < n_frames; q++) { plan[n].poc = q; plan[n].slice_type = 1; plan[n].nal_type = 1; n++; }
    }
    if (n != n_frames) fail(d, "testenc: picture plan");
    int cra_poc = -1;
    if (seq_mode && prm->open_gop > 0) {
      if (!b || prm->long_term_ref) fail(d, "testenc: open_gop needs b_frames and no long_term_ref");
      cra_poc = step * prm->open_gop;
      if (cra_poc >= n_frames) fail(d, "testenc: open_gop names an anchor behind the last picture");
      for (int k = 1; k < n_frames; k++) {
        if (plan[k].poc == cra_poc) { plan[k].slice_type = 2; plan[k].nal_type = 21; }                                   /* CRA_NUT */
        else if (plan[k].slice_type == 0 && plan[k].poc > cra_poc - step && plan[k].poc < cra_poc) plan[k].nal_type = prm->b_ref ? 9 : 8;   /* RASL_R / RASL_N */
      }
    }
    const int nrefs = Max(1, prm->inter_num_refs);
    for (int k = 1; k < n_frames; k++) {   /* own references */
      PicPlan* P = &plan[k];
      int prev_anchors[16], npa = 0;
      if (P->slice_type == 2) continue;   /* (the CRA picture of open_gop: intra; what its RASL pictures need is kept by the loop below) */
      for (int j = k - 1; j >= 0 && npa < 16; j--)
        if (plan[j].slice_type != 0 && plan[j].poc < P->poc && !(cra_poc >= 0 && P->poc > cra_poc && plan[j].poc < cra_poc))
          prev_anchors[npa++] = plan[j].poc;
      /* (coding order of anchors is POC order, so prev_anchors is sorted closest first) */
      if (P->slice_type == 0) {
        if (prm->b_ref && k > 0 && plan[k - 1].slice_type == 0 && plan[k - 1].poc == P->poc - 1) { P->neg_poc[P->n_neg] = P->poc - 1; P->neg_used[P->n_neg++] = 1; }
        for (int j = k - 1; j >= 0; j--) if (plan[j].slice_type != 0 && plan[j].poc > P->poc) { P->pos_poc[0] = plan[j].poc; P->pos_used[0] = 1; P->n_pos = 1; break; }
      }
      int take = Min(npa, nrefs);
      for (int i = 0; i < take; i++) { P->neg_poc[P->n_neg] = prev_anchors[i]; P->neg_used[P->n_neg++] = !(take >= 3 && i == take - 1); }
      if (prm->long_term_ref) {   /* the IDR picture (POC 0): a long-term reference picture of every later picture, never a short-term one */
        int m = 0;
        for (int i = 0; i < P->n_neg; i++) if (P->neg_poc[i] != 0) { P->neg_poc[m] = P->neg_poc[i]; P->neg_used[m++] = P->neg_used[i]; }
        P->n_neg = m;
        P->lt_poc[0] = 0; P->n_lt = 1;
      }
    }
    for (int k = 1; k < n_frames; k++) {   /* keep what later pictures need */
      PicPlan* P = &plan[k];
      for (int j = k + 1; j < n_frames; j++)
        for (int t = 0; t < plan[j].n_neg + plan[j].n_pos; t++) {
          int poc = t < plan[j].n_neg ? plan[j].neg_poc[t] : plan[j].pos_poc[t - plan[j].n_neg];
          int decoded = 0, have = 0;
          for (int q = 0; q < k; q++) if (plan[q].poc == poc) decoded = 1;
          for (int q = 0; q < P->n_neg; q++) if (P->neg_poc[q] == poc) have = 1;
          for (int q = 0; q < P->n_pos; q++) if (P->pos_poc[q] == poc) have = 1;
          if (!decoded || have || poc == P->poc || (prm->long_term_ref && poc == 0)) continue;
          if (poc < P->poc) { if (P->n_neg < 16) { P->neg_poc[P->n_neg] = poc; P->neg_used[P->n_neg++] = 0; } }
          else if (P->n_pos < 16) { P->pos_poc[P->n_pos] = poc; P->pos_used[P->n_pos++] = 0; }
        }
      /* S0 by decreasing, S1 by increasing POC (7.4.8) */
      for (int i = 0; i < P->n_neg; i++) for (int j = i + 1; j < P->n_neg; j++) if (P->neg_poc[j] > P->neg_poc[i]) {
        int t = P->neg_poc[i]; P->neg_poc[i] = P->neg_poc[j]; P->neg_poc[j] = t; uint8_t u = P->neg_used[i]; P->neg_used[i] = P->neg_used[j]; P->neg_used[j] = u; }
      for (int i = 0; i < P->n_pos; i++) for (int j = i + 1; j < P->n_pos; j++) if (P->pos_poc[j] < P->pos_poc[i]) {
        int t = P->pos_poc[i]; P->pos_poc[i] = P->pos_poc[j]; P->pos_poc[j] = t; uint8_t u = P->pos_used[i]; P->pos_used[i] = P->pos_used[j]; P->pos_used[j] = u; }
    }
  }
  for (int k = 0; k < n_frames; k++) {
    enc_picture(e, planes + 3 * plan[k].poc, &plan[k], &stream);
    out[k] = stream.p; out_sizes[k] = stream.n;
    stream.p = NULL; stream.n = stream.cap = 0;
  }
  free(plan);
  free(e->ev); free(e->pcm_blob);
  free_dec(d);
  return 0;
}

int hevc_testenc_encode(const hevc_testenc_params* prm, const uint16_t* const planes[3], uint8_t** out, size_t* out_size,
                        char* errbuf, size_t errbuf_len)
{
  return enc_run(prm, 1, planes, out, out_size, errbuf, errbuf_len, 0);
}

int hevc_testenc_encode_seq(const hevc_testenc_params* prm, int n_frames, const uint16_t* const* planes, uint8_t** out, size_t* out_sizes,
                            char* errbuf, size_t errbuf_len)
{
  if (n_frames < 1) return -1;
  return enc_run(prm, n_frames, planes, out, out_sizes, errbuf, errbuf_len, 1);
}

void hevc_testenc_free(uint8_t* p) { free(p); }
