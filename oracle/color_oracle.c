/*
 * color_oracle.c — CPU ORACLE for the colour stage (test infrastructure, NOT product code).
 *
 * Plain-C restatement of libheif's own colour-conversion operations that lie on the hot path
 * (SURVEY.md §8a rows a9-a15).  Unlike the HEVC oracle this part IS pinned: tests compare it with
 * the real reference ops compiled from /root/reference (oracle/_ref/libref_harness.so) and with the
 * reference's known-answer test tests/conversion.cc:685-725.
 *
 * All planes are passed as uint16_t arrays (8-bit data widened by the caller); strides in samples.
 * Compile with -ffp-contract=off: the reference is built for baseline x86-64 without FMA.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* libheif/nclx.cc:45-72 get_colour_primaries: gx,gy,bx,by,rx,ry,wx,wy */
static int primaries_of(int idx, float p[8])
{
  static const float t[][9] = {
    {1, 0.300f, 0.600f, 0.150f, 0.060f, 0.640f, 0.330f, 0.3127f, 0.3290f},
    {4, 0.21f, 0.71f, 0.14f, 0.08f, 0.67f, 0.33f, 0.310f, 0.316f},
    {5, 0.29f, 0.60f, 0.15f, 0.06f, 0.64f, 0.33f, 0.3127f, 0.3290f},
    {6, 0.310f, 0.595f, 0.155f, 0.070f, 0.630f, 0.340f, 0.3127f, 0.3290f},
    {7, 0.310f, 0.595f, 0.155f, 0.070f, 0.630f, 0.340f, 0.3127f, 0.3290f},
    {8, 0.243f, 0.692f, 0.145f, 0.049f, 0.681f, 0.319f, 0.310f, 0.316f},
    {9, 0.170f, 0.797f, 0.131f, 0.046f, 0.708f, 0.292f, 0.3127f, 0.3290f},
    {10, 0.0f, 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.333333f, 0.33333f},
    {11, 0.265f, 0.690f, 0.150f, 0.060f, 0.680f, 0.320f, 0.314f, 0.351f},
    {12, 0.265f, 0.690f, 0.150f, 0.060f, 0.680f, 0.320f, 0.3127f, 0.3290f},
    {22, 0.295f, 0.605f, 0.155f, 0.077f, 0.630f, 0.340f, 0.3127f, 0.3290f}};
  for (unsigned i = 0; i < sizeof(t) / sizeof(t[0]); i++)
    if ((int)t[i][0] == idx) { memcpy(p, &t[i][1], 8 * sizeof(float)); return 1; }
  memset(p, 0, 8 * sizeof(float));
  return 0;
}

/* libheif/nclx.cc:84-140 get_Kr_Kb */
static void kr_kb(int matrix, int primaries, float* Kr, float* Kb)
{
  *Kr = 0; *Kb = 0;
  if (matrix == 12 || matrix == 13) {
    float p[8];
    primaries_of(primaries, p);
    float gx = p[0], gy = p[1], bx = p[2], by = p[3], rx = p[4], ry = p[5], wx = p[6], wy = p[7];
    float zr = 1 - (rx + ry), zg = 1 - (gx + gy), zb = 1 - (bx + by), zw = 1 - (wx + wy);
    float denom = wy * (rx * (gy * zb - by * zg) + gx * (by * zr - ry * zb) + bx * (ry * zg - gy * zr));
    if (denom == 0.0f) return;
    *Kr = (ry * (wx * (gy * zb - by * zg) + wy * (bx * zg - gx * zb) + zw * (gx * by - bx * gy))) / denom;
    *Kb = (by * (wx * (ry * zg - gy * zr) + wy * (gx * zr - rx * zg) + zw * (rx * gy - gx * ry))) / denom;
    return;
  }
  switch (matrix) {
    case 1: *Kr = 0.2126f; *Kb = 0.0722f; break;
    case 4: *Kr = 0.30f; *Kb = 0.11f; break;
    case 5: case 6: *Kr = 0.299f; *Kb = 0.114f; break;
    case 7: *Kr = 0.212f; *Kb = 0.087f; break;
    case 9: case 10: *Kr = 0.2627f; *Kb = 0.0593f; break;
    default: break;
  }
}

/* libheif/nclx.cc:143-173: out = {r_cr, g_cb, g_cr, b_cb} */
void color_oracle_coeffs(int has_nclx, int matrix, int primaries, float out[4])
{
  float Kr = 0, Kb = 0;
  if (has_nclx) kr_kb(matrix, primaries, &Kr, &Kb);
  if (has_nclx && (Kb != 0 || Kr != 0)) {
    out[0] = 2 * (-Kr + 1);
    out[1] = 2 * Kb * (-Kb + 1) / (Kb + Kr - 1);
    out[2] = 2 * Kr * (-Kr + 1) / (Kb + Kr - 1);
    out[3] = 2 * (-Kb + 1);
  } else {
    out[0] = 1.402f; out[1] = -0.344136f; out[2] = -0.714136f; out[3] = 1.772f;
  }
}

static int clip_int_u8(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
static int clip_int_u16(int x, int maxi) { return x < 0 ? 0 : x > maxi ? maxi : x; }
static int clip_f_u16(float fx, int maxi) /* libheif/common_utils.h:108-114 */
{
  int x = (int)(fx + 0.5f);
  return x < 0 ? 0 : x > maxi ? maxi : x;
}

/* a9: Op_YCbCr420_to_RGB24 / _RGB32 (libheif/color-conversion/yuv2rgb.cc:345-426, :481-562) */
void color_oracle_420_to_rgb24(const uint16_t* y, int ys, const uint16_t* cb, int cbs, const uint16_t* cr, int crs,
                               int w, int h, int has_nclx, int matrix, int primaries,
                               uint8_t* out, int out_stride, int with_alpha)
{
  float c[4];
  color_oracle_coeffs(has_nclx, matrix, primaries, c);
  int r_cr = (int)lround(256 * c[0]), g_cb = (int)lround(256 * c[1]);
  int g_cr = (int)lround(256 * c[2]), b_cb = (int)lround(256 * c[3]);
  int bpp = with_alpha ? 4 : 3;
  for (int yy = 0; yy < h; yy++)
    for (int x = 0; x < w; x++) {
      int cbv = cb[(yy / 2) * cbs + x / 2] - 128, crv = cr[(yy / 2) * crs + x / 2] - 128;
      int r_off = (r_cr * crv + 128) >> 8;
      int g_off = (g_cb * cbv + g_cr * crv + 128) >> 8;
      int b_off = (b_cb * cbv + 128) >> 8;
      int yv = y[yy * ys + x];
      uint8_t* p = out + (size_t)yy * out_stride + bpp * x;
      p[0] = (uint8_t)clip_int_u8(yv + r_off);
      p[1] = (uint8_t)clip_int_u8(yv + g_off);
      p[2] = (uint8_t)clip_int_u8(yv + b_off);
      if (with_alpha) p[3] = 0xFF;
    }
}

/* a10: Op_YCbCr_to_RGB<Pixel> (libheif/color-conversion/yuv2rgb.cc:92-292), planar output.
   chroma: 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 (heif_chroma numeric values) */
void color_oracle_ycbcr_to_rgb_planar(const uint16_t* y, int ys, const uint16_t* cb, int cbs, const uint16_t* cr, int crs,
                                      int w, int h, int bpp, int chroma, int has_nclx, int matrix, int primaries,
                                      int full_range, uint16_t* r, uint16_t* g, uint16_t* b, int os)
{
  int matrix_coeffs = 2, full_range_flag = 1;
  float c[4];
  color_oracle_coeffs(has_nclx, matrix, primaries, c);
  if (has_nclx) { matrix_coeffs = matrix; full_range_flag = full_range; }
  int halfRange = 1 << (bpp - 1);
  int fullRange = (1 << bpp) - 1;
  float limited_range_offset = (float)(16 << (bpp - 8));
  int shiftH = (chroma == 3) ? 0 : 1, shiftV = (chroma == 1) ? 1 : 0;
  for (int yy = 0; yy < h; yy++)
    for (int x = 0; x < w; x++) {
      int cx = x >> shiftH, cy = yy >> shiftV;
      int Y = y[yy * ys + x], Cb = cb[cy * cbs + cx], Cr = cr[cy * crs + cx];
      int R, G, B;
      if (matrix_coeffs == 0) {
        if (full_range_flag) { R = Cr; G = Y; B = Cb; }
        else {
          R = clip_f_u16((Cr - limited_range_offset) * 1.1429f, fullRange);
          G = clip_f_u16((Y - limited_range_offset) * 1.1689f, fullRange);
          B = clip_f_u16((Cb - limited_range_offset) * 1.1429f, fullRange);
        }
      } else if (matrix_coeffs == 8) {
        int cbv = Cb - halfRange, crv = Cr - halfRange;
        R = clip_int_u8(Y - cbv + crv); G = clip_int_u8(Y + cbv); B = clip_int_u8(Y - cbv - crv);
      } else if (matrix_coeffs == 16) {
        int16_t yy16 = (int16_t)Y;
        int16_t cbv = (int16_t)((int16_t)Cb - (int16_t)halfRange), crv = (int16_t)((int16_t)Cr - (int16_t)halfRange);
        int16_t t = (int16_t)(yy16 - (cbv >> 1));
        int16_t gg = (int16_t)(t + cbv);
        int16_t bb = (int16_t)(t - (crv >> 1));
        int16_t rr = (int16_t)(bb + crv);
        R = clip_int_u16(rr * 4, fullRange); G = clip_int_u16(gg * 4, fullRange); B = clip_int_u16(bb * 4, fullRange);
      } else {
        float yv = (float)Y, cbv = (float)(Cb - halfRange), crv = (float)(Cr - halfRange);
        if (!full_range_flag) { yv = (yv - limited_range_offset) * 1.1689f; cbv = cbv * 1.1429f; crv = crv * 1.1429f; }
        R = clip_f_u16(yv + c[0] * crv, fullRange);
        G = clip_f_u16(yv + c[1] * cbv + c[2] * crv, fullRange);
        B = clip_f_u16(yv + c[3] * cbv, fullRange);
      }
      r[yy * os + x] = (uint16_t)R; g[yy * os + x] = (uint16_t)G; b[yy * os + x] = (uint16_t)B;
    }
}

/* a11: Op_RGB_to_RGB24_32 (libheif/color-conversion/rgb2rgb.cc:72-150), no input alpha */
void color_oracle_rgb_planar_to_interleaved8(const uint16_t* r, const uint16_t* g, const uint16_t* b, int is,
                                             int w, int h, uint8_t* out, int out_stride, int want_alpha)
{
  int bpp = want_alpha ? 4 : 3;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint8_t* p = out + (size_t)y * out_stride + bpp * x;
      p[0] = (uint8_t)r[y * is + x]; p[1] = (uint8_t)g[y * is + x]; p[2] = (uint8_t)b[y * is + x];
      if (want_alpha) p[3] = 0xFF;
    }
}

/* a12: Op_YCbCr420_to_RRGGBBaa (libheif/color-conversion/yuv2rgb.cc:622-734), no alpha */
void color_oracle_420_to_rrggbb(const uint16_t* y, int ys, const uint16_t* cb, int cbs, const uint16_t* cr, int crs,
                                int w, int h, int bpp, int has_nclx, int matrix, int primaries, int full_range,
                                uint8_t* out, int out_stride, int little_endian)
{
  int full_range_flag = 1;
  float c[4];
  color_oracle_coeffs(has_nclx, matrix, primaries, c);
  if (has_nclx) full_range_flag = full_range;
  int maxval = (1 << bpp) - 1, le = little_endian ? 1 : 0;
  float limited_range_offset = (float)(16 << (bpp - 8));
  for (int yy = 0; yy < h; yy++)
    for (int x = 0; x < w; x++) {
      float y_ = y[yy * ys + x];
      float cbv = (float)(cb[(yy / 2) * cbs + x / 2] - (1 << (bpp - 1)));
      float crv = (float)(cr[(yy / 2) * crs + x / 2] - (1 << (bpp - 1)));
      if (!full_range_flag) { y_ = (y_ - limited_range_offset) * 1.1689f; cbv = cbv * 1.1429f; crv = crv * 1.1429f; }
      int r = clip_f_u16(y_ + c[0] * crv, maxval);
      int g = clip_f_u16(y_ + c[1] * cbv + c[2] * crv, maxval);
      int b = clip_f_u16(y_ + c[3] * cbv, maxval);
      uint8_t* p = out + (size_t)yy * out_stride + 6 * x;
      p[0 + le] = (uint8_t)(r >> 8); p[2 + le] = (uint8_t)(g >> 8); p[4 + le] = (uint8_t)(b >> 8);
      p[1 - le] = (uint8_t)(r & 0xff); p[3 - le] = (uint8_t)(g & 0xff); p[5 - le] = (uint8_t)(b & 0xff);
    }
}

/* a13: Op_YCbCr420_bilinear_to_YCbCr444 (libheif/color-conversion/chroma_sampling.cc:501-724),
   one chroma plane; including the border loops' `cx / 2` indexing exactly as the reference has it */
void color_oracle_bilinear_420_to_444(const uint16_t* in, int is, int w, int h, uint16_t* out, int os)
{
  /* every output sample is written by the reference only inside these loops; samples it never
     writes (right/bottom border for odd sizes) keep the calloc'd zero of HeifPixelImage */
  for (int y = 0; y < h; y++) memset(out + (size_t)y * os, 0, sizeof(uint16_t) * w);
  out[0] = in[0];
  for (int cx = 0; cx < (w - 1) / 2; cx++) {
    out[2 * cx + 1] = (uint16_t)((3 * in[cx / 2] + 1 * in[cx / 2 + 1] + 2) / 4);
    out[2 * cx + 2] = (uint16_t)((1 * in[cx / 2] + 3 * in[cx / 2 + 1] + 2) / 4);
  }
  if (w % 2 == 0) out[w - 1] = in[w / 2 - 1];
  for (int cy = 0; cy < (h - 1) / 2; cy++) {
    out[(2 * cy + 1) * os] = (uint16_t)((3 * in[cy / 2 * is] + 1 * in[(cy / 2 + 1) * is] + 2) / 4);
    out[(2 * cy + 2) * os] = (uint16_t)((1 * in[cy / 2 * is] + 3 * in[(cy / 2 + 1) * is] + 2) / 4);
  }
  if (h % 2 == 0) out[(h - 1) * os] = in[(h / 2 - 1) * is];
  if (w % 2 == 0)
    for (int cy = 0; cy < (h - 1) / 2; cy++) {
      out[(2 * cy + 1) * os + w - 1] = (uint16_t)((3 * in[cy / 2 * is + w / 2 - 1] + 1 * in[(cy / 2 + 1) * is + w / 2 - 1] + 2) / 4);
      out[(2 * cy + 2) * os + w - 1] = (uint16_t)((1 * in[cy / 2 * is + w / 2 - 1] + 3 * in[(cy / 2 + 1) * is + w / 2 - 1] + 2) / 4);
    }
  if (h % 2 == 0)
    for (int cx = 0; cx < (w - 1) / 2; cx++) {
      out[(h - 1) * os + 2 * cx + 1] = (uint16_t)((3 * in[(h / 2 - 1) * is + cx / 2] + 1 * in[(h / 2 - 1) * is + cx / 2 + 1] + 2) / 4);
      out[(h - 1) * os + 2 * cx + 2] = (uint16_t)((1 * in[(h / 2 - 1) * is + cx / 2] + 3 * in[(h / 2 - 1) * is + cx / 2 + 1] + 2) / 4);
    }
  if (w % 2 == 0 && h % 2 == 0) out[(h - 1) * os + w - 1] = in[(h / 2 - 1) * is + w / 2 - 1];
  for (int y = 1; y < h - 1; y += 2)
    for (int x = 1; x < w - 1; x += 2) {
      int cx = x / 2, cy = y / 2;
      int c00 = in[cy * is + cx], c01 = in[cy * is + cx + 1], c10 = in[(cy + 1) * is + cx], c11 = in[(cy + 1) * is + cx + 1];
      out[(y + 0) * os + x + 0] = (uint16_t)((c00 * 9 + c01 * 3 + c10 * 3 + c11 * 1 + 8) / 16);
      out[(y + 0) * os + x + 1] = (uint16_t)((c00 * 3 + c01 * 9 + c10 * 1 + c11 * 3 + 8) / 16);
      out[(y + 1) * os + x + 0] = (uint16_t)((c00 * 3 + c01 * 1 + c10 * 9 + c11 * 3 + 8) / 16);
      out[(y + 1) * os + x + 1] = (uint16_t)((c00 * 1 + c01 * 3 + c10 * 3 + c11 * 9 + 8) / 16);
    }
}

/* Op_YCbCr422_bilinear_to_YCbCr444 (libheif/color-conversion/chroma_sampling.cc:732-954), one chroma plane: the left
   border copies, the right border copies for even widths (:895-906), the inner pairs are 3-1 / 4 (+2) filtered (:911-927); for odd
   widths the reference's inner loop also writes the last column */
void color_oracle_bilinear_422_to_444(const uint16_t* in, int is, int w, int h, uint16_t* out, int os)
{
  for (int y = 0; y < h; y++) {
    const uint16_t* s = in + (size_t)y * is;
    uint16_t* d = out + (size_t)y * os;
    memset(d, 0, sizeof(uint16_t) * w);
    d[0] = s[0];
    if (w % 2 == 0) d[w - 1] = s[w / 2 - 1];
    for (int x = 1; x < w - 1; x += 2) {
      const int cx = x / 2;
      d[x + 0] = (uint16_t)((s[cx] * 3 + s[cx + 1] * 1 + 2) / 4);
      d[x + 1] = (uint16_t)((s[cx] * 1 + s[cx + 1] * 3 + 2) / 4);
    }
  }
}

/* a14: Op_to_sdr_planes (libheif/color-conversion/hdr_sdr.cc:146-244): v >> (bits-8), no rounding */
void color_oracle_to_sdr(const uint16_t* in, int is, int w, int h, int bits, uint16_t* out, int os)
{
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) out[y * os + x] = (uint16_t)(in[y * is + x] >> (bits - 8));
}
