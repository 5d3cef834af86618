// color.hip — fused colour stage over decoded planes in HBM (hand-written HIP for gfx950).
//
// Restates, bit-exactly, the ColorConversionOperations that libheif's pipeline planner selects for
// HEIC stills:
//   a9   Op_YCbCr420_to_RGB24/_RGB32        libheif/color-conversion/yuv2rgb.cc:345-426, :481-562
//   a10  Op_YCbCr_to_RGB<u8/u16>            libheif/color-conversion/yuv2rgb.cc:92-292
//   a11  Op_RGB_to_RGB24_32                 libheif/color-conversion/rgb2rgb.cc:72-150 (fused into a10)
//   a12  Op_YCbCr420_to_RRGGBBaa            libheif/color-conversion/yuv2rgb.cc:622-734
//   a13  Op_YCbCr420_bilinear_to_YCbCr444   libheif/color-conversion/chroma_sampling.cc:501-724
//   a14  Op_to_sdr_planes                   libheif/color-conversion/hdr_sdr.cc:146-244
//   a15  get_YCbCr_to_RGB_coefficients      libheif/nclx.cc:84-173
//
// Roofline: pure streaming, HBM-bound.  Algorithmic bytes per luma pixel: 1.5*s in + 3*s_out out
// (RGB24 from 8-bit 4:2:0: 4.5 B/px).  Each thread converts a 4x2 luma block so that the chroma
// pair is read once, Y is read as one dword per row and RGB24 leaves as one 12-byte store per row
// (64 lanes -> 768 contiguous bytes per wave-store).
// Float parity: compiled with -ffp-contract=off; the reference build (x86-64 baseline) has no FMA.
#include "hipdec_internal.h"
#include "color_device.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

using namespace hipdec::colordev;


template <typename Pix, int LAYOUT>
__device__ __forceinline__ void rgb_block(const ColorParams& p)
{
  const int bx = blockIdx.x * blockDim.x + threadIdx.x;  // 4-pixel column group
  const int by = blockIdx.y * blockDim.y + threadIdx.y;  // row pair
  const int x0 = bx * 4, y0 = by * 2;
  if (x0 >= p.w || y0 >= p.h) return;
  const int npx = min(4, p.w - x0);
#pragma unroll
  for (int dy = 0; dy < 2; dy++) {
    const int yy = y0 + dy;
    if (yy >= p.h) break;
    HIPDEC_GLOBAL const Pix* yrow = (HIPDEC_GLOBAL const Pix*)((HIPDEC_GLOBAL const uint8_t*)p.y + (size_t)yy * p.ys);   // (address space 1: see color_device.h)
    HIPDEC_GLOBAL const Pix* cbrow = (HIPDEC_GLOBAL const Pix*)((HIPDEC_GLOBAL const uint8_t*)p.cb + (size_t)(yy >> p.shiftV) * p.cbs);
    HIPDEC_GLOBAL const Pix* crrow = (HIPDEC_GLOBAL const Pix*)((HIPDEC_GLOBAL const uint8_t*)p.cr + (size_t)(yy >> p.shiftV) * p.crs);
    int Y[4], CB[4], CR[4];
    if (npx == 4 && sizeof(Pix) == 1 && (((uintptr_t)(yrow + x0)) & 3) == 0) {
      uint32_t v = *(HIPDEC_GLOBAL const uint32_t*)(yrow + x0);
      Y[0] = v & 255; Y[1] = (v >> 8) & 255; Y[2] = (v >> 16) & 255; Y[3] = v >> 24;
    } else if (npx == 4 && sizeof(Pix) == 2 && (((uintptr_t)(yrow + x0)) & 7) == 0) {
      uint2 v = *(HIPDEC_GLOBAL const uint2*)(yrow + x0);
      Y[0] = v.x & 0xffff; Y[1] = v.x >> 16; Y[2] = v.y & 0xffff; Y[3] = v.y >> 16;
    } else {
      for (int i = 0; i < 4; i++) Y[i] = i < npx ? yrow[x0 + i] : 0;
    }
    if (p.arith == AR_MONO) {
#pragma unroll
      for (int i = 0; i < 4; i++) { CB[i] = 0; CR[i] = 0; }
    } else if (p.shiftH) {
      int c0 = x0 >> 1;
      int cbA = cbrow[c0], crA = crrow[c0];
      int cbB = cbA, crB = crA;
      if (npx > 2) { cbB = cbrow[c0 + 1]; crB = crrow[c0 + 1]; }
      CB[0] = CB[1] = cbA; CB[2] = CB[3] = cbB;
      CR[0] = CR[1] = crA; CR[2] = CR[3] = crB;
    } else {
      for (int i = 0; i < 4; i++) { CB[i] = i < npx ? cbrow[x0 + i] : 0; CR[i] = i < npx ? crrow[x0 + i] : 0; }
    }
    int R[4], G[4], B[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (sizeof(Pix) == 2 && (LAYOUT == LO_RGB24 || LAYOUT == LO_RGBA32)) {   // > 8-bit planes to 8-bit interleaved output: Op_to_sdr_planes on either side
        convert_px(p, Y[i] >> p.in_shift, CB[i] >> p.in_shift, CR[i] >> p.in_shift, R[i], G[i], B[i]);
        R[i] >>= p.out_shift; G[i] >>= p.out_shift; B[i] >>= p.out_shift;
      } else convert_px(p, Y[i], CB[i], CR[i], R[i], G[i], B[i]);
    }

    if (LAYOUT == LO_PLANAR) {
      HIPDEC_GLOBAL Pix* r = (HIPDEC_GLOBAL Pix*)((HIPDEC_GLOBAL uint8_t*)p.o0 + (size_t)yy * p.os) + x0;
      HIPDEC_GLOBAL Pix* g = (HIPDEC_GLOBAL Pix*)((HIPDEC_GLOBAL uint8_t*)p.o1 + (size_t)yy * p.os) + x0;
      HIPDEC_GLOBAL Pix* b = (HIPDEC_GLOBAL Pix*)((HIPDEC_GLOBAL uint8_t*)p.o2 + (size_t)yy * p.os) + x0;
      if (npx == 4 && sizeof(Pix) == 1 && ((p.os | (uintptr_t)p.o0 | (uintptr_t)p.o1 | (uintptr_t)p.o2) & 3) == 0) {
        *(HIPDEC_GLOBAL uint32_t*)r = R[0] | (R[1] << 8) | (R[2] << 16) | ((uint32_t)R[3] << 24);
        *(HIPDEC_GLOBAL uint32_t*)g = G[0] | (G[1] << 8) | (G[2] << 16) | ((uint32_t)G[3] << 24);
        *(HIPDEC_GLOBAL uint32_t*)b = B[0] | (B[1] << 8) | (B[2] << 16) | ((uint32_t)B[3] << 24);
      } else if (npx == 4 && sizeof(Pix) == 2 && ((p.os | (uintptr_t)p.o0 | (uintptr_t)p.o1 | (uintptr_t)p.o2) & 7) == 0) {
        *(HIPDEC_GLOBAL uint2*)r = make_uint2(R[0] | (R[1] << 16), R[2] | (R[3] << 16));
        *(HIPDEC_GLOBAL uint2*)g = make_uint2(G[0] | (G[1] << 16), G[2] | (G[3] << 16));
        *(HIPDEC_GLOBAL uint2*)b = make_uint2(B[0] | (B[1] << 16), B[2] | (B[3] << 16));
      } else {
        for (int i = 0; i < npx; i++) { r[i] = (Pix)R[i]; g[i] = (Pix)G[i]; b[i] = (Pix)B[i]; }
      }
    } else if (LAYOUT == LO_RGB24) {
      HIPDEC_GLOBAL uint8_t* o = (HIPDEC_GLOBAL uint8_t*)p.o0 + (size_t)yy * p.os + (size_t)x0 * 3;
      if (npx == 4 && ((p.os | (uintptr_t)p.o0) & 3) == 0) {
        U3 v;
        v.a = R[0] | (G[0] << 8) | (B[0] << 16) | ((uint32_t)R[1] << 24);
        v.b = G[1] | (B[1] << 8) | (R[2] << 16) | ((uint32_t)G[2] << 24);
        v.c = B[2] | (R[3] << 8) | (G[3] << 16) | ((uint32_t)B[3] << 24);
        *(HIPDEC_GLOBAL U3*)o = v;
      } else {
        for (int i = 0; i < npx; i++) { o[3 * i] = (uint8_t)R[i]; o[3 * i + 1] = (uint8_t)G[i]; o[3 * i + 2] = (uint8_t)B[i]; }
      }
    } else if (LAYOUT == LO_RGBA32) {
      HIPDEC_GLOBAL uint8_t* o = (HIPDEC_GLOBAL uint8_t*)p.o0 + (size_t)yy * p.os + (size_t)x0 * 4;
      uint32_t A[4] = {255u, 255u, 255u, 255u};
      if (p.a) {
        HIPDEC_GLOBAL const uint8_t* arow = (HIPDEC_GLOBAL const uint8_t*)p.a + (size_t)yy * p.as + x0;
        if (npx == 4 && (((uintptr_t)arow) & 3) == 0) {
          const uint32_t v = *(HIPDEC_GLOBAL const uint32_t*)arow;
          A[0] = v & 255u; A[1] = (v >> 8) & 255u; A[2] = (v >> 16) & 255u; A[3] = v >> 24;
        } else {
          for (int i = 0; i < npx; i++) A[i] = arow[i];
        }
      }
      if (npx == 4 && ((p.os | (uintptr_t)p.o0) & 15) == 0) {
        uint4 v;
        v.x = R[0] | (G[0] << 8) | (B[0] << 16) | (A[0] << 24);
        v.y = R[1] | (G[1] << 8) | (B[1] << 16) | (A[1] << 24);
        v.z = R[2] | (G[2] << 8) | (B[2] << 16) | (A[2] << 24);
        v.w = R[3] | (G[3] << 8) | (B[3] << 16) | (A[3] << 24);
        *(HIPDEC_GLOBAL uint4*)o = v;
      } else {
        for (int i = 0; i < npx; i++) { o[4 * i] = (uint8_t)R[i]; o[4 * i + 1] = (uint8_t)G[i]; o[4 * i + 2] = (uint8_t)B[i]; o[4 * i + 3] = (uint8_t)A[i]; }
      }
    } else {  // RRGGBB BE / LE, yuv2rgb.cc:717-723
      HIPDEC_GLOBAL uint8_t* o = (HIPDEC_GLOBAL uint8_t*)p.o0 + (size_t)yy * p.os + (size_t)x0 * 6;
      const bool le = LAYOUT == LO_RRGGBB_LE;
      uint16_t s[12];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        int r = R[i], g = G[i], b = B[i];
        if (!le) { r = ((r & 255) << 8) | (r >> 8); g = ((g & 255) << 8) | (g >> 8); b = ((b & 255) << 8) | (b >> 8); }
        s[3 * i] = (uint16_t)r; s[3 * i + 1] = (uint16_t)g; s[3 * i + 2] = (uint16_t)b;
      }
      if (npx == 4 && ((p.os | (uintptr_t)p.o0) & 3) == 0) {
        U3 v0, v1;
        v0.a = s[0] | ((uint32_t)s[1] << 16); v0.b = s[2] | ((uint32_t)s[3] << 16); v0.c = s[4] | ((uint32_t)s[5] << 16);
        v1.a = s[6] | ((uint32_t)s[7] << 16); v1.b = s[8] | ((uint32_t)s[9] << 16); v1.c = s[10] | ((uint32_t)s[11] << 16);
        ((HIPDEC_GLOBAL U3*)o)[0] = v0; ((HIPDEC_GLOBAL U3*)o)[1] = v1;
      } else {
        for (int i = 0; i < npx * 3; i++) { o[2 * i] = (uint8_t)(s[i] & 255); o[2 * i + 1] = (uint8_t)(s[i] >> 8); }
      }
    }
  }
}

template <typename Pix, int LAYOUT>
__global__ __launch_bounds__(256) void k_ycbcr_to_rgb(ColorParams p) { rgb_block<Pix, LAYOUT>(p); }

// all pictures of a batch in ONE launch: blockIdx.z selects the picture's parameter block (grid x / y cover the largest one)
template <typename Pix, int LAYOUT>
__global__ __launch_bounds__(256) void k_ycbcr_to_rgb_batch(const ColorParams* __restrict__ ps)
{
  const ColorParams p = ps[blockIdx.z];   // wave-uniform: scalar loads into SGPRs
  rgb_block<Pix, LAYOUT>(p);
}

// a13: one thread per 4 output samples of one row
template <typename Pix>
__device__ __forceinline__ int bilinear_at(const Pix* in, size_t is /*samples*/, int w, int h, int x, int y)
{
  // chroma_sampling.cc:611-708, expressed per output sample.  `is` is the input stride in samples.
  if (x == 0 && y == 0) return in[0];
  if (y == 0) {  // top border (note the reference's cx / 2 indexing, :620-626)
    if ((x & 1) && (x - 1) / 2 < (w - 1) / 2) { int cx = (x - 1) / 2; return (3 * in[cx / 2] + 1 * in[cx / 2 + 1] + 2) / 4; }
    if (!(x & 1) && (x - 2) / 2 < (w - 1) / 2) { int cx = (x - 2) / 2; return (1 * in[cx / 2] + 3 * in[cx / 2 + 1] + 2) / 4; }
    if (w % 2 == 0 && x == w - 1) return in[w / 2 - 1];
    return 0;
  }
  if (x == 0) {  // left border :635-640
    if ((y & 1) && (y - 1) / 2 < (h - 1) / 2) { int cy = (y - 1) / 2; return (3 * in[(cy / 2) * is] + 1 * in[(cy / 2 + 1) * is] + 2) / 4; }
    if (!(y & 1) && (y - 2) / 2 < (h - 1) / 2) { int cy = (y - 2) / 2; return (1 * in[(cy / 2) * is] + 3 * in[(cy / 2 + 1) * is] + 2) / 4; }
    if (h % 2 == 0 && y == h - 1) return in[(h / 2 - 1) * is];
    return 0;
  }
  if (w % 2 == 0 && x == w - 1) {  // right border :649-656
    if (h % 2 == 0 && y == h - 1) return in[(h / 2 - 1) * is + w / 2 - 1];
    if ((y & 1) && (y - 1) / 2 < (h - 1) / 2) { int cy = (y - 1) / 2; return (3 * in[(cy / 2) * is + w / 2 - 1] + 1 * in[(cy / 2 + 1) * is + w / 2 - 1] + 2) / 4; }
    if (!(y & 1) && (y - 2) / 2 < (h - 1) / 2) { int cy = (y - 2) / 2; return (1 * in[(cy / 2) * is + w / 2 - 1] + 3 * in[(cy / 2 + 1) * is + w / 2 - 1] + 2) / 4; }
    return 0;
  }
  if (h % 2 == 0 && y == h - 1) {  // bottom border :660-667
    const Pix* row = in + (size_t)(h / 2 - 1) * is;
    if ((x & 1) && (x - 1) / 2 < (w - 1) / 2) { int cx = (x - 1) / 2; return (3 * row[cx / 2] + 1 * row[cx / 2 + 1] + 2) / 4; }
    if (!(x & 1) && (x - 2) / 2 < (w - 1) / 2) { int cx = (x - 2) / 2; return (1 * row[cx / 2] + 3 * row[cx / 2 + 1] + 2) / 4; }
    return 0;
  }
  // interior :678-708
  int xb = (x & 1) ? x : x - 1, yb = (y & 1) ? y : y - 1;
  if (xb >= w - 1 || yb >= h - 1) return 0;
  int cx = xb / 2, cy = yb / 2;
  int c00 = in[cy * is + cx], c01 = in[cy * is + cx + 1], c10 = in[(cy + 1) * is + cx], c11 = in[(cy + 1) * is + cx + 1];
  int wx1 = (x == xb) ? 1 : 3, wx0 = 4 - wx1;  // weight of the right / left chroma sample
  int wy1 = (y == yb) ? 1 : 3, wy0 = 4 - wy1;
  return (c00 * wx0 * wy0 + c01 * wx1 * wy0 + c10 * wx0 * wy1 + c11 * wx1 * wy1 + 8) / 16;
}

template <typename Pix>
__global__ __launch_bounds__(256) void k_bilinear_420_to_444(const uint8_t* in, size_t is, int w, int h, uint8_t* out, size_t os)
{
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x0 >= w || y >= h) return;
  const Pix* src = (const Pix*)in;
  Pix* dst = (Pix*)(out + (size_t)y * os);
  const size_t iss = is / sizeof(Pix);
  for (int i = 0; i < 4 && x0 + i < w; i++) dst[x0 + i] = (Pix)bilinear_at<Pix>(src, iss, w, h, x0 + i, y);
}

// Op_YCbCr422_bilinear_to_YCbCr444 (libheif/color-conversion/chroma_sampling.cc:732-954) for one chroma plane: out(0) = in(0); for even widths
// out(w - 1) = in(w / 2 - 1); the pairs (x, x + 1), x odd, between them are (3 a + b + 2) / 4 and (a + 3 b + 2) / 4 of the chroma samples
// a = in(x / 2), b = in(x / 2 + 1) (:911-927).  Four output samples per lane.
template <typename Pix>
__global__ __launch_bounds__(256) void k_bilinear_422_to_444(const uint8_t* in, size_t is, int w, int h, uint8_t* out, size_t os)
{
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x0 >= w || y >= h) return;
  const Pix* src = (const Pix*)(in + (size_t)y * is);
  Pix* dst = (Pix*)(out + (size_t)y * os);
  for (int i = 0; i < 4 && x0 + i < w; i++) {
    const int x = x0 + i;
    int v;
    if (x == 0) v = src[0];
    else if (x == w - 1 && (w & 1) == 0) v = src[w / 2 - 1];
    else {
      const int cx = (x - 1) >> 1, a = src[cx], b = src[cx + 1];     // x odd: first of the pair, x even: second
      v = (x & 1) ? (a * 3 + b + 2) / 4 : (a + b * 3 + 2) / 4;
    }
    dst[x] = (Pix)v;
  }
}

__global__ __launch_bounds__(256) void k_to_sdr(const uint8_t* in, size_t is, int w, int h, int shift, uint8_t* out, size_t os)
{
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x0 >= w || y >= h) return;
  const uint16_t* src = (const uint16_t*)(in + (size_t)y * is);
  uint8_t* dst = out + (size_t)y * os;
  if (x0 + 3 < w && ((is | (uintptr_t)in) & 7) == 0 && ((os | (uintptr_t)out) & 3) == 0) {
    uint2 v = *(const uint2*)(src + x0);
    uint32_t o = ((v.x & 0xffff) >> shift) | (((v.x >> 16) >> shift) << 8) | (((v.y & 0xffff) >> shift) << 16) | (((v.y >> 16) >> shift) << 24);
    *(uint32_t*)(dst + x0) = o;
  } else {
    for (int i = 0; i < 4 && x0 + i < w; i++) dst[x0 + i] = (uint8_t)(src[x0 + i] >> shift);
  }
}

// Op_to_hdr_planes (libheif/color-conversion/hdr_sdr.cc:25-109): 8-bit plane -> out_bits (<= 16) by replicating the bit pattern
__global__ __launch_bounds__(256) void k_to_hdr(const uint8_t* in, size_t is, int w, int h, int out_bits, uint8_t* out, size_t os)
{
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x0 >= w || y >= h) return;
  const uint8_t* src = in + (size_t)y * is;
  uint16_t* dst = (uint16_t*)(out + (size_t)y * os);
  const int shift1 = out_bits - 8, shift2 = 16 - out_bits;
  for (int i = 0; i < 4 && x0 + i < w; i++) { const int v = src[x0 + i]; dst[x0 + i] = (uint16_t)((v << shift1) | (v >> shift2)); }
}

// Op_RRGGBBaa_swap_endianness (libheif/color-conversion/rgb2rgb.cc:647-764): the two bytes of every 16-bit component trade places
__global__ __launch_bounds__(256) void k_swap16(const uint8_t* in, size_t is, int row_bytes, int h, uint8_t* out, size_t os)
{
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;     // byte offset, two components per thread
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x0 >= row_bytes || y >= h) return;
  const uint8_t* src = in + (size_t)y * is + x0;
  uint8_t* dst = out + (size_t)y * os + x0;
  if (x0 + 4 <= row_bytes && ((((uintptr_t)src) | ((uintptr_t)dst)) & 3) == 0) {
    const uint32_t v = *(const uint32_t*)src;
    *(uint32_t*)dst = ((v & 0x00ff00ffu) << 8) | ((v >> 8) & 0x00ff00ffu);
  } else {
    for (int i = 0; i + 1 < 4 && x0 + i + 1 < row_bytes; i += 2) { dst[i] = src[i + 1]; dst[i + 1] = src[i]; }
  }
}

// SMPTE ST 2084 / Rec. ITU-R BT.2100 PQ EOTF on code values: E' = v / (2^bits - 1), Y = (max(E'^(1/m2) - c1, 0) / (c2 - c3 E'^(1/m2)))^(1/m1),
// output linear light normalised to 1.0 = 10000 cd/m2, float32.  NOT in the reference (libheif has no transfer-function maths, SURVEY.md §0
// fact 5): BASELINE.json's config 4 asks for it, the oracle is the published formula in fp64 (tests), tolerance 1e-6 relative (the fp32 result's rounding).
__global__ __launch_bounds__(256) void k_pq_to_linear(const uint8_t* in, size_t is, int n_per_row, int h, int bits, int big_endian, float* out, size_t os)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= n_per_row || y >= h) return;
  uint32_t v = ((const uint16_t*)(in + (size_t)y * is))[x];
  if (big_endian) v = ((v & 255u) << 8) | (v >> 8);
  // evaluated in fp64 (the two pow() calls amplify fp32 rounding to ~5e-5 relative; the MI355X has the fp64 rate to spare on a streaming op)
  const double m1 = 2610.0 / 16384.0, m2 = 2523.0 / 4096.0 * 128.0, c1 = 3424.0 / 4096.0, c2 = 2413.0 / 4096.0 * 32.0, c3 = 2392.0 / 4096.0 * 32.0;
  const double e = (double)v / (double)((1u << bits) - 1u);
  const double p = pow(e, 1.0 / m2);
  const double num = fmax(p - c1, 0.0), den = c2 - c3 * p;
  ((float*)((uint8_t*)out + (size_t)y * os))[x] = (float)pow(num / den, 1.0 / m1);
}

// The same for code values of at most 12 bits through a table: the EOTF is a function of the code value alone, so every workgroup evaluates it
// once per code (fp64, exactly the expression above: the entries ARE the per-sample results) into LDS - 4 KB for 10 bit, 16 KB for 12 - and the
// samples become one LDS read each.  A workgroup covers 16 rows x 1024 samples (64 per thread, 8-byte loads / 16-byte stores), so the table costs
// 4 (16) evaluations per thread against 64 samples.  HBM-bound: 2 B in + 4 B out per sample.  (round 4: the per-sample fp64 pow pair ran at
// 0.5 TB/s on BASELINE config 4.)
__global__ __launch_bounds__(256) void k_pq_to_linear_lut(const uint8_t* in, size_t is, int n_per_row, int h, int bits, int big_endian, float* out, size_t os)
{
  __shared__ float lut[4096];
  const int n_codes = 1 << bits;
  const double m1 = 2610.0 / 16384.0, m2 = 2523.0 / 4096.0 * 128.0, c1 = 3424.0 / 4096.0, c2 = 2413.0 / 4096.0 * 32.0, c3 = 2392.0 / 4096.0 * 32.0;
  for (int v = threadIdx.x; v < n_codes; v += 256) {
    const double e = (double)v / (double)((1u << bits) - 1u);
    const double p = pow(e, 1.0 / m2);
    const double num = fmax(p - c1, 0.0), den = c2 - c3 * p;
    lut[v] = (float)pow(num / den, 1.0 / m1);
  }
  __syncthreads();
  const int x0 = blockIdx.x * 1024 + (int)threadIdx.x * 4;
  const int y0 = blockIdx.y * 16;
  const uint32_t mask = (uint32_t)n_codes - 1u;
  for (int r = 0; r < 16; r++) {
    const int y = y0 + r;
    if (y >= h || x0 >= n_per_row) break;
    const uint16_t* src = (const uint16_t*)(in + (size_t)y * is) + x0;
    float* dst = (float*)((uint8_t*)out + (size_t)y * os) + x0;
    uint32_t v[4];
    const bool full = x0 + 4 <= n_per_row && (((uintptr_t)src & 7u) == 0) && (((uintptr_t)dst & 15u) == 0);
    if (full) { const uint2 w = *(const uint2*)src; v[0] = w.x & 0xffffu; v[1] = w.x >> 16; v[2] = w.y & 0xffffu; v[3] = w.y >> 16; }
    else for (int k = 0; k < 4; k++) v[k] = x0 + k < n_per_row ? src[k] : 0u;
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t c = v[k];
      if (big_endian) c = ((c & 255u) << 8) | (c >> 8);
      if (c <= mask) f[k] = lut[c];
      else {   // a code above 2^bits - 1 (not produced by the colour stage): the formula itself, as k_pq_to_linear
        const double e = (double)c / (double)((1u << bits) - 1u);
        const double p = pow(e, 1.0 / m2);
        f[k] = (float)pow(fmax(p - c1, 0.0) / (c2 - c3 * p), 1.0 / m1);
      }
    }
    if (full) *(float4*)dst = make_float4(f[0], f[1], f[2], f[3]);
    else for (int k = 0; k < 4; k++) if (x0 + k < n_per_row) dst[k] = f[k];
  }
}

// Hybrid log-gamma (ARIB STD-B67 / Rec. ITU-R BT.2100 table 5, the inverse OETF): E = E'^2 / 3 for E' <= 1/2, (exp((E' - c) / a) + b) / 12 above, with
// a = 0.17883277, b = 1 - 4a, c = 1/2 - a ln(4a); per component, scene linear light normalised to 1.0, float32.  (The display's OOTF - a gain over the
// scene luminance with the system gamma of the viewing environment - is the renderer's business and mixes the components; it is not applied.)  Like
// the PQ stage it is NOT in the reference (libheif has no transfer-function maths) and it is the PQ kernels' structure with another curve.
__device__ __forceinline__ double hlg_inverse_oetf(double e)
{
  const double a = 0.17883277, b = 1.0 - 4.0 * a, c = 0.5 - a * log(4.0 * a);
  return e <= 0.5 ? e * e / 3.0 : (exp((e - c) / a) + b) / 12.0;
}
__global__ __launch_bounds__(256) void k_hlg_to_linear(const uint8_t* in, size_t is, int n_per_row, int h, int bits, int big_endian, float* out, size_t os)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= n_per_row || y >= h) return;
  uint32_t v = ((const uint16_t*)(in + (size_t)y * is))[x];
  if (big_endian) v = ((v & 255u) << 8) | (v >> 8);
  ((float*)((uint8_t*)out + (size_t)y * os))[x] = (float)hlg_inverse_oetf((double)v / (double)((1u << bits) - 1u));
}
__global__ __launch_bounds__(256) void k_hlg_to_linear_lut(const uint8_t* in, size_t is, int n_per_row, int h, int bits, int big_endian, float* out, size_t os)
{
  __shared__ float lut[4096];
  const int n_codes = 1 << bits;
  for (int v = threadIdx.x; v < n_codes; v += 256) lut[v] = (float)hlg_inverse_oetf((double)v / (double)((1u << bits) - 1u));
  __syncthreads();
  const int x0 = blockIdx.x * 1024 + (int)threadIdx.x * 4;
  const int y0 = blockIdx.y * 16;
  const uint32_t mask = (uint32_t)n_codes - 1u;
  for (int r = 0; r < 16; r++) {
    const int y = y0 + r;
    if (y >= h || x0 >= n_per_row) break;
    const uint16_t* src = (const uint16_t*)(in + (size_t)y * is) + x0;
    float* dst = (float*)((uint8_t*)out + (size_t)y * os) + x0;
    uint32_t v[4];
    const bool full = x0 + 4 <= n_per_row && (((uintptr_t)src & 7u) == 0) && (((uintptr_t)dst & 15u) == 0);
    if (full) { const uint2 w = *(const uint2*)src; v[0] = w.x & 0xffffu; v[1] = w.x >> 16; v[2] = w.y & 0xffffu; v[3] = w.y >> 16; }
    else for (int k = 0; k < 4; k++) v[k] = x0 + k < n_per_row ? src[k] : 0u;
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t c = v[k];
      if (big_endian) c = ((c & 255u) << 8) | (c >> 8);
      f[k] = c <= mask ? lut[c] : (float)hlg_inverse_oetf((double)c / (double)((1u << bits) - 1u));   // (a code above 2^bits - 1: the formula itself)
    }
    if (full) *(float4*)dst = make_float4(f[0], f[1], f[2], f[3]);
    else for (int k = 0; k < 4; k++) if (x0 + k < n_per_row) dst[k] = f[k];
  }
}

// ---- host side -------------------------------------------------------------------------------

// libheif/nclx.cc:45-72
bool primaries_of(int idx, float p[8])
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
  for (auto& r : t)
    if ((int)r[0] == idx) { memcpy(p, &r[1], 8 * sizeof(float)); return true; }
  memset(p, 0, 8 * sizeof(float));
  return false;
}

// libheif/nclx.cc:84-173.  Host float arithmetic, contraction off, same operation order.
void coefficients(const hipdec_nclx* n, float out[4])
{
  float Kr = 0, Kb = 0;
  if (n && n->has_nclx) {
    int m = n->matrix_coefficients;
    if (m == 12 || m == 13) {
      float p[8];
      primaries_of(n->colour_primaries, p);
      float gx = p[0], gy = p[1], bx = p[2], by = p[3], rx = p[4], ry = p[5], wx = p[6], wy = p[7];
      float zr = 1 - (rx + ry), zg = 1 - (gx + gy), zb = 1 - (bx + by), zw = 1 - (wx + wy);
      float denom = wy * (rx * (gy * zb - by * zg) + gx * (by * zr - ry * zb) + bx * (ry * zg - gy * zr));
      if (denom != 0.0f) {
        Kr = (ry * (wx * (gy * zb - by * zg) + wy * (bx * zg - gx * zb) + zw * (gx * by - bx * gy))) / denom;
        Kb = (by * (wx * (ry * zg - gy * zr) + wy * (gx * zr - rx * zg) + zw * (rx * gy - gx * ry))) / denom;
      }
    } else {
      switch (m) {
        case 1: Kr = 0.2126f; Kb = 0.0722f; break;
        case 4: Kr = 0.30f; Kb = 0.11f; break;
        case 5: case 6: Kr = 0.299f; Kb = 0.114f; break;
        case 7: Kr = 0.212f; Kb = 0.087f; break;
        case 9: case 10: Kr = 0.2627f; Kb = 0.0593f; break;
        default: break;
      }
    }
  }
  if (Kb != 0 || Kr != 0) {
    out[0] = 2 * (-Kr + 1);
    out[1] = 2 * Kb * (-Kb + 1) / (Kb + Kr - 1);
    out[2] = 2 * Kr * (-Kr + 1) / (Kb + Kr - 1);
    out[3] = 2 * (-Kb + 1);
  } else {
    out[0] = 1.402f; out[1] = -0.344136f; out[2] = -0.714136f; out[3] = 1.772f;
  }
}

// Capture mode (hipdec_batch_to_rgb_all): the per-picture entry points run their argument checks and the planner rules
// as usual, but instead of launching they record the parameter block; the recorded blocks then go out as one launch.
struct Captured { ColorParams p; int variant; };
thread_local std::vector<Captured> t_captured;
thread_local bool t_capture = false;

template <typename Pix, int LAYOUT>
int launch_rgb(const ColorParams& p, hipStream_t s)
{
  if (t_capture) { t_captured.push_back(Captured{p, (int)sizeof(Pix) * 16 + LAYOUT}); return 0; }
  dim3 block(64, 4);
  dim3 grid(((p.w + 3) / 4 + 63) / 64, ((p.h + 1) / 2 + 3) / 4);
  hipLaunchKernelGGL((k_ycbcr_to_rgb<Pix, LAYOUT>), grid, block, 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hipdec::set_error(HIPDEC_ERR_DEVICE, "colour kernel launch: %s", hipGetErrorString(e));
  return 0;
}

int fill_common(ColorParams& p, const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w,
                int h, int bpp, int chroma, const hipdec_nclx* nclx)
{
  if (!y || !cb || !cr || w <= 0 || h <= 0) return hipdec::set_error(HIPDEC_ERR_INVALID_ARGUMENT, "colour: bad plane arguments");
  if (chroma < 1 || chroma > 3) return hipdec::set_error(HIPDEC_ERR_INVALID_ARGUMENT, "colour: chroma must be 1 (420), 2 (422) or 3 (444)");
  memset(&p, 0, sizeof(p));
  p.y = (const uint8_t*)y; p.cb = (const uint8_t*)cb; p.cr = (const uint8_t*)cr;
  p.ys = ys; p.cbs = cbs; p.crs = crs; p.w = w; p.h = h; p.bpp = bpp;
  p.shiftH = chroma == 3 ? 0 : 1;
  p.shiftV = chroma == 1 ? 1 : 0;
  float c[4];
  coefficients(nclx, c);
  p.f_r_cr = c[0]; p.f_g_cb = c[1]; p.f_g_cr = c[2]; p.f_b_cb = c[3];
  p.i_r_cr = (int)std::lround(256 * c[0]); p.i_g_cb = (int)std::lround(256 * c[1]);
  p.i_g_cr = (int)std::lround(256 * c[2]); p.i_b_cb = (int)std::lround(256 * c[3]);
  p.full_range = (nclx && nclx->has_nclx) ? nclx->full_range_flag : 1;
  return 0;
}

// arithmetic selection of Op_YCbCr_to_RGB (yuv2rgb.cc:208-282)
int generic_arith(const hipdec_nclx* nclx)
{
  int matrix = (nclx && nclx->has_nclx) ? nclx->matrix_coefficients : 2;
  int full = (nclx && nclx->has_nclx) ? nclx->full_range_flag : 1;
  if (matrix == 0) return full ? AR_GBR_FULL : AR_GBR_LIMITED;
  if (matrix == 8) return AR_YCGCO;
  if (matrix == 16) return AR_YCGCO_RE;
  return AR_FLOAT;
}

template <typename Pix, int LAYOUT>
void launch_rgb_batch(const ColorParams* dev, int n, int max_w, int max_h, hipStream_t s)
{
  dim3 block(64, 4);
  dim3 grid(((max_w + 3) / 4 + 63) / 64, ((max_h + 1) / 2 + 3) / 4, n);
  hipLaunchKernelGGL((k_ycbcr_to_rgb_batch<Pix, LAYOUT>), grid, block, 0, s, dev);
}

}  // namespace

// variant = sizeof(Pix) * 16 + LAYOUT, as recorded by launch_rgb
#define HIPDEC_RGB_VARIANTS(X)                                                                                     \
  X(16 + LO_PLANAR, uint8_t, LO_PLANAR) X(32 + LO_PLANAR, uint16_t, LO_PLANAR) X(16 + LO_RGB24, uint8_t, LO_RGB24) \
  X(16 + LO_RGBA32, uint8_t, LO_RGBA32) X(32 + LO_RRGGBB_BE, uint16_t, LO_RRGGBB_BE) X(32 + LO_RRGGBB_LE, uint16_t, LO_RRGGBB_LE)                     \
  X(32 + LO_RGB24, uint16_t, LO_RGB24) X(32 + LO_RGBA32, uint16_t, LO_RGBA32)

namespace hipdec {

void color_capture_begin()
{
  t_captured.clear();
  t_capture = true;
}

void color_capture_abort()
{
  t_captured.clear();
  t_capture = false;
}

// Ends capture mode WITHOUT launching: uploads the recorded parameter blocks (one per captured call, in call order) and hands back the
// device array — for a kernel of another translation unit that consumes them (SAO with fused RGB emission, filter_kernels.hip).
// *uniform_variant = the kernel variant all blocks share (sizeof(Pix) * 16 + LAYOUT), or -1 when they differ.
int color_capture_take(ColorBatchState& st, hipStream_t s, const void** dev, int* uniform_variant, int* count)
{
  t_capture = false;
  std::vector<Captured> caps;
  caps.swap(t_captured);
  *dev = nullptr; *uniform_variant = -1; *count = (int)caps.size();
  if (caps.empty()) return 0;
  bool same = true;
  for (const auto& c : caps) same = same && c.variant == caps[0].variant;
  const size_t bytes = caps.size() * sizeof(ColorParams);
  std::vector<uint8_t> host(bytes);
  for (size_t i = 0; i < caps.size(); i++) memcpy(host.data() + i * sizeof(ColorParams), &caps[i].p, sizeof(ColorParams));
  if (st.dev_bytes < bytes) {
    if (st.dev) arena_release(st.dev, st.dev_bytes);
    st.dev = nullptr; st.dev_bytes = 0; st.host.clear();
    HIPDEC_CHECK_HIP(arena_acquire(&st.dev, bytes, &st.dev_bytes));
  }
  if (st.host != host) {
    st.prev.swap(st.host);   // (not freed while its copy may be pending)
    st.host.swap(host);
    HIPDEC_CHECK_HIP(hipMemcpyAsync(st.dev, st.host.data(), bytes, hipMemcpyHostToDevice, s));
  }
  *dev = st.dev;
  if (same) *uniform_variant = caps[0].variant;
  return 0;
}
int color_variant_rgb24_u8() { return (int)sizeof(uint8_t) * 16 + LO_RGB24; }

int color_capture_launch(ColorBatchState& st, hipStream_t s)
{
  t_capture = false;
  std::vector<Captured> caps;
  caps.swap(t_captured);
  if (caps.empty()) return 0;
  bool same = true;
  int max_w = 0, max_h = 0;
  for (const auto& c : caps) { same = same && c.variant == caps[0].variant; max_w = c.p.w > max_w ? c.p.w : max_w; max_h = c.p.h > max_h ? c.p.h : max_h; }
  if (!same || caps.size() == 1) {   // mixed kernel variants cannot share a launch
    for (const auto& c : caps) {
      int rc = HIPDEC_ERR_UNSUPPORTED;
      switch (c.variant) {
#define X(id, Pix, LO) case id: rc = launch_rgb<Pix, LO>(c.p, s); break;
        HIPDEC_RGB_VARIANTS(X)
#undef X
        default: break;
      }
      if (rc) return rc;
    }
    return 0;
  }
  const size_t bytes = caps.size() * sizeof(ColorParams);
  std::vector<uint8_t> host(bytes);
  for (size_t i = 0; i < caps.size(); i++) memcpy(host.data() + i * sizeof(ColorParams), &caps[i].p, sizeof(ColorParams));
  if (st.dev_bytes < bytes) {   // from the arena pool: hipFree() would synchronise the device every time a batch is retired
    if (st.dev) arena_release(st.dev, st.dev_bytes);
    st.dev = nullptr; st.dev_bytes = 0; st.host.clear();
    HIPDEC_CHECK_HIP(arena_acquire(&st.dev, bytes, &st.dev_bytes));
  }
  if (st.host != host) {   // steady state (same planes, same outputs): nothing to upload
    st.prev.swap(st.host);   // (not freed while its copy may be pending)
    st.host.swap(host);
    HIPDEC_CHECK_HIP(hipMemcpyAsync(st.dev, st.host.data(), bytes, hipMemcpyHostToDevice, s));
  }
  const ColorParams* dev = (const ColorParams*)st.dev;
  switch (caps[0].variant) {
#define X(id, Pix, LO) case id: launch_rgb_batch<Pix, LO>(dev, (int)caps.size(), max_w, max_h, s); break;
    HIPDEC_RGB_VARIANTS(X)
#undef X
    default: return set_error(HIPDEC_ERR_UNSUPPORTED, "colour batch: unknown kernel variant");
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(HIPDEC_ERR_DEVICE, "colour batch launch: %s", hipGetErrorString(e));
  return 0;
}

void color_batch_state_free(ColorBatchState& st)
{
  if (st.dev) arena_release(st.dev, st.dev_bytes);
  st.dev = nullptr; st.dev_bytes = 0; st.host.clear();
}

}  // namespace hipdec

using namespace hipdec;

extern "C" {

void hipdec_color_coefficients(const hipdec_nclx* nclx, float out[4]) { if (out) coefficients(nclx, out); }   // (nclx NULL: the reference's defaults)

int hipdec_color_420_to_rgb24(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w, int h,
                              const hipdec_nclx* nclx, void* out, size_t out_stride, int with_alpha, void* stream)
{
  if (int rc = ensure_init()) return rc;
  // Op_YCbCr420_to_RGB24::state_after_conversion (yuv2rgb.cc:298-341) refuses these inputs
  if (nclx && nclx->has_nclx) {
    int m = nclx->matrix_coefficients;
    if (m == 0 || m == 8 || m == 11 || m == 14)
      return set_error(HIPDEC_ERR_UNSUPPORTED, "420_to_rgb24: matrix_coefficients %d is not handled by this op", m);
    if (!nclx->full_range_flag) return set_error(HIPDEC_ERR_UNSUPPORTED, "420_to_rgb24: limited range is not handled by this op");
  }
  ColorParams p;
  if (int rc = fill_common(p, y, ys, cb, cbs, cr, crs, w, h, 8, 1, nclx)) return rc;
  if (!out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "420_to_rgb24: out is NULL");
  p.arith = AR_INT88; p.o0 = (uint8_t*)out; p.os = out_stride;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  return with_alpha ? launch_rgb<uint8_t, LO_RGBA32>(p, s) : launch_rgb<uint8_t, LO_RGB24>(p, s);
}

/* RGBA with a real alpha plane (Op_YCbCr420_to_RGB32 with an alpha channel, yuv2rgb.cc:521-553; the float chain copies the alpha
 * plane the same way, rgb2rgb.cc:72-150): `alpha` NULL fills 0xFF */
int hipdec_color_420_to_rgba_alpha(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w, int h,
                                   const hipdec_nclx* nclx, const void* alpha, size_t alpha_stride, int integer_op, int chroma,
                                   void* out, size_t out_stride, void* stream)
{
  if (int rc = ensure_init()) return rc;
  ColorParams p;
  if (int rc = fill_common(p, y, ys, cb, cbs, cr, crs, w, h, 8, integer_op ? 1 : chroma, nclx)) return rc;
  if (!out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "420_to_rgba: out is NULL");
  p.arith = integer_op ? AR_INT88 : generic_arith(nclx);
  p.o0 = (uint8_t*)out; p.os = out_stride; p.a = (const uint8_t*)alpha; p.as = alpha_stride;
  return launch_rgb<uint8_t, LO_RGBA32>(p, stream ? (hipStream_t)stream : default_stream());
}

int hipdec_color_ycbcr_to_rgb_planar(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w,
                                     int h, int bpp, int chroma, const hipdec_nclx* nclx, void* r, void* g, void* b,
                                     size_t out_stride, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (bpp < 8 || bpp > 14) return set_error(HIPDEC_ERR_UNSUPPORTED, "ycbcr_to_rgb: bits per pixel %d outside 8..14", bpp);
  if (nclx && nclx->has_nclx && (nclx->matrix_coefficients == 11 || nclx->matrix_coefficients == 14))
    return set_error(HIPDEC_ERR_UNSUPPORTED, "ycbcr_to_rgb: matrix_coefficients %d unsupported (as in the reference)", nclx->matrix_coefficients);
  ColorParams p;
  if (int rc = fill_common(p, y, ys, cb, cbs, cr, crs, w, h, bpp, chroma, nclx)) return rc;
  if (!r || !g || !b) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "ycbcr_to_rgb: output plane is NULL");
  p.arith = generic_arith(nclx); p.o0 = (uint8_t*)r; p.o1 = (uint8_t*)g; p.o2 = (uint8_t*)b; p.os = out_stride;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  return bpp == 8 ? launch_rgb<uint8_t, LO_PLANAR>(p, s) : launch_rgb<uint16_t, LO_PLANAR>(p, s);
}

int hipdec_color_ycbcr_to_rgb24_float(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w,
                                      int h, int chroma, const hipdec_nclx* nclx, void* out, size_t out_stride, int with_alpha,
                                      void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (nclx && nclx->has_nclx && (nclx->matrix_coefficients == 11 || nclx->matrix_coefficients == 14))
    return set_error(HIPDEC_ERR_UNSUPPORTED, "ycbcr_to_rgb24: matrix_coefficients %d unsupported (as in the reference)", nclx->matrix_coefficients);
  ColorParams p;
  if (int rc = fill_common(p, y, ys, cb, cbs, cr, crs, w, h, 8, chroma, nclx)) return rc;
  if (!out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "ycbcr_to_rgb24: out is NULL");
  p.arith = generic_arith(nclx); p.o0 = (uint8_t*)out; p.os = out_stride;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  return with_alpha ? launch_rgb<uint8_t, LO_RGBA32>(p, s) : launch_rgb<uint8_t, LO_RGB24>(p, s);
}

int hipdec_color_420_to_rrggbb(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w, int h,
                               int bpp, const hipdec_nclx* nclx, void* out, size_t out_stride, int little_endian, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (bpp <= 8 || bpp > 16) return set_error(HIPDEC_ERR_UNSUPPORTED, "420_to_rrggbb: needs more than 8 bits per pixel");
  if (nclx && nclx->has_nclx) {
    int m = nclx->matrix_coefficients;  // yuv2rgb.cc:590-593
    if (m == 0 || m == 8 || m == 11 || m == 14)
      return set_error(HIPDEC_ERR_UNSUPPORTED, "420_to_rrggbb: matrix_coefficients %d is not handled by this op", m);
  }
  ColorParams p;
  if (int rc = fill_common(p, y, ys, cb, cbs, cr, crs, w, h, bpp, 1, nclx)) return rc;
  if (!out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "420_to_rrggbb: out is NULL");
  p.arith = AR_FLOAT; p.o0 = (uint8_t*)out; p.os = out_stride;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  return little_endian ? launch_rgb<uint16_t, LO_RRGGBB_LE>(p, s) : launch_rgb<uint16_t, LO_RRGGBB_BE>(p, s);
}

/* Op_mono_to_RGB24_32 (libheif/color-conversion/monochrome.cc): an 8-bit monochrome plane to interleaved RGB24 / RGBA32, R = G = B = Y; the alpha
 * plane of the image is copied when there is one (NULL: 0xFF) */
int hipdec_color_mono_to_rgb24(const void* y, size_t ys, const void* alpha, size_t alpha_stride, int w, int h, void* out, size_t out_stride,
                               int with_alpha, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (!y || !out || w <= 0 || h <= 0) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "mono_to_rgb24: bad arguments");
  if (alpha && !with_alpha) return set_error(HIPDEC_ERR_UNSUPPORTED, "mono_to_rgb24: dropping an alpha plane is left to the stock ops");
  ColorParams p;
  memset(&p, 0, sizeof(p));
  p.y = (const uint8_t*)y; p.ys = ys; p.cb = p.cr = (const uint8_t*)y; p.cbs = p.crs = ys;   // (never read: AR_MONO)
  p.w = w; p.h = h; p.bpp = 8; p.arith = AR_MONO; p.full_range = 1;
  p.a = (const uint8_t*)alpha; p.as = alpha_stride;
  p.o0 = (uint8_t*)out; p.os = out_stride;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  return with_alpha ? launch_rgb<uint8_t, LO_RGBA32>(p, s) : launch_rgb<uint8_t, LO_RGB24>(p, s);
}

/* > 8-bit planes to 8-bit interleaved RGB(A), one pass, as the two chains the reference's planner builds for it (which one: hipdec_color_plan):
 *   sdr_first = 1: Op_to_sdr_planes on Y, Cb, Cr, then Op_YCbCr420_to_RGB24 / _RGB32 (4:2:0, full range, a matrix the integer op takes)
 *   sdr_first = 0: Op_YCbCr_to_RGB<uint16_t> at the input depth, Op_to_sdr_planes on R, G, B, Op_RGB_to_RGB24_32 (everything else) */
int hipdec_color_hdr_to_rgb24(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w, int h, int bpp, int chroma,
                              const hipdec_nclx* nclx, void* out, size_t out_stride, int with_alpha, int sdr_first, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (bpp <= 8 || bpp > 14) return set_error(HIPDEC_ERR_UNSUPPORTED, "hdr_to_rgb24: bits per pixel %d outside 9..14", bpp);
  if (nclx && nclx->has_nclx && (nclx->matrix_coefficients == 11 || nclx->matrix_coefficients == 14))
    return set_error(HIPDEC_ERR_UNSUPPORTED, "hdr_to_rgb24: matrix_coefficients %d unsupported (as in the reference)", nclx->matrix_coefficients);
  ColorParams p;
  if (sdr_first) {
    if (chroma != 1) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "hdr_to_rgb24: the sdr-first chain is the 4:2:0 integer op's");
    if (nclx && nclx->has_nclx) {   // Op_YCbCr420_to_RGB24::state_after_conversion (yuv2rgb.cc:298-341)
      const int m = nclx->matrix_coefficients;
      if (m == 0 || m == 8 || !nclx->full_range_flag) return set_error(HIPDEC_ERR_UNSUPPORTED, "hdr_to_rgb24: the integer op does not take this colour profile");
    }
    if (int rc = fill_common(p, y, ys, cb, cbs, cr, crs, w, h, 8, 1, nclx)) return rc;
    p.arith = AR_INT88; p.in_shift = bpp - 8;
  } else {
    if (int rc = fill_common(p, y, ys, cb, cbs, cr, crs, w, h, bpp, chroma, nclx)) return rc;
    p.arith = generic_arith(nclx); p.out_shift = bpp - 8;
  }
  if (!out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "hdr_to_rgb24: out is NULL");
  p.o0 = (uint8_t*)out; p.os = out_stride;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  return with_alpha ? launch_rgb<uint16_t, LO_RGBA32>(p, s) : launch_rgb<uint16_t, LO_RGB24>(p, s);
}

/* Op_YCbCr_to_RGB<uint16_t> (yuv2rgb.cc:92-292, nearest-neighbour chroma for 4:2:0 / 4:2:2 inputs) + Op_RGB_HDR_to_RRGGBBaa_BE (rgb2rgb.cc)
 * [+ Op_RRGGBBaa_swap_endianness for little endian] as ONE pass: what the reference's planner chains for > 8-bit planes of any chroma format that
 * the 4:2:0-only op above does not take (4:2:2, 4:4:4, matrix_coefficients 0 / 8); the components keep the input bit depth */
int hipdec_color_ycbcr_to_rrggbb_float(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w, int h,
                                       int bpp, int chroma, const hipdec_nclx* nclx, void* out, size_t out_stride, int little_endian, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (bpp <= 8 || bpp > 14) return set_error(HIPDEC_ERR_UNSUPPORTED, "ycbcr_to_rrggbb: bits per pixel %d outside 9..14", bpp);
  if (nclx && nclx->has_nclx && (nclx->matrix_coefficients == 11 || nclx->matrix_coefficients == 14))
    return set_error(HIPDEC_ERR_UNSUPPORTED, "ycbcr_to_rrggbb: matrix_coefficients %d unsupported (as in the reference)", nclx->matrix_coefficients);
  ColorParams p;
  if (int rc = fill_common(p, y, ys, cb, cbs, cr, crs, w, h, bpp, chroma, nclx)) return rc;
  if (!out) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "ycbcr_to_rrggbb: out is NULL");
  p.arith = generic_arith(nclx); p.o0 = (uint8_t*)out; p.os = out_stride;
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  return little_endian ? launch_rgb<uint16_t, LO_RRGGBB_LE>(p, s) : launch_rgb<uint16_t, LO_RRGGBB_BE>(p, s);
}

int hipdec_color_bilinear_420_to_444(const void* in, size_t is, int w, int h, int bpp, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (!in || !out || w <= 0 || h <= 0) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "bilinear: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  dim3 block(64, 4), grid(((w + 3) / 4 + 63) / 64, (h + 3) / 4);
  if (bpp <= 8) hipLaunchKernelGGL(k_bilinear_420_to_444<uint8_t>, grid, block, 0, s, (const uint8_t*)in, is, w, h, (uint8_t*)out, os);
  else hipLaunchKernelGGL(k_bilinear_420_to_444<uint16_t>, grid, block, 0, s, (const uint8_t*)in, is, w, h, (uint8_t*)out, os);
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipdec_color_bilinear_422_to_444(const void* in, size_t is, int w, int h, int bpp, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (!in || !out || w <= 0 || h <= 0) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "bilinear 4:2:2: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  dim3 block(64, 4), grid(((w + 3) / 4 + 63) / 64, (h + 3) / 4);
  if (bpp <= 8) hipLaunchKernelGGL(k_bilinear_422_to_444<uint8_t>, grid, block, 0, s, (const uint8_t*)in, is, w, h, (uint8_t*)out, os);
  else hipLaunchKernelGGL(k_bilinear_422_to_444<uint16_t>, grid, block, 0, s, (const uint8_t*)in, is, w, h, (uint8_t*)out, os);
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipdec_color_to_sdr(const void* in, size_t is, int w, int h, int bits, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (!in || !out || w <= 0 || h <= 0 || bits <= 8 || bits > 16) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "to_sdr: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  dim3 block(64, 4), grid(((w + 3) / 4 + 63) / 64, (h + 3) / 4);
  hipLaunchKernelGGL(k_to_sdr, grid, block, 0, s, (const uint8_t*)in, is, w, h, bits - 8, (uint8_t*)out, os);
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipdec_color_to_hdr(const void* in, size_t is, int w, int h, int out_bits, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (!in || !out || w <= 0 || h <= 0 || out_bits <= 8 || out_bits > 16) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "to_hdr: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  dim3 block(64, 4), grid(((w + 3) / 4 + 63) / 64, (h + 3) / 4);
  hipLaunchKernelGGL(k_to_hdr, grid, block, 0, s, (const uint8_t*)in, is, w, h, out_bits, (uint8_t*)out, os);
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipdec_color_swap_endianness(const void* in, size_t is, int w, int h, int components, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (!in || !out || w <= 0 || h <= 0 || (components != 3 && components != 4)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "swap_endianness: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  const int row_bytes = w * components * 2;
  dim3 block(64, 4), grid(((row_bytes + 3) / 4 + 63) / 64, (h + 3) / 4);
  hipLaunchKernelGGL(k_swap16, grid, block, 0, s, (const uint8_t*)in, is, row_bytes, h, (uint8_t*)out, os);
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipdec_color_pq_to_linear(const void* in, size_t is, int w, int h, int components, int bits, int big_endian, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (!in || !out || w <= 0 || h <= 0 || components < 1 || components > 4 || bits < 8 || bits > 16)
    return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "pq_to_linear: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  const int n = w * components;
  if (bits <= 12) {
    hipLaunchKernelGGL(k_pq_to_linear_lut, dim3((n + 1023) / 1024, (h + 15) / 16), dim3(256), 0, s, (const uint8_t*)in, is, n, h, bits, big_endian, (float*)out, os);
  } else {
    dim3 block(64, 4), grid((n + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_pq_to_linear, grid, block, 0, s, (const uint8_t*)in, is, n, h, bits, big_endian, (float*)out, os);
  }
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipdec_color_hlg_to_linear(const void* in, size_t is, int w, int h, int components, int bits, int big_endian, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (!in || !out || w <= 0 || h <= 0 || components < 1 || components > 4 || bits < 8 || bits > 16)
    return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "hlg_to_linear: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  const int n = w * components;
  if (bits <= 12) {
    hipLaunchKernelGGL(k_hlg_to_linear_lut, dim3((n + 1023) / 1024, (h + 15) / 16), dim3(256), 0, s, (const uint8_t*)in, is, n, h, bits, big_endian, (float*)out, os);
  } else {
    dim3 block(64, 4), grid((n + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_hlg_to_linear, grid, block, 0, s, (const uint8_t*)in, is, n, h, bits, big_endian, (float*)out, os);
  }
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
