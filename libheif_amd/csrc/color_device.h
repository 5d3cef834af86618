// color_device.h — device-side pieces of the colour stage shared by color.hip (the colour kernels) and filter_kernels.hip (SAO with
// the RGB24 emission fused into its store path): the per-picture parameter block and the per-pixel conversion, i.e. the arithmetic of
// Op_YCbCr420_to_RGB24 (libheif/color-conversion/yuv2rgb.cc:377-421) and Op_YCbCr_to_RGB<Pixel> (yuv2rgb.cc:208-282).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hipdec {
namespace colordev {

enum Arith { AR_INT88 = 0, AR_FLOAT = 1, AR_GBR_FULL = 2, AR_GBR_LIMITED = 3, AR_YCGCO = 4, AR_YCGCO_RE = 5,
             AR_MONO = 6 /* Op_mono_to_RGB24_32 (monochrome.cc): R = G = B = Y, no chroma planes */ };
enum Layout { LO_PLANAR = 0, LO_RGB24 = 1, LO_RGBA32 = 2, LO_RRGGBB_BE = 3, LO_RRGGBB_LE = 4 };

struct ColorParams {
  const uint8_t *y, *cb, *cr;
  size_t ys, cbs, crs;
  uint8_t *o0, *o1, *o2;
  size_t os;
  const uint8_t* a;       // 8-bit alpha plane for the RGBA layout (yuv2rgb.cc:521-553), NULL: filled with 0xFF
  size_t as;
  int w, h, bpp, shiftH, shiftV;
  int arith;
  int i_r_cr, i_g_cb, i_g_cr, i_b_cb;
  float f_r_cr, f_g_cb, f_g_cr, f_b_cb;
  int full_range;
  int in_shift;    // > 8-bit planes through an 8-bit chain: Op_to_sdr_planes (v >> (bits - 8), hdr_sdr.cc:193) applied to Y, Cb, Cr as they are loaded
  int out_shift;   // > 8-bit planes to 8-bit interleaved RGB: Op_to_sdr_planes applied to R, G, B after the conversion at the input depth
};

// A pointer that came out of a parameter block in MEMORY (or out of an integer) is a generic pointer to the compiler: every access through it is a FLAT
// instruction (aperture check per access; it counts in vmcnt AND lgkmcnt, so a wait for LDS also waits for the plane traffic).  These buffers are device
// global memory by contract; only an access through a pointer TYPED as address space 1 makes the compiler believe it (a cast there and back is folded
// away, assumptions are not used): HIPDEC_GLOBAL at the access sites.  Round 5: the batched colour kernels had 40 - 46 FLAT instructions each, k_sao_rgb 12, k_mc 14.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPDEC_HOST_EMU)
#define HIPDEC_GLOBAL __attribute__((address_space(1)))
#else
#define HIPDEC_GLOBAL
#endif

__device__ __forceinline__ int clip_i(int x, int maxi) { return x < 0 ? 0 : (x > maxi ? maxi : x); }
// libheif/common_utils.h:108-114 clip_f_u16: (int32)(fx + 0.5f), then clamp
__device__ __forceinline__ int clip_f(float fx, int maxi)
{
  int x = (int)(fx + 0.5f);
  return x < 0 ? 0 : (x > maxi ? maxi : x);
}

__device__ __forceinline__ void convert_px(const ColorParams& p, int Y, int Cb, int Cr, int& R, int& G, int& B)
{
  const int fullRange = (1 << p.bpp) - 1;
  const int halfRange = 1 << (p.bpp - 1);
  switch (p.arith) {
    case AR_INT88: {  // yuv2rgb.cc:377-421
      int cb = Cb - 128, cr = Cr - 128;
      R = clip_i(Y + ((p.i_r_cr * cr + 128) >> 8), 255);
      G = clip_i(Y + ((p.i_g_cb * cb + p.i_g_cr * cr + 128) >> 8), 255);
      B = clip_i(Y + ((p.i_b_cb * cb + 128) >> 8), 255);
      break;
    }
    case AR_MONO: R = Y; G = Y; B = Y; break;
    case AR_GBR_FULL: R = Cr; G = Y; B = Cb; break;  // yuv2rgb.cc:224-229
    case AR_GBR_LIMITED: {                             // yuv2rgb.cc:230-236
      float off = (float)(16 << (p.bpp - 8));
      R = clip_f(((float)Cr - off) * 1.1429f, fullRange);
      G = clip_f(((float)Y - off) * 1.1689f, fullRange);
      B = clip_f(((float)Cb - off) * 1.1429f, fullRange);
      break;
    }
    case AR_YCGCO: {  // yuv2rgb.cc:237-249 (clip_int_u8 even for >8 bit, as the reference does)
      int cb = Cb - halfRange, cr = Cr - halfRange;
      R = clip_i(Y - cb + cr, 255); G = clip_i(Y + cb, 255); B = clip_i(Y - cb - cr, 255);
      break;
    }
    case AR_YCGCO_RE: {  // yuv2rgb.cc:250-266, int16 arithmetic
      short yy = (short)Y;
      short cb = (short)((short)Cb - (short)halfRange), cr = (short)((short)Cr - (short)halfRange);
      short t = (short)(yy - (cb >> 1));
      short g = (short)(t + cb);
      short b = (short)(t - (cr >> 1));
      short r = (short)(b + cr);
      R = clip_i(r * 4, fullRange); G = clip_i(g * 4, fullRange); B = clip_i(b * 4, fullRange);
      break;
    }
    default: {  // AR_FLOAT  yuv2rgb.cc:267-282 / :699-710
      float yv = (float)Y, cb = (float)(Cb - halfRange), cr = (float)(Cr - halfRange);
      if (!p.full_range) {
        yv = (yv - (float)(16 << (p.bpp - 8))) * 1.1689f;
        cb = cb * 1.1429f;
        cr = cr * 1.1429f;
      }
      R = clip_f(yv + p.f_r_cr * cr, fullRange);
      G = clip_f(yv + p.f_g_cb * cb + p.f_g_cr * cr, fullRange);
      B = clip_f(yv + p.f_b_cb * cb, fullRange);
      break;
    }
  }
}

struct __attribute__((packed, aligned(4))) U3 { uint32_t a, b, c; };

}  // namespace colordev
}  // namespace hipdec
