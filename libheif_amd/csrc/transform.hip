// transform.hip — the transformative item properties on the device: 'irot', 'imir', 'clap' as plane kernels.
//
// libheif applies them on the host after the decoder plugin returned (ImageItem::decode_image, libheif/image-items/image_item.cc:949-1081):
//   irot  HeifPixelImage::rotate_ccw      -> ComponentStorage::rotate_ccw<T>      (libheif/image/pixelimage.cc:1175-1300, :1305-1333)
//   imir  HeifPixelImage::mirror_inplace  -> ComponentStorage::mirror_inplace<T>  (:1358-1430, :1337-1355)
//   clap  HeifPixelImage::crop                                                    (:1433-1530)
// each a per-sample index remap of every plane.  Here: one launch per plane, samples of 1 or 2 bytes.
//   * 90 / 270 degrees are a transpose: 64x64 tiles through LDS so that both the reads (along input rows) and the writes (along output rows)
//     are coalesced — the naive form writes one sample per 64-byte segment;
//   * 180 degrees, the two mirrors and the crop keep the row direction: a remap kernel (x -> x0 + dx * x, y -> y0 + dy * y), four output
//     samples per lane.
// All are HBM-bound: algorithmic bytes = one read + one write of the plane.  The chroma rules of the reference (odd sizes of a subsampled
// image would first be converted to 4:4:4) are applied by hipdec_image_transform (decoder.hip), which owns the plane bookkeeping.
#include <hip/hip_runtime.h>
#include "heif_hipdec.h"
#include "hipdec_internal.h"

namespace hipdec {
namespace {

// out(X, Y), X < ow = h, Y < oh = w:   270: in(col = Y, row = h - 1 - X)      90: in(col = w - 1 - Y, row = X)      (pixelimage.cc:1318-1331)
template <typename T>
__global__ __launch_bounds__(256) void k_rotate_quarter(const T* __restrict__ src, size_t ss, int w, int h, T* __restrict__ dst, size_t ds, int angle)
{
  __shared__ T tile[64][64 + 4 / sizeof(T)];   // row = X offset, column = Y offset
  const int tx = (int)threadIdx.x & 63, ty = (int)threadIdx.x >> 6;
  const int ox = (int)blockIdx.x * 64, oy = (int)blockIdx.y * 64;
  for (int k = 0; k < 16; k++) {
    const int sr = ty + 4 * k, X = ox + sr, Y = oy + tx;
    if (X < h && Y < w) tile[sr][tx] = angle == 270 ? src[(size_t)(h - 1 - X) * ss + (size_t)Y] : src[(size_t)X * ss + (size_t)(w - 1 - Y)];
  }
  __syncthreads();
  for (int k = 0; k < 16; k++) {
    const int yo = ty + 4 * k, X = ox + tx, Y = oy + yo;
    if (X < h && Y < w) dst[(size_t)Y * ds + (size_t)X] = tile[tx][yo];
  }
}

// out(X, Y) = in(x0 + dx * X, y0 + dy * Y), dx, dy = +1 / -1; four consecutive output samples per lane
template <typename T>
__global__ __launch_bounds__(256) void k_remap(const T* __restrict__ src, size_t ss, T* __restrict__ dst, size_t ds, int ow, int oh, int x0, int dx, int y0, int dy)
{
  const int X = ((int)blockIdx.x * 64 + ((int)threadIdx.x & 63)) * 4, Y = (int)blockIdx.y * 4 + ((int)threadIdx.x >> 6);
  if (Y >= oh || X >= ow) return;
  const T* row = src + (size_t)(y0 + dy * Y) * ss;
  T* out = dst + (size_t)Y * ds + X;
  T v[4];
  const int n = ow - X < 4 ? ow - X : 4;
  for (int i = 0; i < 4; i++) v[i] = i < n ? row[x0 + dx * (X + i)] : (T)0;
  for (int i = 0; i < n; i++) out[i] = v[i];
}

template <typename T>
int rotate_plane(const void* in, size_t is, int w, int h, int angle, void* out, size_t os, hipStream_t s)
{
  if (angle == 90 || angle == 270) {
    dim3 grid((unsigned)((h + 63) / 64), (unsigned)((w + 63) / 64));
    hipLaunchKernelGGL(k_rotate_quarter<T>, grid, dim3(256), 0, s, (const T*)in, is / sizeof(T), w, h, (T*)out, os / sizeof(T), angle);
  } else {
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)((h + 3) / 4));
    hipLaunchKernelGGL(k_remap<T>, grid, dim3(256), 0, s, (const T*)in, is / sizeof(T), (T*)out, os / sizeof(T), w, h, w - 1, -1, h - 1, -1);
  }
  return 0;
}

template <typename T>
int remap_plane(const void* in, size_t is, void* out, size_t os, int ow, int oh, int x0, int dx, int y0, int dy, hipStream_t s)
{
  dim3 grid((unsigned)((ow + 255) / 256), (unsigned)((oh + 3) / 4));
  hipLaunchKernelGGL(k_remap<T>, grid, dim3(256), 0, s, (const T*)in, is / sizeof(T), (T*)out, os / sizeof(T), ow, oh, x0, dx, y0, dy);
  return 0;
}

bool bad_plane(const void* in, const void* out, int w, int h, int bps, size_t is, size_t os, size_t out_row_bytes)
{
  return !in || !out || w <= 0 || h <= 0 || (bps != 1 && bps != 2) || is < (size_t)w * (size_t)bps || os < out_row_bytes || (bps == 2 && ((is | os) & 1));
}

}  // namespace
}  // namespace hipdec

using namespace hipdec;

extern "C" {

// ComponentStorage::rotate_ccw<T> (pixelimage.cc:1305-1333).  The output plane is h x w for 90 / 270 degrees.
int hipdec_plane_rotate_ccw(const void* in, size_t is, int w, int h, int bytes_per_sample, int angle, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (angle != 90 && angle != 180 && angle != 270) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "plane_rotate_ccw: angle must be 90, 180 or 270");
  const int ow = angle == 180 ? w : h;
  if (bad_plane(in, out, w, h, bytes_per_sample, is, os, (size_t)(ow > 0 ? ow : 0) * (size_t)bytes_per_sample)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "plane_rotate_ccw: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  if (bytes_per_sample == 1) rotate_plane<uint8_t>(in, is, w, h, angle, out, os, s);
  else rotate_plane<uint16_t>(in, is, w, h, angle, out, os, s);
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

// ComponentStorage::mirror_inplace<T> (pixelimage.cc:1337-1355), out of place; direction = heif_transform_mirror_direction (0 vertical: rows
// swapped, 1 horizontal: columns swapped)
int hipdec_plane_mirror(const void* in, size_t is, int w, int h, int bytes_per_sample, int direction, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (direction != 0 && direction != 1) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "plane_mirror: direction must be 0 (vertical) or 1 (horizontal)");
  if (bad_plane(in, out, w, h, bytes_per_sample, is, os, (size_t)(w > 0 ? w : 0) * (size_t)bytes_per_sample)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "plane_mirror: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  const int x0 = direction == 1 ? w - 1 : 0, dx = direction == 1 ? -1 : 1, y0 = direction == 0 ? h - 1 : 0, dy = direction == 0 ? -1 : 1;
  if (bytes_per_sample == 1) remap_plane<uint8_t>(in, is, out, os, w, h, x0, dx, y0, dy, s);
  else remap_plane<uint16_t>(in, is, out, os, w, h, x0, dx, y0, dy, s);
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

// the per-plane copy of HeifPixelImage::crop (pixelimage.cc:1433-1530): out_w x out_h samples from (left, top) of a w x h plane
int hipdec_plane_crop(const void* in, size_t is, int w, int h, int bytes_per_sample, int left, int top, int out_w, int out_h, void* out, size_t os, void* stream)
{
  if (int rc = ensure_init()) return rc;
  if (left < 0 || top < 0 || out_w <= 0 || out_h <= 0 || (long long)left + out_w > w || (long long)top + out_h > h)
    return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "plane_crop: region outside the plane");
  if (bad_plane(in, out, w, h, bytes_per_sample, is, os, (size_t)out_w * (size_t)bytes_per_sample)) return set_error(HIPDEC_ERR_INVALID_ARGUMENT, "plane_crop: bad arguments");
  hipStream_t s = stream ? (hipStream_t)stream : default_stream();
  if (bytes_per_sample == 1) remap_plane<uint8_t>(in, is, out, os, out_w, out_h, left, 1, top, 1, s);
  else remap_plane<uint16_t>(in, is, out, os, out_w, out_h, left, 1, top, 1, s);
  HIPDEC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
