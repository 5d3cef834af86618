"""Host-side mirror of the reference's decoder-plugin interface for the HEVC path.

`HipDecoder` follows the call sequence libheif drives through heif_decoder_plugin
(libheif/codecs/decoder.cc:355-563): new_decoder -> push_data (xN) -> decode_next_image -> free,
with the same framing contract as libheif/plugins/decoder_libde265.cc:322-368 and the same error
behaviour (truncated framing -> End_of_data, nothing pushed -> no image).  `Batch` is the grid /
throughput entry point (libheif/image-items/grid.cc:405-453 on the device).  Both are thin ctypes
wrappers over the C ABI in include/heif_hipdec.h — no pixel is computed in Python.
"""
import ctypes as C
import numpy as np
from ._capi import HipDecError, ImageInfo, check, load_library, DeviceBuffer


def _bind(lib):
    if getattr(lib, "_dec_bound", False):
        return lib
    vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
    lib.hipdec_decoder_new.argtypes = [C.POINTER(vp), ci, C.c_uint64]
    lib.hipdec_decoder_free.argtypes = [vp]
    lib.hipdec_decoder_push_data.argtypes = [vp, C.c_char_p, sz]
    lib.hipdec_decoder_decode.argtypes = [vp, C.POINTER(ImageInfo)]
    lib.hipdec_decoder_read_plane.argtypes = [vp, ci, vp, sz]
    lib.hipdec_decoder_coalesce_stats.restype = None
    lib.hipdec_decoder_coalesce_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
    lib.hipdec_decoder_chain_stats.restype = None
    lib.hipdec_decoder_chain_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
    lib.hipdec_batch_create.argtypes = [C.POINTER(vp), ci, C.POINTER(C.c_char_p), C.POINTER(sz), C.c_uint64]
    lib.hipdec_batch_create_recycling.argtypes = [C.POINTER(vp), ci, C.POINTER(C.c_char_p), C.POINTER(sz), C.c_uint64, vp]
    lib.hipdec_batch_free.argtypes = [vp]
    lib.hipdec_batch_count.argtypes = [vp]
    lib.hipdec_batch_info.argtypes = [vp, ci, C.POINTER(ImageInfo)]
    lib.hipdec_batch_run.argtypes = [vp, vp]
    lib.hipdec_batch_status.argtypes = [vp]
    lib.hipdec_batch_read_plane.argtypes = [vp, ci, ci, vp, sz]
    lib.hipdec_batch_device_plane.argtypes = [vp, ci, ci, C.POINTER(vp), C.POINTER(sz)]
    lib.hipdec_batch_to_rgb.argtypes = [vp, ci, ci, vp, sz, vp]
    lib.hipdec_batch_to_rgb_all.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(sz), vp]
    lib.hipdec_batch_run_rgb.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(sz), vp]
    lib.hipdec_batch_last_timing_us.argtypes = [vp, C.POINTER(C.c_float)]
    lib.hipdec_batch_item_packed_bytes.restype = sz
    lib.hipdec_batch_item_packed_bytes.argtypes = [vp, ci]
    lib.hipdec_batch_pack_item.argtypes = [vp, ci, vp, sz, vp]
    lib.hipdec_copy2d_d2d.argtypes = [vp, sz, vp, sz, sz, sz, vp]
    lib.hipdec_batch_timing_slots.argtypes = [vp, ci]
    lib.hipdec_batch_slot_timing_us.argtypes = [vp, ci, C.POINTER(C.c_float)]
    lib.hipdec_batch_slot_kernel_timing_us.argtypes = [vp, ci, C.POINTER(C.c_float)]
    lib.hipdec_batch_read_tap.argtypes = [vp, ci, ci, ci, vp, sz]
    lib.hipdec_batch_read_maps.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, sz]
    lib._dec_bound = True
    return lib


def _info_dict(info):
    return {f: getattr(info, f) for f, _ in ImageInfo._fields_}


def coalesce_stats():
    """(decode requests, launch sets, requests that shared a launch set) since the library was loaded"""
    lib = _bind(load_library())
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib.hipdec_decoder_coalesce_stats(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def chain_stats():
    """(look-ahead chains of sequence tracks, launch sets issued for them, launch sets that held several tracks' chains) since the library was loaded"""
    lib = _bind(load_library())
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib.hipdec_decoder_chain_stats(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def set_sequence_pipeline(chains):
    """look-ahead chains of one track in flight at a time (hipdec_set_sequence_pipeline; 1: every chain is waited for where it is launched)"""
    lib = load_library()
    lib.hipdec_set_sequence_pipeline.argtypes = [C.c_int]
    lib.hipdec_set_sequence_pipeline.restype = None
    lib.hipdec_set_sequence_pipeline(int(chains))


def pipeline_stats():
    """(chains that were left in flight when they were enqueued, roll-backs of failed ones) since the library was loaded"""
    lib = load_library()
    lib.hipdec_decoder_pipeline_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.hipdec_decoder_pipeline_stats.restype = None
    a, b = C.c_uint64(), C.c_uint64()
    lib.hipdec_decoder_pipeline_stats(C.byref(a), C.byref(b))
    return a.value, b.value


class DecodedImage:
    def __init__(self, info, planes):
        self.info = info
        self.planes = planes
        self.nclx = (info["colour_primaries"], info["transfer_characteristics"], info["matrix_coeffs"], info["full_range_flag"])


class HipDecoder:
    def __init__(self, strict_decoding=False, max_image_size_pixels=0):
        self._lib = _bind(load_library())
        self._h = C.c_void_p()
        check(self._lib.hipdec_decoder_new(C.byref(self._h), int(strict_decoding), int(max_image_size_pixels)))

    def push_data(self, data: bytes):
        check(self._lib.hipdec_decoder_push_data(self._h, data, len(data)))

    def flush_data(self):
        return None

    def decode_next_image(self):
        """Returns a DecodedImage, or None when there is nothing (more) to deliver."""
        info = ImageInfo()
        rc = self._lib.hipdec_decoder_decode(self._h, C.byref(info))
        if rc == -7:  # HIPDEC_ERR_NO_IMAGE
            return None
        check(rc)
        d = _info_dict(info)
        dt = np.uint16 if d["bit_depth_luma"] > 8 else np.uint8
        planes = []
        for c in range(3 if d["chroma_format_idc"] else 1):
            w, h = (d["width"], d["height"]) if c == 0 else (d["chroma_width"], d["chroma_height"])
            a = np.empty((h, w), dt)
            check(self._lib.hipdec_decoder_read_plane(self._h, c, a.ctypes.data, w * a.itemsize))
            planes.append(a)
        return DecodedImage(d, planes)

    def _read_planes(self, info):
        d = _info_dict(info)
        dt = np.uint16 if d["bit_depth_luma"] > 8 else np.uint8
        planes = []
        for c in range(3 if d["chroma_format_idc"] else 1):
            w, h = (d["width"], d["height"]) if c == 0 else (d["chroma_width"], d["chroma_height"])
            a = np.empty((h, w), dt)
            check(self._lib.hipdec_decoder_read_plane(self._h, c, a.ctypes.data, w * a.itemsize))
            planes.append(a)
        return DecodedImage(d, planes)

    def next_picture(self, flush=False, user_data=None):
        """decode_next_image2 with output order (hipdec_decoder_next_picture): decodes the pushed sample if one is pending and returns
        (DecodedImage, user_data) of the next picture in OUTPUT order, or None while the bumping process holds it back (B pictures)."""
        if user_data is not None:
            self._lib.hipdec_decoder_set_user_data.argtypes = [C.c_void_p, C.c_size_t]
            self._lib.hipdec_decoder_set_user_data.restype = None
            self._lib.hipdec_decoder_set_user_data(self._h, int(user_data))
        self._lib.hipdec_decoder_next_picture.argtypes = [C.c_void_p, C.c_int, C.POINTER(ImageInfo), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
        info, have, ud = ImageInfo(), C.c_int(0), C.c_size_t(0)
        check(self._lib.hipdec_decoder_next_picture(self._h, 1 if flush else 0, C.byref(info), C.byref(have), C.byref(ud)))
        if not have.value:
            return None
        return self._read_planes(info), ud.value

    def free(self):
        if self._h:
            self._lib.hipdec_decoder_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Batch:
    """Many independent coded items (grid tiles, batches of stills) decoded by one set of launches."""

    def __init__(self, streams, max_image_size_pixels=0, recycle=None):
        """recycle: a batch of the same shape whose planes have been consumed; its arena is taken over (a stream of batches)"""
        self._lib = _bind(load_library())
        self._keep = [bytes(s) for s in streams]
        n = len(self._keep)
        arr = (C.c_char_p * n)(*self._keep)
        sizes = (C.c_size_t * n)(*[len(s) for s in self._keep])
        self._h = C.c_void_p()
        if recycle is None:
            check(self._lib.hipdec_batch_create(C.byref(self._h), n, arr, sizes, int(max_image_size_pixels)))
        else:
            check(self._lib.hipdec_batch_create_recycling(C.byref(self._h), n, arr, sizes, int(max_image_size_pixels), recycle._h))
        self.n = n

    def info(self, i):
        info = ImageInfo()
        check(self._lib.hipdec_batch_info(self._h, i, C.byref(info)))
        return _info_dict(info)

    def run(self, stream=None):
        check(self._lib.hipdec_batch_run(self._h, stream))

    def status(self):
        check(self._lib.hipdec_batch_status(self._h))

    def planes(self, i):
        d = self.info(i)
        dt = np.uint16 if d["bit_depth_luma"] > 8 else np.uint8
        out = []
        for c in range(3 if d["chroma_format_idc"] else 1):
            w, h = (d["width"], d["height"]) if c == 0 else (d["chroma_width"], d["chroma_height"])
            a = np.empty((h, w), dt)
            check(self._lib.hipdec_batch_read_plane(self._h, i, c, a.ctypes.data, w * a.itemsize))
            out.append(a)
        return out

    def tap(self, i, c):
        """deblocked (pre-SAO) picture at coded size — debug tap"""
        d = self.info(i)
        dt = np.uint16 if d["bit_depth_luma"] > 8 else np.uint8
        cf = d["chroma_format_idc"]
        subw, subh = (1, 1) if c == 0 else ((1 if cf == 3 else 2), (2 if cf == 1 else 1))
        w, h = d["coded_width"] // subw, d["coded_height"] // subh
        a = np.empty((h, w), dt)
        check(self._lib.hipdec_batch_read_tap(self._h, i, 1, c, a.ctypes.data, w * a.itemsize))
        return a

    def maps(self, i):
        d = self.info(i)
        uw, uh = (d["coded_width"] + 3) // 4, (d["coded_height"] + 3) // 4
        names = ["log2_tb", "log2_cb", "intra_luma", "intra_chroma", "qp_y", "flags"]
        arrs = [np.zeros((uh, uw), np.int8 if n == "qp_y" else np.uint8) for n in names]
        check(self._lib.hipdec_batch_read_maps(self._h, i, *[a.ctypes.data for a in arrs], uw * uh))
        return dict(zip(names, arrs))

    def to_rgb(self, i, out_chroma=10):
        """fused colour stage on the device planes of item i; returns the interleaved rows (uint8)."""
        d = self.info(i)
        bpp = {10: 3, 11: 4, 12: 6, 14: 6}[out_chroma]
        w, h = d["width"], d["height"]
        buf = DeviceBuffer(w * h * bpp)
        check(self._lib.hipdec_batch_to_rgb(self._h, i, out_chroma, buf.ptr, w * bpp, None))
        check(self._lib.hipdec_stream_synchronize(None))
        return buf.to_numpy((h, w * bpp), np.uint8)

    def alloc_rgb(self, out_chroma=10):
        """pre-allocates one interleaved output buffer per item for to_rgb_all()"""
        bpp = {10: 3, 11: 4, 12: 6, 14: 6}[out_chroma]
        self._rgb = []
        for i in range(self.n):
            d = self.info(i)
            self._rgb.append((DeviceBuffer(d["width"] * d["height"] * bpp), d["width"] * bpp, d["height"]))
        self._rgb_chroma = out_chroma
        self._rgb_ptrs = (C.c_void_p * self.n)(*[buf.ptr for buf, _, _ in self._rgb])
        self._rgb_strides = (C.c_size_t * self.n)(*[stride for _, stride, _ in self._rgb])

    def rgb_state(self):
        """the pre-allocated output buffers (device memory owned by Python objects), to hand to another batch of the same shape"""
        return (self._rgb, self._rgb_chroma, self._rgb_ptrs, self._rgb_strides)

    def use_rgb(self, state):
        self._rgb, self._rgb_chroma, self._rgb_ptrs, self._rgb_strides = state

    def to_rgb_all(self, stream=None):
        """asynchronous: fused colour stage over every item's planes into the pre-allocated buffers, ONE launch"""
        check(self._lib.hipdec_batch_to_rgb_all(self._h, self._rgb_chroma, self._rgb_ptrs, self._rgb_strides, stream))

    def run_rgb(self, stream=None):
        """asynchronous: decode + colour stage into the pre-allocated buffers as ONE call (for 8-bit 4:2:0 -> RGB24 the colour conversion is
        fused into the SAO kernel's store path)"""
        check(self._lib.hipdec_batch_run_rgb(self._h, self._rgb_chroma, self._rgb_ptrs, self._rgb_strides, stream))

    def rgb(self, i):
        buf, stride, h = self._rgb[i]
        check(self._lib.hipdec_stream_synchronize(None))
        return buf.to_numpy((h, stride), np.uint8)

    def timing_us(self):
        t = (C.c_float * 5)()
        check(self._lib.hipdec_batch_last_timing_us(self._h, t))
        return dict(parse=t[0], recon=t[1], deblock=t[2], sao=t[3], total=t[4])

    def timing_slots(self, n):
        check(self._lib.hipdec_batch_timing_slots(self._h, n))

    def slot_timing_us(self, slot):
        t = (C.c_float * 5)()
        check(self._lib.hipdec_batch_slot_timing_us(self._h, slot, t))
        return dict(parse=t[0], recon=t[1], deblock=t[2], sao=t[3], total=t[4])

    _KERNELS = ("parse", "residual", "recon", "deblock", "sao", "colour", "decode_total")

    def slot_kernel_timing_us(self, slot):
        """device time per kernel of the run recorded in `slot` (HIP events on the launch stream), microseconds"""
        t = (C.c_float * 8)()
        check(self._lib.hipdec_batch_slot_kernel_timing_us(self._h, slot, t))
        return {k: t[i] for i, k in enumerate(self._KERNELS)}

    def kernel_timing_us(self):
        """of the last run when the batch keeps one timing slot (the default)"""
        return self.slot_kernel_timing_us(0)

    def free(self):
        if self._h:
            self._lib.hipdec_batch_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
