"""Test tooling: drives the REAL reference libheif (oracle/_ref/libheif.so, compiled from
/root/reference by oracle/Makefile.ref — the prebuilt .so travels to the GPU box) through its public C
API with ctypes, with libheif_amd/libheifhip.so loaded as a decoder plugin exactly the way an
application would (heif_load_plugin -> dlopen + dlsym("plugin_info"), libheif/plugins_unix.cc:103-118).
This is the drop-in proof: heif_decode_image() -> libheif -> heif_decoder_plugin -> HIP kernels."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "..", "oracle", "_ref")
_LIB = None
_PLUGIN_LOADED = False

COLORSPACE_YCBCR, COLORSPACE_RGB, COLORSPACE_MONO, COLORSPACE_UNDEFINED = 0, 1, 2, 99
CHROMA_UNDEFINED, CHROMA_MONO, CHROMA_420, CHROMA_RGB, CHROMA_RGBA = 99, 0, 1, 10, 11
CHROMA_RRGGBB_BE, CHROMA_RRGGBB_LE = 12, 14
CHANNEL_Y, CHANNEL_CB, CHANNEL_CR, CHANNEL_INTERLEAVED = 0, 1, 2, 10
COMPRESSION_HEVC = 1


class HeifError(C.Structure):
    _fields_ = [("code", C.c_int), ("subcode", C.c_int), ("message", C.c_char_p)]


class LibheifError(RuntimeError):
    def __init__(self, e):
        super().__init__("libheif error %d.%d: %s" % (e.code, e.subcode, (e.message or b"").decode("latin1")))
        self.code, self.subcode = e.code, e.subcode


# HIPDEC_TEST_LIBHEIF selects the build of the reference to drive: libheif.so (stock) or libheif_hipcolor.so (the same sources with the
# HIP colour op registered in init_ops(), oracle/Makefile.ref) — one per process, both export the same symbols
_NAME = os.environ.get("HIPDEC_TEST_LIBHEIF", "libheif.so")


def available(name=None):
    return os.path.exists(os.path.join(_REF, name or _NAME))


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(_REF, _NAME), mode=C.RTLD_GLOBAL)
        vp = C.c_void_p
        L.heif_load_plugin.restype = HeifError
        L.heif_load_plugin.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.heif_have_decoder_for_format.argtypes = [C.c_int]
        L.heif_context_alloc.restype = vp
        L.heif_context_free.argtypes = [vp]
        L.heif_context_read_from_memory_without_copy.restype = HeifError
        L.heif_context_read_from_memory_without_copy.argtypes = [vp, C.c_char_p, C.c_size_t, vp]
        L.heif_context_get_primary_image_handle.restype = HeifError
        L.heif_context_get_primary_image_handle.argtypes = [vp, C.POINTER(vp)]
        L.heif_context_set_max_decoding_threads.argtypes = [vp, C.c_int]
        L.heif_image_handle_get_width.argtypes = [vp]
        L.heif_image_handle_get_height.argtypes = [vp]
        L.heif_image_handle_release.argtypes = [vp]
        L.heif_decode_image.restype = HeifError
        L.heif_decode_image.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, vp]
        L.heif_image_get_plane_readonly2.restype = C.POINTER(C.c_uint8)
        L.heif_image_get_plane_readonly2.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
        L.heif_image_get_width.argtypes = [vp, C.c_int]
        L.heif_image_get_height.argtypes = [vp, C.c_int]
        L.heif_image_get_bits_per_pixel_range.argtypes = [vp, C.c_int]
        L.heif_image_release.argtypes = [vp]
        L.heif_decoding_options_alloc.restype = vp
        L.heif_decoding_options_free.argtypes = [vp]
        _LIB = L
    return _LIB


def check(e):
    if e.code != 0:
        raise LibheifError(e)


def load_hip_plugin():
    """heif_load_plugin(libheifhip.so): what LIBHEIF_PLUGIN_PATH discovery does for every file in the
    plugin directory."""
    global _PLUGIN_LOADED
    L = lib()
    if not _PLUGIN_LOADED:
        import libheif_amd
        from libheif_amd import _capi
        # this interpreter may `import torch` later (grid tests): keep the process on ONE HIP runtime whatever
        # the test order (see _capi._share_torch_hip_runtime); a C host without torch has nothing to do here
        _capi._share_torch_hip_runtime()
        info = C.c_void_p()
        check(L.heif_load_plugin(libheif_amd.library_path().encode(), C.byref(info)))
        _PLUGIN_LOADED = True
    return L


def _plane(L, img, channel, bytes_per_sample=1, interleave=1):
    stride = C.c_size_t()
    p = L.heif_image_get_plane_readonly2(img, channel, C.byref(stride))
    if not p:
        return None
    w, h = L.heif_image_get_width(img, channel), L.heif_image_get_height(img, channel)
    rowbytes = w * bytes_per_sample * interleave
    buf = np.ctypeslib.as_array(p, shape=(h * stride.value,))[:h * stride.value].reshape(h, stride.value)[:, :rowbytes].copy()
    return buf.view(np.uint16) if bytes_per_sample == 2 else buf


def open_heic(data):
    L = lib()
    ctx = L.heif_context_alloc()
    check(L.heif_context_read_from_memory_without_copy(ctx, data, len(data), None))
    h = C.c_void_p()
    check(L.heif_context_get_primary_image_handle(ctx, C.byref(h)))
    return ctx, h


def primary_size(data):
    L = lib()
    ctx, h = open_heic(data)
    try:
        return L.heif_image_handle_get_width(h), L.heif_image_handle_get_height(h)
    finally:
        L.heif_image_handle_release(h)
        L.heif_context_free(ctx)


def decode(data, colorspace=COLORSPACE_UNDEFINED, chroma=CHROMA_UNDEFINED, max_threads=None, ignore_transformations=False):
    """heif_decode_image() on the primary item.  Returns a dict: YCbCr -> planes [Y, Cb, Cr];
    interleaved RGB -> 'rgb' rows."""
    L = lib()
    ctx, h = open_heic(data)
    img = C.c_void_p()
    try:
        if max_threads is not None:
            L.heif_context_set_max_decoding_threads(ctx, max_threads)
        opts = None
        if ignore_transformations:   # heif_decoding_options: uint8 version, uint8 ignore_transformations, ... (api/libheif/heif_decoding.h:63-71)
            L.heif_decoding_options_alloc.restype = C.c_void_p
            L.heif_decoding_options_free.argtypes = [C.c_void_p]
            opts = C.c_void_p(L.heif_decoding_options_alloc())
            C.cast(opts, C.POINTER(C.c_uint8))[1] = 1
        try:
            check(L.heif_decode_image(h, C.byref(img), colorspace, chroma, opts))
        finally:
            if opts:
                L.heif_decoding_options_free(opts)
        out = {}
        if chroma in (CHROMA_RGB, CHROMA_RGBA):
            out["rgb"] = _plane(L, img, CHANNEL_INTERLEAVED, 1, 3 if chroma == CHROMA_RGB else 4)
        elif chroma in (CHROMA_RRGGBB_BE, CHROMA_RRGGBB_LE):
            out["rgb"] = _plane(L, img, CHANNEL_INTERLEAVED, 1, 6)
        else:
            bpp = L.heif_image_get_bits_per_pixel_range(img, CHANNEL_Y)
            bs = 2 if bpp > 8 else 1
            out["planes"] = [p for p in (_plane(L, img, c, bs) for c in (CHANNEL_Y, CHANNEL_CB, CHANNEL_CR)) if p is not None]
            out["bit_depth"] = bpp
        return out
    finally:
        if img:
            L.heif_image_release(img)
        L.heif_image_handle_release(h)
        L.heif_context_free(ctx)


ERROR_END_OF_SEQUENCE = 13


def decode_track(data, colorspace=COLORSPACE_UNDEFINED, chroma=CHROMA_UNDEFINED, track_id=0, max_images=None, options=None):
    """heif_track_decode_next_image() (api/libheif/heif_sequences.h:218) on the first visual track until End_of_sequence: what an application does
    with an image-sequence file.  libheif's Track_Visual::decode_next_image_sample (sequences/track_visual.cc:175-330) drives the decoder plugin:
    push_data2 per sample with the sample index as user_data, decode_next_image2 polls, flush_data at the end.  Returns the images in the
    order libheif delivers them: [{'planes': [Y, Cb, Cr]} | {'rgb': rows}].  options: a heif_decoding_options* (e.g. with decoder_id pinned,
    codecs/decoder.cc:327-340), or None for libheif's defaults."""
    L = lib()
    vp = C.c_void_p
    L.heif_context_get_track.restype = vp
    L.heif_context_get_track.argtypes = [vp, C.c_uint32]
    L.heif_track_release.argtypes = [vp]
    L.heif_track_decode_next_image.restype = HeifError
    L.heif_track_decode_next_image.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, vp]
    L.heif_context_has_sequence.argtypes = [vp]
    ctx = L.heif_context_alloc()
    out = []
    track = None
    try:
        check(L.heif_context_read_from_memory_without_copy(ctx, data, len(data), None))
        assert L.heif_context_has_sequence(ctx)
        track = vp(L.heif_context_get_track(ctx, track_id))
        assert track
        while max_images is None or len(out) < max_images:
            img = vp()
            e = L.heif_track_decode_next_image(track, C.byref(img), colorspace, chroma, options)
            if e.code == ERROR_END_OF_SEQUENCE:
                break
            check(e)
            try:
                if chroma in (CHROMA_RGB, CHROMA_RGBA):
                    out.append({"rgb": _plane(L, img, CHANNEL_INTERLEAVED, 1, 3 if chroma == CHROMA_RGB else 4)})
                else:
                    bpp = L.heif_image_get_bits_per_pixel_range(img, CHANNEL_Y)
                    bs = 2 if bpp > 8 else 1
                    out.append({"planes": [p for p in (_plane(L, img, c, bs) for c in (CHANNEL_Y, CHANNEL_CB, CHANNEL_CR)) if p is not None], "bit_depth": bpp})
            finally:
                L.heif_image_release(img)
        return out
    finally:
        if track:
            L.heif_track_release(track)
        L.heif_context_free(ctx)
