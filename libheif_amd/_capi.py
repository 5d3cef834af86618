"""ctypes loader for libheifhip.so (C ABI: include/heif_hipdec.h).  Fails loudly when the HIP
extension is missing — nothing in this package computes on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class HipDecError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("hipdec error %d: %s" % (code, message))
        self.code = code
        self.message = message

    def __reduce__(self):   # (picklable: worker processes of the sweep tools hand exceptions back to their parent)
        return (HipDecError, (self.code, self.message))


class ImageInfo(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("chroma_format_idc", C.c_int),
                ("chroma_width", C.c_int), ("chroma_height", C.c_int),
                ("bit_depth_luma", C.c_int), ("bit_depth_chroma", C.c_int),
                ("colour_primaries", C.c_int), ("transfer_characteristics", C.c_int),
                ("matrix_coeffs", C.c_int), ("full_range_flag", C.c_int),
                ("coded_width", C.c_int), ("coded_height", C.c_int),
                ("bitstream_bytes", C.c_size_t), ("num_substreams", C.c_int)]


class Nclx(C.Structure):
    _fields_ = [("has_nclx", C.c_int), ("colour_primaries", C.c_int), ("transfer_characteristics", C.c_int),
                ("matrix_coefficients", C.c_int), ("full_range_flag", C.c_int)]


def library_path():
    """The in-tree library.  Development only: with HIPDEC_DEV_AB=1 (set by tools/ab_*.sh, never by the product) HIPDEC_LIBRARY names another
    build to measure beside this one; without that flag the variable is ignored with a warning - the native decoder takes untrusted input, and
    no environment may silently swap it.  bench.py records the resolved path in its JSON line."""
    override = os.environ.get("HIPDEC_LIBRARY")
    if override:
        if os.environ.get("HIPDEC_DEV_AB") == "1":
            import sys
            print("[libheif_amd] development override: loading %s" % override, file=sys.stderr)
            return override
        import warnings
        warnings.warn("HIPDEC_LIBRARY is set but HIPDEC_DEV_AB=1 is not: ignored, loading the in-tree libheifhip.so")
    return os.path.join(_HERE, "libheifhip.so")


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (soname
    libamdhip64.so.7, the same as /opt/rocm's) but link it by the un-versioned name, so if libheifhip.so
    pulls in the system runtime first a later `import torch` loads a SECOND runtime that sees no GPU and
    device pointers stop being shareable.  When torch is installed, bind to its copy up front.

    Never when a HIP runtime is already mapped (e.g. libheif dlopen()ed the plugin earlier in this process):
    a second runtime loaded RTLD_GLOBAL would interpose its un-initialised hsa_* symbols under the first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return
    except OSError:
        pass
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        p = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
            return


def load_library():
    """Returns the CDLL; raises if the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    _share_torch_hip_runtime()
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("libheifhip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C libheif_amd/csrc`); libheif_amd has no CPU fallback")
    lib = C.CDLL(path)
    vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
    lib.hipdec_last_error.restype = C.c_char_p
    lib.hipdec_version.restype = C.c_char_p
    lib.hipdec_malloc.restype = vp
    lib.hipdec_malloc.argtypes = [sz]
    lib.hipdec_free.argtypes = [vp]
    lib.hipdec_memcpy_h2d.argtypes = [vp, vp, sz]
    lib.hipdec_memcpy_d2h.argtypes = [vp, vp, sz]
    lib.hipdec_memset.argtypes = [vp, ci, sz]
    lib.hipdec_stream_synchronize.argtypes = [vp]
    lib.hipdec_stream_create.restype = vp
    lib.hipdec_stream_destroy.argtypes = [vp]
    np_ = C.POINTER(Nclx)
    lib.hipdec_color_420_to_rgb24.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, np_, vp, sz, ci, vp]
    lib.hipdec_color_ycbcr_to_rgb_planar.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, ci, np_, vp, vp, vp, sz, vp]
    lib.hipdec_color_ycbcr_to_rgb24_float.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, np_, vp, sz, ci, vp]
    lib.hipdec_color_420_to_rrggbb.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, np_, vp, sz, ci, vp]
    lib.hipdec_color_ycbcr_to_rrggbb_float.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, ci, np_, vp, sz, ci, vp]
    lib.hipdec_color_hdr_to_rgb24.argtypes = [vp, sz, vp, sz, vp, sz, ci, ci, ci, ci, np_, vp, sz, ci, ci, vp]
    lib.hipdec_color_mono_to_rgb24.argtypes = [vp, sz, vp, sz, ci, ci, vp, sz, ci, vp]
    lib.hipdec_color_bilinear_422_to_444.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    lib.hipdec_color_bilinear_420_to_444.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    lib.hipdec_color_to_sdr.argtypes = [vp, sz, ci, ci, ci, vp, sz, vp]
    lib.hipdec_color_coefficients.argtypes = [np_, C.POINTER(C.c_float)]
    lib.hipdec_color_coefficients.restype = None
    for name in ("hipdec_decoder_new", "hipdec_batch_create"):
        if not hasattr(lib, name):
            continue
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        raise HipDecError(rc, load_library().hipdec_last_error().decode("utf-8", "replace"))
    return rc


class DeviceBuffer:
    """A hipMalloc'd byte buffer owned through the C ABI."""

    def __init__(self, nbytes):
        lib = load_library()
        self.nbytes = int(nbytes)
        self.ptr = lib.hipdec_malloc(self.nbytes)
        if not self.ptr:
            raise HipDecError(-6, lib.hipdec_last_error().decode())

    @classmethod
    def from_numpy(cls, arr):
        import numpy as np
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        check(load_library().hipdec_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes))
        return b

    def to_numpy(self, shape, dtype):
        import numpy as np
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(load_library().hipdec_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            load_library().hipdec_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
