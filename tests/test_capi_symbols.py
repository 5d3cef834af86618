"""CPU-side check that the C-ABI library loads and exports every symbol include/*.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import pytest
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            names += re.findall(r"HIPDEC_API[^;(]*?\b(hipdec_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import libheif_amd
    lib = libheif_amd.load_library()
    declared = _declared()
    assert len(declared) > 20
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, "C ABI symbols declared in include/ but not exported: %s" % missing


def test_plugin_info_symbol_exported():
    import libheif_amd
    lib = libheif_amd.load_library()
    assert ctypes.c_void_p.in_dll(lib, "plugin_info") is not None


def test_no_gpu_fails_loudly_not_silently():
    """Without a device the product must error out, never fall back to CPU code."""
    import libheif_amd
    lib = libheif_amd.load_library()
    if lib.hipdec_device_count() > 0:
        return
    assert lib.hipdec_init(-1) != 0
    assert b"no HIP device" in lib.hipdec_last_error()


def test_concurrent_batches_setting_validates_its_argument():
    """host-only setting (no GPU touched): how many large batches share the CABAC pool's wave slots"""
    import ctypes as C
    from libheif_amd._capi import library_path
    lib = C.CDLL(library_path())
    lib.hipdec_last_error.restype = C.c_char_p
    assert lib.hipdec_set_concurrent_batches(0) != 0
    assert b"concurrent" in lib.hipdec_last_error()
    assert lib.hipdec_set_concurrent_batches(65) != 0
    assert lib.hipdec_set_concurrent_batches(2) == 0
    assert lib.hipdec_set_concurrent_batches(1) == 0


def test_example_hosts_compile_against_the_public_header():
    """examples/decode_batch.c is plain C over include/heif_hipdec.h (the header must stay C-compatible) and links against the built library;
    without a GPU it stops at the first device call with the library's loud "no CPU fallback" error.  examples/decode_with_libheif.c is
    syntax-checked against the reference's public headers when they are around."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "libheif_amd")
    if not os.path.exists(os.path.join(lib_dir, "libheifhip.so")):
        pytest.skip("libheifhip.so not built")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "decode_batch")
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "decode_batch.c"),
                               "-L", lib_dir, "-lheifhip", "-Wl,-rpath," + lib_dir, "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "usage" in r.stderr
    api = "/root/reference/libheif/api"
    gen = os.path.join(root, "oracle", "_ref", "gen")
    if os.path.isdir(api) and os.path.isdir(gen):
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-fsyntax-only", "-I", api, "-I", gen, os.path.join(root, "examples", "decode_with_libheif.c")])
