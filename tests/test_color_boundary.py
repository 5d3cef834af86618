"""The colour boundary (SURVEY.md §8b "second boundary", §8a row a8): libheif's ColorConversionPipeline with the HIP op registered in
init_ops() (libheif_amd/integration/colorconversion_hip.cc, built into oracle/_ref/libheif_hipcolor.so by oracle/Makefile.ref) against
the STOCK pipeline of the same reference (oracle/_ref/libheif.so).

CPU part: the C planner hipdec_color_plan() makes the decisions of ColorConversionPipeline::construct_pipeline
(colorconversion.cc:279-435) — checked against the chains the real planner builds, probed through the compiled reference — and
the patched library exists and exports the same API.  GPU part: heif_decode_image(..., RGB, interleaved) through the patched libheif
+ the decoder plugin runs the HIP colour kernels (counters of the boundary as proof; profiles/ holds the rocprofv3 kernel trace of
the same test) on the decoder's device-resident planes, bit-exact against the stock pipeline."""
import ctypes as C
import json
import os
import subprocess
import sys
import numpy as np
import pytest

import libheif_amd
from libheif_amd._capi import Nclx
from oracle import pyoracle as orc
import heic_util as hu
import libheif_host as lh

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = {1: "to_sdr", 2: "bilinear", 3: "420_to_rgb24", 4: "420_to_rgb32", 5: "ycbcr_to_rgb", 6: "rgb_to_rgb24_32", 7: "420_to_rrggbb", 8: "bilinear_422",
       9: "rgb_hdr_to_rrggbb_be", 10: "swap_endianness", 11: "mono_to_rgb24_32"}


def _plan(bpp, chroma, has_alpha, nclx, out_chroma, ups, only):
    lib = libheif_amd.load_library()
    lib.hipdec_color_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Nclx), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    ops, n = (C.c_int * 8)(), C.c_int()
    ns = Nclx(1, *[int(v) for v in nclx]) if nclx is not None else Nclx(0, 2, 2, 2, 1)
    rc = lib.hipdec_color_plan(bpp, chroma, int(has_alpha), C.byref(ns), out_chroma, ups, int(only), ops, C.byref(n))
    return rc, [OPS[ops[i]] for i in range(n.value)]


def test_c_planner_matches_the_python_mirror_and_the_reference_rules():
    """every in-scope state: same chain as libheif_amd/color.py:plan (round 1's mirror, itself checked against the compiled
    reference planner in tests/test_color_oracle.py)"""
    from libheif_amd import color
    names = {"Op_to_sdr_planes": "to_sdr", "Op_YCbCr420_bilinear_to_YCbCr444": "bilinear", "Op_YCbCr420_to_RGB24": "420_to_rgb24",
             "Op_YCbCr420_to_RGB32": "420_to_rgb32", "Op_YCbCr_to_RGB<u8>": "ycbcr_to_rgb", "Op_RGB_to_RGB24_32": "rgb_to_rgb24_32",
             "Op_YCbCr420_to_RRGGBBaa": "420_to_rrggbb", "Op_YCbCr422_bilinear_to_YCbCr444": "bilinear_422", "Op_YCbCr_to_RGB<u16>": "ycbcr_to_rgb",
             "Op_RGB_HDR_to_RRGGBBaa_BE": "rgb_hdr_to_rrggbb_be", "Op_RRGGBBaa_swap_endianness": "swap_endianness"}
    n = 0
    for chroma in (1, 2, 3):
      for bpp in (8, 10, 12):
        for nclx in (None, (1, 13, 6, 1), (1, 13, 6, 0), (9, 16, 9, 0), (9, 16, 9, 1), (1, 1, 1, 0), (2, 2, 2, 1), (1, 13, 0, 1), (1, 13, 8, 1), (1, 13, 11, 1)):
            for out in (10, 11, 12, 14):
                for ups, only in ((1, False), (2, False), (2, True), (1, True)):
                    try:
                        want = [names[x] for x in color.plan(bpp, chroma, nclx, out, ups, only)]
                    except libheif_amd.HipDecError:
                        want = None
                    rc, got = _plan(bpp, chroma, False, nclx, out, ups, only)
                    assert (rc == 0) == (want is not None), (chroma, bpp, nclx, out, ups, only, rc, want)
                    if want is not None:
                        assert got == want, (chroma, bpp, nclx, out, ups, only)
                        n += 1
    assert n > 400
    assert _plan(8, 0, False, None, 10, 2, False) == (0, ["mono_to_rgb24_32"]) and _plan(8, 0, True, None, 11, 2, False) == (0, ["mono_to_rgb24_32"])
    assert _plan(8, 0, True, None, 10, 2, False)[0] != 0 and _plan(10, 0, False, None, 10, 2, False)[0] != 0      # dropping alpha / > 8 bit: stock ops
    assert _plan(8, 1, True, (1, 13, 6, 1), 11, 1, False) == (0, ["420_to_rgb32"])       # alpha plane travels with the integer op
    assert _plan(8, 1, True, (1, 13, 6, 0), 11, 1, False) == (0, ["ycbcr_to_rgb", "rgb_to_rgb24_32"])
    assert _plan(8, 1, True, (1, 13, 6, 1), 10, 1, False)[0] != 0                          # dropping alpha: stock ops


def test_patched_reference_is_built_with_the_op():
    if not os.path.isdir("/root/reference") and not lh.available("libheif_hipcolor.so"):
        pytest.fail("oracle/_ref/libheif_hipcolor.so missing: build() must run where /root/reference exists")
    if not lh.available("libheif_hipcolor.so"):
        pytest.skip("oracle/_ref not built")
    so = os.path.join(HERE, "..", "oracle", "_ref", "libheif_hipcolor.so")
    syms = subprocess.run(["nm", "-DC", "--defined-only", so], capture_output=True, text=True).stdout
    assert "Op_YCbCr_to_RGB_hip::convert_colorspace" in syms and "heif_decode_image" in syms
    stock = subprocess.run(["nm", "-DC", "--defined-only", os.path.join(HERE, "..", "oracle", "_ref", "libheif.so")], capture_output=True, text=True).stdout
    assert "Op_YCbCr_to_RGB_hip" not in stock


SRGB = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)


def _run_child(libname, jobs, tmp_path):
    jf, of = str(tmp_path / ("jobs_%s.json" % libname)), str(tmp_path / ("out_%s.npz" % libname))
    json.dump(jobs, open(jf, "w"))
    env = dict(os.environ, HIPDEC_TEST_LIBHEIF=libname, PYTHONPATH=os.pathsep.join([os.path.join(HERE, ".."), HERE]))
    subprocess.run([sys.executable, os.path.join(HERE, "colorboundary_child.py"), jf, of], check=True, env=env, timeout=600)
    return np.load(of)


@pytest.mark.gpu
def test_heif_decode_image_to_rgb_runs_the_hip_colour_op_bit_exact(tmp_path):
    if not (lh.available("libheif.so") and lh.available("libheif_hipcolor.so")):
        pytest.fail("oracle/_ref libraries missing on the GPU box")
    cases = []

    def add(name, heic, chroma, threads=None):
        path = str(tmp_path / (name + ".heic"))
        open(path, "wb").write(heic)
        cases.append(dict(name=name, heic=path, colorspace=lh.COLORSPACE_RGB, chroma=chroma, threads=threads))

    s = orc.encode(orc.synth_image(456, 264, 8, 1, seed=21), **SRGB)
    add("srgb_rgb", hu.build_heic([(s, 456, 264)]), lh.CHROMA_RGB)                         # integer op, planes device-resident
    add("srgb_rgba", hu.build_heic([(s, 456, 264)]), lh.CHROMA_RGBA)
    s = orc.encode(orc.synth_image(322, 200, 8, 1, seed=22), vui_primaries=1, vui_transfer=1, vui_matrix=1, vui_full_range=0)
    add("bt709_limited_rgb", hu.build_heic([(s, 322, 200)]), lh.CHROMA_RGB)                # float chain
    s = orc.encode(orc.synth_image(200, 136, 8, 1, seed=23))
    add("no_vui_rgb", hu.build_heic([(s, 200, 136)]), lh.CHROMA_RGB)                       # unspecified nclx
    s10 = orc.encode(orc.synth_image(264, 200, 10, 1, seed=24), bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16)
    add("main10_rrggbb_le", hu.build_heic([(s10, 264, 200)], bit_depth=10), lh.CHROMA_RRGGBB_LE)
    add("main10_rrggbb_be", hu.build_heic([(s10, 264, 200)], bit_depth=10), lh.CHROMA_RRGGBB_BE)
    # Main10 to 8-bit RGB: limited range takes the generic op at 10 bits and Op_to_sdr_planes behind it, full range Op_to_sdr_planes + the integer op
    add("main10_limited_rgb", hu.build_heic([(s10, 264, 200)], bit_depth=10), lh.CHROMA_RGB)
    s10f = orc.encode(orc.synth_image(264, 200, 10, 1, seed=31), bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16, vui_full_range=1)
    add("main10_full_rgb", hu.build_heic([(s10f, 264, 200)], bit_depth=10), lh.CHROMA_RGB)
    # the chroma formats of round 3: 4:4:4 8-bit and the camera format 4:2:2 10-bit, to 8-bit RGB (Op_to_sdr first for the latter) and to RRGGBB
    s444 = orc.encode(orc.synth_image(264, 200, 8, 3, seed=27), **SRGB)
    add("444_rgb", hu.build_heic([(s444, 264, 200)], chroma_format_idc=3), lh.CHROMA_RGB)
    s422 = orc.encode(orc.synth_image(264, 200, 10, 2, seed=28), bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16)
    add("422_main10_rgb", hu.build_heic([(s422, 264, 200)], bit_depth=10, chroma_format_idc=2), lh.CHROMA_RGB)
    add("422_main10_rrggbb_le", hu.build_heic([(s422, 264, 200)], bit_depth=10, chroma_format_idc=2), lh.CHROMA_RRGGBB_LE)
    s444_10 = orc.encode(orc.synth_image(200, 136, 10, 3, seed=29), bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16)
    add("444_main10_rrggbb_be", hu.build_heic([(s444_10, 200, 136)], bit_depth=10, chroma_format_idc=3), lh.CHROMA_RRGGBB_BE)
    tiles = [(orc.encode(orc.synth_image(128, 128, 8, 1, seed=30 + i), **SRGB), 128, 128) for i in range(6)]
    add("grid_rgb", hu.build_heic(tiles, grid=(2, 3, 380, 250)), lh.CHROMA_RGB, threads=6)  # canvas assembled by libheif: upload path
    master = orc.encode(orc.synth_image(200, 136, 8, 1, seed=25), **SRGB)
    alpha = orc.encode(orc.synth_image(200, 136, 8, 0, seed=26))
    add("alpha_rgba", hu.build_heic([(master, 200, 136, 1), (alpha, 200, 136, 0)], alpha_of={2: 1}), lh.CHROMA_RGBA)

    stock = _run_child("libheif.so", cases, tmp_path)
    hipc = _run_child("libheif_hipcolor.so", cases, tmp_path)
    for c in cases:
        n = c["name"]
        np.testing.assert_array_equal(hipc[n + ".rgb"], stock[n + ".rgb"], err_msg=n)
        conv, resident, launches = [int(v) for v in hipc[n + ".stats"]]
        assert tuple(int(v) for v in stock[n + ".stats"]) == (0, 0, 0), n                   # the stock build never reaches the boundary
        served = int(hipc[n + ".rgbres"][1])                                                # answered from the launch set's resident RGB: no kernel at all
        assert conv == 1 and (launches >= 1 or served == 1), (n, conv, resident, launches, served)
        if n not in ("grid_rgb",):
            assert resident >= 3, (n, resident)                                              # the decoder's own device planes were used
    # the alpha really is the auxiliary image's plane
    a_ref = orc.decode(alpha)["planes"][0]
    np.testing.assert_array_equal(hipc["alpha_rgba.rgb"].reshape(136, 200, 4)[:, :, 3], a_ref)


@pytest.mark.gpu
def test_repeated_rgb_requests_are_served_from_the_launch_sets_resident_rgb(tmp_path):
    """once the host has asked for interleaved RGB24, the decoder's launch sets emit it from the SAO kernel (k_sao_rgb) beside the
    planes and stage it to pinned host memory; the integration op's conversion then is a host copy - no colour kernel queued behind CABAC pools.  The
    pixels stay those of the stock build's ops, for the integer op (full range) and the float chain (limited range) alike; an image whose planes
    libheif edits before the conversion (mirror) must NOT take the shortcut."""
    if not (lh.available("libheif.so") and lh.available("libheif_hipcolor.so")):
        pytest.fail("oracle/_ref libraries missing on the GPU box")
    cases = []

    def add(name, heic, chroma=lh.CHROMA_RGB):
        path = str(tmp_path / (name + ".heic"))
        open(path, "wb").write(heic)
        cases.append(dict(name=name, heic=path, colorspace=lh.COLORSPACE_RGB, chroma=chroma))

    full = orc.encode(orc.synth_image(456, 264, 8, 1, seed=41), **SRGB)
    limited = orc.encode(orc.synth_image(322, 200, 8, 1, seed=42), vui_primaries=1, vui_transfer=1, vui_matrix=1, vui_full_range=0)
    for k in range(3):
        add("full_%d" % k, hu.build_heic([(full, 456, 264)]))
    for k in range(2):
        add("limited_%d" % k, hu.build_heic([(limited, 322, 200)]))
    add("mirrored", hu.build_heic([(full, 456, 264)], transforms=[("imir", 0)]))
    add("rgba_not_served", hu.build_heic([(full, 456, 264)]), lh.CHROMA_RGBA)
    stock = _run_child("libheif.so", cases, tmp_path)
    hipc = _run_child("libheif_hipcolor.so", cases, tmp_path)
    for c in cases:
        np.testing.assert_array_equal(hipc[c["name"] + ".rgb"], stock[c["name"] + ".rgb"], err_msg=c["name"])
    served = {c["name"]: int(hipc[c["name"] + ".rgbres"][1]) for c in cases}
    produced = {c["name"]: int(hipc[c["name"] + ".rgbres"][0]) for c in cases}
    launches = {c["name"]: int(hipc[c["name"] + ".stats"][2]) for c in cases}
    assert served["full_0"] == 0 and launches["full_0"] >= 1                      # nobody had asked yet: the colour kernel ran
    for n in ("full_1", "full_2", "limited_0", "limited_1"):
        assert produced[n] == 1 and served[n] == 1 and launches[n] == 0, (n, produced[n], served[n], launches[n])
    assert served["rgba_not_served"] == 0 and launches["rgba_not_served"] >= 1
    assert served["mirrored"] == 0                                                # the planes were edited in place (or transformed on the device): never the shortcut
