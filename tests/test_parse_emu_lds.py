"""CPU check of the THROUGHPUT build's context layout: parse_core.h compiled for the host with HIPDEC_PARSE_LDS_CTX (context variables, rangeTabLps and
transIdxLps in the emulated LDS, scaled variables, the park / restore repacking) against the oracle, over a slice of test_parse_emu.py's matrix incl.
the work-pool scheduling with forced yields (contexts parked and restored through the LDS image).  The hand-scheduled gfx950 statements of that build
(parse_bins_lds_gfx950.h) run in the GPU tier; their C++ twins run here."""
import ctypes as C
import os
import pytest

import test_parse_emu as T
from oracle import pyoracle as orc


@pytest.fixture()
def lds_emu(monkeypatch):
    T.build_emu("libparse_emu_lds.so")
    L = C.CDLL(os.path.join(T.HERE, "emu", "libparse_emu_lds.so"))
    L.emu_create.restype = C.c_void_p
    L.emu_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.emu_free.argtypes = [C.c_void_p]
    L.emu_run_parse.argtypes = [C.c_void_p]
    L.emu_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.emu_maps.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
    L.emu_coeffs.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3
    L.emu_sao.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3
    monkeypatch.setattr(T, "_LIB", L)
    return L


LDS_CONFIGS = [dict(), dict(stress=1), dict(stress=1, wpp=0, log2_ctb=4, log2_max_tb=4), dict(tile_cols=3, tile_rows=2, wpp=1, loop_filter_across_tiles=0),
               dict(num_slices=4, wpp=0, stress=1), dict(transform_skip=1, stress=1), dict(lossless_pct=30), dict(bit_depth=10), dict(qp=4, stress=1), dict(qp=40),
               dict(pcm_pct=30, pcm_loop_filter_disabled=1, bit_depth=10, stress=1), dict(dependent_segments=3, wpp=1, stress=1),
               dict(dependent_segments=4, wpp=0, num_slices=2, stress=1, log2_ctb=4, log2_max_tb=4)]


@pytest.mark.parametrize("cfg", LDS_CONFIGS, ids=lambda c: ",".join("%s=%s" % kv for kv in c.items()) or "default")
def test_lds_context_build_matches_oracle(cfg, lds_emu):
    bd = cfg.get("bit_depth", 8)
    stream = orc.encode(orc.synth_image(200, 136, bd, 1, seed=203), **cfg)
    status, got = T.run_emu([stream])
    assert status == 0, "device status 0x%x" % status
    T.check_against_oracle(stream, got[0])


@pytest.mark.parametrize("yield_ctbs", [0, 1, 3])
def test_lds_context_build_pool_scheduler(yield_ctbs, lds_emu, monkeypatch):
    monkeypatch.setenv("HIPDEC_PARSE_POOL", "1")
    monkeypatch.setenv("HIPDEC_POOL_YIELD", str(yield_ctbs))
    streams = [orc.encode(orc.synth_image(w, h, 8, 1, seed=50 + i), **cfg) for i, (w, h, cfg) in
               enumerate([(264, 200, dict()), (200, 136, dict(stress=1, log2_ctb=4, log2_max_tb=4)), (328, 72, dict(tile_cols=3, tile_rows=2, wpp=1))])]
    status, got = T.run_emu(streams)
    assert status == 0, "device status 0x%x" % status
    for s, g in zip(streams, got):
        T.check_against_oracle(s, g)
