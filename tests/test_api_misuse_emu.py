"""Hostile callers of the C ABI (include/heif_hipdec.h), on the whole library compiled for the host (tests/emu/libheifhip_emu.so) in a child process:
every one of the header's entry points with zeros and NULLs, then valid objects with bad indices, NULL outputs, too-small strides and calls in the wrong
order.  Nothing may crash; every int-returning entry point that was handed a NULL where it needs an object reports an error; what the reference's plugin
reports for the same situation (decoder_libde265.cc:331-346 End_of_data, :183-199 the pixel limit) keeps its code."""
import os
import subprocess
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU = os.path.join(HERE, "emu", "libheifhip_emu.so")

# entry points for which zeros / NULLs are valid arguments (setters, queries, frees of nothing, zero-byte copies)
HARMLESS = {"hipdec_init", "hipdec_device_count", "hipdec_memcpy_h2d", "hipdec_memcpy_d2h", "hipdec_memset", "hipdec_stream_synchronize", "hipdec_set_arena_cache_bytes",
            "hipdec_set_stage_overlap", "hipdec_set_reserved_wave_slots", "hipdec_batch_count", "hipdec_batch_item_packed_bytes", "hipdec_rccl_available"}


@pytest.fixture(scope="module")
def transcript():
    if not os.path.exists(EMU):
        r = subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu"), "libheifhip_emu.so"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, HIPDEC_LIBRARY=EMU, HIPDEC_DEV_AB="1", HIPEMU_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(HERE, "api_misuse_child.py")], capture_output=True, text=True, timeout=600, env=env)
    lines = r.stdout.splitlines()
    assert r.returncode == 0 and lines and lines[-1] == "DONE", "the child died (signal %d) at: %s\n%s" % (-r.returncode, lines[-1] if lines else "?", r.stderr[-2000:])
    return lines


def test_every_entry_point_survives_zeros_and_nulls(transcript):
    zero = [l.split() for l in transcript if l.startswith("ZERO ")]
    n = int([l for l in transcript if l.startswith("PROTOTYPES")][0].split()[1])
    assert len(zero) == n and n >= 100
    for _, name, ret, pointers, value in zero:
        if ret == "int" and int(pointers) > 0 and name not in HARMLESS:
            assert int(value) < 0, "%s accepted NULL pointers (returned %s)" % (name, value)


def test_bad_indices_null_outputs_small_strides_and_wrong_order_are_errors(transcript):
    bad = [l for l in transcript if l.startswith("UNEXPECTED")]
    assert not bad, "\n".join(bad)
    assert sum(l.startswith("CALL ") for l in transcript) >= 80
