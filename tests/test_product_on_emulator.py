"""The product's HOST ORCHESTRATION on the CPU: tests/emu/libheifhip_emu.so is the whole library - decoder.hip (launch sets, look-ahead chains, both
coalescers, DPB / bumping / RASL handling), runtime.hip (arena, pinned and stream pools), plugin.hip (the heif_decoder_plugin table), the grid and colour
entry points - compiled for the host against the SIMT emulator's synchronous runtime API (tests/emu/shim/hip/hip_runtime.h), with the kernels under the
emulator and the lane-emulated CABAC parser in place of the gfx950 assembly.  Loaded in place of libheifhip.so (HIPDEC_LIBRARY + HIPDEC_DEV_AB=1, the
development override of libheif_amd/_capi.py), it runs the GPU tier's own tests of the C ABI - unchanged - without a GPU: what the `-m gpu` tier checks
on the MI355X about the host code is checked here on every CPU run, including the multi-threaded paths (tracks decoded side by side sharing launch
sets, a corrupt track beside good ones, concurrent decoder instances) and the calls through the real libheif.

Only what depends on timing or on the hardware stays GPU-only: kernel performance, memory ordering between wavefronts (the emulator runs a launch to
completion before the next one starts), and the tests that need torch's CUDA runtime.  `bash tools/gpu_tier_on_emulator.sh` runs the WHOLE GPU tier this
way (457 of 459 tests in ~2 minutes on 16 cores); this module runs the host-orchestration subset inside the CPU tier."""
import os
import subprocess
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_LIB = os.path.join(HERE, "emu", "libheifhip_emu.so")
# the GPU-tier modules that are about host logic (small pictures: seconds under the emulator)
MODULES = ["test_sequence_gpu.py", "test_golden_sequences.py", "test_plugin_dropin.py", "test_resident_planes_gpu.py", "test_color_boundary.py",
           "test_image_ops_boundary.py", "test_tili_gpu.py"]


def _build():
    from test_parse_emu import build_emu
    build_emu("libheifhip_emu.so")
    assert os.path.exists(EMU_LIB)


def _run(args, timeout):
    env = dict(os.environ, HIPDEC_LIBRARY=EMU_LIB, HIPDEC_DEV_AB="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("PYTEST_XDIST_WORKER", None)
    try:
        import xdist  # noqa: F401
        par = ["-n", str(min(8, os.cpu_count() or 1))]
    except ImportError:
        par = []
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "--timeout", "900"] + par + args
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


def test_emulated_library_is_the_whole_c_abi():
    """every symbol include/heif_hipdec.h declares is exported by the host build too (it IS the product's sources), and a still decodes through it"""
    import re
    import ctypes as C
    _build()
    L = C.CDLL(EMU_LIB)
    header = open(os.path.join(ROOT, "include", "heif_hipdec.h")).read()
    names = sorted(set(re.findall(r"HIPDEC_API\s+[^;(]*?\b(hipdec_\w+)\s*\(", header)))
    assert len(names) > 60
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_gpu_tier_host_orchestration_on_the_emulated_library():
    _build()
    r = _run([os.path.join("tests", m) for m in MODULES], timeout=3000)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
