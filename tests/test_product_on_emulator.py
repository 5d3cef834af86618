"""The product's HOST ORCHESTRATION on the CPU: tests/emu/libheifhip_emu.so is the whole library - decoder.hip (launch sets, look-ahead chains, both
coalescers, DPB / bumping / RASL handling), runtime.hip (arena, pinned and stream pools), plugin.hip (the heif_decoder_plugin table), the grid and colour
entry points - compiled for the host against the SIMT emulator's synchronous runtime API (tests/emu/shim/hip/hip_runtime.h), with the kernels under the
emulator and the lane-emulated CABAC parser in place of the gfx950 assembly.  Loaded in place of libheifhip.so (HIPDEC_LIBRARY + HIPDEC_DEV_AB=1, the
development override of libheif_amd/_capi.py), it runs the GPU tier's own tests of the C ABI - unchanged - without a GPU: what the `-m gpu` tier checks
on the MI355X about the host code is checked here on every CPU run, including the multi-threaded paths (tracks decoded side by side sharing launch
sets, a corrupt track beside good ones, concurrent decoder instances) and the calls through the real libheif.

Only what depends on timing or on the hardware stays GPU-only: kernel performance, memory ordering between wavefronts (the emulator runs a launch to
completion before the next one starts), and the tests that need torch's CUDA runtime.  `bash tools/gpu_tier_on_emulator.sh` runs the WHOLE GPU tier this
way (478 of 480 tests in ~2 minutes on 8 cores); this module runs the host-orchestration subset inside the CPU tier."""
import os
import subprocess
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_LIB = os.path.join(HERE, "emu", "libheifhip_emu.so")
# the GPU-tier modules that are about host logic (small pictures: seconds under the emulator)
MODULES = ["test_sequence_gpu.py", "test_sequence_pipeline_gpu.py", "test_golden_sequences.py", "test_plugin_dropin.py", "test_resident_planes_gpu.py", "test_color_boundary.py",
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


MULTI_DEVICE_SCRIPT = r"""
import ctypes as C, json, sys
sys.path.insert(0, "tests")
import numpy as np
import libheif_amd
import test_grid_sharding as T
from oracle import pyoracle as orc
lib = libheif_amd.load_library()
assert lib.hipdec_device_count() == 4
lib.hipdec_grid_transport.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
vui = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
out = []
for devices in (None, [0, 1, 2, 3], [3, 1], [0, 1, 2, 3, 0, 1]):
    g, canvas = T._c_grid_case(2, 3, 128, 128, 380, 250, devices, **vui)      # (asserts the canvas planes against the oracle's tiles)
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    assert lib.hipdec_grid_transport(g._h, C.byref(a), C.byref(b), C.byref(c)) == 0
    rgb = g.to_rgb(10)
    want = orc.color_420_to_rgb24(canvas[0][:250, :380], canvas[1][:125, :190], canvas[2][:125, :190], (1, 13, 6, 1)).reshape(250, -1)
    assert (rgb == want).all()
    g.decode(); g.wait()
    assert (g.planes()[0] == canvas[0][:250, :380]).all()
    g.free()
    out.append([a.value, b.value, c.value])
# a damaged tile that lives on another device than the root's: the decode reports the device error (no hang, no half-written success), and the next
# photo decodes on all four devices again
from libheif_amd import HipDecError
from libheif_amd.grid import GridDecoderC, GridLayout
streams = [orc.encode(orc.synth_image(128, 128, 8, 1, seed=70 + t)) for t in range(6)]
bad = bytearray(streams[2])
for k in range(200, 260):
    bad[k] ^= 0x55
try:
    g = GridDecoderC({t: (bytes(bad) if t == 2 else s) for t, s in enumerate(streams)}, GridLayout(2, 3, 128, 128, 380, 250), None)
    g.decode(); g.wait()
    raise SystemExit("the damaged tile decoded without an error")
except HipDecError as e:
    assert "device decode error" in str(e) or "bitstream" in str(e).lower(), str(e)
g = GridDecoderC({t: s for t, s in enumerate(streams)}, GridLayout(2, 3, 128, 128, 380, 250), None)
g.decode(); g.wait()
assert (g.planes()[0][:128, 256:380] == orc.decode(streams[2])["planes"][0][:, :124]).all()
g.free()
print("TRANSPORT " + json.dumps(out))
"""


def test_grid_shards_over_four_emulated_devices():
    """SURVEY 8e on the host side: HIPEMU_DEVICES=4 gives the emulated runtime four devices, so hipdec_grid_* really places its shards on different
    devices (tile t on shard t mod G), issues them from per-device host threads, pastes into the root canvas across devices and converts the canvas once.
    No MI355X node with more than one GPU was available to any round; this is the partition / paste / life-cycle logic executing with N > 1, not a
    measurement."""
    import json
    _build()
    env = dict(os.environ, HIPDEC_LIBRARY=EMU_LIB, HIPDEC_DEV_AB="1", HIPEMU_DEVICES="4", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", MULTI_DEVICE_SCRIPT], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("TRANSPORT ")][-1]
    transport = json.loads(line[len("TRANSPORT "):])
    # [local, peer, staged] shards: with all four devices (or the explicit list) one shard is the root's own, three reach the canvas by peer access
    assert transport[0] == [1, 3, 0] and transport[1] == [1, 3, 0], transport
    assert transport[2][0] + transport[2][1] + transport[2][2] == 2 and transport[3][0] + transport[3][1] + transport[3][2] == 6, transport


def test_grid_rccl_path_with_three_ranks_over_a_toy_transport(tmp_path):
    """hipdec_grid_*_rccl, the one-process-per-GPU form, with THREE ranks: three processes, each with the emulated library on its own emulated
    device, and tests/emu/libfake_rccl.so (files in a directory) where librccl would be.  Every rank decodes its tiles t mod 3, the grouped send / receive
    gathers the packed tiles on rank 0, which pastes and converts; a damaged tile on rank 1 makes EVERY rank's wait fail; the communicator survives.  The
    RCCL transport itself is not exercised (world size 1 on the GPU box, N ranks first in `bench.py --gpus N`)."""
    from test_parse_emu import build_emu
    _build()
    build_emu("libfake_rccl.so")
    env = dict(os.environ, HIPDEC_LIBRARY=EMU_LIB, HIPDEC_DEV_AB="1", HIPEMU_DEVICES="3", HIPDEC_RCCL_LIBRARY=os.path.join(HERE, "emu", "libfake_rccl.so"),
               TMPDIR=str(tmp_path), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    idfile = str(tmp_path / "unique_id")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "emu", "rccl_rank.py"), str(r), "3", idfile], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(3)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hangs (a collective nobody answers?)")
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK %d OK" % r) in o, "rank %d:\n%s" % (r, o[-3000:])
