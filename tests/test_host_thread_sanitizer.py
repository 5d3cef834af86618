"""ThreadSanitizer over the product's host orchestration (CPU tier, no GPU): tools/emu_tsan_host.sh compiles decoder.hip (both coalescers, look-ahead
chains, DPB holds, the resident-plane registry), runtime.hip (pools), hevc_headers.hip, batch_layout.hip, plugin.hip and grid_rccl.hip with
-fsanitize=thread, links them with the kernels under the SIMT emulator and with tests/emu/tsan_host.cc - application threads that decode the golden
stills, golden tracks and two grid photos (four emulated devices, one issue thread per device) side by side through the C ABI, some inputs damaged - and
every picture must equal the serial pass while the sanitizer reports nothing.  The threads come first (cold mode): the library's lazy initialisation
happens under concurrency, which is where both findings were - the coalescers' knobs and the init flag of runtime.hip.  (What it found when it was first run is in profiles/r05_emulation_sweeps.txt:
the unguarded first-use read of the coalescers' environment knobs.)"""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tsan_usable(tmp_path):
    src = tmp_path / "probe.cc"
    src.write_text("#include <thread>\nint main() { int x = 0; std::thread t([&] { x = 1; }); t.join(); return x - 1; }\n")
    exe = tmp_path / "probe"
    r = subprocess.run(["g++", "-fsanitize=thread", "-o", str(exe), str(src), "-lpthread"], capture_output=True)
    if r.returncode != 0:
        return False
    return subprocess.run([str(exe)], capture_output=True).returncode == 0   # (old libtsan + high-entropy ASLR: "unexpected memory mapping")


@pytest.fixture(scope="module")
def build_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("tsan")
    if not _tsan_usable(d):
        pytest.skip("g++ -fsanitize=thread does not produce a runnable program here")
    return str(d / "build")   # (both tests share the objects: tools/emu_tsan_objects.sh keeps what is newer than the sources)


def test_application_threads_on_the_whole_library_under_thread_sanitizer(build_dir):
    env = dict(os.environ, HIPEMU_DEVICES="4", TSAN_HOST_BUILD=build_dir, ALL="1")   # ALL=1: the host functions beside the kernels are watched too
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "emu_tsan_host.sh"), "8", "3", "1", "15", "1"], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in out, out[-6000:]
    assert r.returncode == 0, out[-3000:]
    assert " 0 MISMATCHES" in out and "serial pass:" in out, out[-3000:]
    assert "grid photos: 4 shards" in out, out[-3000:]


def _reports_about_the_plugin(out):
    """ThreadSanitizer report blocks that name the product's sources.  (The one report stock libheif earns by itself: its grid loop checks `inout_image`
    without the mutex before pasting a tile - image-items/grid.cc:534-560 - so a paste is not ordered behind the other thread's canvas allocation; no frame
    of the plugin is in it.)"""
    blocks = [b for b in out.split("==================") if "WARNING: ThreadSanitizer" in b]
    return [b for b in blocks if "libheif_amd/csrc" in b or "tests/emu" in b]


@pytest.mark.parametrize("rgb", [0, 1], ids=["planes_stock_libheif", "rgb_patched_libheif"])
def test_the_plugin_inside_the_real_libheif_under_thread_sanitizer(build_dir, rgb):
    """the same instrumented build as a plugin of oracle/_ref/libheif*.so: application threads x heif_decode_image() on small HEIC files (stills and a grid
    item) - function table, plane hand-over, coalescer; to RGB through the patched libheif: integration colour op, resident planes, grid hook"""
    lib = os.path.join(ROOT, "oracle", "_ref", "libheif_hipcolor.so" if rgb else "libheif.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref is not built")
    env = dict(os.environ, TSAN_HOST_BUILD=build_dir, RGB=str(rgb), DROPIN_WARMUP_S="1", ALL="1")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "emu_tsan_libheif.sh"), "6", "4"], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    ours = _reports_about_the_plugin(out)
    assert not ours, ours[0][-6000:]
    last = [l for l in r.stdout.strip().splitlines() if l and l[0].isdigit()]
    assert last, out[-3000:]
    decodes, _, _, requests, launch_sets, failed = last[-1].split()[:6]   # (+ mean / p95 ms per call since round 6)
    assert int(failed) == 0 and int(decodes) > 0 and int(requests) > 0, out[-3000:]


def test_random_walks_over_the_decoder_abi(build_dir):
    """tests/emu/api_fuzz.cc: legal calls in any order - poll before push, push after a flush, samples of other streams and random bytes between good ones,
    decode() between polls, plane reads at any time, instances dropped half way - from three threads; the library answers every one (a picture or an error
    code), does not crash, does not hang, and the sanitizer stays silent.  (The AddressSanitizer / LeakSanitizer campaigns of the same program:
    profiles/r05_emulation_sweeps.txt.)"""
    env = dict(os.environ, TSAN_HOST_BUILD=build_dir, SAN="thread", ALL="1", FUZZ_TIMEOUT="600")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "emu_api_fuzz.sh"), "7", "24", "50", "3"], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in out, out[-6000:]
    assert r.returncode == 0 and "no crash" in out, out[-3000:]
