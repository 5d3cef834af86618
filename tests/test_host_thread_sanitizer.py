"""ThreadSanitizer over the product's host orchestration (CPU tier, no GPU): tools/emu_tsan_host.sh compiles decoder.hip (both coalescers, look-ahead
chains, DPB holds, the resident-plane registry), runtime.hip (pools), hevc_headers.hip, batch_layout.hip, plugin.hip and grid_rccl.hip with
-fsanitize=thread, links them with the kernels under the SIMT emulator and with tests/emu/tsan_host.cc - application threads that decode the golden
stills, golden tracks and two grid photos (four emulated devices, one issue thread per device) side by side through the C ABI, some inputs damaged - and
every picture must equal the serial pass while the sanitizer reports nothing.  (What it found when it was first run is in profiles/r05_emulation_sweeps.txt:
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


def test_application_threads_on_the_whole_library_under_thread_sanitizer(tmp_path):
    if not _tsan_usable(tmp_path):
        pytest.skip("g++ -fsanitize=thread does not produce a runnable program here")
    env = dict(os.environ, HIPEMU_DEVICES="4", TSAN_HOST_BUILD=str(tmp_path / "build"))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "emu_tsan_host.sh"), "6", "3", "1", "15"], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in out, out[-6000:]
    assert r.returncode == 0, out[-3000:]
    assert " 0 MISMATCHES" in out and "serial pass:" in out, out[-3000:]
    assert "grid photos: 4 shards" in out, out[-3000:]
