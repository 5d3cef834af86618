"""The SIMT emulator resumes the lanes of a workgroup in index order by default; between two convergence points (barrier, ballot, shuffle) the hardware
promises no order at all.  HIPEMU_LANE_ORDER=1 (reverse) / 2 (pseudo-random, changing with every pass) re-runs the emulated device pipeline with another
order: a kernel whose pixels depend on it - an LDS write another lane reads without a barrier in between - would decode differently.  (The knob is read once
per process, hence the child pytest; tools/emu_random_sweep*.py take it from the environment as well: profiles/r05_emulation_sweeps.txt.)"""
import os
import subprocess
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("order", [1, 2], ids=["reverse", "shuffled"])
def test_emulated_pipeline_does_not_depend_on_the_lane_order(order):
    env = dict(os.environ, HIPEMU_LANE_ORDER=str(order))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(HERE, "test_pipeline_emu.py"),
                        os.path.join(HERE, "test_golden_sequences.py"), "-m", "not gpu", "-k", "not oracle_reproduces"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]
