"""Guard-allocation mode: every buffer handed to the C ABI (dotaclient_amd.engine.DEVICE_ALLOC_HOOK) sits at the END
(or START) of its own exactly-sized HIP virtual-memory mapping, so an overrun of any kernel faults deterministically.
tools/guard_soak.py runs each workload in a subprocess that way (eager epochs and epochs replayed from a raw hipGraph capture) and first
proves the guard itself works (a deliberate one-element overrun must die of a memory access fault).  Second pass: every guarded buffer
starts as 0xFF bytes (NaN / 255 / -1), so a kernel that reads memory nothing has written turns the run NaN or faults."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_workload_is_clean_on_the_guard_allocator():
    env = dict(os.environ, DC_GUARD_ITERS='2')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'guard_soak.py')], capture_output=True, text=True, timeout=1500, env=env)
    sys.stdout.write(r.stdout)
    assert 'selftest(overrun must fault)     OK' in r.stdout, r.stdout + r.stderr[-2000:]
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]


def test_no_kernel_reads_uninitialised_memory():
    env = dict(os.environ, DC_GUARD_ITERS='2', DC_GUARD_FILL='1', DC_GUARD_MODES='end')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'guard_soak.py')], capture_output=True, text=True, timeout=1500, env=env)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
