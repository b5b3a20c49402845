"""The conv kernel issues asynchronous loads from inline asm (direct-A / mode-3 K loops).  hipcc does not know those loads
are in flight; tools/audit_asm_loads.py proves on the compiled assembly that no compiler-generated instruction touches
a destination register between the load and the hand-counted wait that releases it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_compiler_access_to_in_flight_registers():
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "kernels with asm loads audited, 0 with violations" in r.stdout and " 0 kernels" not in r.stdout
