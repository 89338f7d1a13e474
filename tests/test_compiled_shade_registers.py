"""The render kernels that read a row of rays ahead load it with inline assembly the compiler does not keep track of.
Until the statement that waits for those loads has run, nothing may read their destination registers; whether that holds
is decided by the compiler's register allocation, so it is checked on the compiler's assembly output (about a minute)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc to compile the kernels to assembly")
def test_no_read_of_a_row_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_row_regs.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "4 kernels checked, 0 problems" in r.stdout
