"""ASan + UBSan over the host side of the library (tools/sanitize_host.py): the three host translation units and the C++
host mirror are rebuilt with -fsanitize=address,undefined and the CPU boundary tests run against that build.  Sanitizers
run on the CPU build only (no GPU ASan on this pool)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_side_is_clean_under_asan_and_ubsan():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize_host.py")], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
