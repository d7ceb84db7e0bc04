"""The host half of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer: the pair
plan, the option store, the argument checks and the chunk arithmetic live in radio-mapper_amd/csrc/host_plan.hpp,
a HIP-free header that rmx_hip.hip includes; tests/host/test_host_plan.cpp exercises it, built here with g++
-fsanitize=address,undefined (the GPU box allows no sanitizer runs; this is the CPU build)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_plan_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "test_host_plan")
    src = os.path.join(ROOT, "tests", "host", "test_host_plan.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-pthread", "-Wall", "-Wextra", "-Werror", "-o", exe, src]
    subprocess.run(cmd, check=True, cwd=ROOT)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "all checks passed" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
