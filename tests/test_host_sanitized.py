"""The GPU-free part of the library under AddressSanitizer + UndefinedBehaviorSanitizer ("build needs its
own"; GPU ASan is not available on the pool): npp_level.cpp, npp_reach.cpp and the host-only C entries are compiled by
nclone_amd.build_native.build_sanitized() with g++ -fsanitize=address,undefined into a test-only library, and
tests/host_sanitized_driver.py runs every host-only entry point on ~120 fixture levels and on 900 malformed maps (truncated blobs,
out-of-range tile ids, entity coordinates / types / counts that lie, NaNs, random bytes) in a child process with the ASan runtime
preloaded.  A sanitizer report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_and_ubsan():
    from nclone_amd import build_native

    lib = build_native.build_sanitized()
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan runtime next to g++")
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=23"      # CPython itself 'leaks' at exit
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_sanitized_driver.py"), lib], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    assert "sanitized host run" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
