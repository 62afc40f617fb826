"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY section 5 asked for a sanitizer pass on the CPU
side; GPU sanitizers are unavailable on the pool).  oracle/Makefile's libdr3oracle_asan.so target is built and the oracle's own
known-answer tests, the golden-vector reproduction and the port-opt / SSE2-order tests (padded planes, intrinsic loads) run against it in a child process with the ASan runtime preloaded:
any out-of-bounds read of a window, pad or table in the restatement, or signed overflow / misaligned access, aborts that child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_kat_and_golden_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libdr3oracle_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=libasan, DR3O_LIB=os.path.join(ROOT, "oracle", "libdr3oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_golden.py"),
                        os.path.join(ROOT, "tests", "test_oracle_opt.py"), os.path.join(ROOT, "tests", "test_oracle_sse2.py"),
                        "-m", "not gpu", "--deselect", "tests/test_oracle_opt.py::test_opt_is_faster"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert " passed" in r.stdout
