"""CPU: the oracle's C sources (oracle/kf_oracle.c, gru_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C oracle asan` builds liboracle_asan.so from the same sources; a child Python with libasan preloaded loads it through
ORACLE_LIB and re-runs the golden-vector suite (G1-G8) plus the property tests, including the OpenMP-split batch entry
points.  Any heap / stack overflow, use-after-free or undefined operation aborts the child (exit code != 0).  GPU-side
sanitizers are not available on this pool; this is the sanitizer leg of the CPU-side C."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _gcc_file(name):
    r = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_goldens_under_asan_and_ubsan():
    asan, ubsan = _gcc_file("libasan.so"), _gcc_file("libubsan.so")
    if not asan:
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    lib = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    env = dict(os.environ, ORACLE_LIB=lib, LD_PRELOAD=":".join(p for p in (asan, ubsan) if p),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    code = ("import sys, pytest; sys.exit(pytest.main(['-x', '-q', '-p', 'no:cacheprovider', "
            "'tests/test_oracle_golden.py', 'tests/test_oracle_properties.py']))")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    assert " passed" in r.stdout
