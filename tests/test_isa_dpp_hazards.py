"""CPU: the gfx9 DPP read-after-write hazard in the translation units whose DPP instructions are inline assembly.

hipcc's hazard recogniser does not look inside inline assembly: a VALU write of a VGPR needs two wait states before a DPP
operand reads that register from another lane.  Round 5 met it for real: hipcc sank a multiply to zero wait states in front
of a `v_fmac_f64_dpp` of kf_dense_rows.hip and the reference-generated golden G8 failed by 9e-4.  This compiles the two
files to assembly (cross-compilation, no GPU) and scans every DPP instruction (tools/isa_dpp_hazard_scan.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CSRC = os.path.join(ROOT, "optistate_amd", "csrc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src, extra", [("kf_dense_rows.hip", []), ("kf_rows_kernel.hip", ["-fno-slp-vectorize"])])
def test_inline_asm_dpp_sources_are_two_wait_states_old(tmp_path, src, extra):
    out = tmp_path / (src + ".s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", '-DOS_BUILD_ID="scan"', "-S",
                        "--cuda-device-only", "-I" + CSRC, os.path.join(CSRC, src), "-o", str(out)] + extra,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    s = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_dpp_hazard_scan.py"), str(out)], capture_output=True, text=True)
    assert s.returncode == 0 and "0 hazards" in s.stdout, s.stdout[-3000:]
    assert int(s.stdout.strip().splitlines()[-1].split()[0]) > 100          # the scan saw the DPP instructions it is meant for
