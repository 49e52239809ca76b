"""CPU: two gfx9 hazards in the translation units whose DPP / VMEM instructions are inline assembly.

hipcc's hazard recogniser does not look inside inline assembly:
  * a VALU write of a VGPR needs two wait states before a DPP operand reads that register from another lane.  Round 5 met it for
    real: hipcc sank a multiply to zero wait states in front of a `v_fmac_f64_dpp` of kf_dense_rows.hip and the
    reference-generated golden G8 failed by 9e-4;
  * an SGPR written by a VALU instruction (v_readfirstlane, the v_readlane that reloads a spilled SGPR) needs five wait states
    before a VMEM instruction reads it: the AGPR stores of fused_kf_gru_kernel_v2<*, SEQOUT> went to random addresses (GRU l-inf
    1.5e-2) until their statement got its own `s_nop 4`.
This compiles the files to assembly (cross-compilation, no GPU) and scans them (tools/isa_dpp_hazard_scan.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CSRC = os.path.join(ROOT, "optistate_amd", "csrc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src, extra", [("kf_dense_rows.hip", []), ("kf_rows_kernel.hip", ["-fno-slp-vectorize"]), ("mpc_kernels.hip", []), ("mpc_quad.hip", []),
                                        ("fused_kernels.hip", [])])
def test_inline_asm_operands_are_past_their_hazard_windows(tmp_path, src, extra):
    out = tmp_path / (src + ".s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", '-DOS_BUILD_ID="scan"', "-S",
                        "--cuda-device-only", "-I" + CSRC, os.path.join(CSRC, src), "-o", str(out)] + extra,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    s = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_dpp_hazard_scan.py"), str(out)], capture_output=True, text=True)
    assert s.returncode == 0 and "0 hazards" in s.stdout, s.stdout[-3000:]
    last = s.stdout.strip().splitlines()[-1].split()
    assert int(last[0]) + int(last[4]) > 100          # the scan saw the instructions it is meant for
