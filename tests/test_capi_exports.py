"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/optistate_hip.h declares."""
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from optistate_amd import build, _capi
    build.build()
    return _capi.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "optistate_hip.h")).read()
    declared = set(re.findall(r"\b(os_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    from optistate_amd import _capi
    assert declared == set(_capi.EXPORTS)


def test_version_and_arch(lib):
    assert lib.os_version() >= 1
    assert lib.os_build_arch() == b"gfx950"


def test_param_count_matches_reference_configs(lib):
    import ctypes as C
    from optistate_amd._capi import OsGruDims
    # counted by instantiating the reference's RNN
    for dims, n in (((188, 128, 4, 24), 422424), ((188, 64, 4, 24), 125208), ((60, 64, 1, 24), 25752)):
        d = OsGruDims(*dims, 1)
        assert lib.os_gru_param_count(C.byref(d)) == n


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import optistate_amd
    with pytest.raises(RuntimeError):
        optistate_amd.Engine()
    m = optistate_amd.RNN(60, 64, 1, 24, torch.device("cpu"))
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 10, 60))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "optistate_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, fn
