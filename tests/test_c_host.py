"""The C-ABI from a plain-C host (examples/c_abi_demo.c: hipMalloc'd buffers, no Python, no torch in the process).
CPU: the header is valid C99 and the example compiles and links against the built library with -Wall -Wextra -Werror.
GPU: the program runs on seeded inputs; os_kf_run, os_fused_run and os_kf_step results against the float64 oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

LIBDIR = os.path.join(ROOT, "optistate_amd", "lib")


def _build(out):
    from optistate_amd import build
    build.build()
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + LIBDIR, "-loptistate_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_plain_c_host_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path / "c_abi_demo"))
    assert os.path.getsize(exe) > 0
    r = subprocess.run([exe], capture_output=True, text=True)            # no arguments: usage, before anything touches a GPU
    assert r.returncode == 1 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,L", [(96, 20, 1), (700, 12, 2)])
def test_plain_c_host_results_match_the_oracle(tmp_path, B, T, L):
    import torch
    from optistate_amd import RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    exe = _build(str(tmp_path / "c_abi_demo"))
    d = synth_numpy(B, T, seed=77)
    Q, R = Q_FITTED, R_FITTED
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R)
    rows = np.concatenate([ref["x"], d["accel"].astype(np.float64), d["f"].astype(np.float64), ref["p_rot"],
                           d["dp"].astype(np.float64), d["imu"].astype(np.float64)], axis=2)
    mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0)
    torch.manual_seed(5)
    m = RNN(60, 64, L, 24, torch.device("cpu"))
    ref_out, _, _ = orc.gru_forward((rows - mn) / (mx - mn), orc.flatten_state_dict(m.state_dict(), L), 60, 64, L, 24)
    w = flatten_state_dict(m.state_dict(), L).cpu().numpy().astype(np.float32)
    soa = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32).transpose(1, 2, 0))             # [B][T][F] -> [T][F][B]
    c = np.asarray(d["contact"], dtype=np.uint32)
    packed = np.ascontiguousarray((c[..., 0] | (c[..., 1] << 8) | (c[..., 2] << 16) | (c[..., 3] << 24)).T.astype(np.uint32))      # [T][B]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as fp:
        np.array([B, T, L, w.size], dtype=np.int32).tofile(fp)
        for k in ("p", "f", "dp", "imu"):
            soa(d[k]).tofile(fp)
        packed.tofile(fp)
        soa(d["accel"]).tofile(fp)
        np.ascontiguousarray(d["x0"].astype(np.float32).T).tofile(fp)
        np.tile(Q.astype(np.float32).reshape(144, 1), (1, B)).tofile(fp)
        Q.astype(np.float32).tofile(fp); R.astype(np.float32).tofile(fp)
        np.stack([mn, mx]).astype(np.float32).tofile(fp)
        w.tofile(fp)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "built for gfx950" in r.stdout
    with open(fout, "rb") as fp:
        x_kf = np.fromfile(fp, np.float32, T * 12 * B).reshape(T, 12, B).transpose(2, 0, 1)
        st_kf = np.fromfile(fp, np.int32, B)
        x_fu = np.fromfile(fp, np.float32, T * 12 * B).reshape(T, 12, B).transpose(2, 0, 1)
        out = np.fromfile(fp, np.float32, B * 24).reshape(B, 24)
        st_fu = np.fromfile(fp, np.int32, B)
        sx = np.fromfile(fp, np.float64, 12)
        sz = np.fromfile(fp, np.float64, 10)
    assert not (st_kf & 15).any() and not (st_fu & 15).any()
    assert np.abs(x_kf - ref["x"]).max() < 1e-4
    assert np.abs(x_fu - ref["x"]).max() < 1e-4
    assert np.abs(out - ref_out).max() < 1e-5
    # os_kf_step is float64 end to end; its inputs here are the float32 streams and the float32 Q / R of the file, widened
    Q32, R32 = Q.astype(np.float32).astype(np.float64), R.astype(np.float32).astype(np.float64)
    one = orc.kf_run_batch(d["p"][:1, :1], d["f"][:1, :1], d["dp"][:1, :1], d["imu"][:1, :1], d["contact"][:1, :1], d["x0"][:1], Q32[None], Q32, R32)
    assert np.abs(sx - one["x"][0, 0]).max() < 1e-10
    import ctypes as C
    Lb = orc.lib()
    od, z = np.zeros(4), np.zeros(10)
    c64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    dd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    p0, dp0, imu0 = c64(d["p"][0, 0]), c64(d["dp"][0, 0]), c64(d["imu"][0, 0])
    Lb.ok_get_odom(dd(p0), dd(dp0), np.ascontiguousarray(d["contact"][0, 0], dtype=np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)), dd(imu0), dd(od))
    Lb.ok_set_meas(dd(imu0), dd(od), dd(z))
    assert np.abs(sz - z).max() < 1e-12
