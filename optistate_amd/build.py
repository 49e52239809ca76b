"""Builds optistate_amd/lib/liboptistate_hip.so from csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU).

    python -m optistate_amd.build [--force]

The library carries a build id (`os_build_id()`): "<hash of every source / header + the compile flags>-<hash of `hipcc
--version`>".  build() recompiles what is stale BY CONTENT, never by mtime: every object has a key file next to it holding
the hash of its source, of the headers it includes (scanned transitively from the `#include "..."` lines), of its flags and
of the toolchain; an object is rebuilt when that key differs (a pushed / restored tree with arbitrary mtimes cannot link an old
object under a new build id).  _capi.load() recomputes the source half from the files on disk and refuses a library built
from other sources (the .so is git-ignored and travels to the GPU box as a file: nothing else ties it to the tree it sits in).
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liboptistate_hip.so")
SOURCES = ["capi.hip", "kf_kernels.hip", "kf_rows_kernel.hip", "kf_dense_rows.hip", "kf_step.hip", "gru_kernels.hip", "gru_bf16_kernels.hip", "gru_wide_kernel.hip", "fused_kernels.hip", "gru_train_kernels.hip", "vit_kernels.hip", "mpc_kernels.hip", "mpc_quad.hip"]
HEADERS = ["kf_device.hpp", "kf_dense_rows.hpp", "kf_args.hpp", "kf_rows_chain.inc", "gru_common.hpp", "gru_device.hpp", "launch.hpp", "mpc_common.hpp", os.path.join("..", "..", "include", "optistate_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed"]
# per-file additions (the reason is in the file's header)
# mpc_quad.hip: machine LICM hoists the materialisation of 64-bit literals (polynomial coefficients of expm1 / sincos, weights) out of the
# persistent kernels' outer loops and then SPILLS them -- a scratch reload where two v_mov would do (kf_mpc_rows_kernel at two wavefronts
# per SIMD: 416 -> 176 B of scratch, B = 8,192 4.4e7 -> 5.1e7 steps/s; the drain phase of mpc_solve_quad_kernel: 28 -> 0 B)
EXTRA_FLAGS = {"kf_rows_kernel.hip": ["-fno-slp-vectorize"], "mpc_quad.hip": ["-mllvm", "-disable-machine-licm"]}


def _all_flags():
    return " ".join(FLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(EXTRA_FLAGS.items())])


def source_id():
    """Hash of every source and header (contents) and of the compile flags: the half of the build id that the loader can
    recompute without a toolchain."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(_all_flags().encode())
    return h.hexdigest()[:16]


def toolchain_id():
    try:
        v = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    except OSError:
        v = "no hipcc"
    return hashlib.sha256(v.encode()).hexdigest()[:8]


def _includes(path, seen=None):
    """Files reached from `path` through #include "..." (quoted includes only: the project's own headers)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path, "r", errors="replace") as fh:
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), re.M):
            _includes(os.path.join(os.path.dirname(path), m.group(1)), seen)
    return seen


def object_key(src, tool_id):
    """What an object file is a function of: its source, the headers it includes, its flags, the toolchain."""
    h = hashlib.sha256()
    for f in sorted(_includes(os.path.join(CSRC, src))):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(src, [])).encode())
    h.update(tool_id.encode())
    return h.hexdigest()[:24]


def _key_of(obj):
    try:
        return open(obj + ".key").read().strip()
    except OSError:
        return ""


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    bid = source_id() + "-" + toolchain_id()
    stamp = os.path.join(objdir, "build_id.txt")
    old = open(stamp).read().strip() if os.path.exists(stamp) else ""
    tool = bid.split("-")[-1]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        key = object_key(src, tool)
        # capi.hip carries the id string: recompiled whenever the id moves (seconds)
        if force or not os.path.exists(o) or _key_of(o) != key or (src == "capi.hip" and old != bid):
            jobs.append((s, o, key))

    def cc(job):
        s, o, key = job
        if os.path.exists(o + ".key"):
            os.remove(o + ".key")
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + [f'-DOS_BUILD_ID="{bid}"', "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        with open(o + ".key", "w") as fh:
            fh.write(key)
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    link_key = hashlib.sha256("".join(_key_of(o) for o in objs).encode() + bid.encode()).hexdigest()[:24]
    if force or jobs or not os.path.exists(LIB) or _key_of(LIB) != link_key:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        with open(LIB + ".key", "w") as fh:
            fh.write(link_key)
    with open(stamp, "w") as fh:
        fh.write(bid)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
