"""Builds optistate_amd/lib/liboptistate_hip.so from csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU).

    python -m optistate_amd.build [--force]

The library carries a build id (`os_build_id()`): "<hash of every source / header + the compile flags>-<hash of `hipcc
--version`>".  build() recompiles what is stale by content, not only by mtime: a changed flag or a ROCm upgrade rebuilds
everything; _capi.load() recomputes the source half from the files on disk and refuses a library built from other sources
(the .so is git-ignored and travels to the GPU box as a file: nothing else ties it to the tree it sits in).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liboptistate_hip.so")
SOURCES = ["capi.hip", "kf_kernels.hip", "kf_rows_kernel.hip", "kf_step.hip", "gru_kernels.hip", "fused_kernels.hip", "gru_train_kernels.hip", "vit_kernels.hip", "mpc_kernels.hip"]
HEADERS = ["kf_device.hpp", "kf_args.hpp", "kf_rows_chain.inc", "gru_common.hpp", "launch.hpp", os.path.join("..", "..", "include", "optistate_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed"]
# per-file additions (the reason is in the file's header)
EXTRA_FLAGS = {"kf_rows_kernel.hip": ["-fno-slp-vectorize"]}


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


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    bid = source_id() + "-" + toolchain_id()
    stamp = os.path.join(objdir, "build_id.txt")
    old = open(stamp).read().strip() if os.path.exists(stamp) else ""
    # another toolchain or other flags: every object is stale whatever its mtime says
    if old.split("-")[-1] != bid.split("-")[-1] or _flags_changed(objdir):
        force = True
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        # capi.hip carries the id string: recompiled whenever the id moves (seconds)
        if force or _stale(o, [s] + hdrs) or (src == "capi.hip" and old != bid):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + [f'-DOS_BUILD_ID="{bid}"', "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(os.path.join(objdir, "flags.txt"), "w") as fh:
        fh.write(_all_flags())
    with open(stamp, "w") as fh:
        fh.write(bid)
    return LIB


def _flags_changed(objdir):
    f = os.path.join(objdir, "flags.txt")
    return not os.path.exists(f) or open(f).read() != _all_flags()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
