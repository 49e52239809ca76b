#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in the shipped liboptistate_hip.so, read from the code objects themselves.

    python tools/codeobj_report.py [--md profiles/rNN_scratch.md] [--lib path]

The library's `.hip_fatbin` section holds one clang offload bundle per translation unit; each bundle's gfx950 entry is an
AMDGPU ELF whose NT_AMDGPU_METADATA note lists, per kernel: vgpr / agpr / sgpr counts, vgpr / sgpr SPILL counts, the private
(scratch) segment size per lane and the static LDS size.  `llvm-readelf --notes` prints that note as YAML-like text; this
parses it.  No GPU needed.  tests/test_codeobj_budget.py holds a named list of kernels at scratch 0 with it.
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "optistate_amd", "lib", "liboptistate_hip.so")
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(path, name):
    """Bytes of an ELF64 section (little endian) without any external tool."""
    with open(path, "rb") as fh:
        data = fh.read()
    assert data[:4] == b"\x7fELF" and data[4] == 2, "not an ELF64 file"
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    sh = lambda i: struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
    stroff = sh(shstrndx)[4]
    for i in range(shnum):
        n, _, _, _, off, size = sh(i)[:6]
        end = data.index(b"\0", stroff + n)
        if data[stroff + n:end].decode() == name:
            return data[off:off + size]
    raise KeyError(name)


def code_objects(lib=LIB, arch="gfx950"):
    """The gfx950 code objects inside the library's fat binary, one per translation unit."""
    fat = _section(lib, ".hip_fatbin")
    out, pos = [], 0
    while True:
        pos = fat.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", fat, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", fat, q)
            triple = fat[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if arch in triple and size:
                out.append(fat[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


_KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
         "group_segment_fixed_size", "max_flat_workgroup_size")


def kernels_of(co_bytes):
    """[{name, vgpr_count, ...}] from one code object's metadata note."""
    with tempfile.NamedTemporaryFile(suffix=".co") as tf:
        tf.write(co_bytes); tf.flush()
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", tf.name], capture_output=True, text=True, check=True).stdout
    # the note is YAML: kernel entries are the items of `amdhsa.kernels:` ("  - .key:" at indent 2, further keys at indent 4);
    # argument entries sit deeper and are skipped by their indent
    ks, cur, on = [], None, False
    for line in txt.splitlines():
        if line.startswith("amdhsa.kernels:"):
            on = True
            continue
        if on and line and not line.startswith(" "):
            on = False
        if not on:
            continue
        m = re.match(r"^(  - | {4})\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        if m.group(1) == "  - ":
            cur = {}
            ks.append(cur)
        key, val = m.group(2), m.group(3).strip()
        if key == "symbol":
            cur["symbol"] = val.strip("'\"")
        elif key in _KEYS:
            cur[key] = int(val)
    res = []
    for k in ks:
        if "symbol" not in k:
            continue
        sym = k["symbol"][:-3] if k["symbol"].endswith(".kd") else k["symbol"]
        d = {kk: k.get(kk, 0) for kk in _KEYS}
        d["symbol"] = sym
        res.append(d)
    return res


def demangle(names):
    r = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else ["c++filt"],
                       input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.splitlines() if r.returncode == 0 else names
    return [re.sub(r"^void\s+", "", re.sub(r"\(.*\)$", "", o)) for o in out]


def report(lib=LIB):
    rows = []
    for co in code_objects(lib):
        rows += kernels_of(co)
    for r, n in zip(rows, demangle([r["symbol"] for r in rows])):
        r["name"] = n
    return sorted(rows, key=lambda r: r["name"])


def to_markdown(rows):
    lines = ["| kernel | VGPR | AGPR | SGPR | VGPR spills | SGPR spills | scratch B/lane | LDS B |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| `{r['name']}` | {r['vgpr_count']} | {r['agpr_count']} | {r['sgpr_count']} | {r['vgpr_spill_count']} | "
                     f"{r['sgpr_spill_count']} | {r['private_segment_fixed_size']} | {r['group_segment_fixed_size']} |")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--md", default=None, help="write the table to this markdown file")
    ap.add_argument("--nonzero", action="store_true", help="only kernels with scratch or spills")
    a = ap.parse_args()
    rows = report(a.lib)
    if a.nonzero:
        rows = [r for r in rows if r["private_segment_fixed_size"] or r["vgpr_spill_count"] or r["sgpr_spill_count"]]
    md = to_markdown(rows)
    if a.md:
        # a hand-written preamble (everything up to the line "All kernels:") survives regeneration
        head = (f"# Register / scratch budget of every kernel in liboptistate_hip.so\n\nGenerated by `tools/codeobj_report.py` from the "
                f"gfx950 code objects inside the library ({len(rows)} kernels).\n\n")
        if os.path.exists(a.md):
            old = open(a.md).read()
            if "\nAll kernels:\n" in old:
                head = old[:old.index("\nAll kernels:\n")] + "\nAll kernels:\n\n"
        with open(a.md, "w") as fh:
            fh.write(head + md + "\n")
    print(md)


if __name__ == "__main__":
    sys.exit(main())
