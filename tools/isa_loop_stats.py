#!/usr/bin/env python3
"""Static instruction mix of a kernel's hot loop from hipcc's assembly (development aid).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s file.hip
    python tools/isa_loop_stats.py k.s <kernel-name-substring>

Finds the longest backward-branch span in the kernel (the T loop) and counts instruction classes in it: MFMA, other
VALU (split: transcendental, packed, accvgpr moves, v_mov, lane ops), SALU, LDS, VMEM (split: scratch), waitcnt/nop.
With one wavefront per SIMD the non-MFMA VALU count x ~5.3 cycles is what the loop pays beside the MFMAs."""
import re
import sys
from collections import Counter


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    _, lo, hi = best
    c = Counter()
    for l in body[lo:hi + 1]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";", "//")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
                c["valu.trans"] += 1
            elif op.startswith("v_pk_"):
                c["valu.packed"] += 1
            elif op.startswith("v_accvgpr"):
                c["valu.accvgpr"] += 1
            elif op.startswith("v_mov"):
                c["valu.mov"] += 1
            elif op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")):
                c["valu.lane"] += 1
            elif "f64" in op:
                c["valu.f64"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("scratch_") or (op.startswith("buffer_") and "offen" in l and "s[0:3]" in l):
            c["vmem.scratch"] += 1
        elif op.startswith(("buffer_", "global_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_nop"):
            c["nop"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    print(f"loop lines {lo}..{hi} of kernel ({hi - lo} lines)")
    for k in sorted(c):
        print(f"  {k:14s} {c[k]}")
    print(f"  est. cycles/wave-step at 1 wave/SIMD: mfma {c['mfma'] * 64} + valu {(c['valu'] - c['valu.trans']) * 5.3 + c['valu.trans'] * 9:.0f}")


if __name__ == "__main__":
    main()
