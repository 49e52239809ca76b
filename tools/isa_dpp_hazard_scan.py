#!/usr/bin/env python3
"""Scans hipcc assembly for two gfx9 hazards that inline assembly must handle itself (hipcc's hazard recogniser does not look
inside an asm statement):
  * a VALU write of a VGPR needs two wait states before a DPP instruction reads that VGPR through its DPP operand (src0);
  * a VALU write of an SGPR (v_readfirstlane, v_readlane -- every reload of a spilled SGPR --, v_cmp with an SGPR destination)
    needs five wait states before a VMEM instruction reads that SGPR (descriptor or scalar offset).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s file.hip
    python tools/isa_dpp_hazard_scan.py k.s [kernel-name-substring]
"""
import re
import sys


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def regs(tok):
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().splitlines()
    kern, bad, ndpp, nvmem = None, 0, 0, 0
    hist = []          # (wait states this instruction provides to later ones, set of VGPRs written by a VALU)
    shist = []         # the same for SGPRs written by a VALU instruction
    for ln, l in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, hist, shist = m.group(1), [], []
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        ops = [o.strip() for o in t[len(op):].split(",")] if len(t) > len(op) else []
        if "_dpp" in op or " row_" in t or " quad_perm" in t:
            ndpp += 1
            src = regs(ops[1].split()[0]) if len(ops) > 1 else set()
            need, i = 2, len(hist) - 1
            while need > 0 and i >= 0:
                ws, wr = hist[i]
                if wr & src and (not pat or pat in (kern or "")):
                    print(f"{path}:{ln}: {kern}: DPP reads v{sorted(src)} written {2 - need} wait states earlier: {t}")
                    bad += 1
                    break
                need -= ws
                i -= 1
        if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            nvmem += 1
            used = set()
            for o in ops:
                used |= sregs(o.split()[0]) if o.split() else set()
            need, i = 5, len(shist) - 1
            while need > 0 and i >= 0:
                ws, wr = shist[i]
                if wr & used and (not pat or pat in (kern or "")):
                    print(f"{path}:{ln}: {kern}: VMEM reads s{sorted(wr & used)} written by a VALU {5 - need} wait states earlier: {t}")
                    bad += 1
                    break
                need -= ws
                i -= 1
        swr = set()
        if op.startswith(("v_readlane", "v_readfirstlane")) and ops:
            swr = sregs(ops[0].split()[0])
        elif op.startswith("v_cmp") and op.endswith("_e64") and ops:
            swr = sregs(ops[0].split()[0])
        shist.append((int(ops[0]) + 1 if (op == "s_nop" and ops) else 1, swr))
        shist = shist[-12:]
        if op == "s_nop":
            hist.append((int(ops[0]) + 1 if ops else 1, set()))
        elif op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            hist.append((1, regs(ops[0].split()[0]) if ops else set()))
        elif op.startswith(("s_waitcnt", ";;#")):
            continue
        else:
            hist.append((1, set()))
        hist = hist[-8:]
    print(f"{ndpp} DPP instructions and {nvmem} VMEM instructions scanned, {bad} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
