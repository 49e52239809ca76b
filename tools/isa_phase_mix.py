#!/usr/bin/env python3
"""Static instruction mix per phase of the QP solvers: compiles mpc_quad.hip / mpc_kernels.hip with their timestamp macros (OSQ_STAMP /
OSM_STAMP) turned into assembly comments and counts, between consecutive marks of one function, the instructions by class (f64 VALU,
DPP, moves / selects, v_readlane / v_writelane, AGPR moves, SALU, branches, s_nop, s_waitcnt, LDS, scratch, global memory).  Runs in the
container (no GPU).  The layout of basic blocks is the compiler's: a phase with internal branches may be split over several regions.
usage: tools/isa_phase_mix.py quad|wave [extra hipcc flags, e.g. -DOSQ_OCC=1]"""
import collections, os, re, subprocess, sys, tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "quad"
flags = sys.argv[2:]
src, macro, func = {"quad": ("mpc_quad.hip", "OSQ_STAMP", "_ZN3osq21mpc_solve_quad_kernelILi2ELi2EEE"),
                    "wave": ("mpc_kernels.hip", "OSM_STAMP", "_ZN3osmL19mpc_solve_wave_callILi2EEE")}[which]
text = open(os.path.join(R, "optistate_amd", "csrc", src)).read()
text = text.replace(f"#define {macro}(i)\n#endif", f'#define {macro}(i) asm volatile("; PHASEMARK " #i);\n#endif')
with tempfile.TemporaryDirectory() as d:
    cpy, asm = os.path.join(d, "marked.hip"), os.path.join(d, "marked.s")
    open(cpy, "w").write(text)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                    "-I" + os.path.join(R, "optistate_amd", "csrc"), *flags, cpy, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(func) and l.rstrip().endswith(":") or (l.startswith(func) and ": ;" in l))
end = next((i for i in range(start + 1, len(lines)) if ".amdhsa_kernel" in lines[i] or re.match(r"^_Z\w+:", lines[i])), len(lines))
L = lines[start:end]


def cls(l):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        return None
    op = l.split()[0]
    if op.startswith("v_"):
        if "dpp" in l: return "dpp"
        if "accvgpr" in op: return "agpr"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
        if op.startswith(("v_mov", "v_cndmask")): return "mov/sel"
        if "f64" in op: return "f64"
        return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch"): return "scratch"
    if op.startswith(("global", "buffer", "flat")): return "vmem"
    return "other"


marks = [(0, "entry")] + [(i, l.split("PHASEMARK")[1].strip()) for i, l in enumerate(L) if "PHASEMARK" in l] + [(len(L), "end")]
print(f"{src}: {func}  ({len(L)} lines)")
for (a, na), (b, nb) in zip(marks, marks[1:]):
    c = collections.Counter(filter(None, (cls(l) for l in L[a:b])))
    if c:
        print(f"  {na:>6} -> {nb:<6} {sum(c.values()):5d}  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))
