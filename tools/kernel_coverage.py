"""Which kernels of the shipped library did a profiled run launch?  Reads every rocprofv3 `*kernel_stats.csv` under a directory
(one per traced process) and compares the kernel names with the library's code objects (tools/codeobj_report.py).
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cov -- python3 -m pytest $R/tests -m gpu -q
                  python3 tools/kernel_coverage.py gpurun_out/cov > gpurun_out/kernel_coverage.md"""
import collections
import csv
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj_report as cr   # noqa: E402


def norm(name):
    name = re.sub(r"^void ", "", name.strip())
    depth = 0
    for i, ch in enumerate(name):                       # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main(d):
    calls = collections.Counter()
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            calls[norm(r["Name"])] += int(r["Calls"])
    lib = sorted(r["name"] for r in cr.report())
    hit = [k for k in lib if calls.get(k)]
    miss = [k for k in lib if not calls.get(k)]
    print(f"# Kernels of liboptistate_hip.so launched by the profiled run ({len(files)} traced processes)\n")
    print(f"{len(hit)} of {len(lib)} kernels launched; never launched: {len(miss)}\n")
    print("| never launched |\n|---|")
    for k in miss:
        print(f"| `{k}` |")
    print("\n| launched | calls |\n|---|---|")
    for k in hit:
        print(f"| `{k}` | {calls[k]} |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cov")
