#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run (kernel_stats.csv) into a short committed summary.
usage: tools/summarize_prof.py gpurun_out/prof_TAG profiles/TAG_kernel_stats.md "command line" """
import csv, glob, os, sys
src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
f = max(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # newest run
rows = list(csv.DictReader(open(f)))
with open(dst, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\nsource: `{os.path.relpath(f)}` (MI355X, gfx950)\n\n")
    o.write("| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        name = r["Name"]
        if len(name) > 110:
            name = name[:107] + "..."
        if float(r["Percentage"]) < 0.05:
            continue
        o.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.2f} | {float(r['MinNs'])/1e6:.4f} | {float(r['MaxNs'])/1e6:.4f} |\n")
print(open(dst).read())
