#!/bin/bash
# rocprofv3 --kernel-trace --stats of any python script of this repo -> gpurun_out/<name>_kernel_stats.md
# usage (GPU box): bash tools/kstats_any.sh <name> <script.py> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; N=$1; shift; S=$1; shift
O=$R/gpurun_out/kstats_$N; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $R/$S "$@" > $O/stdout.txt 2> $O/prof.log)
python3 - <<PY > $R/gpurun_out/${N}_kernel_stats.md
import csv, glob
f = glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True)
print("# rocprofv3 --kernel-trace --stats -- python3 $S $*\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
if f:
    for r in list(csv.DictReader(open(f[0])))[:18]:
        print("| %s | %s | %.3f | %.1f | %.1f | %.1f | %s |" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
cat $R/gpurun_out/${N}_kernel_stats.md
