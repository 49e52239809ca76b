#!/bin/bash
# Kernel-trace one bench mode and print the per-kernel averages (development aid; run on the GPU box through gpurun).
# usage: tools/ktrace.sh <tag> <bench args...>      -> gpurun_out/<tag>/{bench.json,k_kernel_stats.csv}
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/bench.py "$@" --cpu-seconds 0 --parity-samples 0 > $O/bench.json 2> $O/err.log)
python3 - <<PY
import csv, json
try:
    print("ms_per_step", round(json.load(open("$O/bench.json"))["ms_per_step"], 4))
except Exception as e:
    print("bench line unreadable:", e)
rows = list(csv.DictReader(open("$O/k_kernel_stats.csv")))
for r in rows[:14]:
    print("%-72s %5s %9.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
