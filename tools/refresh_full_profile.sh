#!/bin/bash
# Re-collect only the config-5 evidence (kernel stats under rocprofv3 + the plain bench line) into gpurun_out/<tag>/.
# usage: tools/refresh_full_profile.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof_full --output-format csv -- python3 $R/bench.py --mode full --cpu-seconds 0 > $O/bench_full_under_rocprof.json 2> $O/prof_full.log)
python3 - <<PY > $O/full_kernel_stats.md
import csv, glob
f = glob.glob("$O/prof_full/**/*kernel_stats.csv", recursive=True)
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode full --cpu-seconds 0\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
if f:
    for r in list(csv.DictReader(open(f[0])))[:16]:
        print("| %s | %s | %.3f | %.1f | %.1f | %.1f | %s |" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
python3 bench.py --mode full > $O/bench_full.json 2> $O/bench.err
rm -rf $O/prof_full
ls $O
