#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04v; mkdir -p $O; cd $R
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $R/tools/train_small_batch.py 100 > $O/tsb.json 2> $O/prof.log)
python3 - <<PY
import csv, glob
f = glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:22]:
        print("| %s | %s | %.3f | %.1f |" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
cat $O/tsb.json
