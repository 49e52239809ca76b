#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters...>"   -- one rocprofv3 --pmc pass over tools/run_fused_once.py
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $@ --output-format csv -d $OUT -- python3 $R/tools/run_fused_once.py 2 > $OUT.log 2>&1
tail -2 $OUT.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "fused_kf_gru" in k or "kf_run_sym" in k:
        acc[k[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
