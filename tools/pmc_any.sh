#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<counters...>" <python-script> [args...]   -- one rocprofv3 --pmc pass, per-kernel means
TAG=$1; CNT=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 "$@" > $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = max(len(v) for v in d.values())
    if sum(d.get("SQ_WAVE_CYCLES", [0])) < 1e6 and "GRBM" not in str(d.keys()): continue
    print(k, "launches", n)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} mean={sum(v)/len(v):.4g}")
PY
