#!/bin/bash
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $@ --output-format csv -d $OUT -- python3 $R/bench.py --mode train --steps 2 --warmup 1 > $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "bwd_sweep" in k or "gru_layer" in k or "dw_kernel" in k:
        acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
