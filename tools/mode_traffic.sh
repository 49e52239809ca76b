#!/bin/bash
# usage: tools/mode_traffic.sh <mode>   -- FETCH_SIZE / WRITE_SIZE passes over `bench.py --mode <mode>`; raw CSVs under gpurun_out/pmc_<mode>_*
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc_any.sh $1_$c "$c" $GRAFT_REPO_ROOT/bench.py --mode $1 --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_$1_%s/**/*counter_collection.csv" % c, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = (sum(v) / len(v), len(v))
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0]):
    fe, wr = d.get("FETCH_SIZE", (0, 0)), d.get("WRITE_SIZE", (0, 0))
    if fe[0] + wr[0] < 1000: continue
    print(f"{k:72s} n={fe[1]:3d} read {fe[0] / 0.5 * 1024 / 1e6:8.1f} MB  write {wr[0] * 1024 / 1e6:8.1f} MB")
PY
