#!/bin/bash
# HBM traffic of the bench kernels from PMC counters, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do
# not fit one pass), plus the calibration kernels with known byte counts.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/traffic
mkdir -p $OUT
rm -f $OUT/traffic_raw.json          # never leave a stale result for traffic_to_json.py to pick up
[ -x $R/tools/micro/traffic_cal ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/micro/traffic_cal.hip -o $R/tools/micro/traffic_cal
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -- $R/tools/micro/traffic_cal > $OUT/cal_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/run_$c -- python3 $R/tools/run_fused_once.py 2 > $OUT/run_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/rows_$c -- python3 $R/tools/run_rows_once.py 2 > $OUT/rows_$c.log 2>&1      # configs[1]: B = 4096, T = 1000
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for tag in ("cal", "run", "rows"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (tag, c), recursive=True)
        if not f: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            acc[r["Kernel_Name"].split("(")[0][-48:]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k][c] = sum(v) / len(v)
for k, d in res.items():
    if any(s in k for s in ("rd_dword", "wr_dword", "fused_kf_gru", "kf_run_sym", "kf_run_rows2")):
        print(k, {c: "%.4g" % v for c, v in d.items()})
# provenance, recorded where and when the counters were read: the content keys of the objects the measured kernels live in
import sys, datetime
sys.path.insert(0, "$R")
from optistate_amd import build as b, _capi
tool = b.toolchain_id()
res["_provenance"] = {"keys": {s: b.object_key(s, tool) for s in ("fused_kernels.hip", "kf_kernels.hip", "kf_rows_kernel.hip")},
                      "build_id": _capi.load().os_build_id().decode(), "collected": datetime.date.today().isoformat()}
json.dump(res, open("$OUT/traffic_raw.json", "w"), indent=1)
PY
