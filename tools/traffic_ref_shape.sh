#!/bin/bash
# HBM traffic of the reference-shape path (GRU(188,128,4,24), B = 65,536, T = 100) from PMC counters, one counter per pass, with
# the same in-run calibration as tools/traffic_pass.sh.  Prints bytes per launch and kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/traffic_ref
mkdir -p $OUT
[ -x $R/tools/micro/traffic_cal ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/micro/traffic_cal.hip -o $R/tools/micro/traffic_cal
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -- $R/tools/micro/traffic_cal > $OUT/cal_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/run_$c -- python3 $R/tools/run_ref_shape_once.py 2 > $OUT/run_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
res = collections.defaultdict(dict)
for tag in ("cal", "run"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (tag, c), recursive=True)
        if not f: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k][c] = (sum(v) / len(v), len(v))
GiB_KiB = float(1 << 20)
fr = res["rd_dword"]["FETCH_SIZE"][0] / GiB_KiB; wr = res["wr_dword"]["WRITE_SIZE"][0] / GiB_KiB
print("calibration: FETCH_SIZE reports %.3f of the bytes read, WRITE_SIZE %.3f of the bytes written" % (fr, wr))
steps = 65536 * 100
for k, d in res.items():
    if any(s in k for s in ("gru_layer", "kf_run_sym", "gru_head")):
        rb = d.get("FETCH_SIZE", (0, 0))[0] / fr * 1024; wb = d.get("WRITE_SIZE", (0, 0))[0] / wr * 1024
        print("%-42s launches %d  read %.3f GB  written %.3f GB per launch = %.0f B per (trajectory, step)" % (k, d.get("FETCH_SIZE", (0, 0))[1], rb / 1e9, wb / 1e9, (rb + wb) / steps))
PY
