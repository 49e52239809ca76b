#!/usr/bin/env python3
"""gpurun_out/traffic/traffic_raw.json (tools/traffic_pass.sh: one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass, each with
the 1 GiB calibration kernels) -> profiles/traffic.json: HBM bytes per launch of the bench kernels.

Correction as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a coalesced streaming read.  The factor is not assumed: it is measured in the same run on a
1 GiB dword-per-lane read / write (tools/micro/traffic_cal.hip), this code base's access pattern."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "traffic", "traffic_raw.json")
if not os.path.exists(raw_path):        # tools/traffic_pass.sh deletes the previous result first: a failed pass must not publish stale traffic
    sys.exit(f"traffic_to_json: {raw_path} is missing (did tools/traffic_pass.sh fail?)")
raw = json.load(open(raw_path))
if "rd_dword" not in raw or "wr_dword" not in raw:
    sys.exit("traffic_to_json: calibration kernels missing from the raw counters")
GiB_KiB = float(1 << 20)
fr = raw["rd_dword"]["FETCH_SIZE"] / GiB_KiB
wr = raw["wr_dword"]["WRITE_SIZE"] / GiB_KiB
out = {"_calibration": {"fetch_reported_over_actual": fr, "write_reported_over_actual": wr,
                        "correction": "bytes = (FETCH_SIZE / fetch_ratio + WRITE_SIZE / write_ratio) * 1024"}}
prov = raw.get("_provenance", {})
SRC = {"fused_kf_gru_kernel_v2": "fused_kernels.hip", "fused_kf_gru_bf16_kernel": "fused_kernels.hip", "fused_kf_gru_kernel": "fused_kernels.hip",
       "kf_run_sym_kernel": "kf_kernels.hip", "kf_run_rows2_kernel": "kf_rows_kernel.hip"}
SHAPES = {"kf_run_rows2_kernel": "B=4096, T=1000 (BASELINE configs[1]), per launch"}
for key, name in (("fused_kf_gru_kernel_v2", "fused_kf_gru_kernel_v2"), ("fused_kf_gru_bf16_kernel", "fused_kf_gru_bf16_kernel"),
                  ("kf_run_sym_kernel", "kf_run_sym_kernel"), ("fused_kf_gru_kernel<", "fused_kf_gru_kernel"),
                  ("kf_run_rows2_kernel", "kf_run_rows2_kernel")):
    hit = [k for k in raw if key in k and not k.startswith("_")]
    if not hit:
        continue
    d = raw[hit[0]]
    rb, wb = d.get("FETCH_SIZE", 0.0) / fr * 1024, d.get("WRITE_SIZE", 0.0) / wr * 1024
    out[name] = rb + wb
    out[name + "_detail"] = {"read_bytes": rb, "write_bytes": wb, "shape": SHAPES.get(name, "B=65536, T=100 (bench shape), per launch"),
                             # provenance (bench.py load_traffic): the content key of the object this kernel was compiled into, as
                             # recorded ON THE GPU BOX by tools/traffic_pass.sh when the counters were read, and the date
                             "source_key": prov.get("keys", {}).get(SRC.get(name)), "source": SRC.get(name),
                             "build_id": prov.get("build_id"), "collected": prov.get("collected")}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
