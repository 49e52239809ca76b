#!/usr/bin/env python3
"""gpurun_out/traffic/traffic_raw.json (tools/traffic_pass.sh) -> profiles/traffic.json with the gfx950 corrections applied.
FETCH_SIZE is reported in KiB and at exactly 1/2 of the bytes for coalesced dword-per-lane reads (calibrated in the same run:
rd_dword reads 1 GiB and reports 524,288 KiB; MI355X_MICROARCH.md section HBM says the same for 16 B/lane); WRITE_SIZE is exact
(wr_dword: 1 GiB -> 1,048,576 KiB)."""
import json, sys
raw = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traffic/traffic_raw.json"))
cal_r = raw[[k for k in raw if "rd_dword" in k][0]]["FETCH_SIZE"] * 1024 / 2 ** 30
cal_w = raw[[k for k in raw if "wr_dword" in k][0]]["WRITE_SIZE"] * 1024 / 2 ** 30
out = {"_calibration": {"fetch_reported_over_actual": cal_r, "write_reported_over_actual": cal_w,
                        "correction": "bytes = (FETCH_SIZE / fetch_ratio + WRITE_SIZE / write_ratio) * 1024"}}
for k, v in raw.items():
    for name in ("fused_kf_gru_kernel", "kf_run_sym_kernel"):
        if name in k:
            out[name] = (v["FETCH_SIZE"] / cal_r + v["WRITE_SIZE"] / cal_w) * 1024
            out[name + "_detail"] = {"read_bytes": v["FETCH_SIZE"] / cal_r * 1024, "write_bytes": v["WRITE_SIZE"] / cal_w * 1024,
                                     "shape": "B=65536, T=100 (bench shape), per launch"}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
