#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV: per queue, the kernels' start/end (us from the first start) of the last N dispatches, and how much of the
time between the first and last of them had 0 / 1 / 2+ kernels in flight.  argv: <kernel_trace.csv> [N]"""
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-N:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(f'q{r["Queue_Id"]:>3} {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} -> {(int(r["End_Timestamp"])-t0)/1e3:9.1f} us  {r["Kernel_Name"][:60]}')
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in rows] + [(int(r["End_Timestamp"]), -1) for r in rows])
depth = 0; last = ev[0][0]; acc = {}
for t, d in ev:
    acc[min(depth, 2)] = acc.get(min(depth, 2), 0) + t - last
    depth += d; last = t
tot = sum(acc.values())
print({k: round(v / tot, 3) for k, v in sorted(acc.items())}, "of", round(tot / 1e3, 1), "us with 0 / 1 / 2+ kernels in flight")
