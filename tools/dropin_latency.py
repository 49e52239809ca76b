#!/usr/bin/env python3
"""Latency of the drop-in Kalman_Filter class at B = 1: microseconds per time step for
(a) the reference's four-call sequence get_odom / set_measurements / predict / update (three launches) and
(b) Kalman_Filter.step (one launch), on a synthetic trajectory.  The reference's own NumPy path: 327 us/step."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optistate_amd import Kalman_Filter                      # noqa: E402
from optistate_amd.synth import synth_numpy                   # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
d = synth_numpy(1, T, seed=3)
col = lambda k, t, n: d[k][0, t].astype(np.float64).reshape(n, 1)
res = {}
for mode in ("four_calls", "step", "step_with_K"):
    kf = Kalman_Filter()
    for rep in range(2):                                      # first pass warms up (context, staging block, code object)
        t0 = time.perf_counter()
        for t in range(T):
            p = col("p", t, 12)
            if mode == "four_calls":
                od = kf.get_odom(p, col("dp", t, 12), d["contact"][0, t].reshape(4, 1), col("imu", t, 6))
                kf.set_measurements(col("imu", t, 6), od)
                kf.predict(p, col("f", t, 12))
                kf.update()
            else:
                kf.step(p, col("f", t, 12), col("dp", t, 12), col("imu", t, 6), d["contact"][0, t].reshape(4, 1), want_K=mode == "step_with_K")
        el = time.perf_counter() - t0
    res[mode + "_us_per_step"] = el / T * 1e6
res["steps"] = T
res["reference_numpy_us_per_step"] = 327.0
print(json.dumps(res))
