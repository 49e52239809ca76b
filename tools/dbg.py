import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from optistate_amd import Kalman_Filter
from optistate_amd.synth import synth_numpy
d=synth_numpy(1,300,seed=1)
kf=Kalman_Filter()
def step(t):
    p=d["p"][0,t].astype(np.float64).reshape(12,1)
    od=kf.get_odom(p,d["dp"][0,t].reshape(12,1),d["contact"][0,t].reshape(4,1),d["imu"][0,t].reshape(6,1))
    kf.set_measurements(d["imu"][0,t].reshape(6,1),od)
    kf.predict(p,d["f"][0,t].reshape(12,1)); kf.update()
for t in range(20): step(t)
t0=time.perf_counter()
for t in range(20,220): step(t)
print("drop-in class, 4 calls per step: %.0f us/step"%((time.perf_counter()-t0)/200*1e6))
t0=time.perf_counter()
for t in range(20,220):
    p=d["p"][0,t].astype(np.float64).reshape(12,1)
    kf.estimate_state_mpc(d["imu"][0,t].reshape(6,1),p,d["dp"][0,t].reshape(12,1),np.zeros((12,1)),d["contact"][0,t].reshape(4,1),f=d["f"][0,t])
print("drop-in class, estimate_state_mpc (one launch + odom): %.0f us/step"%((time.perf_counter()-t0)/200*1e6))
