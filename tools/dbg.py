import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from optistate_amd import Kalman_Filter
g=np.load('tests/golden/kf_g3_traj.npz')
kf=Kalman_Filter()
kf.x[:]=g["x0"][0].reshape(12,1); kf.Q=g["Q1"].copy(); kf.R=g["R1"].copy(); kf.P=g["Q1"].copy()
for t in range(2):
    p=g["p"][0,t].astype(np.float64).reshape(12,1)
    od=kf.get_odom(p,g["dp"][0,t].reshape(12,1),g["contact"][0,t].reshape(4,1),g["imu"][0,t].reshape(6,1))
    kf.set_measurements(g["imu"][0,t].reshape(6,1),od)
    print('z err',np.abs(kf.z.ravel()-g["s1_b0_z"][t]).max())
    kf.predict(p,g["f"][0,t].reshape(12,1))
    print('prior err',np.abs(kf.x_model.ravel()-g["s1_b0_x_prior"][t]).max(), 'P', kf.P_trace)
    kf.update()
    print('post err',np.abs(kf.x.ravel()-g["s1_b0_x"][t]), 'ptrace', kf.P_trace, g["s1_b0_P_trace"][t], 'kg', kf.K_gain, g["s1_b0_K_gain"][t])
    if t in (0,1): print('K err', np.abs(kf.K-g[f"s1_b0_K{t}"]).max()); print(np.round(kf.K-g[f"s1_b0_K{t}"],4))
