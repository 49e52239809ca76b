import sys; sys.path.insert(0,'.')
import numpy as np, torch
from optistate_amd import Engine
from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
from oracle import c_oracle as orc
B,T=150,12
d=synth_numpy(B,T,seed=33); rng=np.random.default_rng(5)
body_ref=np.zeros((B,T,12),dtype=np.float32); body_ref[...,0:3]=d["imu"][...,0:3]+rng.normal(0,0.01,(B,T,3)).astype(np.float32)
ref=orc.kf_run_batch(d["p"],d["f"],d["dp"],d["imu"],d["contact"],d["x0"],np.tile(Q_FITTED,(B,1,1)),Q_FITTED,R_FITTED,body_ref=body_ref,mode=1)
eng=Engine(0); eng.set_noise(Q_FITTED,R_FITTED)
s={k:eng.pack(torch.as_tensor(d[k])) for k in ("p","f","dp","imu")}; c=eng.pack_contact(torch.as_tensor(d["contact"]))
x=torch.as_tensor(d["x0"].T.copy()).cuda(); P=torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144,1),(1,B))).cuda()
r=eng.kf_run(s["p"],s["f"],s["dp"],s["imu"],c,x,P,body_ref=eng.pack(torch.as_tensor(body_ref)),dense_fd=True,sequential=False)
xo=eng.unpack(r["x_out"]).cpu().numpy()
err=np.abs(xo-ref["x"])
print('per-state max err',err.max(axis=(0,1)).round(5))
print('per-step max err',err.max(axis=(0,2)).round(5))
print('P trace ref', ref["P_trace"][0,:5], 'x scale', np.abs(ref["x"]).max(axis=(0,1)).round(3))
