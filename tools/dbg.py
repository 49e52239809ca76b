import sys; sys.path.insert(0,'.')
import numpy as np, torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
eng=Engine(0); eng.set_noise(Q_FITTED,R_FITTED)
torch.manual_seed(5); m=RNN(60,64,1,24,torch.device("cpu"))
eng.load_gru(flatten_state_dict(m.state_dict(),1),60,64,1,24)
B,T=64,1
d=synth_numpy(B,T,seed=21)
s={k:eng.pack(torch.as_tensor(d[k])) for k in ("p","f","dp","imu","accel")}
c=eng.pack_contact(torch.as_tensor(d["contact"]))
mm=torch.stack([torch.full((60,),-3.0),torch.full((60,),3.0)]).cuda()
outs=[]
for tk in (False,True):
    x=torch.as_tensor(d["x0"].T.copy()).cuda(); P=torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144,1),(1,B))).cuda()
    r=eng.fused_run(s["p"],s["f"],s["dp"],s["imu"],c,s["accel"],mm,x,P,two_kernel=tk); torch.cuda.synchronize()
    outs.append(r["out"].cpu().numpy())
o=outs[0]; ref=outs[1]
for rrow in range(28,64,3):
    dist=np.abs(ref-o[rrow]).max(1); print(rrow,'best match ref row',dist.argmin(), dist.min(), 'own', dist[rrow])
