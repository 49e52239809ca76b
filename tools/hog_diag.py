import os, sys, time, subprocess
sys.path.insert(0, os.getcwd())
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.engine import _ptr
os.environ["OS_GRU_VEC"] = "0"
dims = (188, 128, 4, 24)
torch.manual_seed(2)
m = RNN(*dims, torch.device("cpu"))
flat = flatten_state_dict(m.state_dict(), 4, "cuda")
x = torch.rand(64, 10, 188, device="cuda")
eng = Engine(0); eng.load_gru(flat, *dims)
out = torch.empty((64, 24), device="cuda")
eng.lib.os_gru_forward(eng._h, 64, 10, _ptr(x), _ptr(out), None, eng._stream()); torch.cuda.synchronize()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "tools/micro/cu_hog.hip", "-o", "/tmp/cu_hog"], check=True)
nb = sys.argv[1] if len(sys.argv) > 1 else "240"
hog = subprocess.Popen(["/tmp/cu_hog", nb, sys.argv[2] if len(sys.argv) > 2 else "6000"], stdout=subprocess.PIPE, text=True)
print(hog.stdout.readline().strip()); time.sleep(0.3)
t0 = time.perf_counter()
rc = eng.lib.os_gru_forward(eng._h, 64, 10, _ptr(x), _ptr(out), None, eng._stream())
t1 = time.perf_counter()
print("stacked call rc", rc, "after", t1 - t0, "s; hog alive", hog.poll() is None)
eng.lib.os_gru_set_stack(eng._h, 0)
rc = eng.lib.os_gru_forward(eng._h, 64, 10, _ptr(x), _ptr(out), None, eng._stream()); torch.cuda.synchronize()
print("per-layer call rc", rc, "after", time.perf_counter() - t1, "s; hog alive", hog.poll() is None)
hog.wait()
