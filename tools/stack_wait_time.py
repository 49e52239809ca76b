import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
os.environ["OS_STACK_DBG_DROP"] = "1,3"; os.environ["OS_GRU_VEC"] = "0"
if len(sys.argv) > 1: os.environ["OS_STACK_DBG_POLLS"] = sys.argv[1]
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.engine import _ptr
dims = (188, 128, 4, 24)
torch.manual_seed(2)
m = RNN(*dims, torch.device("cpu"))
flat = flatten_state_dict(m.state_dict(), 4, "cuda")
x = torch.rand(64, 10, 188, device="cuda")
eng = Engine(0); eng.load_gru(flat, *dims)
out = torch.empty((64, 24), device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
rc = eng.lib.os_gru_forward(eng._h, 64, 10, _ptr(x), _ptr(out), None, eng._stream())
print("polls", os.environ.get("OS_STACK_DBG_POLLS", "default 2^22"), "rc", rc, "seconds", time.perf_counter() - t0, eng.lib.os_last_error(eng._h)[:60])
