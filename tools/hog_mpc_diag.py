"""Development: does an os_kf_mpc_run before it change what the starvation scenario of tests/test_gpu_contention.py sees?  (Yes, with THREE
parts: two extra streams of the process, and the wide GRU kernel beside the hog is then not dispatched at all until the hog leaves -- no
expired wait, no fallback, the kernel's own result 2.7 s later.)  argv: none | all | <case index 0..3>"""
import os, sys, time, subprocess, shutil, torch
ROOT = "/root/repo"; sys.path.insert(0, ROOT)
from optistate_amd import Engine, RNN
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
which = sys.argv[1] if len(sys.argv) > 1 else "fused"
os.environ["OS_MPC_PERSISTENT"] = "0"
dev = torch.device("cuda:0")
if which != "none":
    cases = [(4096, 8, "1", True), (4096, 6, "1", False), (32768, 5, "2", False), (49168, 4, "3", True)]
    if which.isdigit():
        cases = [cases[int(which)]]
    for B, T, shards, mix in cases:
        os.environ["OS_MPC_SHARDS"] = shards
        d = synth_torch(B, T, dev, seed=77)
        c4 = d["contact"].clone()
        if mix:
            c4[:, :, 5::7] = 0; c4[:, 1, 5::7] = 1
            c4[2:, :, 3::11] = 0
            if shards == "1":
                c4[3, :, 8::13] = 1; c4[3, 0, 8::13] = 0
        ref = torch.zeros((T, 12, B), device=dev); ref[:, 5] = 0.28; ref[:, 9] = 0.1
        for fuse in ("0", "1"):
            os.environ["OS_MPC_FUSE_KF"] = fuse
            e = Engine(0); e.set_noise(Q_DEFAULT, R_DEFAULT)
            contact = e.contact_soa_to_packed(c4)
            r = e.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, d["x0"].clone(), d["P0"].clone(), want_iters=True, want_p_rot=True)
            e.profile(True)
            e.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, d["x0"].clone(), d["P0"].clone())
            e.profile_read()
        torch.cuda.synchronize()
        print("mpc case done:", B, T, shards, mix, e.kernel_name("mpc"))
    for k in ("OS_MPC_SHARDS", "OS_MPC_FUSE_KF", "OS_MPC_PERSISTENT"):
        os.environ.pop(k, None)
exe = "/tmp/cu_hog"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "micro", "cu_hog.hip"), "-o", exe], check=True)
torch.manual_seed(9)
m = RNN(188, 128, 4, 24, torch.device("cuda")).to("cuda").eval()
x = torch.rand(64, 10, 188, device="cuda")
with torch.no_grad():
    m(x); eng = m._engine
    print("kernel", eng.kernel_name("gru_layer"))
    eng.set_stack_mode(0); ref = m(x).clone(); eng.set_stack_mode(1)
    w = m(x).clone()
torch.cuda.synchronize()
print("idle: wide vs per-layer equal:", torch.equal(w, ref))
fb0 = eng.stack_fallbacks
hog = subprocess.Popen([exe, "240", "3000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(hog.stdout.readline().strip()); time.sleep(0.3)
t0 = time.time()
with torch.no_grad(): out = m(x)
torch.cuda.synchronize(); el = time.time() - t0
hog.wait(timeout=60)
print(f"under hog: {el:.2f} s, fallbacks {eng.stack_fallbacks - fb0}, equal {torch.equal(out, ref)}, max diff {float((out - ref).abs().max()):.3e}, nonfinite {int((~torch.isfinite(out)).sum())}")
with torch.no_grad(): again = m(x)
print("after hog: equal", torch.equal(again, ref), "kernel", eng.kernel_name("gru_layer"))
