#!/bin/bash
# Same-box A/B of round 3's kernels (sources exported by `git archive ac10c19` into build_ab/r03src) against the current library,
# through the same Python host code (OPTISTATE_HIP_LIB): training step, sliding windows, the reference's model shape.
R=${GRAFT_REPO_ROOT:-/root/repo}; S=$R/build_ab/r03src/optistate_amd/csrc; D=$R/build_ab/r03obj; mkdir -p $D
cd $S
for f in capi kf_kernels kf_rows_kernel kf_step gru_kernels fused_kernels gru_train_kernels vit_kernels mpc_kernels; do
  X=; [ $f = kf_rows_kernel ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $X -DOS_BUILD_ID='"r03-ab"' -c $f.hip -o $D/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboptistate_r03.so $D/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
cd $R
pick() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: round(v["ms_per_launch"], 4) for n, v in d.get("kernels", {}).items()}
print(sys.argv[2], "value %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]), k)
PY
}
O=$R/gpurun_out/ab_r03_r04; mkdir -p $O
for rep in 1 2; do
for lib in r03 r04; do
  [ $lib = r03 ] && export OPTISTATE_HIP_LIB=$D/liboptistate_r03.so || unset OPTISTATE_HIP_LIB
  python3 bench.py --mode train --steps 20 --warmup 3 --cpu-seconds 0 > $O/train_${lib}_$rep.json 2>/dev/null; pick $O/train_${lib}_$rep.json "train 20/3 $lib"
  python3 bench.py --mode train --cpu-seconds 0 > $O/train_default_${lib}_$rep.json 2>/dev/null; pick $O/train_default_${lib}_$rep.json "train 10/2 $lib"
  python3 bench.py --mode windows --cpu-seconds 0 > $O/windows_${lib}_$rep.json 2>/dev/null; pick $O/windows_${lib}_$rep.json "windows $lib"
  python3 bench.py --hidden 128 --layers 4 --latent 128 --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 0 --no-second-noise > $O/ref_${lib}_$rep.json 2>/dev/null; pick $O/ref_${lib}_$rep.json "GRU(188,128,4) $lib"
  python3 bench.py --hidden 64 --layers 4 --steps 5 --warmup 1 --cpu-seconds 0 --parity-samples 0 --no-second-noise > $O/h64l4_${lib}_$rep.json 2>/dev/null; pick $O/h64l4_${lib}_$rep.json "GRU(60,64,4) $lib"
done; done
