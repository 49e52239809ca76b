#!/bin/bash
# A/B: seq_out stores of gru_layer_stage_kernel non-temporal (aux 2, default) against the default cache policy (aux 0): time + HBM traffic
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/st0; mkdir -p $D
cd $R/optistate_amd/csrc
for f in capi kf_kernels kf_rows_kernel kf_step gru_kernels fused_kernels gru_train_kernels vit_kernels mpc_kernels; do
  X=; [ $f = gru_kernels ] && X=-DOS_STAGE_STORE_AUX=0; [ $f = kf_rows_kernel ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $X -DOS_BUILD_ID='"st0"' -c $f.hip -o $D/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboptistate_st0.so $D/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
cd $R
for lib in nt st0 nt st0; do
  [ $lib = st0 ] && export OPTISTATE_HIP_LIB=$D/liboptistate_st0.so || unset OPTISTATE_HIP_LIB
  python3 bench.py --hidden 128 --layers 4 --latent 128 --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 0 --no-second-noise 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['kernels']['gru_layer']['ms_per_launch'], d['kernels']['kf']['ms_per_launch'])"
done
export OPTISTATE_HIP_LIB=$D/liboptistate_st0.so
bash tools/traffic_ref_shape.sh 2>&1 | tail -4
