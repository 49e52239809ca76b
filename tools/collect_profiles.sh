#!/bin/bash
# One GPU call that (re)collects the round's evidence under gpurun_out/<tag>/: PMC traffic -> profiles/traffic.json, the
# default bench line, rocprofv3 --kernel-trace --stats of the same command, SQ counters of the fused kernel, and the other
# bench lines.  usage: tools/collect_profiles.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
export GRAFT_REPO_ROOT=$R          # the sub-scripts below resolve their paths from it
mkdir -p $O
cd $R
bash tools/traffic_pass.sh > $O/traffic_pass.log 2>&1
python3 tools/traffic_to_json.py > $O/traffic.json 2>> $O/traffic_pass.log
[ -s $O/traffic.json ] && cp profiles/traffic.json $O/traffic_profiles.json
python3 bench.py > $O/bench.json 2> $O/bench.err
stats() {   # $1 = name, rest = bench args
  n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof_$n --output-format csv -- python3 $R/bench.py "$@" > $O/bench_${n}_under_rocprof.json 2> $O/prof_$n.log)
  python3 - <<PY > $O/${n}_kernel_stats.md
import csv, glob
f = glob.glob("$O/prof_$n/**/*kernel_stats.csv", recursive=True)
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py $*\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
if f:
    for r in list(csv.DictReader(open(f[0])))[:16]:
        print("| %s | %s | %.3f | %.1f | %.1f | %.1f | %s |" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
}
stats default --cpu-seconds 0 --parity-samples 0 --no-second-noise
stats train --mode train --cpu-seconds 0
stats full --mode full --cpu-seconds 0
python3 bench.py --noise fitted > $O/bench_noise_fitted.json 2>> $O/bench.err
python3 bench.py --hostile > $O/bench_hostile.json 2>> $O/bench.err
python3 bench.py --mode kf > $O/bench_kf_B65536.json 2>> $O/bench.err
python3 bench.py --mode kf --batch 4096 --seq 1000 --steps 5 > $O/bench_kf_B4096_T1000.json 2>> $O/bench.err
python3 tools/dropin_latency.py 2000 > $O/dropin_latency.json 2>> $O/bench.err
python3 bench.py --mode train > $O/bench_train.json 2>> $O/bench.err
python3 bench.py --mode full > $O/bench_full.json 2>> $O/bench.err
python3 bench.py --mode windows > $O/bench_windows.json 2>> $O/bench.err
python3 bench.py --split-bf16 > $O/bench_split_bf16.json 2>> $O/bench.err
# the reference's model shape on the opt-in split-bf16 layer kernel (gru_layer_bf16_kernel), 3 and 2 terms
python3 bench.py --split-bf16 3 --hidden 128 --layers 4 --latent 128 > $O/bench_split_bf16_hidden128layers4latent128.json 2>> $O/bench.err
python3 bench.py --split-bf16 2 --hidden 128 --layers 4 --latent 128 > $O/bench_split_bf16x2_hidden128layers4latent128.json 2>> $O/bench.err
python3 bench.py --mode mpc --steps 3 --cpu-seconds 5 > $O/bench_mpc.json 2>> $O/bench.err
python3 bench.py --mode mpc --batch 8 --seq 4000 --steps 3 --cpu-seconds 0 > $O/bench_mpc_B8_T4000.json 2>> $O/bench.err
bash tools/pmc_pass.sh ${TAG}_sq "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY" > $O/pmc_sq.txt 2>&1
bash tools/pmc_pass.sh ${TAG}_grbm "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" > $O/pmc_grbm.txt 2>&1
python3 tools/ab_fused.py 6 > $O/ab_fused.txt 2>&1

# round 4: the reference's model shapes with the repaired k-pair pipeline and gru_layer_stage_kernel (and with it switched off), PMC
# traffic and SQ counters of the layer kernels, in-kernel timestamps of the layer and fused kernels, the wave-per-trajectory layout
REF="--hidden 128 --layers 4 --latent 128 --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise"
stats ref_shape $REF
python3 bench.py $REF > $O/bench_hidden128layers4latent128.json 2>> $O/bench.err
OS_GRU_STAGE=0 python3 bench.py $REF > $O/bench_hidden128layers4latent128_nostage.json 2>> $O/bench.err
python3 bench.py --hidden 128 --layers 4 --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $O/bench_hidden128layers4.json 2>> $O/bench.err
python3 bench.py --hidden 64 --layers 4 --steps 5 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $O/bench_hidden64layers4.json 2>> $O/bench.err
bash tools/traffic_ref_shape.sh > $O/traffic_ref_shape.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_ref -- python3 $R/tools/run_ref_shape_once.py 1 > $O/pmc_ref.log 2>&1)
python3 - <<PY > $O/pmc_ref_shape.txt
import csv, glob, collections
f = glob.glob("$O/pmc_ref/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
if f:
    for r in csv.DictReader(open(f[0])):
        if "gru_layer" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
bash tools/layer_ts.sh 2>&1 | grep -v "^$" | grep "==\|cycles per step" > $O/layer_timestamps_raw.txt
bash tools/fused_ts.sh 2>&1 | grep "cycles per step" > $O/fused_timestamps_raw.txt
python3 bench.py --mode kf --batch 4096 --seq 1000 --steps 3 --warmup 1 --cpu-seconds 0 --wave-per-trajectory > $O/bench_kf_wave_B4096_T1000.json 2>> $O/bench.err
python3 bench.py --mode kf --steps 3 --warmup 1 --cpu-seconds 0 --wave-per-trajectory > $O/bench_kf_wave_B65536.json 2>> $O/bench.err
bash tools/rows_ts.sh > $O/rows2_timestamps_raw.txt 2>&1
python3 tools/rows_crossover.py 2>/dev/null | grep -v "^RCCL" > $O/rows_crossover.txt
# round 4, small batches: the layer-pipelined stack launch, the single-workgroup vector kernel (the reference's own evaluation call
# and training batch), config 5 without the stack launch
python3 tools/stack_timing.py 200 > $O/stack_timing_raw.md 2>> $O/bench.err
python3 tools/dropin_rnn_latency.py 2000 > $O/dropin_rnn_latency.json 2>> $O/bench.err
python3 tools/train_small_batch.py 200 > $O/train_small_batch.json 2>> $O/bench.err
python3 bench.py --mode train --batch 64 --steps 50 --cpu-seconds 0 > $O/bench_train_B64.json 2>> $O/bench.err
OS_GRU_STACK=0 python3 bench.py --mode full > $O/bench_full_nostack.json 2>> $O/bench.err
bash tools/vec_ts.sh 2>&1 | grep "gru_vec_kernel" | sort -u > $O/vec_timestamps_raw.txt
# round 5: the window-stream inference entry (and the materialised form beside it), its kernel table; the dense-F_d float64 filter on
# the row layout; the QP line's fp64 roofline + SQ counters; the world > 1 paths on this one GPU (gloo, host-staged collectives);
# the training step's ablations
stats windows --mode windows --cpu-seconds 0
python3 bench.py --mode windows --materialise > $O/bench_windows_materialised.json 2>> $O/bench.err
python3 tools/windows_probe.py > $O/windows_probe.txt 2>> $O/bench.err
python3 tools/baseline_probe.py > $O/dense_kf_and_mpc_probe.json 2>> $O/bench.err
python3 bench.py --mode mpc --seq 100 --steps 2 --warmup 1 --cpu-seconds 0 > $O/bench_mpc_T100.json 2>> $O/bench.err
OS_MPC_SHARDS=1 bash tools/pmc_any.sh ${TAG}_mpc "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" $R/tools/mpc_run_bench.py 65536 10 > $O/pmc_mpc.txt 2>&1      # (one part: counters per launch of the whole batch)
python3 bench.py --gpus 2 --share-gpu --steps 5 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $O/bench_share_gpu2_fused.json 2>> $O/bench.err
python3 bench.py --gpus 2 --share-gpu --mode train --steps 10 --warmup 2 --cpu-seconds 0 > $O/bench_share_gpu2_train.json 2>> $O/bench.err
for e in "" "OS_TRAIN_DBG_NOSAVE=1" "OS_DW_DBG=1"; do env $e python3 bench.py --mode train --steps 20 --cpu-seconds 0 > $O/bench_train_ablation_${e%%=*}.json 2>> $O/bench.err; done
# the opt-in split-bf16 layer kernel: whole-model times against the fp32 stage kernel, phase timestamps, counters, the bare MFMA mix
python3 tools/bf16_layer_probe.py > $O/bf16_layer_probe.txt 2>> $O/bench.err
bash tools/bf16_layer_ts.sh > $O/bf16_layer_timestamps_raw.txt 2>&1
bash tools/pmc_any.sh ${TAG}_bf16 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" $R/tools/bf16_layer_probe.py 65536 100 > $O/pmc_bf16_layer.txt 2>&1
[ -x tools/micro/mfma_bf16_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/micro/mfma_bf16_rate tools/micro/mfma_bf16_rate.hip
tools/micro/mfma_bf16_rate > $O/mfma_bf16_rate.txt 2>&1
# the small-batch kernels on four CUs per (layer, tile): against one CU per tile, phase timestamps, the batch-64 training step's kernels
python3 tools/wide_probe.py 200 2>/dev/null | grep "^|" > $O/wide_probe.md
bash tools/wide_ts.sh 2>&1 | grep "cycles per step" | sort | tail -4 > $O/wide_timestamps_raw.txt
bash tools/wide_bwd_ts.sh 2>&1 | grep "cycles per step" | sort -u | tail -4 > $O/wide_bwd_timestamps_raw.txt
stats train_B64 --mode train --batch 64 --steps 200 --warmup 20 --cpu-seconds 0
# round 6: the shards of a fixed 65,536 batch (tile shapes of the single fused kernel), the strong-scaling form of the bench line at the
# shard sizes (one rank each: what a rank of an N-GPU job runs), the reference's caller loops on the drop-in classes, the QP's counters
python3 tools/shard_sweep.py --iters 10 --out $O/shard_sweep.md > $O/shard_sweep.log 2>&1
for b in 32768 16384 8192; do python3 bench.py --scaling strong --batch $b --steps 20 --warmup 3 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $O/bench_strong_shard_B$b.json 2>> $O/bench.err; done
python3 bench.py --gpus 2 --share-gpu --scaling strong --steps 5 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $O/bench_share_gpu2_strong.json 2>> $O/bench.err
python3 tools/dropin_loops.py --out $O/dropin_loops.json > $O/dropin_loops.log 2>&1
OS_MPC_QUAD=0 python3 bench.py --mode mpc --steps 3 --cpu-seconds 0 > $O/bench_mpc_wave_per_qp.json 2>> $O/bench.err
OS_MPC_SHARDS=1 bash tools/pmc_any.sh ${TAG}_mpcq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES" $R/tools/mpc_run_bench.py 65536 10 > $O/pmc_mpc_lds.txt 2>&1
python3 tools/mpc_iter_stats.py > $O/mpc_iter_stats.txt 2>> $O/bench.err
python3 tools/stack_wait_time.py > $O/stack_wait_time.txt 2>&1
# round 6, second half: the QP with per-problem records, the filter step inside the QP launch, two concurrent parts of the batch -- the
# line in one part and with the separate filter launch, kernel tables of both, the identity check, in-kernel stamps (one wavefront
# alone, and the wavefront-per-QP solver inside the persistent kernel at the reference's shape), the B = 4,096 line
OS_MPC_SHARDS=1 python3 bench.py --mode mpc --steps 3 --cpu-seconds 0 > $O/bench_mpc_one_part.json 2>> $O/bench.err
OS_MPC_SHARDS=1 OS_MPC_FUSE_KF=0 python3 bench.py --mode mpc --steps 3 --cpu-seconds 0 > $O/bench_mpc_separate_filter.json 2>> $O/bench.err
python3 bench.py --mode mpc --batch 4096 --steps 3 --cpu-seconds 0 > $O/bench_mpc_B4096.json 2>> $O/bench.err
OS_MPC_ROWS=0 python3 bench.py --mode mpc --batch 4096 --steps 3 --cpu-seconds 0 > $O/bench_mpc_B4096_wave_per_trajectory.json 2>> $O/bench.err
# (the row-per-trajectory kernel's range: the line at 8,192 and 16,384, and the three forms side by side)
python3 bench.py --mode mpc --batch 8192 --steps 3 --cpu-seconds 0 > $O/bench_mpc_B8192.json 2>> $O/bench.err
python3 bench.py --mode mpc --batch 16384 --steps 3 --cpu-seconds 0 > $O/bench_mpc_B16384.json 2>> $O/bench.err
for b in 4096 8192 16384; do python3 tools/mpc_rows_check.py $b 100 2>/dev/null | grep -v amdgpu.ids; done > $O/mpc_rows_check.txt
OS_MPC_SHARDS=1 bash tools/kstats_any.sh ${TAG}_mpc1 tools/mpc_run_bench.py 65536 40 > /dev/null 2>&1; cp gpurun_out/${TAG}_mpc1_kernel_stats.md $O/mpc_one_part_kernel_stats.md
OS_MPC_SHARDS=1 OS_MPC_FUSE_KF=0 bash tools/kstats_any.sh ${TAG}_mpc0 tools/mpc_run_bench.py 65536 40 > /dev/null 2>&1; cp gpurun_out/${TAG}_mpc0_kernel_stats.md $O/mpc_separate_filter_kernel_stats.md
bash tools/kstats_any.sh ${TAG}_mpc2 tools/mpc_run_bench.py 65536 40 > /dev/null 2>&1; cp gpurun_out/${TAG}_mpc2_kernel_stats.md $O/mpc_two_parts_kernel_stats.md
python3 tools/mpc_fuse_check.py 65536 40 2 2>/dev/null | grep "^B=" > $O/mpc_fuse_check.txt
L=$(bash tools/ts_lib.sh qts1 liboptistate_qts.so mpc_quad -DOSQ_TS -DOSQ_X_FIXED=40 -DOSQ_OCC=1 | tail -1)
OPTISTATE_HIP_LIB=$L python3 tools/quad_one_wave.py 64 2>&1 | grep "cycles per" | tail -1 > $O/quad_timestamps_raw.txt
L=$(bash tools/ts_lib.sh mts liboptistate_mts.so mpc_kernels -DOSM_TS | tail -1)
OPTISTATE_HIP_LIB=$L python3 bench.py --mode mpc --batch 8 --seq 4000 --steps 1 --warmup 1 --cpu-seconds 0 2>&1 | grep "persistent kernel, cycles" | tail -1 > $O/mpc_persistent_timestamps_raw.txt
ls $O
