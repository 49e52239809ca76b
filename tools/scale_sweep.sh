#!/bin/bash
# One call on an N-GPU node: bench.py --gpus {1,2,4,8} for the fused inference path (weak scaling, no data-path collective) and
# the data-parallel training step (one flat fp32 bucket all-reduce per step over RCCL), tabulated with the efficiency the
# driver would compute from the same lines.  usage: bash tools/scale_sweep.sh [max_gpus] [out_dir]
# (bench.py --gpus N starts its own ranks: torch.distributed.run on 127.0.0.1, one rank per GPU.)
R=${GRAFT_REPO_ROOT:-/root/repo}
MAXG=${1:-$(python3 -c "import torch; print(torch.cuda.device_count())")}
O=${2:-$R/gpurun_out/scale}
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
for mode in fused train; do
  for n in 1 2 4 8; do
    [ $n -le $MAXG ] || continue
    extra=""; [ $mode = train ] && extra="--mode train"
    timeout 900 python3 bench.py --gpus $n $extra --steps 10 --warmup 3 --cpu-seconds 0 --parity-samples 0 --no-second-noise \
        > $O/${mode}_n$n.json 2> $O/${mode}_n$n.err || echo "bench $mode n=$n failed (rc $?)" >&2
  done
done
# the latency-bound 1.69 MB gradient bucket under each RCCL protocol, at the largest world size
for proto in LL LL128 Simple; do
  n=$MAXG; [ $n -ge 2 ] || continue
  timeout 900 python3 bench.py --gpus $n --mode train --rccl-proto $proto --steps 10 --warmup 3 --cpu-seconds 0 \
      > $O/train_n${n}_$proto.json 2> $O/train_n${n}_$proto.err || echo "bench train n=$n proto=$proto failed (rc $?)" >&2
done
python3 - "$O" <<'PY'
import glob, json, os, sys
o = sys.argv[1]
print("| mode | GPUs | value | unit | ms/step | efficiency vs N x 1-GPU | all-reduce us | RCCL world |")
print("|---|---|---|---|---|---|---|---|")
for mode in ("fused", "train"):
    base = None
    for n in (1, 2, 4, 8):
        f = os.path.join(o, f"{mode}_n{n}.json")
        if not os.path.exists(f) or os.path.getsize(f) == 0:
            continue
        d = json.loads(open(f).read().strip().splitlines()[-1])
        base = base or d["value"] / d["n_gpus"]
        eff = d["value"] / (d["n_gpus"] * base)
        ar = d.get("allreduce_us")
        print(f"| {mode} | {d['n_gpus']} | {d['value']:.4g} | {d['unit']} | {d['ms_per_step']:.3f} | {eff:.3f} | "
              f"{'-' if ar is None else '%.1f' % ar} | {d.get('rccl_world_size')} |")
print()
print("| train, RCCL protocol | GPUs | ms/step | all-reduce us | replicas identical |")
print("|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(o, "train_n*_*.json"))):
    if os.path.getsize(f) == 0:
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ar = d.get("allreduce_us")
    print(f"| {d.get('rccl', {}).get('NCCL_PROTO')} | {d['n_gpus']} | {d['ms_per_step']:.3f} | {'-' if ar is None else '%.1f' % ar} | "
          f"{d.get('parity', {}).get('ok')} |")
PY
