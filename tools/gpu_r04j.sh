#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -m pytest tests/test_gpu_gru.py -m gpu -q -x 2>&1 | tail -5
pick() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: (round(v["ms_per_launch"], 4), v["launches_per_step"], v["kernel"]) for n, v in d.get("kernels", {}).items()}
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), k, d.get("parity", {}).get("ok"))
PY
}
python bench.py --hidden 64 --layers 4 --steps 5 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $OUT/bench_h64l4_r04j.json 2>/dev/null; pick $OUT/bench_h64l4_r04j.json
OS_GRU_STAGE=0 python bench.py --hidden 64 --layers 4 --steps 5 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $OUT/bench_h64l4_nostage_r04j.json 2>/dev/null; pick $OUT/bench_h64l4_nostage_r04j.json
python bench.py --hidden 128 --layers 4 --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $OUT/bench_h128l4_r04j.json 2>/dev/null; pick $OUT/bench_h128l4_r04j.json
bash tools/fused_ts.sh 2>&1 | grep "cycles per step" | tee $OUT/fused_ts_r04j.log
