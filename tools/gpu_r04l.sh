#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -m pytest tests/test_gpu_gru.py tests/test_gpu_train.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -4
pick() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: (round(v["ms_per_launch"], 4), v["launches_per_step"], v["kernel"]) for n, v in d.get("kernels", {}).items()}
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), k, d.get("parity", {}).get("ok"))
PY
}
python bench.py --mode train --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_train_r04l.json 2>/dev/null; pick $OUT/bench_train_r04l.json
python bench.py --mode windows --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/bench_windows_r04l.json 2>/dev/null; pick $OUT/bench_windows_r04l.json
python bench.py --hidden 128 --layers 4 --latent 128 --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $OUT/bench_ref_r04l.json 2>/dev/null; pick $OUT/bench_ref_r04l.json
python bench.py --hidden 64 --layers 4 --steps 5 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $OUT/bench_h64l4_r04l.json 2>/dev/null; pick $OUT/bench_h64l4_r04l.json
