#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -m pytest tests/test_gpu_kf.py -m gpu -q -x 2>&1 | tail -4
pick() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: (round(v["ms_per_launch"], 4), v["kernel"]) for n, v in d.get("kernels", {}).items()}
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.4f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), k, d.get("parity", {}).get("state_linf"))
PY
}
python bench.py --mode kf --batch 4096 --seq 1000 --steps 3 --warmup 1 --cpu-seconds 0 --wave-per-trajectory > $OUT/bench_kf_wave_B4096.json 2>$OUT/bench_kf_wave.err; pick $OUT/bench_kf_wave_B4096.json
python bench.py --mode kf --batch 4096 --seq 1000 --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/bench_kf_rows_B4096.json 2>/dev/null; pick $OUT/bench_kf_rows_B4096.json
python bench.py --mode kf --batch 65536 --seq 100 --steps 3 --warmup 1 --cpu-seconds 0 --wave-per-trajectory > $OUT/bench_kf_wave_B65536.json 2>>$OUT/bench_kf_wave.err; pick $OUT/bench_kf_wave_B65536.json
python bench.py --mode kf --batch 65536 --seq 100 --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/bench_kf_sym_B65536.json 2>/dev/null; pick $OUT/bench_kf_sym_B65536.json
tail -3 $OUT/bench_kf_wave.err
