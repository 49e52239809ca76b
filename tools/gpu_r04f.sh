#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
pick() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: (round(v["ms_per_launch"], 4), v["launches_per_step"], v["kernel"]) for n, v in d.get("kernels", {}).items()}
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), k)
PY
}
for ah in 1 0; do
  OS_GRU_AHEAD=$ah python bench.py --mode train --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_train_ah$ah.json 2>/dev/null; pick $OUT/bench_train_ah$ah.json
  OS_GRU_AHEAD=$ah python bench.py --mode windows --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/bench_windows_ah$ah.json 2>/dev/null; pick $OUT/bench_windows_ah$ah.json
done
