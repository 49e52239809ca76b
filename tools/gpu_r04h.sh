#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
pick() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: round(v["ms_per_launch"], 4) for n, v in d.get("kernels", {}).items()}
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], k, "loss", d.get("final_loss"))
PY
}
for cfg in "0 512" "1 512" "2 512" "4 512" "6 512" "0 256" "2 256" "0 1024" "2 1024" "0 384" "2 640"; do
  set -- $cfg
  OS_DW_DBG=$1 OS_DW_RPS=$2 python bench.py --mode train --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_train_dbg.json 2>/dev/null; pick $OUT/bench_train_dbg.json "dbg=$1 rps=$2"
done
