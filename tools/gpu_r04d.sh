#!/bin/bash
# round 4, visit d: tests touched by the status bit / K_gain / provenance work, the traffic passes (with provenance), bench lines
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_kf.py tests/test_gpu_errors.py tests/test_gpu_bench_contract.py tests/test_gpu_mpc.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_r04d.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -s -k "gimbal" 2>&1 | tail -6 | tee -a $OUT/pytest_r04d.log
bash tools/traffic_pass.sh > $OUT/traffic_pass_r04d.log 2>&1; tail -8 $OUT/traffic_pass_r04d.log
python3 tools/traffic_to_json.py > /dev/null && cp profiles/traffic.json $OUT/traffic_r04d.json
python bench.py --steps 10 --warmup 2 > $OUT/bench_r04d.json 2> $OUT/bench_r04d.err; python - <<PY
import json
d = json.loads(open("$OUT/bench_r04d.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
PY
python bench.py --mode kf --batch 4096 --seq 1000 --steps 5 --warmup 1 --cpu-seconds 0 > $OUT/bench_kf_r04d.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$OUT/bench_kf_r04d.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_source"))
PY
