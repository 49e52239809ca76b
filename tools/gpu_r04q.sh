#!/bin/bash
# round 4, visit q: the layer-pipelined stack kernel -- tests (under a timeout), timing table, config 5
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_gpu_gru.py -x -q -m gpu > $O/r04q_gru_tests.log 2>&1; echo "gru tests rc=$?"
tail -5 $O/r04q_gru_tests.log
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/r04q_all_tests.log 2>&1; echo "all gpu tests rc=$?"; tail -3 $O/r04q_all_tests.log
OS_GRU_STACK=0 timeout 300 python3 bench.py --mode full --cpu-seconds 0 > $O/r04q_full_nostack.json 2>/dev/null
timeout 300 python3 bench.py --mode full --cpu-seconds 0 > $O/r04q_full_stack.json 2>/dev/null
python3 - <<PY
import json
for n in ("r04q_full_nostack", "r04q_full_stack"):
    try:
        j = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, j["value"], j["ms_per_step"], j.get("parity", {}).get("ok"), {k: v for k, v in j.get("phases_ms", {}).items()} if "phases_ms" in j else "")
    except Exception as ex:
        print(n, "failed", ex)
PY
