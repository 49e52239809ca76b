#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/r04w_tests.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/r04w_tests.log | tail -1
timeout 300 python3 tools/train_small_batch.py 200 | tee $O/r04w_tsb_auto.json
OS_TRAIN_OVERLAP=0 timeout 300 python3 tools/train_small_batch.py 200 | tee $O/r04w_tsb_off.json
timeout 300 python3 bench.py --mode train --cpu-seconds 0 --steps 20 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train 8192', j['ms_per_step'], j['roofline']['frac'])"
