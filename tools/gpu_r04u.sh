#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python3 -m pytest tests/test_gpu_train.py tests/test_gpu_gru.py -x -q -m gpu > $O/r04u_tests.log 2>&1; echo "rc=$?"; tail -3 $O/r04u_tests.log
timeout 300 python3 bench.py --mode train --batch 64 --cpu-seconds 0 --steps 50 > $O/r04u_bench_train_b64.json 2>/dev/null; python3 -c "
import json; j=json.loads(open('$O/r04u_bench_train_b64.json').read().strip().splitlines()[-1]); print('train 64', j['ms_per_step'], j.get('phases_ms') or j.get('kernels') or list(j.keys()))"
