#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_gru.py tests/test_gpu_vit.py tests/test_gpu_advice.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python3 tools/dropin_rnn_latency.py 2000 > $O/r04t_dropin_rnn_latency.json 2>/dev/null; cat $O/r04t_dropin_rnn_latency.json
timeout 600 bash tools/vec_ts.sh 2>&1 | grep "gru_vec_kernel" | tail -2 > $O/r04t_vec_ts.txt; cat $O/r04t_vec_ts.txt
