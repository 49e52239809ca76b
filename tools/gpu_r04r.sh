#!/bin/bash
# round 4, visit r: gru_vec_kernel -- tests, drop-in RNN latency
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_gru.py -x -q -m gpu > $O/r04r_gru_tests.log 2>&1; echo "gru tests rc=$?"
tail -5 $O/r04r_gru_tests.log
timeout 300 python3 tools/dropin_rnn_latency.py 2000 > $O/r04r_dropin_rnn_latency.json 2>$O/r04r_lat.err; cat $O/r04r_dropin_rnn_latency.json; tail -3 $O/r04r_lat.err
