#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/final_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/final_tests.log | tail -2
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python3 bench.py > $O/final_bench.json 2>$O/final_bench.err; tail -c 1500 $O/final_bench.json
