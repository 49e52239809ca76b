#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mpc_small -- python3 $R/tools/mpc_run_bench.py 64 1000 > $OUT/prof_mpc_small.log 2>&1
tail -1 $OUT/prof_mpc_small.log
find $OUT/prof_mpc_small -name "*kernel_stats.csv" | head -1 | xargs -I{} head -6 {} | cut -c1-150
