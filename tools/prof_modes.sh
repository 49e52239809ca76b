#!/bin/bash
# rocprofv3 kernel-trace summaries of the non-default bench modes.  Usage: tools/prof_modes.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in mpc kf; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$mode -- python3 $R/bench.py --mode $mode --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/prof_${TAG}_$mode.log 2>&1
  tail -1 $OUT/prof_${TAG}_$mode.log | cut -c1-300
  find $OUT/prof_${TAG}_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {}
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_gru -- python3 $R/tools/run_gru_once.py 3 128 4 > $OUT/prof_${TAG}_gru.log 2>&1
find $OUT/prof_${TAG}_gru -name "*kernel_stats.csv" | head -1 | xargs -I{} head -6 {}
