#!/bin/bash
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_train -- python3 $R/bench.py --mode train --steps 5 --warmup 1 > $OUT/prof_${TAG}_train.log 2>&1
grep -h '^{"metric"' $OUT/prof_${TAG}_train.log | cut -c1-250
find $OUT/prof_${TAG}_train -name "*kernel_stats.csv" | head -1 | xargs -I{} head -14 {} | cut -c1-160
