#!/bin/bash
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_full -- python3 $R/bench.py --mode full --steps 5 --warmup 1 > $OUT/prof_${TAG}_full.log 2>&1
grep -h '^{"metric"' $OUT/prof_${TAG}_full.log | cut -c1-200
find $OUT/prof_${TAG}_full -name "*kernel_stats.csv" | head -1 | xargs -I{} head -16 {} | cut -c1-150
