#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Usage: tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu_$TAG.log
python bench.py --steps 10 --warmup 2 2> $OUT/bench_$TAG.err | tee $OUT/bench_$TAG.json
tail -3 $OUT/bench_$TAG.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 > $OUT/prof_$TAG.log 2>&1
tail -2 $OUT/prof_$TAG.log
find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
