cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r03b/pytest_gpu.log
timeout 300 python tools/dropin_latency.py 2000 > gpurun_out/r03b/dropin_latency.json 2> gpurun_out/r03b/dropin_latency.err
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03b/prof_dropin -- python3 $GRAFT_REPO_ROOT/tools/dropin_latency.py 300 > $GRAFT_REPO_ROOT/gpurun_out/r03b/prof_dropin.log 2>&1)
find gpurun_out/r03b/prof_dropin -name "*kernel_stats.csv" | head -1 | xargs -I{} head -5 {} > gpurun_out/r03b/dropin_kernel_stats.csv
find gpurun_out/r03b/prof_dropin -type f ! -name "*kernel_stats.csv" -delete
tail -12 gpurun_out/r03b/pytest_gpu.log; cat gpurun_out/r03b/dropin_latency.json; cat gpurun_out/r03b/dropin_kernel_stats.csv
