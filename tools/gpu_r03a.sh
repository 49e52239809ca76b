cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r03a/pytest_gpu.log
timeout 300 python tools/dropin_latency.py 2000 > gpurun_out/r03a/dropin_latency.json 2> gpurun_out/r03a/dropin_latency.err
timeout 600 python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
timeout 600 python bench.py --noise fitted > gpurun_out/r03a/bench_fitted.json 2>> gpurun_out/r03a/bench.err
timeout 600 python bench.py --hostile > gpurun_out/r03a/bench_hostile.json 2>> gpurun_out/r03a/bench.err
timeout 600 python bench.py --mode kf > gpurun_out/r03a/bench_kf.json 2>> gpurun_out/r03a/bench.err
timeout 600 python bench.py --mode kf --batch 4096 --seq 1000 --steps 5 > gpurun_out/r03a/bench_kf_4096.json 2>> gpurun_out/r03a/bench.err
tail -5 gpurun_out/r03a/pytest_gpu.log; cat gpurun_out/r03a/dropin_latency.json; head -c 600 gpurun_out/r03a/bench.json
