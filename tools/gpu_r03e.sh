cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kf.py tests/test_gpu_advice.py tests/test_gpu_bench_contract.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -15 > $O/pytest.log
timeout 600 python bench.py --mode kf --batch 4096 --seq 1000 --steps 5 --no-second-noise > $O/bench_kf_4096.json 2>> $O/bench.err
OS_KF_ROWS_V1=1 timeout 600 python bench.py --mode kf --batch 4096 --seq 1000 --steps 5 --no-second-noise --cpu-seconds 0 --parity-samples 0 > $O/bench_kf_4096_v1.json 2>> $O/bench.err
timeout 600 python bench.py --mode kf --batch 8192 --seq 200 --steps 5 --no-second-noise --cpu-seconds 0 > $O/bench_kf_8192.json 2>> $O/bench.err
tail -8 $O/pytest.log; head -c 300 $O/bench_kf_4096.json; echo; head -c 300 $O/bench_kf_4096_v1.json; echo; head -c 300 $O/bench_kf_8192.json
