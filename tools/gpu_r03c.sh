cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kf.py tests/test_gpu_fullsize.py tests/test_gpu_advice.py tests/test_gpu_bf16.py tests/test_gpu_pipeline.py tests/test_gpu_errors.py -m gpu -q 2>&1 | tail -30 > $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --mode kf --no-second-noise > $O/bench_kf.json 2>> $O/bench.err
timeout 300 python tools/dropin_latency.py 2000 > $O/dropin_latency.json 2>> $O/bench.err
bash tools/pmc_pass.sh r03c_sq "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY" > $O/pmc_sq.txt 2>&1
tail -8 $O/pytest_gpu.log; head -c 400 $O/bench.json; echo; cat $O/pmc_sq.txt | tail -25; cat $O/dropin_latency.json
