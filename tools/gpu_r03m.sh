cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_bf16.py tests/test_gpu_pipeline.py tests/test_gpu_advice.py -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 | cut -c1-250 > $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2>> $O/bench.err
timeout 600 python bench.py --split-bf16 --cpu-seconds 0 --parity-samples 4096 > $O/bench_bf16.json 2>> $O/bench.err
bash tools/pmc_pass.sh r03m_sq "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY" > $O/pmc_sq.txt 2>&1
bash tools/fused_ts.sh 2>&1 | grep "fused_v2 cycles" > $O/fused_ts.txt
cat $O/pytest.log; for f in $O/bench.json $O/bench_bf16.json; do python3 -c "
import json; d=json.load(open('$f')); print('$f', '%.4g'%d['value'], 'ms %.4f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], d['roofline']['avg_launch_ms'], (d.get('parity') or {}).get('state_linf'), (d.get('parity') or {}).get('gru_linf'))"; done; tail -20 $O/pmc_sq.txt | cut -c1-100; cat $O/fused_ts.txt
