cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
timeout 120 python3 tools/debug_rows.py 6 4 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tail -3 > $O/debug_rows.txt
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6 | cut -c1-250 > $O/pytest.log
timeout 600 python bench.py --mode kf --batch 4096 --seq 1000 --steps 5 --no-second-noise --cpu-seconds 0 > $O/bench_kf_4096.json 2>> $O/bench.err
bash tools/rows_ts.sh > $O/rows_ts.txt 2>&1
cat $O/debug_rows.txt $O/pytest.log; python3 -c "
import json; d=json.load(open('$O/bench_kf_4096.json')); print('%.4g'%d['value'], 'ms %.4f'%d['ms_per_step'], d['parity']['state_linf'])"; tail -8 $O/rows_ts.txt
