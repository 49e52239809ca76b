cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_fullsize.py tests/test_gpu_advice.py tests/test_gpu_train.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -8 > $O/pytest.log
timeout 600 python bench.py --mode kf --no-second-noise --cpu-seconds 0 > $O/bench_kf.json 2>> $O/bench.err
OS_KF_SYM_PRE=0 timeout 600 python bench.py --mode kf --no-second-noise --cpu-seconds 0 --parity-samples 0 > $O/bench_kf_nopre.json 2>> $O/bench.err
bash tools/sym_ts.sh > $O/sym_ts.txt 2>&1
timeout 600 python bench.py --mode train --cpu-seconds 0 --force-dist > $O/bench_train_forcedist.json 2>> $O/bench.err
tail -5 $O/pytest.log | cut -c1-300; for f in $O/bench_kf.json $O/bench_kf_nopre.json $O/bench_train_forcedist.json; do python3 -c "
import json; d=json.load(open('$f')); print('$f', '%.4g'%d['value'], 'ms %.4f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], (d.get('parity') or {}).get('state_linf'), d.get('allreduce_us'), d.get('allreduce'))"; done; tail -2 $O/sym_ts.txt; grep -v amdgpu.ids $O/bench.err | tail -5 | cut -c1-300
