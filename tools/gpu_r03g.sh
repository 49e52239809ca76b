cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.log
timeout 600 python bench.py --mode kf --no-second-noise > $O/bench_kf.json 2> $O/bench.err
OS_KF_SYM_PRE=0 timeout 600 python bench.py --mode kf --no-second-noise --cpu-seconds 0 --parity-samples 0 > $O/bench_kf_nopre.json 2>> $O/bench.err
timeout 600 python bench.py --mode kf --batch 131072 --no-second-noise --cpu-seconds 0 --parity-samples 0 > $O/bench_kf_131072.json 2>> $O/bench.err
tail -4 $O/pytest.log; for f in $O/bench_kf*.json; do python3 -c "
import json; d=json.load(open('$f')); print('$f', '%.4g'%d['value'], 'frac %.3f'%d['roofline']['frac'], 'ms %.4f'%d['roofline']['avg_launch_ms'], d.get('parity',{}).get('state_linf'))"; done
