cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
bash tools/sym_ts.sh > $O/sym_ts.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kf.py -m gpu -q 2>&1 | tail -6 > $O/pytest.log
timeout 600 python bench.py --mode train --cpu-seconds 0 > $O/bench_train.json 2>> $O/bench.err
timeout 600 python bench.py --mode train --cpu-seconds 0 --force-dist > $O/bench_train_forcedist.json 2>> $O/bench.err
timeout 600 python bench.py --mode kf --no-second-noise --cpu-seconds 0 --parity-samples 0 > $O/bench_kf.json 2>> $O/bench.err
cat $O/sym_ts.txt | tail -3; tail -4 $O/pytest.log | cut -c1-200; for f in $O/bench_train.json $O/bench_train_forcedist.json $O/bench_kf.json; do python3 -c "
import json; d=json.load(open('$f')); print('$f', '%.4g'%d['value'], 'ms %.4f'%d['ms_per_step'], d.get('allreduce_us'), d.get('allreduce'))"; done; tail -3 $O/bench.err | cut -c1-300
