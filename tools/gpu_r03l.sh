cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k split 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 | cut -c1-220 > $O/pytest.log
timeout 600 python bench.py --mode train --cpu-seconds 0 --force-dist > $O/bench_train_forcedist.json 2>> $O/bench.err
timeout 600 python bench.py --mode train --cpu-seconds 0 --force-dist --no-split-allreduce > $O/bench_train_forcedist_nosplit.json 2>> $O/bench.err
timeout 600 python bench.py --mode train --cpu-seconds 0 > $O/bench_train.json 2>> $O/bench.err
cat $O/pytest.log; for f in $O/bench_train*.json; do wc -l $f; python3 -c "
import json; d=json.load(open('$f')); print('$f', '%.4g'%d['value'], 'ms %.4f'%d['ms_per_step'], d.get('allreduce_us'), d.get('allreduce'))"; done
