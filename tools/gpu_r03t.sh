cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6 | cut -c1-250 > $O/pytest.log
timeout 600 python bench.py --mode train --steps 20 --cpu-seconds 0 > $O/bench_train.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o train -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 20 --cpu-seconds 0 > /dev/null 2>> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
cat $O/pytest.log; python3 -c "
import json,glob,csv
d=json.load(open('$O/bench_train.json')); print('train', '%.4g'%d['value'], 'ms %.4f'%d['ms_per_step'])
f=glob.glob('$O/prof/**/*kernel_stats.csv', recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:6]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
"
