cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 | cut -c1-250 > $O/pytest.log
timeout 600 python bench.py --steps 10 --no-second-noise --cpu-seconds 0 > $O/bench_default.json 2>> $O/bench.err
timeout 600 python bench.py --mode kf --steps 10 --no-second-noise --cpu-seconds 0 > $O/bench_kf.json 2>> $O/bench.err
cat $O/pytest.log; python3 -c "
import json
for f in ('bench_default','bench_kf'):
    d=json.load(open('$O/%s.json'%f)); print(f, '%.4g'%d['value'], 'ms %.4f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], d['parity']['state_linf'])"
