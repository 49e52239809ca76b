cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
for i in 1 2; do
  for v in prev varA varB; do
    OPTISTATE_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/liboptistate_$v.so timeout 300 python bench.py --steps 10 --no-second-noise --cpu-seconds 0 --parity-samples 0 > $O/${v}_$i.json 2>> $O/bench.err
  done
  timeout 300 python bench.py --steps 10 --no-second-noise --cpu-seconds 0 --parity-samples 0 > $O/cur_$i.json 2>> $O/bench.err
done
python3 -c "
import json
for i in (1,2):
  for f in ('prev','varA','varB','cur'):
    d=json.load(open('$O/%s_%d.json'%(f,i))); print(f, i, '%.4g'%d['value'], 'kernel ms %.4f'%d['roofline']['avg_launch_ms'], 'frac %.3f'%d['roofline']['frac'])"
