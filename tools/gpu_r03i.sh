cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_advice.py tests/test_gpu_vit.py tests/test_gpu_example.py -m gpu -q 2>&1 | tail -6 > $O/pytest.log
timeout 900 python bench.py --hidden 128 --layers 4 --latent 128 --steps 5 --cpu-seconds 0 --parity-samples 4096 --no-second-noise > $O/bench_h128l4lat128.json 2>> $O/bench.err
bash tools/traffic_ref_shape.sh > $O/traffic_ref.txt 2>&1
tail -4 $O/pytest.log; python3 -c "
import json; d=json.load(open('$O/bench_h128l4lat128.json')); print('%.4g'%d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['gru_linf'], {k:(v['ms_per_launch'],v['launches_per_step']) for k,v in d['kernels'].items()})"; tail -8 $O/traffic_ref.txt
