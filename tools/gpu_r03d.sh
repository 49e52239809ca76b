cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
python tools/debug_hostile.py fitted > $O/dbg_new.txt 2>&1
OPTISTATE_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/liboptistate_old.so python tools/debug_hostile.py fitted > $O/dbg_old.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kf.py tests/test_gpu_advice.py tests/test_gpu_bench_contract.py -m gpu -q -k "config2 or rows or 4096 or small or contract or dropin or hostile or sym_lane or g4 or g3" 2>&1 | tail -15 > $O/pytest.log
timeout 600 python bench.py --mode kf --batch 4096 --seq 1000 --steps 5 --no-second-noise > $O/bench_kf_4096.json 2>> $O/bench.err
OS_KF_ROWS_V1=1 timeout 600 python bench.py --mode kf --batch 4096 --seq 1000 --steps 5 --no-second-noise --cpu-seconds 0 --parity-samples 0 > $O/bench_kf_4096_v1.json 2>> $O/bench.err
cat $O/dbg_new.txt | head -40; cat $O/dbg_old.txt | head -12; tail -8 $O/pytest.log; head -c 300 $O/bench_kf_4096.json; echo; head -c 300 $O/bench_kf_4096_v1.json
