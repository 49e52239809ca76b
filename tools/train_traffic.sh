cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03y
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc_any.sh train_$c "$c" $GRAFT_REPO_ROOT/bench.py --mode train --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/r03y/train_$c.txt 2>&1
done
grep -A2 "dw3_kernel\|bwd_sweep\|gru_layer_ahead" gpurun_out/r03y/train_FETCH_SIZE.txt | head -30; grep -A2 "dw3_kernel\|bwd_sweep\|gru_layer_ahead" gpurun_out/r03y/train_WRITE_SIZE.txt | head -30
