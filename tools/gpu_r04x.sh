#!/bin/bash
# stress: the progress-counter kernels (gru_stack_kernel, bwd_sweep_stack_kernel) while ANOTHER process keeps every CU busy with the
# fused kernel: no deadlock whatever is resident (bounded waits; in-order dispatch), and the numbers stay right
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 240 python3 bench.py --steps 6000 --warmup 2 --cpu-seconds 0 --parity-samples 0 --no-second-noise > $O/r04x_hog.json 2>/dev/null &
HOG=$!
sleep 20
for i in 1 2 3; do
  timeout 200 python3 -m pytest tests/test_gpu_gru.py tests/test_gpu_train.py -x -q -m gpu -k "pipelined_stack or stacked_backward or boundaries" 2>&1 | tail -1
done
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
echo "hog done"
