cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/pytest_gpu_r03.log
bash tools/collect_profiles.sh r03 > gpurun_out/collect_r03.log 2>&1
tail -6 gpurun_out/pytest_gpu_r03.log | cut -c1-200; ls gpurun_out/r03 | head -50
