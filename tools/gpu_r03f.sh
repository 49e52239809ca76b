cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
bash tools/rows_ts.sh > $O/rows_ts.txt 2>&1
./tools/micro/mfma_shadow > $O/mfma_shadow.txt 2>&1
for cfg in "--hidden 128 --layers 4" "--hidden 64 --layers 4" "--hidden 128 --layers 4 --latent 128"; do
  n=$(echo $cfg | tr -d ' -'); 
  timeout 900 python bench.py $cfg --steps 5 --cpu-seconds 0 --parity-samples 4096 --no-second-noise > $O/bench_$n.json 2>> $O/bench.err
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$n -- python3 $GRAFT_REPO_ROOT/bench.py $cfg --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 0 --no-second-noise > $GRAFT_REPO_ROOT/$O/prof_$n.log 2>&1)
  find $O/prof_$n -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {} > $O/kernel_stats_$n.csv
  rm -rf $O/prof_$n
done
cat $O/rows_ts.txt | tail -12; cat $O/mfma_shadow.txt; for f in $O/bench_*.json; do head -c 250 $f; echo; done; cat $O/kernel_stats_*.csv
