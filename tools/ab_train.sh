#!/bin/bash
for v in "" NO_PW NO_MFMA; do
  if [ -n "$v" ]; then export OPTISTATE_HIP_LIB=$GRAFT_REPO_ROOT/optistate_amd/lib/exp/lib_$v.so; else unset OPTISTATE_HIP_LIB; fi
  echo "variant=${v:-full}"
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 > /dev/null 2>&1
  grep "bwd_sweep" $(find /tmp/ab_$v -name "*kernel_stats.csv") | cut -d, -f1-4 | cut -c1-120
done
