#!/bin/bash
for v in "" $@; do
  if [ -n "$v" ]; then export OPTISTATE_HIP_LIB=$GRAFT_REPO_ROOT/optistate_amd/lib/exp/lib_$v.so; else unset OPTISTATE_HIP_LIB; fi
  echo "variant=${v:-default} $(python tools/quick_bench.py --iters 5 2>&1 | grep 'two_kernel=False')"
done
