#!/bin/bash
# Development build of the whole library with extra -D flags on some sources (the in-kernel timestamp variants): every csrc/*.hip
# goes into build_ab/<dir> -- the shipped object of `python -m optistate_amd.build` (csrc/build/) where there is one, the flagged
# sources always recompiled with the extra flags -- and is linked as <libname>.
# usage: bash tools/ts_lib.sh <dir> <libname.so> <source[,source...] without .hip> <flags...>     (prints the library's path)
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/$1; LIB=$2; SRC=",$3,"; shift 3
mkdir -p $D
cd $R/optistate_amd/csrc
for f in $(ls *.hip | sed 's/\.hip$//'); do
  X=; [ $f = kf_rows_kernel ] && X="-fno-slp-vectorize"; [ $f = mpc_quad ] && X="-mllvm -disable-machine-licm"       # (optistate_amd/build.py EXTRA_FLAGS)
  case $SRC in *,$f,*) X="$X $*"; rm -f $D/$f.o;; esac
  # (an unflagged source: always the shipped object -- a copy left in build_ab/ by an earlier build of the library would link a stale kernel)
  [ -z "$X" -o "$X" = "-fno-slp-vectorize" -o "$X" = "-mllvm -disable-machine-licm" ] && [ -f build/$f.o ] && cp build/$f.o $D/$f.o
  [ -f $D/$f.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $X -DOS_BUILD_ID='"ts-build"' -c $f.hip -o $D/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/$LIB $D/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib && echo $D/$LIB
