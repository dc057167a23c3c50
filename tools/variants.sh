#!/bin/bash
# Build experiment variants of the back end (extra -D flags on k_basepass_as.hip only) into
# toyrenderer_amd/lib/exp/<name>/libtrhip.so, and (on the GPU box) time each with bench.py.
#   bash tools/variants.sh build name1 "-DFOO -DBAR" name2 "-DBAZ" ...
#   bash tools/variants.sh run name1 name2 ... -- [bench args]
# Experiment builds are never shipped: lib/exp is git-ignored and deleted after use.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/toyrenderer_amd/csrc
LIB=$ROOT/toyrenderer_amd/lib
MODE=$1; shift
if [ "$MODE" = build ]; then
  while [ -n "$1" ]; do
    name=$1; defs=$2; shift 2
    mkdir -p $LIB/exp/$name
    F=${VFILE:-k_basepass_as}          # VFILE=k_gpuculling: the flags go to the instance-pass kernels instead
    /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function --offload-arch=gfx950 $defs -c $CS/$F.hip -o $LIB/exp/$name/$F.o
    OBJS=""
    for o in trhip_core k_gpuculling k_basepass_as k_hzb k_updateinstance k_raster k_giprobe; do
      if [ $o = $F ]; then OBJS="$OBJS $LIB/exp/$name/$F.o"; else OBJS="$OBJS $LIB/obj/$o.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB/exp/$name/libtrhip.so $OBJS
    rm $LIB/exp/$name/$F.o
    echo built $name "($defs)"
  done
else
  NAMES=()
  while [ "$1" != "--" ] && [ -n "$1" ]; do NAMES+=("$1"); shift; done
  shift || true
  for n in "${NAMES[@]}"; do
    if [ "$n" = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
    python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$n', d['value'], d['ms_per_step'], r['avg_launch_ms'] if r else None)"
  done
fi
