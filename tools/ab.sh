#!/bin/bash
# A/B of back-end builds on the GPU box: bench numbers per variant (names under toyrenderer_amd/lib/exp/, or "base").
#   bash tools/ab.sh [-e VAR=value] name1 name2 ... -- [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$R/toyrenderer_amd/lib
NAMES=()
while [ "$1" != "--" ] && [ -n "$1" ]; do NAMES+=("$1"); shift; done
shift || true
for n in "${NAMES[@]}"; do
  if [ "$n" = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  python3 $R/bench.py --steps ${AB_STEPS:-50} --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$n', '$TRHIP_AS_BLOCKS_PER_CU', 'Gm/s', d['value'], 'frame ms', d['ms_per_step'], 'cull ms', r['avg_launch_ms'], 'frac', r['frac'])"
done
