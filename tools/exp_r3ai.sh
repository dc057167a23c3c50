#!/bin/bash
# Transform update through LDS: instances per workgroup (64 / 128 / 256 / 512).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
LIB=$R/toyrenderer_amd/lib
for rep in 1 2; do
for n in base upd64 upd128 upd512; do
  if [ $n = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  python bench.py --config C3 --animate --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n', d['value'], d['ms_per_step'], [v for k,v in r['per_kernel_ms'].items() if 'updateinstance' in k], d['lists_digest'][:8])"
done
done
