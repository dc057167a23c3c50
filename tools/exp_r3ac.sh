#!/bin/bash
# List build: size of a "super" (batches per count workgroup): 256 (shipped) / 128 / 64 / 32 -- a faster count lets the expansion
# start earlier, beside the HZB build instead of beside the late instance pass.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
LIB=$R/toyrenderer_amd/lib
for rep in 1 2; do
for n in base sup7 sup6 sup5; do
  if [ $n = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['per_kernel_ms']
g=lambda s: [v for n,v in k.items() if n.endswith(s)]
print('$n: frame', d['ms_per_step'], 'cull', r['avg_launch_ms'], 'count', g('#count'), 'expand', g('#expand'), 'fused', g('#fused'), 'depth_tile', g('#depth_tile'), d['lists_digest'][:8])"
done
done
