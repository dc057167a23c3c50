#!/bin/bash
# The early list build's expansion (side stream, 122 MB of stores) next to the late phase: throttled by its grid size.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for v in 0 1 2 4; do
  export TRHIP_EXPAND_BLOCKS_PER_CU=$v
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['per_kernel_ms']
print('expand blocks/CU $v: frame', d['ms_per_step'], 'cull', r['avg_launch_ms'], 'fused<1>', k.get('gpuculling_CS_GPUCulling LATE_CULL=1#fused'), 'expand', k.get('basepass_AS_Main LATE_CULL=0#expand'), 'depth_tile', k.get('ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=1#depth_tile'))"
done
done
