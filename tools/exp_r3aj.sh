#!/bin/bash
# Run-to-run spread of the cull kernel on ONE box: six fresh processes, 200 timed frames each (events, no profiler).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2 3 4 5 6; do
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('run $i: frame', d['ms_per_step'], 'cull', r['avg_launch_ms'], 'classify', r['per_kernel_ms']['gpuculling_CS_GPUCulling LATE_CULL=0#classify'])"
done
rocm-smi --showclocks 2>/dev/null | head -20
