#!/bin/bash
# What spatial coherence of the meshlets INSIDE a mesh is worth: C3 (meshlet positions random in the mesh) against C3m (the same
# meshlets in Z-order, as a meshlet builder emits them), cull kernel by trace + the L1->L2 request count.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/ab_trace.sh base 2>&1 | cut -c1-60
BENCH_ARGS="--config C3m" bash tools/ab_trace.sh base 2>&1 | cut -c1-60
bash tools/ab_trace.sh base 2>&1 | cut -c1-60
BENCH_ARGS="--config C3m" bash tools/ab_trace.sh base 2>&1 | cut -c1-60
python3 bench.py --config C3m --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('C3m bench:', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['config']['meshlets_tested_per_frame'], d['config']['visible_per_frame'])"
bash tools/pmc.sh r3ag --config C3m -- "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" 2>&1 | grep -A6 "true, true, true, true"
