#!/bin/bash
# With the meshlet cull stream the lookups are nearly free: is the tile order (binning in the instance pass) still worth its cost?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  bash tools/ab_trace.sh base 2>&1 | cut -c1-200
  TRHIP_AS_NO_PERM=1 bash tools/ab_trace.sh base 2>&1 | cut -c1-200
done
for rep in 1 2; do
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile order frame', d['ms_per_step'])"
TRHIP_AS_NO_PERM=1 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('list order frame', d['ms_per_step'])"
done
