#!/bin/bash
# Late list build: recorded both ways, chosen on the device (shipped) against the three kernels only (TRHIP_NO_DUAL_LIST=1).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dual  frame', d['ms_per_step'])"
TRHIP_NO_DUAL_LIST=1 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('three frame', d['ms_per_step'])"
done
