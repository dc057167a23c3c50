#!/bin/bash
# usage: bash tools/sweep.sh VAR v1 v2 ... -- [bench args]   (prints value, ms/step, cull kernel ms per setting)
VAR=$1; shift
VALS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do VALS+=("$1"); shift; done
shift
for v in "${VALS[@]}"; do
  export $VAR=$v
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$VAR=$v', d['value'], d['ms_per_step'], r['avg_launch_ms'])"
done
