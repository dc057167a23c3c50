#!/bin/bash
# Round-3 first measurement batch: VALU issue rates, the stream in isolation, and the real kernel by flags x record order.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3a
mkdir -p $OUT
cd $R
timeout -k 10 120 tools/valu_rate > $OUT/valu_rate.txt 2>&1 && echo valu done
timeout -k 10 300 tools/streamring > $OUT/streamring.txt 2>&1 && echo stream done
for f in 1 5 7; do
  for p in tile list; do
    if [ $p = list ]; then export TRHIP_AS_NO_PERM=1; else unset TRHIP_AS_NO_PERM; fi
    timeout -k 10 200 python3 bench.py --flags $f --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('flags $f order $p', 'Gm/s', d['value'], 'frame ms', d['ms_per_step'], 'cull ms', r['avg_launch_ms'], 'frac', r['frac'], 'tested', r['meshlets_per_launch'])" >> $OUT/flags_order.txt
  done
done
cat $OUT/flags_order.txt
