#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3p
mkdir -p $OUT
cd $R
run() { python3 bench.py --emulate-ranks 8 --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('ms', d['ms_per_step'], ' '.join('%s=%.1f'%(k.split('#')[-1]+('L' if 'LATE_CULL=1' in k else ''), v*1e3) for k,v in sorted((r.get('per_kernel_ms') or {}).items(), key=lambda x:-x[1])[:14]))"; }
echo "no exchange, default" >> $OUT/emul.txt; TR_NO_GATHER=1 run >> $OUT/emul.txt
echo "no exchange, table from 2^16 groups" >> $OUT/emul.txt; TR_NO_GATHER=1 TRHIP_TABLE_MIN_GROUPS=65536 run >> $OUT/emul.txt
echo "exchange (1-rank RCCL group)" >> $OUT/emul.txt; run >> $OUT/emul.txt
echo "exchange, loopback volume" >> $OUT/emul.txt; TR_EMULATE_LOOPBACK=1 run >> $OUT/emul.txt
cat $OUT/emul.txt
