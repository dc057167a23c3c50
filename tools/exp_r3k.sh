#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.log
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); python3 $R/tools/timeline.py $f 1 > $OUT/frame_timeline.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
python3 -c "
import json
for n in ('bench','bench20'):
    d=json.load(open('$OUT/'+n+'.json')); r=d['roofline']
    print(n, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frame_frac'])
    if n=='bench':
        for k,v in sorted(r['per_kernel_ms'].items(), key=lambda x:-x[1]): print('   %8.4f  %s'%(v,k))
"
cat $OUT/frame_timeline.txt
tail -3 $OUT/bench20.log
