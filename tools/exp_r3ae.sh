#!/bin/bash
# Timeline of a rank's frame at 8 emulated ranks, with the exchange (1-rank RCCL group, one slot unpacked) and without.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/emul8
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/x -- python3 $R/bench.py --emulate-ranks 8 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/x.log 2>&1
python3 $R/tools/timeline.py $(find $OUT/x -name "*kernel_trace.csv" | head -1) 1 > $OUT/timeline_exchange.txt
TR_NO_GATHER=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/n -- python3 $R/bench.py --emulate-ranks 8 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/n.log 2>&1
python3 $R/tools/timeline.py $(find $OUT/n -name "*kernel_trace.csv" | head -1) 1 > $OUT/timeline_noexchange.txt
rm -rf $OUT/x $OUT/n
cat $OUT/timeline_exchange.txt; echo; cat $OUT/timeline_noexchange.txt
