#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3s
mkdir -p $OUT
cd $R
AB_STEPS=100 bash tools/ab.sh base nostore base nostore -- > $OUT/ab.txt 2>&1
for f in 1 5; do AB_STEPS=60 bash tools/ab.sh base nostore -- --flags $f | sed "s/^/flags $f /" >> $OUT/ab.txt; done
cat $OUT/ab.txt
