#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3o
mkdir -p $OUT
cd $R
for f in 0 7; do AB_STEPS=60 bash tools/ab.sh base inst0 base inst0 -- --flags $f | sed "s/^/flags $f /" >> $OUT/ab.txt; done
cat $OUT/ab.txt
