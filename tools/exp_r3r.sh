#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3r
mkdir -p $OUT
cd $R
for f in 0; do AB_STEPS=60 bash tools/ab.sh base inst0 nostore both base inst0 nostore both -- --flags $f | sed "s/^/flags $f /" >> $OUT/ab.txt; done
TRHIP_AS_NO_PERM=1 AB_STEPS=60 bash tools/ab.sh base both -- --flags 0 | sed "s/^/flags 0 listorder /" >> $OUT/ab.txt
cat $OUT/ab.txt
