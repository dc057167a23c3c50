#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3l
mkdir -p $OUT
cd $R
AB_STEPS=100 bash tools/ab.sh base prio3 prio1 b16 base prio3 prio1 b16 -- > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
