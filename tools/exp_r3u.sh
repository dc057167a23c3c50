#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3u
mkdir -p $OUT
cd $R
AB_STEPS=100 bash tools/ab.sh base nostore local dense base nostore local dense -- > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
