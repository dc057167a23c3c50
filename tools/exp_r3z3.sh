#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3z
mkdir -p $OUT
cd $R
bash tools/ab_trace.sh base lksc0 lksc1 base lksc0 lksc1 base lksc0 lksc1 base lksc0 lksc1 2>&1 | cut -c1-40 | tee $OUT/ab3.txt
