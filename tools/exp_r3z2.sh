#!/bin/bash
# Cache-policy bits on the table lookup (global_load_ushort): does a lookup that does not allocate in the L1 stop costing the stream?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3z
mkdir -p $OUT
cd $R
bash tools/ab_trace.sh base lksc0 lksc1 lksc01 lknt lkntsc1 base lksc0 lksc1 lksc01 lknt lkntsc1 2>&1 | cut -c1-40 | tee $OUT/ab2.txt
