#!/bin/bash
# The table lookup as an atomic with return (executed at the L2, holds no L1 line): does it stop competing with the stream?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/abtrace
cd $R
bash tools/ab_trace.sh base lkatom base lkatom 2>&1 | cut -c1-40
grep -o '"lists_digest": "[0-9a-f]*"' $OUT/t1.json $OUT/t2.json
