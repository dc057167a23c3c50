#!/bin/bash
# Record -> wave mapping inside a window: interleaved (shipped) vs blocked (an instance's four groups in two consecutive steps of one wave).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/abtrace
cd $R
bash tools/ab_trace.sh base blk base blk base blk 2>&1 | cut -c1-40
grep -o '"lists_digest": "[0-9a-f]*"' $OUT/t1.json $OUT/t2.json
