#!/bin/bash
# What makes the table lookups cost 50 us: per instruction, per distinct line (L1 tag work) or per L1 miss?
#   lk1: all lanes of a lookup read ONE line; lk2: the lanes' own entries folded into 4 KB (as many distinct lines, all L1 hits);
#   nomem / nomemnolk: the same question without the HBM stream.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3z
mkdir -p $OUT
cd $R
bash tools/ab_trace.sh base nolk lk1 lk2 nomem nomemnolk base nolk lk1 lk2 nomem nomemnolk 2>&1 | cut -c1-40 | tee $OUT/ab.txt
