#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for b in 4 3 2; do echo "blocks $b"; TRHIP_AS_BLOCKS_PER_CU=$b bash tools/ab_trace.sh base 2>&1 | cut -c1-60; done
