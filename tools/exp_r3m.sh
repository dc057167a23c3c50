#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3m
mkdir -p $OUT
cd $R
for b in 4 3 2 4 3; do TRHIP_AS_BLOCKS_PER_CU=$b AB_STEPS=100 bash tools/ab.sh base -- | sed "s/^/blocks $b /" >> $OUT/ab.txt; done
cat $OUT/ab.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c4_full_size" > $OUT/c4.txt 2>&1; echo "c4 rc $?"; tail -5 $OUT/c4.txt
