#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3d
mkdir -p $OUT
cd $R
LIB=$R/toyrenderer_amd/lib
for f in 0 1 2 3 4 5 6 7; do AB_STEPS=30 bash tools/ab.sh base -- --flags $f | sed "s/^/flags $f /" >> $OUT/flags.txt; done
AB_STEPS=50 bash tools/ab.sh base nomem nomem_nl nomem4 -- > $OUT/ab.txt 2>&1
TRHIP_AS_BLOCKS_PER_CU=5 TRHIP_LIB=$LIB/exp/stamps/libtrhip.so LD_LIBRARY_PATH=$LIB/exp/stamps timeout -k 10 200 python3 tools/stamps.py > $OUT/stamps.txt 2> $OUT/stamps.err
cat $OUT/flags.txt $OUT/ab.txt $OUT/stamps.txt
