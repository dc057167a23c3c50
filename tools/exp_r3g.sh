#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3g
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py -x -q -m gpu > $OUT/parity.txt 2>&1; echo "parity rc $?" | tee -a $OUT/parity.txt
tail -3 $OUT/parity.txt
AB_STEPS=100 bash tools/ab.sh prev base nomem prev base -- > $OUT/ab.txt 2>&1
for f in 0 1 5; do AB_STEPS=30 bash tools/ab.sh prev base -- --flags $f | sed "s/^/flags $f /" >> $OUT/ab.txt; done
cat $OUT/ab.txt
