#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3i
mkdir -p $OUT
cd $R
LIB=$R/toyrenderer_amd/lib
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py -x -q -m gpu > $OUT/parity.txt 2>&1; echo "parity rc $?" | tee -a $OUT/parity.txt
tail -3 $OUT/parity.txt
# mutation: without the uncertainty band the boundary test must fail
TRHIP_LIB=$LIB/exp/noband/libtrhip.so LD_LIBRARY_PATH=$LIB/exp/noband timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "cone_test_at_its_decision_boundary" > $OUT/mutation.txt 2>&1; echo "mutation rc $? (expected: non-zero)" | tee -a $OUT/mutation.txt
tail -4 $OUT/mutation.txt
AB_STEPS=100 bash tools/ab.sh late base d3b24 nomem late base d3b24 -- > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
