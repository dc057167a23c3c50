#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3t
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py tests/test_gpu_exchange.py -x -q -m gpu > $OUT/parity.txt 2>&1; echo "parity rc $?" | tee -a $OUT/parity.txt
tail -3 $OUT/parity.txt
bash tools/ab_kernels.sh prev base prev base > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
