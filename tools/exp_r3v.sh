#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3v
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py tests/test_gpu_exchange.py -x -q -m gpu > $OUT/parity.txt 2>&1; echo "parity rc $?" | tee -a $OUT/parity.txt
tail -3 $OUT/parity.txt
bash tools/ab_trace.sh nowriter base nostore nowriter base nostore 2>&1 | cut -c1-120 | tee $OUT/abtrace.txt
