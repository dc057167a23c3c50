#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3q
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_host_path.py -x -q -m gpu -k "group_capacity or different_kinds" > $OUT/exch.txt 2>&1; echo "rc $?" | tee -a $OUT/exch.txt
tail -5 $OUT/exch.txt
