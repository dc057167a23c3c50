#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3j
mkdir -p $OUT
cd $R
LIB=$R/toyrenderer_amd/lib
for n in base late; do
  if [ $n = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu > $OUT/parity_$n.txt 2>&1; echo "$n rc $?"
  grep -E "^FAILED|passed|failed" $OUT/parity_$n.txt | cut -c1-150
done
