#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3q
mkdir -p $OUT /tmp/q2tmp
cd $R
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 23456 tests/mr_host_ranks.py q2 /tmp/q2tmp > $OUT/q2_direct.txt 2>&1; echo "rc $?"
grep -v "^\s*$" $OUT/q2_direct.txt | grep -v "Gloo\|warn" | tail -30
