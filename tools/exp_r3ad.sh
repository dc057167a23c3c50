#!/bin/bash
# stepQuotients' range check and the zero-weight flag as lane masks (170 VALU per step instead of 173): parity + A/B against the previous kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 700 python3 -m pytest tests -x -q -m gpu -k "parity or primitives or full_size" > gpurun_out/parity.log 2>&1; tail -3 gpurun_out/parity.log
bash tools/ab_trace.sh base prev base prev base prev 2>&1 | cut -c1-40
