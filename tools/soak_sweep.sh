#!/bin/bash
# Soak: the randomised small-scene sweep of tests/test_gpu_parity.py (and the host-mirror one) on N x 48 further random
# configurations, every output word against the oracle.   bash tools/soak_sweep.sh [N=20]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-20}
OUT=$R/gpurun_out/soak
mkdir -p $OUT
cd $R
fail=0
for i in $(seq 1 $N); do
  # (TR_SWEEP_TABLE=1 in the environment: every pass at a record capacity of 2^19 -> the footprint-table cull kernel)
  TR_SWEEP_OFFSET=$((i * 48)) timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomised_small_scene_sweep" > $OUT/sweep_$i.log 2>&1 || { fail=1; echo "offset $((i * 48)) FAILED"; tail -20 $OUT/sweep_$i.log; break; }
  echo "offset $((i * 48)): $(tail -1 $OUT/sweep_$i.log)"
done
exit $fail
