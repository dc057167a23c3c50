#!/bin/bash
# The whole -m gpu suite + smoke, output under gpurun_out/suite/ (progress visible in the file while it runs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/suite
mkdir -p $OUT
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; rc=$?
tail -5 $OUT/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
exit $rc
