#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats + separate PMC passes (never combined with
# sys/hip traces).  Usage (from repo root, via gpurun):  bash tools/profile.sh <tag> [bench args...]
set -e
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS --no-profile > /dev/null 2> $OUT/pmc_sq.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS --no-profile > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS --no-profile > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py $ARGS --no-profile > /dev/null 2> $OUT/pmc_tcc.log
find $OUT -name "*.csv" | head -30
