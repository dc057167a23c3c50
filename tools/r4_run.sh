#!/bin/bash
# Round-4 GPU runs, one parametrised script (no more one-off exp_*.sh files).
#   bash tools/r4_run.sh <step> [args]      output under gpurun_out/r4/<step>*/
# steps (each is what one gpurun call chains with &&):
#   suite                         pytest -m gpu + smoke
#   ab <names...>                 bench.py (events, steady state) per back-end build: base = shipped, others under lib/exp/
#   trace <name> [bench args]     rocprofv3 --kernel-trace of bench.py --no-profile; prints the early cull launch durations in
#                                 launch order (first 12, every 16th, last 12) + per-kernel averages of the steady-state launches
#   pmc <name> "<group>" ...      one rocprofv3 --pmc pass per counter group on <name>
#   events                        per-launch HIP-event durations of the dominant kernel in launch order (tools/events_order.py)
#   emul M [bench args]           rank 0's share at M emulated ranks: without / with the exchange / with the loop-back collective
#   bench [bench args]            the plain bench line -> gpurun_out/r4/bench*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$R/toyrenderer_amd/lib
OUT=$R/gpurun_out/r4
mkdir -p $OUT
STEP=$1; shift
use() { if [ "$1" = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$1/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$1; fi; }
case $STEP in
suite)
  cd $R
  timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; rc=$?
  tail -5 $OUT/gpu_tests.log
  python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
  exit $rc ;;
ab)
  for n in "$@"; do
    use $n
    python3 $R/bench.py --steps ${AB_STEPS:-50} --warmup 5 --no-cpu-baseline --no-animated-leg ${BENCH_ARGS} 2>$OUT/ab_$n.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('${LABEL:-$n}', 'Gm/s', d['value'], 'frame ms', d['ms_per_step'], 'cull ms', r['avg_launch_ms'], 'frac', r['frac'], 'digest', d['lists_digest'][:12])" | tee -a $OUT/ab.txt
  done ;;
trace)
  n=$1; shift
  use $n
  cd /tmp && export TMPDIR=/tmp
  rm -rf $OUT/trace_$n
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$n -- python3 $R/bench.py --steps 40 --warmup 5 --prime-steps 128 --no-cpu-baseline --no-profile "$@" > $OUT/trace_$n.json 2> $OUT/trace_$n.log
  python3 $R/tools/trace_order.py $OUT/trace_$n $n | tee -a $OUT/trace.txt ;;
pmc)
  n=$1; shift
  use $n
  cd /tmp && export TMPDIR=/tmp
  i=0
  for G in "$@"; do
    rm -rf $OUT/pmc_${n}_g$i
    timeout -k 5 150 rocprofv3 --pmc $G --output-format csv -d $OUT/pmc_${n}_g$i -- python3 $R/bench.py --steps 3 --warmup 2 --prime-steps 8 --no-cpu-baseline --no-profile ${BENCH_ARGS} > /dev/null 2> $OUT/pmc_${n}_g$i.log || tail -5 $OUT/pmc_${n}_g$i.log
    i=$((i+1))
  done
  python3 $R/tools/pmc_summary.py $OUT "pmc_${n}_g" $n | tee -a $OUT/pmc.txt ;;
events)
  use ${1:-base}
  python3 $R/tools/events_order.py 2>$OUT/events.err | tee -a $OUT/events.txt ;;
emul)
  # rank 0's share at M emulated ranks: no exchange / exchange with a 1-rank RCCL group / loop-back collective (real unpack volume)
  M=${1:-8}; shift || true
  use ${EMUL_LIB:-base}
  for mode in none one loop; do
    case $mode in none) E="TR_NO_GATHER=1";; one) E="";; loop) E="TR_EMULATE_LOOPBACK=1";; esac
    env $E python3 $R/bench.py --emulate-ranks $M --steps 100 --warmup 10 --no-cpu-baseline --no-profile "$@" 2>$OUT/emul_$mode.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${LABEL:-emul} M $M', '$mode', 'ms', d['ms_per_step'], 'Gm/s(rank)', d['value'])" | tee -a $OUT/emul.txt
  done ;;
bench)
  use base
  python3 $R/bench.py "$@" 2>$OUT/bench.err | tee $OUT/bench_$(date +%H%M%S).json ;;
*) echo "unknown step $STEP"; exit 2 ;;
esac
