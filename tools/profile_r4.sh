#!/bin/bash
# Round-4 evidence run on the GPU box (from the repo root, through gpurun; needs the variant builds count, stamps, r3stamps,
# noband under toyrenderer_amd/lib/exp -- tools/variants.sh, see tools/README.md).  One run, after the last change to the
# hashed kernel sources; tools/collect_r4.py (run once on the box, for traffic.json, and once here) turns gpurun_out/r4_final/
# into profiles/r4/final_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$R/toyrenderer_amd/lib
OUT=$R/gpurun_out/r4_final
rm -rf $OUT; mkdir -p $OUT
use() { if [ "$1" = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$1/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$1; fi; }
use base
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"; date +%T
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile > $OUT/final_bench_under_rocprof.json 2> $OUT/trace.log   # (--no-profile: the timed frames only -- the last frame of the trace is a steady-state one)
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $f 1 > $OUT/final_frame_timeline.txt
python3 $R/tools/trace_order.py $OUT/trace final > $OUT/final_cull_launch_order.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/final_kernel_stats.csv
rm -rf $OUT/trace
echo "== PMC"; date +%T
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32" \
         "GRBM_GUI_ACTIVE"; do
  timeout -k 5 150 rocprofv3 --pmc $G --output-format csv -d $OUT/pmc/g$i -- python3 $R/bench.py --steps 3 --warmup 2 --prime-steps 8 --no-cpu-baseline --no-profile > /dev/null 2> $OUT/pmc_g$i.log || tail -3 $OUT/pmc_g$i.log
  i=$((i+1))
done
echo "== traffic.json from this box's trace + PMC passes, then the bench lines (they report traffic / trace only with it in place)"; date +%T
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-animated-leg > $OUT/final_bench.json 2> /dev/null     # (placeholder: algorithmic bytes per launch for collect_r4.py)
(cd $R && python3 tools/collect_r4.py > $OUT/collect_on_box.txt 2>&1) || tail -5 $OUT/collect_on_box.txt
python3 $R/bench.py > $OUT/final_bench.json 2> $OUT/final_bench.log
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/final_bench_20steps.json 2> $OUT/final_bench_20steps.log
echo "== stamps, counts, events, sync cost"; date +%T
for v in r3stamps stamps; do use $v; echo "== $v"; TRHIP_AS_BLOCKS_PER_CU=5 python3 $R/tools/stamps.py 2>/dev/null | tail -9; done > $OUT/stamps.txt
use count; python3 $R/tools/count_slow.py 2>/dev/null | tail -2 > $OUT/deferred_counts.txt
use base
python3 $R/tools/events_order.py > $OUT/events_vs_trace.txt 2> $OUT/events.err
timeout -k 10 200 $R/tools/sync_cost > $OUT/sync_cost.txt 2>&1
echo "== boundary tests on the shipped build and on the build without bands"; date +%T
cd $R
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "projection_filter or cone_test_at" 2>&1 | tail -3 > $OUT/boundary_tests_shipped.txt
use noband; python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "projection_filter" 2>&1 | grep -E "^FAILED|passed|failed" > $OUT/boundary_tests_noband.txt
use base
cd /tmp
echo "== animated, emulated rank shares"; date +%T
python3 $R/bench.py --config C4 --animate --steps 20 --warmup 3 --no-cpu-baseline --no-profile > $OUT/c4_animate_bench.json 2> $OUT/c4_animate_bench.log
for m in 2 4 8; do
  for mode in none one loop; do
    case $mode in none) E="TR_NO_GATHER=1";; one) E="TR_X=1";; loop) E="TR_EMULATE_LOOPBACK=1";; esac
    env $E python3 $R/bench.py --emulate-ranks $m --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C3 M $m', '$mode', 'ms per frame', d['ms_per_step'])"
  done
done > $OUT/emulated_rank_share.txt
for mode in none one loop; do
  case $mode in none) E="TR_NO_GATHER=1";; one) E="TR_X=1";; loop) E="TR_EMULATE_LOOPBACK=1";; esac
  env $E python3 $R/bench.py --emulate-ranks 8 --config C4 --animate --steps 40 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C4 animated M 8', '$mode', 'ms per frame', d['ms_per_step'])"
done >> $OUT/emulated_rank_share.txt
date +%T
ls $OUT
