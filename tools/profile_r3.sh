#!/bin/bash
# Round-3 evidence run on the GPU box (from the repo root, through gpurun): bench lines (default and the driver's short run),
# kernel trace + stats, frame timeline, PMC passes of the dominant kernel (one rocprofv3 run per counter group, never
# combined with other traces), effective clock, C3 / C4 animated diagnostics, emulated rank shares.  -> gpurun_out/r3_final/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/final_bench.json 2> $OUT/final_bench.log
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/final_bench_20steps.json 2> $OUT/final_bench_20steps.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/final_bench_under_rocprof.json 2> $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); python3 $R/tools/timeline.py $f 1 > $OUT/final_frame_timeline.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/final_kernel_stats.csv
rm -rf $OUT/trace
cd $R
bash tools/pmc.sh r3final -- "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
   "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
   "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
   "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32" \
   "GRBM_GUI_ACTIVE" > $OUT/pmc_print.txt 2>&1
bash tools/clock_probe.sh base > $OUT/clock.txt 2>&1
python3 $R/bench.py --config C4 --animate --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c4_animate_bench.json 2> $OUT/c4_animate_bench.log
python3 $R/bench.py --config C3 --animate --steps 100 --warmup 10 --no-cpu-baseline > $OUT/c3_animate_bench.json 2> $OUT/c3_animate_bench.log
for m in 2 4 8; do
  TR_NO_GATHER=1 python3 $R/bench.py --emulate-ranks $m --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('M', $m, 'shard frame, no exchange: ms', d['ms_per_step'])"
  python3 $R/bench.py --emulate-ranks $m --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('M', $m, 'shard frame + exchange (1-rank RCCL group, one slot unpacked): ms', d['ms_per_step'])"
done > $OUT/emulated_rank_share.txt
ls $OUT
