#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3x
mkdir -p $OUT
cd $R
LIB=$R/toyrenderer_amd/lib
for n in base nostore local; do
  if [ $n = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  bash tools/pmc.sh r3x_$n -- "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum" "GRBM_GUI_ACTIVE" 2>&1 | grep -A40 "true, true, true, true" | head -30 > $OUT/pmc_$n.txt
  echo "== $n"; cat $OUT/pmc_$n.txt
done
