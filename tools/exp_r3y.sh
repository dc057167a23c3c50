#!/bin/bash
# What the table lookups still cost the round-3 kernel (the bound on LDS-staged table rows): base vs -DTR_NO_LOOKUP by kernel
# trace, and the L1->L2 request counts of both.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3y
mkdir -p $OUT
cd $R
bash tools/ab_trace.sh base nolk base nolk 2>&1 | tee $OUT/ab.txt
LIB=$R/toyrenderer_amd/lib
for n in base nolk; do
  if [ $n = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  bash tools/pmc.sh r3y_$n -- "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" 2>&1 | grep -A14 "true, true, true, true" > $OUT/pmc_$n.txt
  echo "== $n"; cat $OUT/pmc_$n.txt
done
