#!/bin/bash
# A/B of back-end builds: frame ms and the HIP-event times of the named kernels of the early phase.
#   bash tools/ab_kernels.sh name1 name2 ... (names under toyrenderer_amd/lib/exp/, or "base")
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$R/toyrenderer_amd/lib
for n in "$@"; do
  if [ "$n" = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_ms']
print('%-8s frame %.4f ' % ('$n', d['ms_per_step']) + ' '.join('%s %.1f' % (x.replace('gpuculling_CS_GPUCulling LATE_CULL=','i').replace('basepass_AS_Main LATE_CULL=','m').replace('ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=1','spd'), v * 1000) for x, v in k.items()))"
done
