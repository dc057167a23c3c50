#!/bin/bash
# A/B by the kernel trace: average duration of the early cull kernel and of the frame's other kernels over a 40-frame run
# under rocprofv3 --kernel-trace --stats (names under toyrenderer_amd/lib/exp/, or "base").   bash tools/ab_trace.sh base prev base prev
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$R/toyrenderer_amd/lib
OUT=$R/gpurun_out/abtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for n in "$@"; do
  i=$((i+1))
  if [ $n = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  rm -rf $OUT/t$i
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$i -- python3 $R/bench.py --steps 40 --warmup 5 --prime-steps 64 --no-cpu-baseline --no-profile ${BENCH_ARGS} > $OUT/t$i.json 2> $OUT/t$i.log
  python3 - "$OUT/t$i" "$n" <<'PY'
import csv, glob, sys
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/*/*kernel_stats.csv')
if not f:
    print(name, 'no stats'); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
out = []
for r in rows:
    k = r['Name'].replace('(anonymous namespace)::', '')
    short = k.split('(')[0]
    if 'meshletCullKernel' in k and 'true, true, true, true' in k: short = 'CULL'
    elif 'meshletCullKernel' in k: short = 'cullLate'
    out.append((short, float(r['AverageNs']) / 1e3, int(r['Calls'])))
cull = [x for x in out if x[0] == 'CULL']
print('%-8s' % name, 'cull %.1f us (%d calls) |' % (cull[0][1], cull[0][2]) if cull else '', ' '.join('%s %.1f' % (a[:14], b) for a, b, c in out if a != 'CULL' and c >= 20)[:400])
PY
done
