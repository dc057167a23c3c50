#!/bin/bash
# Effective shader clock of the cull kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, for the shipped
# library and experiment builds (names under toyrenderer_amd/lib/exp).   bash tools/clock_probe.sh base nomem ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$R/toyrenderer_amd/lib
OUT=$R/gpurun_out/clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ $n = base ]; then unset TRHIP_LIB; export LD_LIBRARY_PATH=$LIB; else export TRHIP_LIB=$LIB/exp/$n/libtrhip.so; export LD_LIBRARY_PATH=$LIB/exp/$n; fi
  rm -rf $OUT/$n
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2> $OUT/$n.log
  python3 - "$OUT/$n" "$n" <<'PY'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(d + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = []
for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and 'meshletCullKernel' in r['Kernel_Name'] and 'true, true, true, true' in r['Kernel_Name']:
            k, ns = dur.get(r['Dispatch_Id'], (None, None))
            if ns:
                rows.append((float(r['Counter_Value']) / 8.0, ns))
rows = rows[3:]
if rows:
    cyc = sorted(c for c, _ in rows)[len(rows) // 2]
    ns = sorted(n for _, n in rows)[len(rows) // 2]
    print(f"{name}: early cull launches {len(rows)}, median {cyc / 1e3:.0f} k cycles per XCD in {ns / 1e3:.1f} us -> {cyc / ns:.3f} GHz")
else:
    print(name, "no rows")
PY
done
