#!/bin/bash
# (EXTRA_PMC="<counters>": further counters of the same launches.)
# Diagnostic: effective shader clock of the early cull kernel, launch by launch: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs)
# and the kernel trace's duration from ONE rocprofv3 run (--pmc with --kernel-trace only), N processes back to back.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4; mkdir -p $OUT
export LD_LIBRARY_PATH=$R/toyrenderer_amd/lib
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 ${1:-4}); do
  rm -rf $OUT/pmc_clk
  timeout -k 5 150 rocprofv3 --pmc GRBM_GUI_ACTIVE ${EXTRA_PMC} --kernel-trace --output-format csv -d $OUT/pmc_clk -- python3 $R/bench.py --steps 40 --warmup 5 --prime-steps 128 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python3 - <<PY
import csv, glob, statistics
import collections
cyc, dur, extra = {}, {}, collections.defaultdict(dict)
for f in glob.glob("$OUT/pmc_clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "meshletCullKernel<true, true, true, true>" in r["Kernel_Name"]:
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cyc[int(r["Dispatch_Id"])] = float(r["Counter_Value"]) / 8
            else: extra[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
for f in glob.glob("$OUT/pmc_clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "meshletCullKernel<true, true, true, true>" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = sorted(set(cyc) & set(dur)); ids = ids[len(ids) // 2:]
c = statistics.median(cyc[i] for i in ids); d = statistics.median(dur[i] for i in ids)
print("process $i: early cull, steady-state launches (%d): %.1f us, %.0f cycles per XCD -> %.3f GHz" % (len(ids), d, c, c / d / 1e3)
      + "".join("  %s %.0f" % (k, statistics.median(v[i] for i in ids if i in v)) for k, v in sorted(extra.items())))
PY
done
