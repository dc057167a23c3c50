#!/usr/bin/env python3
"""Copy the evidence of tools/profile_r4.sh (gpurun_out/r4_final) into profiles/r4/ and derive traffic.json -- HBM bytes per launch
from the PMC passes, the kernel's steady-state duration in the kernel trace and its VALU-active fraction, all tied to the
kernel's sources (bench.kernel_source_sha16: bench.py reports them only while the hash matches)."""
import collections, csv, glob, json, os, re, shutil, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
src, dst = os.path.join(ROOT, "gpurun_out", "r4_final"), os.path.join(ROOT, "profiles", "r4")
os.makedirs(dst, exist_ok=True)
for f in ("final_bench.json", "final_bench_20steps.json", "final_bench_under_rocprof.json", "final_frame_timeline.txt", "final_kernel_stats.csv",
          "final_cull_launch_order.txt", "stamps.txt", "deferred_counts.txt", "events_vs_trace.txt", "sync_cost.txt",
          "boundary_tests_shipped.txt", "boundary_tests_noband.txt", "c4_animate_bench.json", "emulated_rank_share.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    else:
        print("missing", f)
d = collections.defaultdict(lambda: collections.defaultdict(list))
# gpurun MERGES a run's files into the local gpurun_out/: an earlier collection's CSVs (other file names) are still there.  One
# file per counter group: the newest.
newest = {}
for f in glob.glob(os.path.join(src, "pmc", "g*", "**", "*counter_collection.csv"), recursive=True):
    g = os.path.relpath(f, os.path.join(src, "pmc")).split(os.sep)[0]
    if g not in newest or os.path.getmtime(f) > os.path.getmtime(newest[g]): newest[g] = f
for f in newest.values():
    for r in csv.DictReader(open(f)):
        if "meshletCullKernel" in r["Kernel_Name"]:
            d[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
with open(os.path.join(dst, "final_pmc_cull_kernel.txt"), "w") as o:
    o.write("rocprofv3 --pmc <group> -- python3 bench.py --steps 3 --warmup 2 --prime-steps 8 --no-cpu-baseline --no-profile, one run per counter group\n"
            "(tools/profile_r4.sh).  Columns: first launch (cleared HZB), median of the others.\n")
    for k, v in d.items():
        o.write(k + "\n")
        for c in sorted(v):
            x = [y for _, y in sorted(v[c])]
            o.write("   %-40s %14.0f %14.0f\n" % (c, x[0], statistics.median(x[1:])))
K = [k for k in d if "true, true, true, true" in k][0]
med = lambda c: statistics.median([y for _, y in sorted(d[K][c])][1:])
rd, rd32, wr, wr64 = med("TCC_EA0_RDREQ_sum"), med("TCC_EA0_RDREQ_32B_sum"), med("TCC_EA0_WRREQ_sum"), med("TCC_EA0_WRREQ_64B_sum")
b = json.load(open(os.path.join(dst, "final_bench.json")))
alg = b["roofline"]["algorithmic_bytes_per_launch"]
hbm = int((rd - rd32) * 128 + rd32 * 32 + wr64 * 64 + (wr - wr64) * 32)
cyc = med("GRBM_GUI_ACTIVE") / 8
valu_active = med("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cyc
m = re.search(r"second half ([0-9.]+)", open(os.path.join(dst, "final_cull_launch_order.txt")).read())
trace_ms = float(m.group(1)) / 1e3 if m else None
note = ("reads: {:.3f} M requests of 128 B ({:.0f} of 32 B) = {:.3f} GB; writes: {:.3f} M requests, {} of them 64 B, the others 32 B = {:.1f} MB.  Request sizes are read "
        "directly, so the gfx950 FETCH_SIZE half-count does not apply; requests served by the Infinity Cache are included.  Algorithmic bytes of the launch: "
        "{:.3f} GB -> traffic / algorithmic = {:.2f}").format(rd / 1e6, rd32, ((rd - rd32) * 128 + rd32 * 32) / 1e9, wr / 1e6, int(wr64), (wr64 * 64 + (wr - wr64) * 32) / 1e6, alg / 1e9, hbm / alg)
json.dump({"source": "tools/profile_r4.sh: rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum (own pass), TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum (own pass), "
                     "SQ_ACTIVE_INST_VALU ... (own pass), GRBM_GUI_ACTIVE (own pass); bench.py --steps 3 --warmup 2, config C3, flags 7; profiles/r4/final_pmc_cull_kernel.txt, "
                     "meshletCullKernel<true,true,true,true> = the early launch; median of the steady-state launches.  avg_launch_ms_trace: rocprofv3 --kernel-trace of bench.py --steps 20 "
                     "--warmup 5, average of the second half of the launches (profiles/r4/final_cull_launch_order.txt)",
           "kernel": "basepass_AS_Main LATE_CULL=0#cull", "config": "C3", "culling_flags": 7, "kernel_source_sha16": bench.kernel_source_sha16(),
           "read_requests_128B": int(rd - rd32), "read_requests_32B": int(rd32), "write_requests_total": int(wr), "write_requests_64B": int(wr64),
           "hbm_bytes_per_launch": hbm, "avg_launch_ms_trace": trace_ms, "valu_active": round(valu_active, 3), "note": note},
          open(os.path.join(dst, "traffic.json"), "w"), indent=1)
steps = b["roofline"]["meshlets_per_launch"] / 64
print("kernel source sha16", bench.kernel_source_sha16())
print("hbm bytes", hbm, "ratio %.3f" % (hbm / alg), "trace ms", trace_ms)
print("per step: VALU %.1f (trans %.1f) SALU %.1f LDS %.1f" % (med("SQ_INSTS_VALU") / steps, med("SQ_INSTS_VALU_TRANS_F32") / steps, med("SQ_INSTS_SALU") / steps, med("SQ_INSTS_LDS") / steps))
print("VALU active %.3f  TA busy %.3f  TCP pending stall %.3f  L2 requests from L1 %.2f M  cycles per XCD %.0f  LDS bank conflict cycles %.1f M" % (
    valu_active, med("TA_TA_BUSY_sum") / 256 / cyc, med("TCP_PENDING_STALL_CYCLES_sum") / 256 / cyc, med("TCP_TCC_READ_REQ_sum") / 1e6, cyc, med("SQ_LDS_BANK_CONFLICT") / 1e6))
