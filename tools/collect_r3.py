#!/usr/bin/env python3
"""Copy the evidence of tools/profile_r3.sh (gpurun_out/r3_final, gpurun_out/pmc_r3final) into profiles/r3/ and derive
traffic.json (tied to the kernel's sources: bench.kernel_source_sha16) + the per-step figures from the PMC passes (median of
the steady-state launches)."""
import collections, csv, glob, json, os, shutil, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
src, dst = os.path.join(ROOT, "gpurun_out", "r3_final"), os.path.join(ROOT, "profiles", "r3")
os.makedirs(dst, exist_ok=True)
for f in ("final_bench.json", "final_bench_20steps.json", "final_bench_under_rocprof.json", "final_frame_timeline.txt", "final_kernel_stats.csv",
          "c4_animate_bench.json", "c3_animate_bench.json", "emulated_rank_share.txt", "clock.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
d = collections.defaultdict(lambda: collections.defaultdict(list))
newest = {}
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_r3final", "g*", "*", "*counter_collection.csv")):
    g = f.split(os.sep)[-3]
    if g not in newest or os.path.getmtime(f) > os.path.getmtime(newest[g]):
        newest[g] = f
for f in sorted(newest.values()):
    for r in csv.DictReader(open(f)):
        if "meshletCullKernel" in r["Kernel_Name"]:
            d[r["Kernel_Name"][:86]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
with open(os.path.join(dst, "final_pmc_cull_kernel.txt"), "w") as o:
    o.write("rocprofv3 --pmc <group> -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile, one run per counter group (tools/profile_r3.sh);\n"
            "the launches of a run: the priming frames, then warm-up and timed ones.  Columns: first launch (cleared HZB), median of the others.\n")
    for k, v in d.items():
        o.write(k + "\n")
        for c in sorted(v):
            x = [y for _, y in sorted(v[c])]
            o.write("   %-40s %14.0f %14.0f\n" % (c, x[0], statistics.median(x[1:])))
K = [k for k in d if "true, true, true, true" in k][0]
med = lambda c: statistics.median([y for _, y in sorted(d[K][c])][1:])
rd, wr, wr64 = med("TCC_EA0_RDREQ_sum"), med("TCC_EA0_WRREQ_sum"), med("TCC_EA0_WRREQ_64B_sum")
b = json.load(open(os.path.join(dst, "final_bench.json")))
alg = b["roofline"]["algorithmic_bytes_per_launch"]
hbm = int(rd * 128 + wr64 * 64 + (wr - wr64) * 32)
note = ("reads: all requests are 128 B (RDREQ_32B = {:.0f}): {:.3f} M x 128 B = {:.3f} GB; writes: {:.3f} M requests, {} of them 64 B, the others 32 B = {:.1f} MB.  "
        "Request sizes are read directly, so the gfx950 FETCH_SIZE half-count does not apply; requests served by the Infinity Cache are included.  "
        "Algorithmic bytes of the launch: {:.3f} GB -> traffic / algorithmic = {:.2f}").format(
            med("TCC_EA0_RDREQ_32B_sum"), rd / 1e6, rd * 128 / 1e9, wr / 1e6, int(wr64), (wr64 * 64 + (wr - wr64) * 32) / 1e6, alg / 1e9, hbm / alg)
json.dump({"source": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum (own pass) and TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum (own pass); "
                     "bench.py --steps 3 --warmup 2, config C3, flags 7; profiles/r3/final_pmc_cull_kernel.txt, meshletCullKernel<true,true,true,true> = the early launch; median of the steady-state launches",
           "kernel": "basepass_AS_Main LATE_CULL=0#cull", "config": "C3", "culling_flags": 7, "kernel_source_sha16": bench.kernel_source_sha16(),
           "read_requests_128B": int(rd), "read_requests_32B": int(med("TCC_EA0_RDREQ_32B_sum")),
           "write_requests_total": int(wr), "write_requests_64B": int(wr64), "hbm_bytes_per_launch": hbm, "note": note}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
steps = b["roofline"]["meshlets_per_launch"] / 64
cyc = med("GRBM_GUI_ACTIVE") / 8
print("kernel source sha16", bench.kernel_source_sha16())
print("hbm bytes", hbm, "ratio %.3f" % (hbm / alg))
print("per step: VALU %.1f (trans %.1f) SALU %.1f" % (med("SQ_INSTS_VALU") / steps, med("SQ_INSTS_VALU_TRANS_F32") / steps, med("SQ_INSTS_SALU") / steps))
print("VALU active %.3f  TA busy %.3f  TCP pending stall %.3f  L2 requests from L1 %.2f M  cycles per XCD %.0f" % (
    med("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cyc, med("TA_TA_BUSY_sum") / 256 / cyc, med("TCP_PENDING_STALL_CYCLES_sum") / 256 / cyc, med("TCP_TCC_READ_REQ_sum") / 1e6, cyc))
rows = list(csv.DictReader(open(os.path.join(dst, "final_kernel_stats.csv"))))
for r in rows[:22]:
    print("%-70s %5s %10.1f us" % (r["Name"][:70].replace("(anonymous namespace)::", ""), r["Calls"], float(r["AverageNs"]) / 1e3))
