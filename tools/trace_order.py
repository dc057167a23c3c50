#!/usr/bin/env python3
"""Kernel trace (rocprofv3 --kernel-trace --output-format csv) -> the early cull kernel's durations IN LAUNCH ORDER and the
steady-state average of every kernel of the frame.   python tools/trace_order.py <dir> [label]"""
import csv, glob, sys, collections
d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not files:
    print(label, "no kernel trace under", d); sys.exit(0)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    name = k.split("(")[0]
    if "meshletCullKernel" in name:
        name = "CULL" if "true, true, true, true" in name or "1, 1, 1, 1" in name else "cullLate" if "<" in name else name
    return name
per = collections.defaultdict(list)
for r in rows:
    per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cull = per.get("CULL", [])
n = len(cull)
print(f"{label}: {n} early cull launches; in launch order (us):")
if n:
    print("  first 12:", " ".join(f"{x:.0f}" for x in cull[:12]))
    print("  every 16th:", " ".join(f"{x:.0f}" for x in cull[::16]))
    print("  last 12:", " ".join(f"{x:.0f}" for x in cull[-12:]))
    steady = cull[n // 2:]
    print(f"  average all {sum(cull) / n:.1f}, second half {sum(steady) / len(steady):.1f}, min {min(cull):.1f}, max {max(cull):.1f}")
print("  steady-state (second half of each kernel's launches) averages, us:")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 8:
        h = v[len(v) // 2:]
        print(f"    {k[:44]:44s} {sum(h) / len(h):8.1f}  x{len(v)}")
