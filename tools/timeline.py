#!/usr/bin/env python3
"""Print the kernel timeline of the last frames from a rocprofv3 --kernel-trace CSV (start/end relative
to the frame's first kernel, stream/queue id), to see what overlaps.  usage: tools/timeline.py trace.csv [frames]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
k.sort()
# a frame starts at each early instance pass (instanceClassifyKernel<0>, or instanceFusedKernel<0> for small passes)
starts = [i for i, x in enumerate(k) if any(t in x[2] for t in ("instanceClassifyKernel<0>", "instanceClassifyKernelILi0", "instanceFusedKernel<0>", "instanceFusedKernelILi0"))]
if len(starts) < nframes + 1:
    print("not enough frames", len(starts)); sys.exit(0)
a, b = starts[-nframes - 1], starts[-1]
t0 = k[a][0]
for s, e, name, q, st in k[a:b]:
    import re
    mm = re.search(r"(\w+(?:<[^()]*>)?)\(", name.replace("(anonymous namespace)::", ""))
    short = (mm.group(1) if mm else name)[-60:]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f} us  q{q} s{st}  {short}")
print("frame span us:", (k[b][0] - t0) / 1e3 / nframes)
