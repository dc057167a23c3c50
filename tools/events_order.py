#!/usr/bin/env python3
"""HIP-event duration of the dominant kernel, launch by launch: (a) the 5 frames right after an idle gap with every launch
bracketed (what round 3's bench measured), (b) steady state with only the dominant kernel bracketed (what the bench measures
now).  Explains the 8 % between round 3's events (0.373 ms) and the kernel trace (0.341 ms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from toyrenderer_amd import synth, host, rhi
import bench
DOM = bench.DOMINANT
torch.cuda.set_device(0)
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
spec = synth.config_spec("C3")
view = synth.make_view(eye=(0.0, 0.0, 0.0), prev_eye=(0.05, 0.0, 0.1), prev_yaw=0.002)
depth = synth.gen_depth(view, 200)
cap = spec.num_instances * 4 + 1
r = host.Renderer(render=(view.renderW, view.renderH), device_index=0, stream=side.cuda_stream, max_groups=cap, max_transient_bytes=8 << 30)
dev = rhi.Device(handle=r.device())
bench.build_shard(spec, 0, 1, r, threads=8)
r.set_culling(7); r.set_gpu_timers(False); r.upload_depth(depth)
def frames(n):
    for _ in range(n):
        r.set_camera(view); r.frame()
frames(256); dev.wait_idle()
for label, filt, idle_ms in (("all launches bracketed, after 50 ms idle", None, 50), ("all launches bracketed, steady state", None, 0),
                             ("dominant only, after 50 ms idle", DOM, 50), ("dominant only, steady state", DOM, 0)):
    if idle_ms:
        dev.wait_idle(); time.sleep(idle_ms / 1e3)
    else:
        frames(128)
    dev.profile_filter(filt)
    seq = []
    # per-frame figures: reset between frames would synchronise; instead profile k frames at a time with growing k
    for k in (1, 1, 1, 1, 1, 5, 10, 20, 40):
        dev.profile_reset(); dev.profile_enable(True)
        frames(k)
        dev.profile_enable(False)
        if idle_ms:
            dev.wait_idle()
        n, ms = dev.profile()[DOM]
        seq.append((k, ms / n))
    print(label + ": " + "  ".join(f"{k}f:{m * 1e3:.0f}us" for k, m in seq), flush=True)
t0 = time.perf_counter(); frames(128); dev.wait_idle(); t1 = time.perf_counter()
dev.profile_filter(DOM); dev.profile_reset(); dev.profile_enable(True)
t2 = time.perf_counter(); frames(128); dev.wait_idle(); t3 = time.perf_counter()
dev.profile_enable(False)
print(f"frame ms without / with the one event pair per frame: {(t1 - t0) / 128 * 1e3:.4f} / {(t3 - t2) / 128 * 1e3:.4f}")
r.shutdown()
