#!/usr/bin/env python3
"""Soak: many frames of a config with a static camera; the outputs of every k-th frame must hash to the steady-state value.
Catches rare ordering bugs (side stream hand-overs, ticket-chained kernels) that a handful of test frames would miss.
usage: python tools/soak.py [config=C3] [frames=2000] [every=100]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toyrenderer_amd import host, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
spec = synth.config_spec(cfg)
view = synth.make_view(eye=(0.0, 0.0, 0.0), prev_eye=(0.05, 0.0, 0.1), prev_yaw=0.002)
scene = synth.make_scene(spec)
depth = synth.gen_depth(view, 200)
cap = spec.num_instances * ((2 * spec.meshlets_lod0 + 31) // 32) + 1
r = host.Renderer(render=(view.renderW, view.renderH), max_groups=cap, max_transient_bytes=8 << 30)
r.load_scene(scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
r.set_culling(7)
r.upload_depth(depth)


def digest():
    res = r.results()
    h = hashlib.sha1()
    for s in range(4):
        if res[s] is None:
            continue
        for k in ("records", "visMask", "visibleList", "drawArgs", "dispatchArgs"):
            h.update(np.ascontiguousarray(res[s][k]).tobytes())
    h.update(r.download_hzb().tobytes())
    return h.hexdigest()


want, bad, t0 = None, 0, time.time()
for f in range(frames):
    r.set_camera(view)
    r.frame()
    if f == 9:
        want = digest()
    elif f > 9 and f % every == every - 1:
        got = digest()
        if got != want:
            bad += 1
            print(f"frame {f}: digest {got} != steady state {want}", flush=True)
        if time.time() - t0 > 60:
            print(f"frame {f} ...", flush=True); t0 = time.time()
r.shutdown()
print(f"{cfg}: {frames} frames, checked every {every}: {'OK' if not bad else str(bad) + ' MISMATCHES'} (steady-state digest {want})")
sys.exit(1 if bad else 0)
