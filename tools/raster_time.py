#!/usr/bin/env python3
"""Time the two-phase frame with self-rendered depth (FrameDriver(raster_depth=True)) on a generated glTF scene with real
geometry, per kernel (back-end profile).  usage: python tools/raster_time.py [num_spheres] [width height]"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from scene_gen import write_city_gltf  # noqa: E402
from toyrenderer_amd import gltf_lite, rhi, synth  # noqa: E402
from toyrenderer_amd import interop as I  # noqa: E402
from toyrenderer_amd.frame import FrameDriver, GpuScene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
render = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
with tempfile.TemporaryDirectory() as d:
    s = gltf_lite.load(write_city_gltf(Path(d), num_spheres=n, num_cutouts=n // 8))
# world matrices on the host (this is a timing tool; the transform pass is timed elsewhere)
from toyrenderer_amd import frame as F  # noqa: E402
dev = rhi.Device(0)
inst = s.instances.copy()
nodes = s.nodes
for i in range(len(inst)):
    k = int(s.primToNode[i])
    M = np.eye(4, dtype=np.float64)
    while k != 0xFFFFFFFF:
        t = nodes[k]
        R = synth.quat_to_matrix(tuple(t["m_Rotation"]))
        L = np.diag(list(t["m_Scale"]) + [1.0]) @ R
        L[3, :3] = t["m_Position"]
        M = M @ L
        k = int(t["m_ParentNodeIdx"])
    inst["m_WorldMatrix"][i] = M.astype(np.float32)
gs = GpuScene(dev, inst, s.meshData, s.meshlets, s.opaqueIds, s.alphaMaskIds)
gs.set_geometry(s.vertices, s.meshletVertexIds, s.meshletTriangles)
cam = s.cameras[0]
P = synth.perspective_rh_reverse_z_infinite(cam.yfov, render[0] / render[1], cam.znear)
V = synth.world_to_view((0.0, 0.0, 0.0), cam.orientation)
view = synth.View(V, V.copy(), P, float(np.float32(cam.znear)), *render)
drv = FrameDriver(dev, gs, view, record_capacity=1 << 16, culling_flags=7, raster_depth=True)
for _ in range(3):
    drv.record(); drv.run()
dev.wait_idle()
dev.profile_reset(); dev.profile_enable(True)
for _ in range(5):
    drv.record(); drv.run()
dev.wait_idle()
prof = dev.profile()
res = drv.results()
print(f"{len(inst)} instances, {len(s.meshlets)} meshlets in meshes, render {render}; visible early {int(res[0]['drawArgs'][0])}, late {int(res[1]['drawArgs'][0]) if res[1] else 0}")
for k_, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k_:70s} {ms / cnt * 1e3:9.1f} us x{cnt}")
depth = drv.depth.download_mip(0)
print("covered pixels:", int(np.count_nonzero(depth)), "of", depth.size)
