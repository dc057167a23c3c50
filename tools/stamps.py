"""Diagnostic: per-segment cycle shares of the meshlet cull kernel (TR_STAMPS build only)."""
import ctypes as C, os, sys, json, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyrenderer_amd import host, rhi, synth
import bench
spec = synth.config_spec(os.environ.get("CFG", "C3"))
view = synth.make_view(eye=(0.0, 0.0, 0.0), prev_eye=(0.05, 0.0, 0.1), prev_yaw=0.002)
depth = synth.gen_depth(view, 200)
cap = spec.num_instances * 4 + 1
r = host.Renderer(render=(view.renderW, view.renderH), max_groups=cap, max_transient_bytes=8 << 30)
bench.build_shard(spec, 0, 1, r)
r.set_culling(7); r.upload_depth(depth)
L = rhi.load()
for _ in range(3):
    r.set_camera(view); r.frame()
r.wait_idle()
out = (C.c_ulonglong * 8)()
L.trhip_debug_read_stamps(out, 1)
for _ in range(3):
    r.set_camera(view); r.frame()
r.wait_idle()
L.trhip_debug_read_stamps(out, 1)
v = np.array(list(out), float)
names = ["0 between batches (drain, fixups, mask store)", "1 prologue (entry -> instance block -> LDS)", "2 wait for the slot + transform + frustum", "3 quotients + cone", "4 lookup + prefetch issue", "5 lookup wait + resolve", "6 ballot + mask to LDS", "7 loop overhead"]
tot = v.sum()
WAVES = 256 * 4 * int(os.environ.get("TRHIP_AS_BLOCKS_PER_CU", "4"))
for n, x in zip(names, v):
    print(f"{n:48s} {x/tot*100:6.2f} %   {x/3/WAVES:12.0f} cycles per wave per frame (both cull launches)")
print("total cycles per wave per frame", tot / 3 / WAVES)
r.shutdown()
