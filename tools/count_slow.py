import ctypes as C, os, sys
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
r.set_camera(view); r.frame()
r.wait_idle()
L.trhip_debug_read_stamps(out, 1)
print("batches", out[0], "with fixups", out[1], "deferred lookups", out[2], "overflows", out[3])
print("wave-steps on the fast arithmetic path", out[4], "on the exact path", out[5], "(needs -DTR_COUNT_PATHS)")
r.shutdown()
