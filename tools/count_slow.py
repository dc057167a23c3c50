"""Diagnostic (needs a -DTR_COUNT_PATHS build in TRHIP_LIB): how many meshlets of a frame the deferred mode of the cull kernel
sends to their exact re-evaluation, in how many fix passes."""
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
res = r.results()
tested = sum(bench.gs_num_meshlets(spec, res[s]["records"]) for s in (0, 1) if res[s] is not None)
print(f"one frame ({tested} meshlets tested): deferred meshlets {out[0]} ({100.0 * out[0] / max(tested, 1):.3f} %) in {out[1]} waves' regions, batches redone exactly in place {out[2]}, meshlets re-evaluated in place (region full) {out[3]}")
print("wave-steps of the texel-path kernel on the fast arithmetic path", out[4], "on the exact path", out[5], "(needs -DTR_COUNT_PATHS)")
r.shutdown()
