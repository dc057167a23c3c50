"""ctypes binding of the C++ host mirror (include/trhost.h -> lib/libtoyrenderer_host.so):
Graphic / Scene / RenderGraph / BasePassRenderers driving the HIP kernels through the C ABI.
This is the drop-in path; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import interop as I
from . import rhi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtoyrenderer_host.so")

HOST_SYMBOLS = [
    "trhost_last_error", "trhost_initialize", "trhost_shutdown", "trhost_load_scene", "trhost_upload_meshlets", "trhost_load_nodes",
    "trhost_set_node_transforms", "trhost_set_instance_update_range", "trhost_set_camera", "trhost_set_culling", "trhost_set_limits", "trhost_upload_depth",
    "trhost_upload_hzb_mip", "trhost_download_hzb_mip", "trhost_hzb_info", "trhost_frame", "trhost_wait_idle",
    "trhost_pass_buffers", "trhost_instance_buffer", "trhost_device", "trhost_render_graph_stats", "trhost_renderer_times",
    "trhost_heap_sim", "trhost_set_shard_late_exchange", "trhost_set_gpu_timers",
    "trhost_rccl_allgather", "trhost_exchange_create", "trhost_exchange_run", "trhost_exchange_wait", "trhost_exchange_outputs",
    "trhost_exchange_destroy", "trhost_load_geometry", "trhost_set_raster_depth", "trhost_download_depth",
    "trhost_load_scene_cached", "trhost_scene_list_sizes", "trhost_rccl_allreduce_max_u32", "trhost_load_gi_probes", "trhost_gi_probe_buffers",
    "trhost_set_renderer_queue", "trhost_render_graph_frame_stats",
]

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)   # trhost_allgather_fn


class ExchangeDesc(C.Structure):
    _fields_ = [("world", C.c_uint32), ("rank", C.c_uint32), ("slot_groups", C.c_uint32), ("group_capacity", C.c_uint32),
                ("list_capacity", C.c_uint64), ("pass_slot_mask", C.c_uint32), ("overlap", C.c_int),
                ("slots_allgather", C.c_void_p), ("slots_user", C.c_void_p), ("late_allgather", C.c_void_p), ("late_user", C.c_void_p),
                ("list_presence_mask", C.c_uint32), ("depth_allreduce_max", C.c_void_p), ("depth_user", C.c_void_p),
                ("slot_runs", C.c_uint32), ("global_group_capacity", C.c_uint32)]


DEPTH_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)   # trhost_exchange_desc.depth_allreduce_max
SHARD_LATE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int)   # trhost_shard_late_fn


class PassBuffers(C.Structure):
    _fields_ = [("ran", C.c_int), ("records", C.c_void_p), ("dispatch_args", C.c_void_p), ("vis_mask", C.c_void_p),
                ("visible_list", C.c_void_p), ("draw_args", C.c_void_p), ("late_count", C.c_void_p), ("late_args", C.c_void_p)]


class HostError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    rhi.load()   # libtrhip.so first (RPATH $ORIGIN also finds it)
    if not os.path.exists(LIB_PATH):
        raise HostError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.trhost_last_error.restype = C.c_char_p
    L.trhost_initialize.argtypes = [C.c_int, u32, u32, vp]
    L.trhost_shutdown.restype = None
    L.trhost_load_scene.argtypes = [vp, u32, vp, u32, vp, u64, vp, u32, vp, u32]
    L.trhost_upload_meshlets.argtypes = [u64, vp, u64]
    L.trhost_load_nodes.argtypes = [vp, u32, vp]
    L.trhost_set_node_transforms.argtypes = [vp, u32]
    L.trhost_set_instance_update_range.argtypes = [u32, u32]
    L.trhost_set_camera.argtypes = [vp, vp, vp, C.c_float]
    L.trhost_set_culling.argtypes = [C.c_int] * 5
    L.trhost_set_limits.argtypes = [u32, u64]
    L.trhost_upload_depth.argtypes = [vp, u32, u32]
    L.trhost_load_geometry.argtypes = [vp, u64, vp, u64, vp, u64]
    L.trhost_load_scene_cached.argtypes = [C.c_char_p, vp, u32, vp, u32, vp, u32]
    L.trhost_set_raster_depth.argtypes = [C.c_int]
    L.trhost_download_depth.argtypes = [vp, u64]
    L.trhost_upload_hzb_mip.argtypes = [u32, vp, u64]
    L.trhost_download_hzb_mip.argtypes = [u32, vp, u64]
    L.trhost_hzb_info.argtypes = [C.POINTER(u32)] * 3
    L.trhost_pass_buffers.argtypes = [u32, C.POINTER(PassBuffers)]
    L.trhost_instance_buffer.argtypes = [C.POINTER(vp)]
    L.trhost_device.restype = vp
    L.trhost_render_graph_stats.argtypes = [C.POINTER(u32), C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
    L.trhost_renderer_times.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.trhost_set_gpu_timers.argtypes = [C.c_int]
    L.trhost_exchange_create.argtypes = [C.POINTER(ExchangeDesc)]
    L.trhost_scene_list_sizes.argtypes = [C.POINTER(u32), C.POINTER(u32)]
    L.trhost_load_gi_probes.argtypes = [vp, vp, u32, C.c_float, C.c_int]
    L.trhost_set_renderer_queue.argtypes = [C.c_char_p, C.c_int]
    L.trhost_render_graph_frame_stats.argtypes = [C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
    L.trhost_gi_probe_buffers.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.trhost_exchange_outputs.argtypes = [u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.trhost_set_shard_late_exchange.argtypes = [SHARD_LATE_FN, vp]
    L.trhost_heap_sim.argtypes = [u64, vp, u32, vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise HostError(load().trhost_last_error().decode(errors="replace"))


def heap_sim(heap_size: int, ops):
    """RenderGraph::Heap allocator without a GPU (tests)."""
    ops = np.asarray(ops, np.int64)
    res = np.zeros(len(ops), np.uint64)
    used, peak, nb = C.c_uint64(), C.c_uint64(), C.c_uint32()
    _check(load().trhost_heap_sim(heap_size, ops.ctypes.data, len(ops), res.ctypes.data, C.byref(used), C.byref(peak), C.byref(nb)))
    return res, int(used.value), int(peak.value), int(nb.value)


def _download(handle, dtype, count):
    out = np.empty(count, dtype)
    if count:
        rc = rhi.load().trhip_buffer_download(C.c_void_p(handle), 0, out.ctypes.data, out.nbytes)
        if rc != 0:
            raise HostError(rhi.load().trhip_last_error().decode())
    return out


class Renderer:
    """One process-wide renderer context (Graphic / Scene are singletons in the reference)."""

    def __init__(self, render=(3840, 2160), device_index=0, stream: int | None = None, max_groups: int | None = None,
                 max_transient_bytes: int | None = None):
        L = load()
        _check(L.trhost_initialize(device_index, render[0], render[1], C.c_void_p(stream) if stream else None))
        self.render = render
        self.max_groups = 65535
        if max_groups or max_transient_bytes:
            _check(L.trhost_set_limits(max_groups or 0, max_transient_bytes or 0))
            if max_groups:
                self.max_groups = max_groups
        w, h, m = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(L.trhost_hzb_info(C.byref(w), C.byref(h), C.byref(m)))
        self.hzb_w, self.hzb_h, self.hzb_mips = w.value, h.value, m.value

    def load_scene(self, instances, meshData, meshlets, opaqueIds, alphaMaskIds, num_meshlets: int | None = None):
        """meshlets=None + num_meshlets: allocate only; stream the data in with upload_meshlets()."""
        a = [np.ascontiguousarray(x) for x in (instances, meshData)]
        ml = np.ascontiguousarray(meshlets) if meshlets is not None else None
        op = np.ascontiguousarray(opaqueIds, np.uint32)
        am = np.ascontiguousarray(alphaMaskIds, np.uint32)
        _check(load().trhost_load_scene(a[0].ctypes.data, len(a[0]), a[1].ctypes.data, len(a[1]),
                                        ml.ctypes.data if ml is not None else None, len(ml) if ml is not None else int(num_meshlets),
                                        op.ctypes.data if op.size else None, op.size, am.ctypes.data if am.size else None, am.size))
        self.num_opaque, self.num_alpha = op.size, am.size

    def load_scene_cached(self, cached_data_path: str, instances, opaqueIds, alphaMaskIds):
        """Meshes, meshlets and geometry from a `<scene>_CachedData.bin` v3 (read natively), instances / id lists from the caller."""
        a = np.ascontiguousarray(instances)
        op = np.ascontiguousarray(opaqueIds, np.uint32)
        am = np.ascontiguousarray(alphaMaskIds, np.uint32)
        _check(load().trhost_load_scene_cached(os.fsencode(cached_data_path), a.ctypes.data, len(a), op.ctypes.data if op.size else None, op.size,
                                               am.ctypes.data if am.size else None, am.size))
        self.num_opaque, self.num_alpha = op.size, am.size

    def upload_meshlets(self, first: int, meshlets):
        ml = np.ascontiguousarray(meshlets)
        _check(load().trhost_upload_meshlets(int(first), ml.ctypes.data, len(ml)))

    def load_nodes(self, nodes, prim_to_node):
        n = np.ascontiguousarray(nodes); p = np.ascontiguousarray(prim_to_node, np.uint32)
        _check(load().trhost_load_nodes(n.ctypes.data, len(n), p.ctypes.data))

    def set_node_transforms(self, nodes):
        n = np.ascontiguousarray(nodes)
        _check(load().trhost_set_node_transforms(n.ctypes.data, len(n)))

    def set_instance_update_range(self, first: int, count: int):
        """Multi-GPU: the per-frame transform update covers instances [first, first + count) only (this rank's shard)."""
        _check(load().trhost_set_instance_update_range(int(first), int(count)))

    def set_camera(self, view):
        w = np.ascontiguousarray(view.worldToView, np.float32); p = np.ascontiguousarray(view.prevWorldToView, np.float32)
        c = np.ascontiguousarray(view.viewToClip, np.float32)
        _check(load().trhost_set_camera(w.ctypes.data, p.ctypes.data, c.ctypes.data, float(view.nearPlane)))

    def set_culling(self, flags=7, freeze=False, force_mesh_lod=-1):
        _check(load().trhost_set_culling(flags & 1, (flags >> 1) & 1, (flags >> 2) & 1, int(freeze), int(force_mesh_lod)))
        self.flags = flags

    def upload_depth(self, depth):
        d = np.ascontiguousarray(depth, np.float32)
        _check(load().trhost_upload_depth(d.ctypes.data, d.shape[1], d.shape[0]))

    def load_geometry(self, vertices, meshlet_vertex_ids, meshlet_triangles):
        """The buffers the mesh shader reads (RawVertexFormat vertices, meshlet vertex ids, packed meshlet triangles)."""
        from . import interop as I
        v = np.ascontiguousarray(vertices, I.RawVertexFormat)
        vid, tri = np.ascontiguousarray(meshlet_vertex_ids, np.uint32), np.ascontiguousarray(meshlet_triangles, np.uint32)
        _check(load().trhost_load_geometry(v.ctypes.data, len(v), vid.ctypes.data, len(vid), tri.ctypes.data, len(tri)))

    def load_gi_probes(self, positions, states, radius: float, hide_inactive: bool = False):
        """GI debug view: probe world positions [n,3] and states [n] (1 = inactive); GIDebugRenderer culls them every frame."""
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3); st = np.ascontiguousarray(states, np.float32)
        assert len(p) == len(st)
        _check(load().trhost_load_gi_probes(p.ctypes.data if len(p) else None, st.ctypes.data if len(st) else None, len(p), float(radius), int(hide_inactive)))

    def gi_probe_results(self):
        """(positions[k,3], DrawIndexedIndirectArguments (5 words), instance -> probe index [k]) of the last frame."""
        self.wait_idle()
        h = [C.c_void_p() for _ in range(3)]
        _check(load().trhost_gi_probe_buffers(*[C.byref(x) for x in h]))
        args = _download(h[1].value, np.uint32, 5)
        k = int(args[1])
        return _download(h[0].value, np.float32, 3 * k).reshape(-1, 3), args, _download(h[2].value, np.uint32, k)

    def set_raster_depth(self, on: bool = True):
        """The frame rasterises the depth of its own visible meshlets instead of taking the uploaded depth image."""
        _check(load().trhost_set_raster_depth(int(on)))

    def download_depth(self) -> np.ndarray:
        self.wait_idle()
        d = np.empty((self.render[1], self.render[0]), np.float32)
        _check(load().trhost_download_depth(d.ctypes.data, d.nbytes))
        return d

    def upload_hzb(self, texels, offsets):
        for k in range(self.hzb_mips):
            mw, mh = max(self.hzb_w >> k, 1), max(self.hzb_h >> k, 1)
            t = np.ascontiguousarray(texels[offsets[k]:offsets[k] + mw * mh], np.uint16)
            _check(load().trhost_upload_hzb_mip(k, t.ctypes.data, t.nbytes))

    def download_hzb(self) -> np.ndarray:
        parts = []
        for k in range(self.hzb_mips):
            t = np.empty(max(self.hzb_w >> k, 1) * max(self.hzb_h >> k, 1), np.uint16)
            _check(load().trhost_download_hzb_mip(k, t.ctypes.data, t.nbytes))
            parts.append(t)
        return np.concatenate(parts)

    def frame(self):
        _check(load().trhost_frame())

    def wait_idle(self):
        _check(load().trhost_wait_idle())

    def set_gpu_timers(self, enable: bool):
        _check(load().trhost_set_gpu_timers(int(bool(enable))))

    def set_shard_late_exchange(self, fn):
        """Multi-GPU hook (include/trhost.h): fn(hip_stream, late_count_ptr, shard_info_ptr, bucket, phase) runs inside
        frame(): phase 0 after each early instance cull, phase 1 before each late one; None removes it."""
        if fn is None:
            self._shard_late_cb = C.cast(None, SHARD_LATE_FN)
        else:
            self._shard_late_cb = SHARD_LATE_FN(lambda _u, s, c, i, b, ph: fn(int(s or 0), int(c), int(i), int(b), int(ph)))
        _check(load().trhost_set_shard_late_exchange(self._shard_late_cb, None))

    def pass_buffers(self, slot: int) -> PassBuffers:
        pb = PassBuffers()
        _check(load().trhost_pass_buffers(slot, C.byref(pb)))
        return pb

    def results(self):
        """Read back every pass slot of the last frame (tests)."""
        self.wait_idle()
        out = {}
        late = None
        for s in range(4):
            pb = self.pass_buffers(s)
            if not pb.ran:
                out[s] = None
                continue
            args = _download(pb.dispatch_args, np.uint32, 4)
            G = int(min(args[0], args[3], self.max_groups))
            draw = _download(pb.draw_args, np.uint32, 3)
            V = int(min(draw[0], self.max_groups * 32))
            out[s] = dict(dispatchArgs=args[:3].copy(), validRecords=int(args[3]),
                          records=_download(pb.records, I.MeshletAmplificationData, G),
                          visMask=_download(pb.vis_mask, np.uint32, G),
                          visibleList=_download(pb.visible_list, np.uint32, V), drawArgs=draw)
            late = pb
        if late is not None and (self.flags & 2):
            out["lateCount"] = int(_download(late.late_count, np.uint32, 1)[0])
            out["lateArgs"] = _download(late.late_args, np.uint32, 3)
        return out

    def instances(self, count) -> np.ndarray:
        h = C.c_void_p()
        _check(load().trhost_instance_buffer(C.byref(h)))
        self.wait_idle()
        return _download(h.value, I.BasePassInstanceConstants, count)

    def render_graph_stats(self):
        nh, res, used, npass = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        _check(load().trhost_render_graph_stats(C.byref(nh), C.byref(res), C.byref(used), C.byref(npass)))
        return dict(heaps=nh.value, reserved=res.value, used=used.value, passes=npass.value)

    def set_renderer_queue(self, name: str, compute: bool):
        """Async compute: the renderer records for the compute queue (a second stream) instead of the graphics queue."""
        _check(load().trhost_set_renderer_queue(name.encode(), int(bool(compute))))

    def render_graph_frame_stats(self):
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(load().trhost_render_graph_frame_stats(C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(compute_queue_passes=a.value, cross_queue_waits=b.value, transient_bytes=c.value, aliased_bytes=d.value)

    def renderer_times(self, name: str):
        c, g = C.c_float(), C.c_float()
        _check(load().trhost_renderer_times(name.encode(), C.byref(c), C.byref(g)))
        return float(c.value), float(g.value)

    def device(self) -> int:
        return load().trhost_device()

    def shutdown(self):
        load().trhost_shutdown()
