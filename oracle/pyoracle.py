"""ctypes front-end of the CPU oracle (oracle/tr_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/tr_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (toyrenderer_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtr_oracle.so")

MAX_MIPS = 16

# ---- wire formats (reference source/shaders/ShaderInterop.h; sizes SURVEY.md section 9) ----
MATRIX = np.dtype((np.float32, (4, 4)))
INSTANCE_DT = np.dtype([("world", np.float32, (4, 4)), ("prevWorld", np.float32, (4, 4)),
                        ("meshDataIdx", np.uint32), ("materialDataIdx", np.uint32), ("pad", np.float32, (2,))])
MESHLOD_DT = np.dtype([("meshletDataBufferIdx", np.uint32), ("numMeshlets", np.uint32),
                       ("error", np.float32), ("pad", np.uint32)])
MESHDATA_DT = np.dtype([("sphere", np.float32, (4,)), ("lods", MESHLOD_DT, (8,)), ("numLODs", np.uint32),
                        ("globalVertexBufferIdx", np.uint32), ("globalIndexBufferIdx", np.uint32)])
MESHLET_DT = np.dtype([("sphere", np.float32, (4,)), ("cone", np.uint32), ("vertexIDsIdx", np.uint32),
                       ("indexIDsIdx", np.uint32), ("vertexAndTriangleCount", np.uint32)])
RECORD_DT = np.dtype([("instanceConstIdx", np.uint32), ("meshLOD", np.uint32), ("meshletGroupOffset", np.uint32)])
NODE_DT = np.dtype([("parent", np.uint32), ("position", np.float32, (3,)), ("rotation", np.float32, (4,)),
                    ("scale", np.float32, (3,)), ("pad", np.uint32)])
CULLCONSTS_DT = np.dtype([("nbInstances", np.uint32), ("cullingFlags", np.uint32), ("hzbDim", np.uint32, (2,)),
                          ("frustum", np.float32, (4,)), ("worldToView", np.float32, (4, 4)),
                          ("prevWorldToView", np.float32, (4, 4)), ("nearPlane", np.float32), ("P00", np.float32),
                          ("P11", np.float32), ("forcedMeshLOD", np.uint32), ("meshLODTarget", np.float32)])
BASEPASSCONSTS_DT = np.dtype([("worldToClip", np.float32, (4, 4)), ("prevWorldToClip", np.float32, (4, 4)),
                              ("worldToView", np.float32, (4, 4)), ("frustum", np.float32, (4,)),
                              ("hzbDim", np.uint32, (2,)), ("P00", np.float32), ("P11", np.float32),
                              ("nearPlane", np.float32), ("cullingFlags", np.uint32), ("debugMode", np.uint32),
                              ("pad0", np.uint32), ("outputResolution", np.uint32, (2,)),
                              ("visualizeMinMip", np.uint32), ("writeSamplerFeedback", np.uint32)])
assert INSTANCE_DT.itemsize == 144 and MESHDATA_DT.itemsize == 156 and MESHLOD_DT.itemsize == 16
assert MESHLET_DT.itemsize == 32 and RECORD_DT.itemsize == 12 and NODE_DT.itemsize == 48
assert CULLCONSTS_DT.itemsize == 180 and BASEPASSCONSTS_DT.itemsize == 256


class HZB(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mips", C.c_uint32), ("pad", C.c_uint32),
                ("texels", C.c_void_p), ("mipOffset", C.c_uint64 * MAX_MIPS)]


class FrameDesc(C.Structure):
    _fields_ = [("instances", C.c_void_p), ("meshData", C.c_void_p), ("meshlets", C.c_void_p),
                ("opaqueIds", C.c_void_p), ("numOpaque", C.c_uint32),
                ("alphaMaskIds", C.c_void_p), ("numAlphaMask", C.c_uint32),
                ("worldToView", C.c_float * 16), ("prevWorldToView", C.c_float * 16), ("viewToClip", C.c_float * 16),
                ("nearPlane", C.c_float), ("renderHeight", C.c_uint32),
                ("cullingFlags", C.c_uint32), ("forceMeshLOD", C.c_int32), ("freezeCullingCamera", C.c_uint32),
                ("maxGroups", C.c_uint32),
                ("hzbTexels", C.c_void_p), ("hzbW", C.c_uint32), ("hzbH", C.c_uint32), ("hzbMips", C.c_uint32),
                ("hzbMipOffset", C.c_uint64 * MAX_MIPS),
                ("depth", C.c_void_p), ("depthW", C.c_uint32), ("depthH", C.c_uint32),
                ("threads", C.c_uint32),
                ("shardLate", C.c_uint32), ("shardLateBase", C.c_uint32 * 2), ("shardLateTotal", C.c_uint32 * 2),
                ("rasterDepth", C.c_uint32), ("worldToClip", C.c_float * 16),
                ("vertices", C.c_void_p), ("meshletVertexIds", C.c_void_p), ("meshletTriangles", C.c_void_p)]


class FrameOut(C.Structure):
    _fields_ = [("records", C.c_void_p * 4), ("recordCapacity", C.c_uint32),
                ("dispatchArgs", (C.c_uint32 * 3) * 4), ("validRecords", C.c_uint32 * 4),
                ("visMask", C.c_void_p * 4), ("visibleList", C.c_void_p * 4), ("listCapacity", C.c_uint64),
                ("drawArgs", (C.c_uint32 * 3) * 4), ("lateCount", C.c_uint32 * 2), ("lateIds", C.c_void_p * 2),
                ("lateArgs", (C.c_uint32 * 3) * 2), ("meshletsTested", C.c_uint64 * 4), ("passRan", C.c_uint32 * 4)]


def build(force: bool = False) -> str:
    """Compile oracle/tr_oracle.c -> oracle/_build/libtr_oracle.so (gcc, seconds)."""
    src = [os.path.join(_HERE, f) for f in ("tr_oracle.c", "tr_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_hzb_layout.restype = C.c_uint32
        L.orc_hzb_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_frustum_cull.restype = C.c_int
        L.orc_frustum_cull.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.orc_occlusion_cull.restype = C.c_int
        L.orc_occlusion_cull.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_cone_cull.restype = C.c_int
        L.orc_cone_cull.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_float]
        L.orc_sample_hzb_min.restype = C.c_float
        L.orc_sample_hzb_min.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.orc_hzb_level.restype = C.c_int
        L.orc_hzb_level.argtypes = [C.c_float, C.c_float, C.c_uint32]
        L.orc_max_scale.restype = C.c_float
        L.orc_max_scale.argtypes = [C.c_void_p]
        L.orc_sphere_to_world.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_to_view.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_occlusion_footprints.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_float,
                                               C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_unpack_cone_view.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_select_lod.restype = C.c_uint32
        L.orc_select_lod.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_uint32, C.c_float]
        L.orc_make_world_matrix.argtypes = [C.c_void_p] * 4
        L.orc_culling_frustum.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_instance_cull.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_void_p]
        L.orc_build_late_args.argtypes = [C.c_uint32, C.c_void_p]
        L.orc_gi_probe_cull.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_gi_probe_cull.restype = C.c_uint32
        L.orc_meshlet_cull.restype = C.c_uint64
        L.orc_meshlet_cull.argtypes = [C.c_void_p] * 5 + [C.c_uint32, C.c_uint32] + [C.c_void_p] * 4
        L.orc_hzb_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_void_p]
        L.orc_update_instance_consts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_frame.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data if a is not None else None


def f32_to_f16(x: float) -> int:
    return int(lib().orc_f32_to_f16(float(np.float32(x))))


def f16_to_f32(h: int) -> float:
    return float(lib().orc_f16_to_f32(int(h)))


def hzb_layout(w: int, h: int):
    offs = (C.c_uint64 * MAX_MIPS)()
    total = C.c_uint64(0)
    mips = lib().orc_hzb_layout(w, h, offs, C.byref(total))
    return mips, [int(offs[i]) for i in range(mips)], int(total.value)


class HzbTexture:
    """R16F mip chain in one uint16 array (layout: OrcHZB in tr_oracle.h)."""

    def __init__(self, w: int, h: int, texels: np.ndarray | None = None):
        self.w, self.h = int(w), int(h)
        self.mips, self.offsets, self.total = hzb_layout(w, h)
        self.texels = np.zeros(self.total, np.uint16) if texels is None else np.ascontiguousarray(texels, np.uint16)
        assert self.texels.size == self.total

    def struct(self) -> HZB:
        s = HZB()
        s.width, s.height, s.mips = self.w, self.h, self.mips
        s.texels = _p(self.texels)
        for i, o in enumerate(self.offsets):
            s.mipOffset[i] = o
        return s

    def mip(self, k: int) -> np.ndarray:
        mw, mh = max(self.w >> k, 1), max(self.h >> k, 1)
        return self.texels[self.offsets[k]:self.offsets[k] + mw * mh].reshape(mh, mw)

    def build_from_depth(self, depth: np.ndarray):
        depth = np.ascontiguousarray(depth, np.float32)
        H, W = depth.shape
        offs = (C.c_uint64 * MAX_MIPS)(*self.offsets)
        lib().orc_hzb_build(_p(depth), W, H, _p(self.texels), self.w, self.h, self.mips, offs)


def frustum_cull(c, r, f) -> bool:
    c = np.ascontiguousarray(c, np.float32); f = np.ascontiguousarray(f, np.float32)
    return bool(lib().orc_frustum_cull(_p(c), float(np.float32(r)), _p(f)))


def occlusion_cull(c, r, near, P00, P11, hzb: HzbTexture) -> bool:
    c = np.ascontiguousarray(c, np.float32)
    s = hzb.struct()
    return bool(lib().orc_occlusion_cull(_p(c), float(np.float32(r)), float(np.float32(near)),
                                         float(np.float32(P00)), float(np.float32(P11)), C.addressof(s)))


def cone_cull(c, r, axis, cutoff) -> bool:
    c = np.ascontiguousarray(c, np.float32); axis = np.ascontiguousarray(axis, np.float32)
    return bool(lib().orc_cone_cull(_p(c), float(np.float32(r)), _p(axis), float(np.float32(cutoff))))


def sample_hzb_min(hzb: HzbTexture, u, v, level) -> float:
    s = hzb.struct()
    return float(lib().orc_sample_hzb_min(C.addressof(s), float(np.float32(u)), float(np.float32(v)), float(level)))


def hzb_level(width, height, mips) -> int:
    return int(lib().orc_hzb_level(float(np.float32(width)), float(np.float32(height)), int(mips)))


def max_scale(world) -> float:
    w = np.ascontiguousarray(world, np.float32)
    return float(lib().orc_max_scale(_p(w)))


def unpack_cone_view(packed: int, world, worldToView):
    w = np.ascontiguousarray(world, np.float32); v = np.ascontiguousarray(worldToView, np.float32)
    axis = np.zeros(3, np.float32); cutoff = np.zeros(1, np.float32)
    lib().orc_unpack_cone_view(int(packed), _p(w), _p(v), _p(axis), _p(cutoff))
    return axis, float(cutoff[0])


def make_world_matrix(pos, rot, scale) -> np.ndarray:
    p = np.ascontiguousarray(pos, np.float32); q = np.ascontiguousarray(rot, np.float32)
    s = np.ascontiguousarray(scale, np.float32); out = np.zeros((4, 4), np.float32)
    lib().orc_make_world_matrix(_p(p), _p(q), _p(s), _p(out))
    return out


def culling_frustum(viewToClip) -> np.ndarray:
    m = np.ascontiguousarray(viewToClip, np.float32); out = np.zeros(4, np.float32)
    lib().orc_culling_frustum(_p(m), _p(out))
    return out


def instance_cull(consts: np.ndarray, late: bool, instances, ids, meshData, hzb: HzbTexture | None,
                  records, dispatchArgs, lateCount, lateIds, lateArgsX: int, maxGroups: int = 65535):
    """In-place on records/dispatchArgs/lateCount/lateIds (numpy). Returns validRecords."""
    s = hzb.struct() if hzb is not None else HZB()
    valid = C.c_uint32(0)
    lib().orc_instance_cull(_p(consts), int(late), _p(instances), _p(ids), _p(meshData),
                            C.addressof(s), _p(records), _p(dispatchArgs), _p(lateCount), _p(lateIds),
                            int(lateArgsX), int(maxGroups), C.addressof(valid))
    return int(valid.value)


def gi_probe_cull(consts: np.ndarray, positions, states, hzb: HzbTexture, draw_args=None):
    """CS_VisualizeGIProbesCulling (giprobevisualization.hlsl:16-69) with probe positions / states as inputs.
    consts: interop.GIProbeVisualizationUpdateConsts[1].  Returns (positions[k,3], drawArgs[5], instanceToProbe[k])."""
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    st = np.ascontiguousarray(states, np.float32)
    n = int(consts["m_NumProbes"][0])
    assert len(pos) >= n and len(st) >= n
    args = np.zeros(5, np.uint32) if draw_args is None else np.ascontiguousarray(draw_args, np.uint32).copy()
    base = int(args[1])
    out_pos = np.zeros((base + n, 3), np.float32)
    out_idx = np.zeros(base + n, np.uint32)
    h = hzb.struct()
    k = lib().orc_gi_probe_cull(_p(consts), _p(pos), _p(st), C.byref(h), _p(out_pos), _p(args), _p(out_idx))
    return out_pos[base:base + k], args, out_idx[base:base + k]


def build_late_args(count: int) -> np.ndarray:
    out = np.zeros(3, np.uint32)
    lib().orc_build_late_args(int(count), _p(out))
    return out


def meshlet_cull(consts: np.ndarray, instances, meshData, meshlets, records, g0: int, g1: int,
                 hzb: HzbTexture | None, want_list: bool = True):
    """Returns (visMask[g1], visibleList, tested)."""
    s = hzb.struct() if hzb is not None else HZB()
    mask = np.zeros(max(g1, 1), np.uint32)
    lst = np.zeros(max((g1 - g0) * 32, 1), np.uint32) if want_list else None
    cur = C.c_uint64(0)
    tested = lib().orc_meshlet_cull(_p(consts), _p(instances), _p(meshData), _p(meshlets), _p(records),
                                    int(g0), int(g1), C.addressof(s), _p(mask),
                                    _p(lst) if want_list else None, C.addressof(cur) if want_list else None)
    return mask[:g1], (lst[:cur.value] if want_list else None), int(tested)


def update_instance_consts(nodes, primToNode, instances):
    lib().orc_update_instance_consts(_p(nodes), _p(primToNode), _p(instances), len(instances))


def occlusion_footprints(centres: np.ndarray, radii: np.ndarray, view: dict, hzb_dims) -> np.ndarray:
    """Test helper: per sphere (world-space centre, identity instance transform) {level, x0, y0, fracX == 0, fracY == 0}
    of its occlusion lookup (tr_oracle.c orc_occlusion_footprints)."""
    c = np.ascontiguousarray(centres, np.float32).reshape(-1, 3)
    r = np.ascontiguousarray(radii, np.float32).reshape(-1)
    assert len(c) == len(r)
    V = np.ascontiguousarray(view["worldToView"], np.float32).reshape(4, 4)
    P = np.ascontiguousarray(view["viewToClip"], np.float32).reshape(4, 4)
    hw, hh = int(hzb_dims[0]), int(hzb_dims[1])
    mips = max(hw, hh).bit_length()
    out = np.zeros((len(c), 5), np.int32)
    lib().orc_occlusion_footprints(_p(c), _p(r), len(c), _p(V), float(P[0, 0]), float(P[1, 1]), hw, hh, mips, _p(out))
    return out


RawVertexFormat = np.dtype([("m_Position", np.float32, (3,)), ("m_PackedNormal", np.uint32), ("m_TexCoord", np.uint16, (2,))])


def raster_depth(consts: np.ndarray, scene: dict, vertices, meshletVertexIds, meshletTriangles, records, visibleList, depth: np.ndarray):
    """orc_raster_depth: max-merges the depth of the listed visible meshlets into `depth` (float32 [H, W], in place).
    consts: BasePassConstants (m_WorldToClip, m_OutputResolution, m_NearPlane are read)."""
    k = np.ascontiguousarray(consts)
    v = np.ascontiguousarray(vertices, RawVertexFormat)
    vid = np.ascontiguousarray(meshletVertexIds, np.uint32)
    tri = np.ascontiguousarray(meshletTriangles, np.uint32)
    rec = np.ascontiguousarray(records)
    lst = np.ascontiguousarray(visibleList, np.uint32)
    assert depth.dtype == np.float32 and depth.flags.c_contiguous
    assert depth.shape == (int(k["m_OutputResolution"].reshape(-1)[1]), int(k["m_OutputResolution"].reshape(-1)[0]))
    L = lib()
    L.orc_raster_depth.argtypes = [C.c_void_p] * 9 + [C.c_uint32, C.c_void_p]
    L.orc_raster_depth(_p(k), _p(scene["instances"]), _p(scene["meshData"]), _p(scene["meshlets"]), _p(v), _p(vid), _p(tri), _p(rec), _p(lst),
                       len(lst), _p(depth))
    return depth


class FrameResult:
    pass


def frame(scene: dict, view: dict, hzb: HzbTexture, depth: np.ndarray | None, *, cullingFlags=7, forceMeshLOD=-1,
          freeze=False, maxGroups=65535, threads=1, record_capacity=None, list_capacity=None, shard_late=None,
          raster=None) -> FrameResult:
    """Run BasePassRenderer::RenderBasePass on the CPU.  `scene`: instances, meshData, meshlets,
    opaqueIds, alphaMaskIds (numpy, wire dtypes).  `view`: worldToView, prevWorldToView, viewToClip,
    nearPlane, renderHeight.  hzb is updated in place (it is the previous frame's on entry)."""
    inst, md, ml = scene["instances"], scene["meshData"], scene["meshlets"]
    op = np.ascontiguousarray(scene.get("opaqueIds", np.zeros(0, np.uint32)), np.uint32)
    am = np.ascontiguousarray(scene.get("alphaMaskIds", np.zeros(0, np.uint32)), np.uint32)
    d = FrameDesc()
    d.instances, d.meshData, d.meshlets = _p(inst), _p(md), _p(ml)
    d.opaqueIds, d.numOpaque = _p(op) if op.size else None, op.size
    d.alphaMaskIds, d.numAlphaMask = _p(am) if am.size else None, am.size
    for name in ("worldToView", "prevWorldToView", "viewToClip"):
        m = np.ascontiguousarray(view[name], np.float32).reshape(16)
        getattr(d, name)[:] = [float(x) for x in m]
    d.nearPlane = float(view["nearPlane"]); d.renderHeight = int(view["renderHeight"])
    d.cullingFlags, d.forceMeshLOD, d.freezeCullingCamera, d.maxGroups = int(cullingFlags), int(forceMeshLOD), int(freeze), int(maxGroups)
    d.hzbTexels, d.hzbW, d.hzbH, d.hzbMips = _p(hzb.texels), hzb.w, hzb.h, hzb.mips
    for i, o in enumerate(hzb.offsets):
        d.hzbMipOffset[i] = o
    if depth is not None:
        assert raster is None or (depth.dtype == np.float32 and depth.flags.c_contiguous and depth.flags.writeable)
        depth = np.ascontiguousarray(depth, np.float32)
        d.depth, d.depthH, d.depthW = _p(depth), depth.shape[0], depth.shape[1]
    if raster is not None:
        w2c, rv, rvid, rtri = raster
        rv = np.ascontiguousarray(rv, RawVertexFormat)
        rvid, rtri = np.ascontiguousarray(rvid, np.uint32), np.ascontiguousarray(rtri, np.uint32)
        d.rasterDepth = 1
        d.worldToClip[:] = [float(x) for x in np.ascontiguousarray(w2c, np.float32).reshape(16)]
        d.vertices, d.meshletVertexIds, d.meshletTriangles = _p(rv), _p(rvid), _p(rtri)
    d.threads = int(threads)
    if shard_late is not None:      # multi-GPU checker: ((base_opaque, total_opaque), (base_alpha, total_alpha))
        d.shardLate = 1
        for b in (0, 1):
            d.shardLateBase[b], d.shardLateTotal[b] = int(shard_late[b][0]), int(shard_late[b][1])

    if record_capacity is None:
        lods = md[md.dtype.names[1]]
        lod_groups = (lods[lods.dtype.names[1]].astype(np.uint64) + 31) // 32
        per_mesh = lod_groups.max(axis=1) if len(md) else np.zeros(0, np.uint64)
        record_capacity = int(per_mesh[inst[inst.dtype.names[2]]].sum()) + 1 if len(inst) else 1
        record_capacity = min(record_capacity, max(int(maxGroups), 1))
    if list_capacity is None:
        list_capacity = record_capacity * 32
    o = FrameOut()
    res = FrameResult()
    res.records = [np.zeros(record_capacity, RECORD_DT) for _ in range(4)]
    res.visMask = [np.zeros(record_capacity, np.uint32) for _ in range(4)]
    res.visibleList = [np.zeros(list_capacity, np.uint32) for _ in range(4)]
    res.lateIds = [np.zeros(max(op.size, 1), np.uint32), np.zeros(max(am.size, 1), np.uint32)]
    for s in range(4):
        o.records[s], o.visMask[s], o.visibleList[s] = _p(res.records[s]), _p(res.visMask[s]), _p(res.visibleList[s])
    o.lateIds[0], o.lateIds[1] = _p(res.lateIds[0]), _p(res.lateIds[1])
    o.recordCapacity, o.listCapacity = record_capacity, list_capacity
    lib().orc_frame(C.addressof(d), C.addressof(o))
    res.dispatchArgs = np.array([[o.dispatchArgs[s][i] for i in range(3)] for s in range(4)], np.uint32)
    res.drawArgs = np.array([[o.drawArgs[s][i] for i in range(3)] for s in range(4)], np.uint32)
    res.validRecords = np.array([o.validRecords[s] for s in range(4)], np.uint32)
    res.lateCount = np.array([o.lateCount[0], o.lateCount[1]], np.uint32)
    res.lateArgs = np.array([[o.lateArgs[s][i] for i in range(3)] for s in range(2)], np.uint32)
    res.meshletsTested = np.array([o.meshletsTested[s] for s in range(4)], np.uint64)
    res.passRan = np.array([o.passRan[s] for s in range(4)], np.uint32)
    for s in range(4):
        G = int(min(res.dispatchArgs[s][0], res.validRecords[s]))
        res.records[s] = res.records[s][:G]
        res.visMask[s] = res.visMask[s][:G]
        res.visibleList[s] = res.visibleList[s][:min(int(res.drawArgs[s][0]), list_capacity)]
    res.lateIds[0] = res.lateIds[0][:int(res.lateCount[0])]
    res.lateIds[1] = res.lateIds[1][:int(res.lateCount[1])]
    return res
