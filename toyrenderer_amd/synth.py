"""Synthetic scenes for the visibility path (SURVEY.md 8(d) "Synthetic inputs").

Deterministic, chunked numpy generators (PCG64 keyed by (seed, chunk)) so that a 100 M-meshlet
scene can be streamed to the GPU chunk by chunk and any chunk can be regenerated on the host for
an oracle spot check.  Value ranges follow the reference's producers: cone packing
Visual.cpp:421-431 (axis u8 = (a+1)/2*255 truncated, cutoff u8 = 2*cone_cutoff_s8, even, <=254),
LOD error accumulation Visual.cpp:488, id lists Scene.cpp:282-362, projection Scene.cpp:124-125 +
MathUtilities.cpp:10-16 (RH, reverse-Z, infinite far).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import interop as I

SEED_SCENE = 0x5EED0001
SEED_CAMERA = 0x5EED0002


# ----------------------------------------------------------------------------- camera / view
def perspective_rh_reverse_z_infinite(fov_y: float, aspect: float, near: float) -> np.ndarray:
    """XMMatrixPerspectiveFovRH (SimpleMath.inl:2193-2199) followed by ModifyPerspectiveMatrix
    (MathUtilities.cpp:3-38) with reverse-Z + infinite far: _33 = 0, _43 = near."""
    h = np.float32(1.0) / np.float32(math.tan(0.5 * fov_y))
    w = np.float32(h / np.float32(aspect))
    P = np.zeros((4, 4), np.float32)
    P[0, 0], P[1, 1] = w, h
    P[2, 2], P[2, 3] = 0.0, -1.0
    P[3, 2] = near
    return P


def quat_to_matrix(q) -> np.ndarray:
    x, y, z, w = [float(v) for v in q]
    R = np.eye(4, dtype=np.float64)
    R[0, :3] = [1 - 2 * y * y - 2 * z * z, 2 * x * y + 2 * z * w, 2 * x * z - 2 * y * w]
    R[1, :3] = [2 * x * y - 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z + 2 * x * w]
    R[2, :3] = [2 * x * z + 2 * y * w, 2 * y * z - 2 * x * w, 1 - 2 * x * x - 2 * y * y]
    return R


def world_to_view(eye, orientation_quat) -> np.ndarray:
    """Scene.cpp:121-122: ViewToWorld = R(q) * T(eye); WorldToView = inverse."""
    M = quat_to_matrix(orientation_quat)
    M[3, :3] = np.asarray(eye, np.float64)
    return np.linalg.inv(M).astype(np.float32)


@dataclass
class View:
    worldToView: np.ndarray
    prevWorldToView: np.ndarray
    viewToClip: np.ndarray
    nearPlane: float
    renderW: int
    renderH: int

    @property
    def hzb_dims(self):
        return I.hzb_dims(self.renderW, self.renderH)

    def as_dict(self):
        return dict(worldToView=self.worldToView, prevWorldToView=self.prevWorldToView, viewToClip=self.viewToClip,
                    nearPlane=self.nearPlane, renderHeight=self.renderH, renderWidth=self.renderW)


def make_view(eye=(0.0, 0.0, 0.0), yaw=0.0, prev_eye=None, prev_yaw=None, fov_deg=45.0, render=(3840, 2160), near=0.1) -> View:
    """Camera of SURVEY 8(d): RH, looks down -Z, fovY 45 deg (Scene.h:52), near 0.1."""
    def q(y):
        return (0.0, math.sin(0.5 * y), 0.0, math.cos(0.5 * y))
    prev_eye = eye if prev_eye is None else prev_eye
    prev_yaw = yaw if prev_yaw is None else prev_yaw
    P = perspective_rh_reverse_z_infinite(math.radians(fov_deg), render[0] / render[1], near)
    return View(world_to_view(eye, q(yaw)), world_to_view(prev_eye, q(prev_yaw)), P, float(np.float32(near)), render[0], render[1])


# ----------------------------------------------------------------------------- scene description
@dataclass
class SceneSpec:
    """K unique meshes x instances.  `unique=True` gives every instance its own mesh (C3/C4:
    every meshlet is read exactly once per frame -> HBM-bound by construction)."""
    num_meshes: int
    num_instances: int
    meshlets_lod0: int = 128          # LOD0 meshlets per mesh (may be jittered)
    jitter_meshlets: bool = False     # +-25 % per mesh, ragged last group
    max_lods: int = 1                 # 1..8
    unique: bool = False
    alpha_mask_fraction: float = 0.0
    nonuniform_scale_fraction: float = 0.1
    # instance placement box in front of a camera at the origin looking down -Z
    box_x: float = 100.0
    box_y: float = 56.0
    z_near: float = 5.0
    z_far: float = 200.0
    mesh_size: float = 2.0
    seed: int = SEED_SCENE
    chunk_meshes: int = 32768
    # order of the meshlets inside a (mesh, LOD) range: "random" (positions drawn independently: consecutive meshlets are
    # anywhere in the mesh -- the BASELINE configs) or "morton" (the same meshlets sorted along a Z-order curve through the
    # mesh's box, the spatial coherence a meshlet builder's output has: a group of 32 covers a fraction of the mesh)
    meshlet_order: str = "random"


def _rng(seed: int, stream: int, chunk: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed) & 0xFFFFFFFF, int(stream), int(chunk)]))


def lod_meshlet_counts(spec: SceneSpec, m0: np.ndarray, nl: np.ndarray) -> np.ndarray:
    """[K, 8] meshlets per LOD: LOD k ~ m0 * 0.65^k, at least 1."""
    k = np.arange(I.kMaxNumMeshLODs)
    c = np.maximum(1, np.floor(m0[:, None] * (0.65 ** k)[None, :]).astype(np.int64))
    c[k[None, :] >= nl[:, None]] = 0
    return c


def gen_mesh_table(spec: SceneSpec):
    """MeshData[K] plus the global meshlet count.  Cheap (no per-meshlet work)."""
    K = spec.num_meshes
    r = _rng(spec.seed, 1, 0)
    m0 = np.full(K, spec.meshlets_lod0, np.int64)
    if spec.jitter_meshlets:
        m0 = np.maximum(1, (m0 * r.uniform(0.75, 1.25, K)).astype(np.int64))
    nl = np.ones(K, np.int64) if spec.max_lods <= 1 else r.integers(1, spec.max_lods + 1, K)
    counts = lod_meshlet_counts(spec, m0, nl)
    per_mesh = counts.sum(axis=1)
    base = np.concatenate([[0], np.cumsum(per_mesh)[:-1]])
    lod_base = base[:, None] + np.concatenate([np.zeros((K, 1), np.int64), np.cumsum(counts, axis=1)[:, :-1]], axis=1)
    total = int(per_mesh.sum())
    assert total < 2 ** 32
    md = np.zeros(K, I.MeshData)
    radius = np.float32(spec.mesh_size * (0.5 * math.sqrt(3.0) + 0.2))
    md["m_BoundingSphere"][:, 3] = radius
    md["m_NumLODs"] = nl
    md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"] = np.where(counts > 0, lod_base, 0)
    md["m_MeshLODDatas"]["m_NumMeshlets"] = counts
    # error: LOD0 0, then growth x1.5 from 1e-3*radius (Visual.cpp:488)
    err = np.zeros((K, 8), np.float32)
    e = np.float32(1e-3) * radius
    for k in range(1, 8):
        err[:, k] = e
        e = np.float32(e * np.float32(1.5))
    err[np.arange(8)[None, :] >= nl[:, None]] = 0
    md["m_MeshLODDatas"]["m_Error"] = err
    return md, total


def gen_meshlets_for_meshes(spec: SceneSpec, md: np.ndarray, mesh_begin: int, mesh_end: int) -> np.ndarray:
    """MeshletData for meshes [mesh_begin, mesh_end) -- contiguous in the global meshlet buffer
    starting at md[mesh_begin].lod[0].meshletDataBufferIdx.  mesh_begin must be a multiple of
    spec.chunk_meshes (chunk-keyed RNG)."""
    assert mesh_begin % spec.chunk_meshes == 0 and mesh_end <= min(mesh_begin + spec.chunk_meshes, spec.num_meshes)
    n = int(md["m_MeshLODDatas"]["m_NumMeshlets"][mesh_begin:mesh_end].sum())
    r = _rng(spec.seed, 2, mesh_begin // spec.chunk_meshes)
    ml = np.zeros(n, I.MeshletData)
    s = np.float32(spec.mesh_size)
    u = r.random((n, 4), np.float32)
    bits = r.integers(0, 2 ** 32, (n, 4), dtype=np.uint32)
    ml["m_BoundingSphere"][:, :3] = (u[:, :3] - np.float32(0.5)) * s
    # radius ~log-uniform in [0.01, 0.32) * mesh size: 0.01 * (1 + u) * 2^k, k in 0..3 (exact scaling)
    k = (bits[:, 3] >> 30).astype(np.int32)
    ml["m_BoundingSphere"][:, 3] = np.ldexp((np.float32(1.0) + u[:, 3]) * (np.float32(0.01) * s), k).astype(np.float32)
    # cone: axis bytes arbitrary (the shader normalises after the adjugate transform, basepass.hlsl:103),
    # cutoff byte = 2 * cone_cutoff_s8: even, <= 254 (Visual.cpp:424)
    ml["m_ConeAxisAndCutoff"] = bits[:, 0] & np.uint32(0xFEFFFFFF)
    ml["m_MeshletVertexIDsBufferIdx"] = bits[:, 1] >> 1
    ml["m_MeshletIndexIDsBufferIdx"] = bits[:, 2] >> 1
    ml["m_VertexAndTriangleCount"] = ((bits[:, 3] & 63) + 1) | ((((bits[:, 3] >> 8) % 96) + 1) << 8)
    if spec.meshlet_order == "morton":
        counts = md["m_MeshLODDatas"]["m_NumMeshlets"][mesh_begin:mesh_end].astype(np.int64).ravel()   # (mesh, LOD) ranges in buffer order
        seg = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
        q = np.clip(((u[:, :3]) * np.float32(1024.0)).astype(np.int64), 0, 1023)

        def spread(v):                                     # 10 bits -> every third bit
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        ml = ml[np.lexsort((key, seg))]
    elif spec.meshlet_order != "random":
        raise ValueError(spec.meshlet_order)
    return ml


def meshlet_chunks(spec: SceneSpec, md: np.ndarray):
    """Yield (global_meshlet_offset, MeshletData chunk) covering the whole meshlet buffer."""
    for b in range(0, spec.num_meshes, spec.chunk_meshes):
        e = min(b + spec.chunk_meshes, spec.num_meshes)
        yield int(md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][b, 0]), gen_meshlets_for_meshes(spec, md, b, e)


def gen_instances(spec: SceneSpec, begin: int = 0, end: int | None = None, chunk: int = 1 << 18) -> np.ndarray:
    """BasePassInstanceConstants for instances [begin, end); begin must be chunk aligned."""
    end = spec.num_instances if end is None else end
    assert begin % chunk == 0
    out = np.zeros(end - begin, I.BasePassInstanceConstants)
    for cb in range(begin, end, chunk):
        ce = min(cb + chunk, end)
        n = ce - cb
        r = _rng(spec.seed, 3, cb // chunk)
        pos = np.empty((n, 3), np.float32)
        pos[:, 0] = r.uniform(-spec.box_x, spec.box_x, n)
        pos[:, 1] = r.uniform(-spec.box_y, spec.box_y, n)
        pos[:, 2] = -r.uniform(spec.z_near, spec.z_far, n)
        q = r.standard_normal((n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        sc = np.repeat(r.uniform(0.5, 2.0, (n, 1)), 3, axis=1)
        nu = r.random(n) < spec.nonuniform_scale_fraction
        sc[nu] = r.uniform(0.5, 2.0, (int(nu.sum()), 3))
        x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.empty((n, 3, 3))
        R[:, 0, 0] = 1 - 2 * y * y - 2 * z * z; R[:, 0, 1] = 2 * x * y + 2 * z * w; R[:, 0, 2] = 2 * x * z - 2 * y * w
        R[:, 1, 0] = 2 * x * y - 2 * z * w; R[:, 1, 1] = 1 - 2 * x * x - 2 * z * z; R[:, 1, 2] = 2 * y * z + 2 * x * w
        R[:, 2, 0] = 2 * x * z + 2 * y * w; R[:, 2, 1] = 2 * y * z - 2 * x * w; R[:, 2, 2] = 1 - 2 * x * x - 2 * y * y
        W = np.zeros((n, 4, 4), np.float32)
        W[:, :3, :3] = (R * sc[:, None, :]).astype(np.float32)   # R*S (toyrenderer_common.hlsli:196-203): column j scaled by s_j
        W[:, 3, :3] = pos
        W[:, 3, 3] = 1.0
        o = out[cb - begin:ce - begin]
        o["m_WorldMatrix"] = W
        o["m_PrevWorldMatrix"] = W
        if spec.unique:
            o["m_MeshDataIdx"] = np.arange(cb, ce, dtype=np.uint32) % np.uint32(spec.num_meshes)
        else:
            o["m_MeshDataIdx"] = r.integers(0, spec.num_meshes, n, dtype=np.uint32)
        o["m_MaterialDataIdx"] = r.integers(0, 64, n, dtype=np.uint32)
    return out


def gen_id_lists(spec: SceneSpec):
    """Opaque / alpha-mask primitive id lists (Scene.cpp:282-362): a partition of [0, N)."""
    ids = np.arange(spec.num_instances, dtype=np.uint32)
    if spec.alpha_mask_fraction <= 0:
        return ids, np.zeros(0, np.uint32)
    r = _rng(spec.seed, 4, 0)
    am = r.random(spec.num_instances) < spec.alpha_mask_fraction
    return ids[~am].copy(), ids[am].copy()


@dataclass
class Scene:
    spec: SceneSpec
    instances: np.ndarray
    meshData: np.ndarray
    meshlets: np.ndarray
    opaqueIds: np.ndarray
    alphaMaskIds: np.ndarray
    total_meshlets: int = 0

    def as_oracle(self) -> dict:
        return dict(instances=self.instances, meshData=self.meshData, meshlets=self.meshlets,
                    opaqueIds=self.opaqueIds, alphaMaskIds=self.alphaMaskIds)


def make_scene(spec: SceneSpec) -> Scene:
    """Materialise a whole scene on the host (small / medium configs, tests)."""
    md, total = gen_mesh_table(spec)
    ml = np.zeros(total, I.MeshletData)
    for off, chunk in meshlet_chunks(spec, md):
        ml[off:off + len(chunk)] = chunk
    inst = gen_instances(spec)
    op, am = gen_id_lists(spec)
    return Scene(spec, inst, md, ml, op, am, total)


def animated_nodes(spec: SceneSpec, frame: int, groups: int = 4096, nodes: np.ndarray | None = None):
    """Node hierarchy of the animated configs (BASELINE configs[4]): one node per instance hanging below one of `groups`
    group nodes; every frame all rotations, scales and positions change (seeded by `frame`).  Returns
    (NodeLocalTransform[n + groups], primitive -> node)."""
    n = spec.num_instances
    if nodes is None:
        nodes = np.zeros(n + groups, I.NodeLocalTransform)
        nodes["m_ParentNodeIdx"][:n] = n + _rng(spec.seed, 11, 0).integers(0, groups, n)
        nodes["m_ParentNodeIdx"][n:] = 0xFFFFFFFF
    r = _rng(spec.seed, 12, frame)
    q = r.standard_normal((n + groups, 4), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    nodes["m_Rotation"] = q
    nodes["m_Scale"][:n] = r.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    nodes["m_Scale"][n:] = 1.0
    nodes["m_Position"][:n] = r.standard_normal((n, 3), dtype=np.float32) * np.float32(3.0)
    gp = np.empty((groups, 3), np.float32)
    gp[:, 0] = r.uniform(-spec.box_x, spec.box_x, groups); gp[:, 1] = r.uniform(-spec.box_y, spec.box_y, groups)
    gp[:, 2] = -r.uniform(spec.z_near, spec.z_far, groups)
    nodes["m_Position"][n:] = gp
    return nodes, np.arange(n, dtype=np.uint32)


# ----------------------------------------------------------------------------- named configs
def config_spec(name: str) -> SceneSpec:
    """BASELINE.json configs made concrete."""
    if name == "C0":   # cornell-like: 3 primitives, 3-4 meshlets
        return SceneSpec(num_meshes=3, num_instances=3, meshlets_lod0=2, jitter_meshlets=True, max_lods=1,
                         box_x=1.0, box_y=1.0, z_near=3.0, z_far=6.0, mesh_size=1.0)
    if name == "C1":   # ~50 k meshlets: 400 instances x ~125
        return SceneSpec(num_meshes=40, num_instances=400, meshlets_lod0=125, jitter_meshlets=True, max_lods=4,
                         alpha_mask_fraction=0.1)
    if name == "C2":   # C1 meshes x1000 instancing ~ 50 M meshlets (cache-resident meshlet data)
        return SceneSpec(num_meshes=40, num_instances=400_000, meshlets_lod0=125, jitter_meshlets=True, max_lods=4,
                         alpha_mask_fraction=0.1)
    if name == "C3":   # 100 M unique meshlets, 781 250 x 128, one LOD (tested count exact)
        return SceneSpec(num_meshes=781_250, num_instances=781_250, meshlets_lod0=128, max_lods=1, unique=True)
    if name == "C4":   # BASELINE configs[4] at full size: 1 B unique meshlets, 7 812 500 x 128 (32 GB of MeshletData: fits one 288-GB MI355X)
        return SceneSpec(num_meshes=7_812_500, num_instances=7_812_500, meshlets_lod0=128, max_lods=1, unique=True)
    if name == "C4r":  # one rank's share of C4 (1 B meshlets over 8 GPUs): 125 M unique meshlets; transforms animated by the caller
        return SceneSpec(num_meshes=976_562, num_instances=976_562, meshlets_lod0=128, max_lods=1, unique=True)
    if name == "C3e":  # 1/8 of C3: the per-GPU share of the 8-GPU run (overhead proxy on one GPU)
        return SceneSpec(num_meshes=97_656, num_instances=97_656, meshlets_lod0=128, max_lods=1, unique=True)
    if name == "C3r":  # one rank's share of C3 on 8 GPUs: 97 656 x 128 = 12.5 M meshlets, 390 625 groups (< 2^19: the small-pass code paths)
        return SceneSpec(num_meshes=97_656, num_instances=97_656, meshlets_lod0=128, max_lods=1, unique=True)
    if name == "C3m":  # DIAGNOSTIC: C3 with the meshlets of every mesh in Z-order (what spatial coherence inside a mesh is worth to the HZB lookups)
        return SceneSpec(num_meshes=781_250, num_instances=781_250, meshlets_lod0=128, max_lods=1, unique=True, meshlet_order="morton")
    if name == "C3s":  # 1/64 of C3 for quick GPU parity runs
        return SceneSpec(num_meshes=12_208, num_instances=12_208, meshlets_lod0=128, max_lods=1, unique=True)
    raise KeyError(name)


# ----------------------------------------------------------------------------- synthetic depth
def gen_depth(view: View, num_occluders: int = 200, seed: int = SEED_CAMERA, scale: float = 1.0) -> np.ndarray:
    """Analytic occluder field: screen-space boxes at constant view depth, reverse-Z
    (depth = near / z, far = 0, GraphicConstants kFarDepth).  float32 [renderH, renderW]."""
    W, H = view.renderW, view.renderH
    r = _rng(seed, 5, 0)
    depth = np.zeros((H, W), np.float32)
    for _ in range(num_occluders):
        z = float(r.uniform(15.0, 160.0))
        w = int(r.uniform(0.01, 0.06) * scale * W); h = int(r.uniform(0.02, 0.10) * scale * H)
        x0 = int(r.uniform(-0.05, 0.95) * W); y0 = int(r.uniform(-0.05, 0.95) * H)
        x1, y1 = min(W, x0 + w), min(H, y0 + h)
        x0, y0 = max(0, x0), max(0, y0)
        if x1 <= x0 or y1 <= y0:
            continue
        d = np.float32(view.nearPlane) / np.float32(z)
        np.maximum(depth[y0:y1, x0:x1], d, out=depth[y0:y1, x0:x1])
    return depth
