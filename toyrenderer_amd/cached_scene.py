"""`<scene>_CachedData.bin` (version 3): the on-disk form of the path's STATIC inputs -- the reference's mesh-processing
cache (source/SceneLoading.cpp:57-79 layout, :706-781 reader, :1090-1145 writer).  The file is the raw arrays the cull
and the mesh shader consume, back to back, after a 32-byte header:

    Header            8 x u32  {version = 3, meshoptimizer version, #vertices, #indices, #meshes,
                                #meshlet vertex ids, #meshlet triangles (packed), #meshlets}
    RawVertexFormat   20 B each   (ShaderInterop.h:278-283)
    indices           u32 each    (GraphicConstants.h:34), per mesh, relative to the mesh's first vertex
    MeshData          156 B each  (ShaderInterop.h:182-189)
    meshlet vertex ids    u32 each, indices into the global vertex buffer (basepass.hlsl:151-152)
    meshlet triangles     u32 each, a | b << 8 | c << 16 (basepass.hlsl:178-184)
    MeshletData       32 B each   (ShaderInterop.h:191-198)
    MeshSpecificData  32 B each   {#indices, #vertices, AABB centre xyz, AABB extents xyz} (SceneLoading.cpp:73-78)
    [animation key frames: only when the glTF has animations; sized by the glTF, kept here as an opaque tail]

What the file does NOT hold: instances, nodes, materials, cameras -- those come from the glTF next to it
(gltf_lite.load).  A file written by the reference (real meshoptimizer meshlets and LODs) therefore drops into
`apply()` below and replaces this build's own LOD-0 meshlets with the reference's; no such file ships with the
reference snapshot (it is generated on first load), so the round trip through this module's own writer is what the
tests pin (tests/test_cached_scene.py)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import interop as I

VERSION = 3                                                                 # SceneLoading.cpp:59
Header = np.dtype([("m_Version", np.uint32), ("m_MeshOptVersion", np.uint32), ("m_NumVertices", np.uint32), ("m_NumIndices", np.uint32),
                   ("m_NumMeshes", np.uint32), ("m_NumMeshletVertexIdxOffsets", np.uint32), ("m_NumMeshletIndices", np.uint32),
                   ("m_NumMeshletDatas", np.uint32)])
MeshSpecificData = np.dtype([("m_NumIndices", np.uint32), ("m_NumVertices", np.uint32),
                             ("m_AABBCenter", np.float32, (3,)), ("m_AABBExtents", np.float32, (3,))])
assert Header.itemsize == 32 and MeshSpecificData.itemsize == 32


@dataclass
class CachedData:
    meshopt_version: int
    vertices: np.ndarray            # RawVertexFormat
    indices: np.ndarray             # u32
    meshData: np.ndarray            # MeshData
    meshletVertexIds: np.ndarray    # u32
    meshletTriangles: np.ndarray    # u32 packed
    meshlets: np.ndarray            # MeshletData
    meshSpecific: np.ndarray        # MeshSpecificData
    tail: bytes = b""               # animation key frames (layout given by the glTF's animations)

    def validate(self):
        """The cross-references the cull and the mesh shader follow blindly; raises ValueError naming the first bad one."""
        md, ml = self.meshData, self.meshlets
        if len(self.meshSpecific) != len(md):
            raise ValueError(f"{len(self.meshSpecific)} MeshSpecificData for {len(md)} meshes")
        for i, m in enumerate(md):
            n = int(m["m_NumLODs"])
            if not 1 <= n <= I.kMaxNumMeshLODs:
                raise ValueError(f"mesh {i}: {n} LODs")
            lods = m["m_MeshLODDatas"][:n]
            end = lods["m_MeshletDataBufferIdx"].astype(np.int64) + lods["m_NumMeshlets"]
            if end.max(initial=0) > len(ml):
                raise ValueError(f"mesh {i}: meshlet range ends at {int(end.max())} of {len(ml)}")
        if len(ml):
            nv = (ml["m_VertexAndTriangleCount"] & 0xFF).astype(np.int64)
            nt = ((ml["m_VertexAndTriangleCount"] >> 8) & 0xFF).astype(np.int64)
            if nv.max() > 64 or nt.max() > 96:                                                   # ShaderInterop.h:19-20
                raise ValueError(f"meshlet with {int(nv.max())} vertices / {int(nt.max())} triangles")
            if (ml["m_MeshletVertexIDsBufferIdx"] + nv).max() > len(self.meshletVertexIds):
                raise ValueError("meshlet vertex-id range past the end of the buffer")
            if (ml["m_MeshletIndexIDsBufferIdx"] + nt).max() > len(self.meshletTriangles):
                raise ValueError("meshlet triangle range past the end of the buffer")
        if len(self.meshletVertexIds) and int(self.meshletVertexIds.max()) >= len(self.vertices):
            raise ValueError("meshlet vertex id past the end of the vertex buffer")
        return self


def read(path: str) -> CachedData:
    """LoadCachedData (SceneLoading.cpp:706-781): header, then the arrays in file order.  Raises ValueError on a version
    other than 3 or a truncated file (the reference `check()`s the same conditions)."""
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < Header.itemsize:
        raise ValueError(f"{path}: {len(raw)} bytes, no header")
    h = np.frombuffer(raw, Header, 1)[0]
    if int(h["m_Version"]) != VERSION:
        raise ValueError(f"{path}: cached data version {int(h['m_Version'])}, this reader handles {VERSION}")
    off = Header.itemsize

    def take(dtype, n):
        nonlocal off
        dtype = np.dtype(dtype)
        nbytes = dtype.itemsize * int(n)
        if off + nbytes > len(raw):
            raise ValueError(f"{path}: truncated ({len(raw)} bytes, need {off + nbytes})")
        a = np.frombuffer(raw, dtype, int(n), off).copy()
        off += nbytes
        return a
    c = CachedData(int(h["m_MeshOptVersion"]),
                   take(I.RawVertexFormat, h["m_NumVertices"]), take(np.uint32, h["m_NumIndices"]), take(I.MeshData, h["m_NumMeshes"]),
                   take(np.uint32, h["m_NumMeshletVertexIdxOffsets"]), take(np.uint32, h["m_NumMeshletIndices"]),
                   take(I.MeshletData, h["m_NumMeshletDatas"]), take(MeshSpecificData, h["m_NumMeshes"]))
    c.tail = raw[off:]
    return c.validate()


def write(path: str, c: CachedData):
    """WriteCachedData (SceneLoading.cpp:1090-1145)."""
    c.validate()
    h = np.zeros(1, Header)
    h["m_Version"], h["m_MeshOptVersion"] = VERSION, c.meshopt_version
    h["m_NumVertices"], h["m_NumIndices"], h["m_NumMeshes"] = len(c.vertices), len(c.indices), len(c.meshData)
    h["m_NumMeshletVertexIdxOffsets"], h["m_NumMeshletIndices"], h["m_NumMeshletDatas"] = len(c.meshletVertexIds), len(c.meshletTriangles), len(c.meshlets)
    with open(path, "wb") as f:
        f.write(h.tobytes())
        for a, dt in ((c.vertices, I.RawVertexFormat), (c.indices, np.uint32), (c.meshData, I.MeshData), (c.meshletVertexIds, np.uint32),
                      (c.meshletTriangles, np.uint32), (c.meshlets, I.MeshletData), (c.meshSpecific, MeshSpecificData)):
            f.write(np.ascontiguousarray(a, dt).tobytes())
        f.write(c.tail)


def from_scene(scene, meshopt_version: int = 0) -> CachedData:
    """The cacheable half of a gltf_lite.LoadedScene.  The index buffer is rebuilt from the LOD-0 meshlets (the path
    itself never reads it); meshopt_version 0 marks a file whose meshlets are this build's own."""
    md = scene.meshData.copy()
    ml = scene.meshlets
    spec = np.zeros(len(md), MeshSpecificData)
    idx_parts, cursor = [], 0
    vb = md["m_GlobalVertexBufferIdx"].astype(np.int64)
    order = np.argsort(vb, kind="stable")
    vend = np.empty(len(md), np.int64)
    vend[order] = np.append(vb[order][1:], len(scene.vertices))
    for i, m in enumerate(md):
        lod0 = m["m_MeshLODDatas"][0]
        tris = []
        for k in range(int(lod0["m_MeshletDataBufferIdx"]), int(lod0["m_MeshletDataBufferIdx"]) + int(lod0["m_NumMeshlets"])):
            nv, nt = int(ml[k]["m_VertexAndTriangleCount"]) & 0xFF, (int(ml[k]["m_VertexAndTriangleCount"]) >> 8) & 0xFF
            vids = scene.meshletVertexIds[int(ml[k]["m_MeshletVertexIDsBufferIdx"]):][:nv]
            packed = scene.meshletTriangles[int(ml[k]["m_MeshletIndexIDsBufferIdx"]):][:nt]
            local = np.stack([packed & 0xFF, (packed >> 8) & 0xFF, (packed >> 16) & 0xFF], 1)
            tris.append(vids[local].astype(np.int64) - vb[i])
        flat = np.concatenate(tris).reshape(-1).astype(np.uint32) if tris else np.zeros(0, np.uint32)
        md["m_GlobalIndexBufferIdx"][i] = cursor
        cursor += len(flat)
        idx_parts.append(flat)
        pos = scene.vertices["m_Position"][vb[i]:vend[i]]
        spec["m_NumIndices"][i], spec["m_NumVertices"][i] = len(flat), len(pos)
        if len(pos):
            lo, hi = pos.min(0), pos.max(0)
            spec["m_AABBCenter"][i], spec["m_AABBExtents"][i] = (lo + hi) * np.float32(0.5), (hi - lo) * np.float32(0.5)
    return CachedData(meshopt_version, scene.vertices, np.concatenate(idx_parts) if idx_parts else np.zeros(0, np.uint32), md,
                      scene.meshletVertexIds, scene.meshletTriangles, ml, spec)


def apply(scene, c: CachedData):
    """The glTF gives instances / nodes / cameras, the cache gives geometry (SceneLoading.cpp: the cached path skips
    Mesh::Initialize and uploads the cached arrays): returns the scene with the cache's meshes and meshlets."""
    import dataclasses
    if len(c.meshData) != len(scene.meshData):
        raise ValueError(f"cache holds {len(c.meshData)} meshes, the glTF {len(scene.meshData)} (SceneLoading.cpp:726)")
    return dataclasses.replace(scene, meshData=c.meshData.copy(), meshlets=c.meshlets.copy(), vertices=c.vertices.copy(),
                               meshletVertexIds=c.meshletVertexIds.copy(), meshletTriangles=c.meshletTriangles.copy())
